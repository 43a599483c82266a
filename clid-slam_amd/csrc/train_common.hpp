// Pieces shared by the training kernels (train.hip: numerical-eikonal fused iteration; train_analytic.hip:
// analytic-eikonal iteration): workspace carving, decimation bookkeeping, decoder-gradient accumulation on
// the matrix cores, block flush of the accumulators.
#pragma once
#include "common.hpp"

namespace clid {

constexpr int kPartialStride = 840;  // 833 decoder grads | bce sum | eik sum | pad
constexpr int kMaxBwdBlocks = 1024;

constexpr int kRecFloatsPerTask = 192;  // qinfo[8] | qdesc[8] | win[8][8]: what the search phase hands the decode phase

struct TrainWs {
  float* partial;  // [kMaxBwdBlocks][kPartialStride]
  float* rec;      // [chunk iterations][tasks][kRecFloatsPerTask]: the hoisted searches of clid_mapping_run
};

// upper bound on wave tasks for Q queries (bundles carry 7-8 queries, plain tasks up to 8; the sparsest case is
// decimation 3: one bundle + one single-sample task per 3 samples = 2Q/9 tasks)
__host__ inline size_t max_tasks(int Q) { return (size_t)Q / 4 + 16; }
// one record buffer holds the searches of a chunk of iterations: 16 iterations at the task bound (typically
// ~32 at the actual task count), capped at 1 GiB for very large batches, never less than one iteration
constexpr int kMaxChunkIters = 32;
__host__ inline size_t rec_buffer_floats(int Q) {
  const size_t one = max_tasks(Q) * kRecFloatsPerTask;
  size_t it = (size_t(1) << 28) / one;
  if (it > 16) it = 16;
  if (it < 1) it = 1;
  return one * it;
}

__host__ __device__ inline int fd_first(long long batch_offset, int decim) {
  const int r = (int)(batch_offset % decim);
  return r == 0 ? 0 : decim - r;
}
__host__ __device__ inline int fd_count(int bs, long long batch_offset, int decim) {
  const int first = fd_first(batch_offset, decim);
  return first >= bs ? 0 : (bs - first + decim - 1) / decim;
}

__host__ inline TrainWs carve(float* ws, int Q) {
  TrainWs t;
  size_t o = 0;
  auto take = [&](size_t n) {
    float* p = ws + o;
    o += (n + 3) & ~size_t(3);
    return p;
  };
  t.partial = take((size_t)kMaxBwdBlocks * kPartialStride);
  t.rec = take(rec_buffer_floats(Q));
  return t;
}

// ---- wave tasks of the training kernels (search records are laid out per task) -------------------------
struct QDesc {
  int p;      // position in this rank's batch, -1 = padding
  int axis;   // -1 batch sample, 0..2 shifted copy
  float sign;
};

// Task -> query mapping (no integer division on the common path).  The local batch is cut into lattice
// blocks of `decim` positions starting at a decimated sample pj = first + j*decim:
//   bundle task j        : the 6 shifted copies of pj, pj itself, and pj - 1 (the sample in front of it)
//   plain task (b, c)    : positions pj_b + 1 + 8c .. pj_b + 8 + 8c of block b (offsets <= decim - 2),
//                          b = -1 covers the samples in front of the first decimated one
//   tail task            : the last position of the last block (nobody's "pj - 1")
// n_fd == 0 (no eikonal term): task t simply covers positions 8t .. 8t+7.
struct TaskMap {
  int bs, n_fd, first, decim;
  int chunks;   // plain tasks per lattice block = ceil((decim - 2) / 8)
  int n_plain;  // (n_fd + 1) * chunks
  int n_tasks;
};
__host__ __device__ inline TaskMap make_task_map(int bs, int n_fd, int first, int decim) {
  TaskMap m;
  m.bs = bs; m.n_fd = n_fd; m.first = first; m.decim = decim;
  if (n_fd == 0) {
    m.chunks = 1; m.n_plain = (bs + 7) / 8; m.n_tasks = m.n_plain;
  } else {
    m.chunks = decim > 2 ? (decim - 2 + 7) / 8 : 0;
    m.n_plain = (n_fd + 1) * m.chunks;
    m.n_tasks = n_fd + m.n_plain + 1;  // + tail task
  }
  return m;
}
__host__ __device__ __forceinline__ QDesc task_query(const TaskMap& tm, int task, int round, int grp) {
  QDesc q;
  q.axis = -1;
  q.sign = 0.f;
  q.p = -1;
  const int slot = round * 4 + grp;
  if (tm.n_fd == 0) {
    const int p = task * 8 + slot;
    q.p = p < tm.bs ? p : -1;
  } else if (task < tm.n_fd) {  // bundle: A = x+,x-,y+,y- ; B = z+, z-, sample, the sample in front of it
    const int pj = tm.first + task * tm.decim;
    if (round == 0) {
      q.p = pj; q.axis = grp >> 1; q.sign = (grp & 1) ? -1.f : 1.f;
    } else if (grp < 2) {
      q.p = pj; q.axis = 2; q.sign = grp ? -1.f : 1.f;
    } else if (grp == 2) {
      q.p = pj;
    } else {
      q.p = tm.decim >= 2 ? pj - 1 : -1;  // -1 for the very first sample: padding
    }
  } else if (task < tm.n_fd + tm.n_plain) {
    const int t = task - tm.n_fd;
    const int b = (tm.chunks == 1) ? t : t / tm.chunks;
    const int c = t - b * tm.chunks;
    const int off = 1 + c * 8 + slot;
    const int p = tm.first + (b - 1) * tm.decim + off;
    q.p = (off <= tm.decim - 2 && p >= 0 && p < tm.bs) ? p : -1;
  } else {
    const int p = tm.first + tm.n_fd * tm.decim - 1;
    q.p = (slot == 0 && p < tm.bs && tm.decim >= 2) ? p : -1;
  }
  return q;
}


// ---- decoder-gradient accumulation -------------------------------------------------------------------
// dW1 [64 x 11] = sum_q dh_q (x) f_q is a GEMM whose reduction runs over the QUERIES, so it goes on the
// matrix cores: v_mfma_f32_16x16x4_f32 (exact fp32, == an fmaf chain) with
//   A[i = lane&15][k = lane>>4] = dh of hidden unit 16u + i of query k   (the lane's own dh[u])
//   B[k = lane>>4][j = lane&15] = f_j of query k, j < 11;  1 for j == 11 (=> column 11 accumulates db1)
// i.e. one instruction per 16-hidden tile consumes the 4 queries of the wave with no data movement, and
// the accumulator D[row = 4*(lane>>4) + r][col = lane&15] is already summed over the wave's queries.
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MlpAcc {
  f32x4 dW1[CLID_HPL];   // tile u: hidden 16u + 4*(lane>>4) + r, column lane&15 (0..10 dW1, 11 db1)
  float dW2[CLID_HPL];   // hidden lane16 + 16u, this group's queries only
  float db2;             // lane16 == 0 only
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int u = 0; u < CLID_HPL; ++u) {
      dW1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      dW2[u] = 0.f;
    }
    db2 = 0.f;
  }
};

constexpr int kRedFloats = CLID_MLP_PARAMS + 3;  // 833 grads | bce | eik | pad

// Block reduction of the waves' accumulators: plain LDS stores into per-wave rows, one barrier, then
// a 4-way sum and one coalesced global store.  (LDS float atomics -- ds_add_f32 -- retire at ~1-2
// lanes/clk on gfx950: measured 20 us for this flush, so they are avoided.)
// With a frozen decoder (freeze_model after `freeze_after_frame`, the steady state of a run) only the two loss
// sums leave the block.
__device__ __forceinline__ void flush_mlp_acc(const MlpAcc& acc, float bce, float eik, float* red /*LDS [4][kRedFloats]*/,
                                              float* __restrict__ out /* [kPartialStride] */, bool train_decoder = true) {
  const int lane = threadIdx.x & 63, lane16 = lane & 15, grp = lane >> 4, wave = threadIdx.x >> 6;
  float* mine = red + wave * kRedFloats;
  if (!train_decoder) {
    const float v1 = cross_group_sum(bce), v2 = cross_group_sum(eik);
    if (lane == 0) {
      mine[CLID_MLP_PARAMS] = v1;
      mine[CLID_MLP_PARAMS + 1] = v2;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
      float s = 0.f;
      for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) s += red[wv * kRedFloats + CLID_MLP_PARAMS + threadIdx.x];
      out[CLID_MLP_PARAMS + threadIdx.x] = s;
    }
    return;
  }
#pragma unroll
  for (int u = 0; u < CLID_HPL; ++u) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int h = CLID_G * u + 4 * grp + r;
      if (lane16 < CLID_D) mine[h * CLID_D + lane16] = acc.dW1[u][r];
      else if (lane16 == CLID_D) mine[CLID_H * CLID_D + h] = acc.dW1[u][r];
    }
    const float w2 = cross_group_sum(acc.dW2[u]);
    if (lane < CLID_G) mine[CLID_H * CLID_D + CLID_H + lane16 + CLID_G * u] = w2;
  }
  {
    const float v0 = cross_group_sum(acc.db2), v1 = cross_group_sum(bce), v2 = cross_group_sum(eik);
    if (lane == 0) {
      mine[CLID_MLP_PARAMS - 1] = v0;
      mine[CLID_MLP_PARAMS] = v1;
      mine[CLID_MLP_PARAMS + 1] = v2;
    }
  }
  __syncthreads();
  const int nw = blockDim.x >> 6;
  for (int i = threadIdx.x; i < CLID_MLP_PARAMS + 2; i += blockDim.x) {
    float s = 0.f;
    for (int wv = 0; wv < nw; ++wv) s += red[wv * kRedFloats + i];
    out[i] = s;
  }
}

// decoder backward for one query given dz = scale * dL/dsdf; returns df (replicated)
__device__ __forceinline__ void mlp_backward(const MlpLds& s, const float (&f)[CLID_D],
                                             const float (&pre)[CLID_HPL], float dz, int lane16,
                                             bool train_decoder, MlpAcc& acc, float (&df)[CLID_D]) {
  float dh[CLID_HPL];
  const int l16 = lane16 + opaque_zero();  // keep the weights in LDS (see opaque_zero)
#pragma unroll
  for (int u = 0; u < CLID_HPL; ++u) {
    const int h = l16 + CLID_G * u;
    const bool on = pre[u] > 0.f;
    dh[u] = on ? dz * s.w[CLID_H * CLID_D + CLID_H + h] : 0.f;
    if (train_decoder) acc.dW2[u] += on ? dz * pre[u] : 0.f;
  }
  if (train_decoder) {
    float fb = (lane16 == CLID_D) ? 1.0f : 0.f;
#pragma unroll
    for (int c = 0; c < CLID_D; ++c) fb = (lane16 == c) ? f[c] : fb;
#pragma unroll
    for (int u = 0; u < CLID_HPL; ++u)
      acc.dW1[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(dh[u], fb, acc.dW1[u], 0, 0, 0);
    if (lane16 == 0) acc.db2 += dz;
  }
#pragma unroll
  for (int c = 0; c < CLID_D; ++c) {
    float part = 0.f;
#pragma unroll
    for (int u = 0; u < CLID_HPL; ++u) part = fmaf(s.w[(l16 + CLID_G * u) * CLID_D + c], dh[u], part);
    df[c] = group_sum(part);
  }
}

}  // namespace clid

// ---- Adam (torch.optim.Adam._single_tensor_adam; ATen's op order) ----------------------------------
//   m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, 1-b2); denom = sqrt(v)/sqrt(bc2) + eps; p.addcdiv_(m, denom, -lr/bc1)
namespace clid {
struct AdamK {
  float one_m_b1, b2, one_m_b2, bc2_sqrt, eps, neg_step, wd;
};
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, const AdamK& k, float wd) {
  if (wd != 0.f) g = fmaf(wd, p, g);
  m = fadd(m, fmul(k.one_m_b1, fsub(g, m)));
  v = fadd(fmul(v, k.b2), fmul(fmul(k.one_m_b2, g), g));
  const float denom = fadd(fdiv(sqrtf(v), k.bc2_sqrt), k.eps);
  p = fadd(p, fdiv(fmul(k.neg_step, m), denom));
}
__host__ inline AdamK adam_scalars(float lr, float b1, float b2, float eps, float wd, int step) {
  AdamK k;
  k.one_m_b1 = (float)(1.0 - (double)b1);
  k.b2 = b2;
  k.one_m_b2 = (float)(1.0 - (double)b2);
  k.eps = eps;
  k.wd = wd;
  if (step < 1) {  // "step 0": nothing to apply yet (the fused loop's first launch)
    k.bc2_sqrt = 1.f;
    k.neg_step = 0.f;
    return k;
  }
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  k.bc2_sqrt = (float)sqrt(bc2);
  k.neg_step = (float)(-((double)lr / bc1));
  return k;
}

// ---- the one-launch iteration (k_decode_tile<.., FUSED>): Adam applied ON READ ------------------------------------
// Iteration t's launch needs theta_t = Adam(theta_{t-1}, g_{t-1}).  Instead of a separate Adam launch between two
// decode launches (a 5 us kernel + a 3.4 us boundary at the ncd128 batch size), the tile waves apply the pending update to
// the few rows they gather, from read-only copies of (theta, m, v, g) of the previous iteration, while extra blocks of the
// SAME launch sweep all rows and write (theta, m, v) of this iteration into the other half of a ping-pong pair.  Nothing
// inside the launch depends on anything written by it, so no grid-wide synchronisation is needed; the kernel boundary to
// the next iteration is the only barrier.  Gradients rotate over three accumulation buffers: read g_{t-1}, accumulate
// g_t, zero the buffer of g_{t+1}.  Decoder gradients and the two loss sums are added by atomics into 8 copies of an
// 848-float head of the accumulation buffer (same-line contention: tools/ubench_atomic.hip), summed by the reader.
constexpr int kGHeadCopies = 8;
constexpr int kGHeadStride = 848;                       // 833 decoder gradients | bce | eik | pad, 64-byte aligned
constexpr int kGHead = kGHeadCopies * kGHeadStride;     // floats in front of the accumulation rows
struct FusedIter {
  const float* th_old; float* th_new;                   // features [(M+1)*F]
  const float* m_old; float* m_new;
  const float* v_old; float* v_new;
  const float* g_prev; float* g_cur; float* g_next;     // accumulation buffers [kGHead + (M+1)*16]
  const float* wd_old; float* wd_new;                   // decoder W1|b1|W2|b2 flat [848] (trainable decoder only)
  const float* mw_old; float* mw_new;
  const float* vw_old; float* vw_new;
  float* cert; int n_cert;                              // local_point_certainties: += column 8 of g_prev
  long long n_feat;
  float* loss_prev;                                     // [4] loss row of the previous iteration (NULL at the first)
  float inv_n_main, inv_n_eik, weight_e;
  int step;                                             // Adam step applied on read (0 = none)
  int n_tile_blocks;                                    // blocks [0, n_tile_blocks) run tiles, the rest sweep rows
  AdamK k;
};

// theta_t of one float4 of a feature row from the previous iteration's state (identical arithmetic for the readers and
// for the sweep, so what a tile wave uses is bit for bit what the sweep stores)
__device__ __forceinline__ float4 adam_on_read4(const FusedIter& fi, long long i4, float4* m_out = nullptr, float4* v_out = nullptr,
                                                bool* idle_out = nullptr) {
  float4 P = *reinterpret_cast<const float4*>(fi.th_old + i4);
  bool idle = true;
  if (fi.step > 0) {
    const long long row = i4 >> 3;
    const float4 G = *reinterpret_cast<const float4*>(fi.g_prev + kGHead + row * CLID_GRAD_ROW16 + (i4 & 7));
    float4 M = *reinterpret_cast<const float4*>(fi.m_old + i4), V = *reinterpret_cast<const float4*>(fi.v_old + i4);
    idle = fi.k.wd == 0.f && G.x == 0.f && G.y == 0.f && G.z == 0.f && G.w == 0.f && M.x == 0.f && M.y == 0.f && M.z == 0.f &&
           M.w == 0.f && V.x == 0.f && V.y == 0.f && V.z == 0.f && V.w == 0.f;
    if (!idle) {
      adam_update(P.x, G.x, M.x, V.x, fi.k, fi.k.wd);
      adam_update(P.y, G.y, M.y, V.y, fi.k, fi.k.wd);
      adam_update(P.z, G.z, M.z, V.z, fi.k, fi.k.wd);
      adam_update(P.w, G.w, M.w, V.w, fi.k, fi.k.wd);
    }
    if (m_out) *m_out = M;
    if (v_out) *v_out = V;
  }
  if (idle_out) *idle_out = idle;
  return P;
}
__device__ __forceinline__ float adam_on_read1(const FusedIter& fi, long long i) {
  float P = fi.th_old[i];
  if (fi.step > 0) {
    const float G = fi.g_prev[kGHead + (i >> 3) * CLID_GRAD_ROW16 + (i & 7)];
    float M = fi.m_old[i], V = fi.v_old[i];
    if (!(fi.k.wd == 0.f && G == 0.f && M == 0.f && V == 0.f)) adam_update(P, G, M, V, fi.k, fi.k.wd);
  }
  return P;
}
// the sweep over the feature rows: thread t owns float4 number t.  Writes theta / m / v of this iteration (rows idle since
// the call began are identical in both halves of the ping-pong pair and are left alone), merges the certainty column of
// g_prev (np.py:714) and zeroes the row of g_next.
__device__ __forceinline__ void adam_sweep_thread(const FusedIter& fi, long long t4, bool write_state, bool zero_next) {
  const long long i4 = t4 * 4;
  if (i4 >= fi.n_feat) return;
  const long long row = i4 >> 3;
  const bool second = (i4 & 7) == 4;
  if (zero_next && fi.g_next) {
    float* z = fi.g_next + kGHead + row * CLID_GRAD_ROW16 + (i4 & 7);
    *reinterpret_cast<float4*>(z) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (second) {
      *reinterpret_cast<float4*>(z + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(z + 8) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (fi.step <= 0) return;
  float4 M, V;
  bool idle;
  const float4 P = adam_on_read4(fi, i4, &M, &V, &idle);
  if (!idle) {
    *reinterpret_cast<float4*>(fi.th_new + i4) = P;
    if (write_state) {
      *reinterpret_cast<float4*>(fi.m_new + i4) = M;
      *reinterpret_cast<float4*>(fi.v_new + i4) = V;
    }
  }  // (a row idle since the call began holds the same value in both halves of the ping-pong pair: nothing to copy)
  if (second && fi.cert && row < fi.n_cert) {
    const float inc = fi.g_prev[kGHead + row * CLID_GRAD_ROW16 + CLID_F];
    if (inc != 0.f) fi.cert[row] += inc;
  }
}
// decoder parameter i (flat layout) of this iteration from the previous state + the 8 head copies of g_prev
__device__ __forceinline__ float decoder_on_read(const FusedIter& fi, int i, float* m_out = nullptr, float* v_out = nullptr) {
  float P = fi.wd_old[i];
  if (fi.step > 0) {
    float gsum = 0.f;
#pragma unroll
    for (int c = 0; c < kGHeadCopies; ++c) gsum += fi.g_prev[c * kGHeadStride + i];
    float M = fi.mw_old[i], V = fi.vw_old[i];
    adam_update(P, gsum, M, V, fi.k, 0.f);
    if (m_out) *m_out = M;
    if (v_out) *v_out = V;
  }
  return P;
}
// total / bce / eikonal of the previous iteration from the head copies of g_prev (one thread)
__device__ __forceinline__ void finish_loss_prev(const FusedIter& fi) {
  if (!fi.loss_prev) return;
  float bce = 0.f, eik = 0.f;
  for (int c = 0; c < kGHeadCopies; ++c) {
    bce += fi.g_prev[c * kGHeadStride + CLID_MLP_PARAMS];
    eik += fi.g_prev[c * kGHeadStride + CLID_MLP_PARAMS + 1];
  }
  bce *= fi.inv_n_main;
  eik *= fi.inv_n_eik;
  fi.loss_prev[0] += bce + fi.weight_e * eik;
  fi.loss_prev[1] += bce;
  fi.loss_prev[2] += eik;
}
}  // namespace clid

// ---- launches under the optional per-kernel timing of clid_profile_enable (train.hip) -------------------------
// Inside a prof_begin / prof_end bracket the kernel goes out through hipExtLaunchKernelGGL with the bracket's start and
// stop events, which then hold the DISPATCH's begin / end time stamps (the same clock rocprofv3 --kernel-trace reads).
#include <hip/hip_ext.h>
bool clid_prof_take(hipEvent_t* a, hipEvent_t* b);
#define CLID_KLAUNCH(kernel, grid, block, shmem, stream, ...)                                        \
  do {                                                                                               \
    hipEvent_t ea__, eb__;                                                                           \
    if (clid_prof_take(&ea__, &eb__))                                                                \
      hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, ea__, eb__, 0, __VA_ARGS__);         \
    else                                                                                             \
      hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                           \
  } while (0)

// host-side launchers of the tile (matrix-core) decode kernels (train_tile.hip); prec 0 = fp32, 1 = bf16 operands
int clid_launch_decode_tile(const clid_map_view* mv, const clid_train_args* a, float* partial, const clid::TaskMap& tmap,
                            const float* rec, int prec, hipStream_t s, const clid::FusedIter* fused = nullptr);
int clid_launch_fused_final(const clid::FusedIter* fi, float* W1, float* b1, float* W2, float* b2, int train_decoder, hipStream_t s);
int clid_decode_tile_blocks(int n_tasks);
// host-side launchers of the analytic-eikonal iteration (train_analytic.hip)
int clid_launch_train_analytic(const clid_map_view* mv, const clid_train_args* a, float* partial, hipStream_t s);
int clid_train_analytic_blocks(int bs);
