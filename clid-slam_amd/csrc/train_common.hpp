// Pieces shared by the training kernels (train.hip: numerical-eikonal fused iteration; train_analytic.hip:
// analytic-eikonal iteration): workspace carving, decimation bookkeeping, decoder-gradient accumulation on
// the matrix cores, block flush of the accumulators.
#pragma once
#include "common.hpp"

namespace clid {

constexpr int kPartialStride = 840;  // 833 decoder grads | bce sum | eik sum | pad
#ifndef CLID_MAX_BWD_BLOCKS
#define CLID_MAX_BWD_BLOCKS 1024
#endif
constexpr int kMaxBwdBlocks = CLID_MAX_BWD_BLOCKS;

constexpr int kRecFloatsPerTask = 192;  // qinfo[8] | qdesc[8] | win[8][8]: what the search phase hands the decode phase
// Behind an iteration's task records: one block of kTileNumWords 32-bit words per TILE (= two consecutive tasks, the unit of
// the matrix-core decode kernel): the tile's (query, neighbour) pairs numbered per distinct map row -- state-independent,
// so it is resolved once next to the searches instead of in every decode launch's dependent chain (k_search_tiles):
//   [0, 96)   map row id of the tile's row number r (r < count)        [96] count
//   [100, 124) 96 bytes: row number of pair (query q, neighbour k) at byte 6 q + k, 255 = no neighbour
// Only iterations of at most kTileLargeFrom tiles carry number blocks: their decode launch runs one tile per wave and is
// one tile's dependent chain long, so taking the numbering out of it pays (14.5 -> 12.5 us at 16 384 samples); larger
// launches number in place (the SIMD's other waves hide it; numbering in the search launch cost 65 536 / 262 144-sample
// iterations +6 / +24 us there for -1 / -8 us in the decode).
constexpr int kTileNumWords = 128, kTileNumCount = 96, kTileNumBytes = 100;
#ifndef CLID_TILE_LARGE_FROM
#define CLID_TILE_LARGE_FROM 2048
#endif
constexpr int kTileLargeFrom = CLID_TILE_LARGE_FROM;  // tiles
__host__ __device__ inline bool tiles_prenumbered(int n_tasks) { return (n_tasks + 1) / 2 <= kTileLargeFrom; }
// ... and behind those the iteration's DEFERRED flags (one 32-bit word per task; per tile when the iteration is pre-numbered):
// 1 = the cell-directory search launch leaves the unit to the probing kernels -- a query point outside the directory's box
// (csrc/train.hip search_task); the launch that follows gathers the set flags.  Every unit writes its word: nothing to reset.
__host__ __device__ inline size_t rec_deferred_offset(int n_tasks) {
  return (size_t)n_tasks * kRecFloatsPerTask + (tiles_prenumbered(n_tasks) ? (size_t)((n_tasks + 1) / 2) * kTileNumWords : 0);
}
__host__ __device__ inline size_t rec_floats_per_iter(int n_tasks) {
  return rec_deferred_offset(n_tasks) + (((size_t)n_tasks + 4 + 3) & ~(size_t)3);
}

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
  const size_t one = max_tasks(Q) * (kRecFloatsPerTask + kTileNumWords / 2);  // (records + number blocks at the task bound)
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

// 16-lane kernels (lane16 = 2 k + half, 4 query points per wave and round) on 16-float accumulation rows: the round's 24
// (query, neighbour) pairs leave through LDS, so that 9 consecutive lanes add one pair's [8 feature gradients | certainty
// increment] = ONE 64-byte request at the memory-side atomic units instead of 8 + 1 (those units retire ~17 G requests/s
// whatever the width, tools/ubench_atomic.hip; 30 requests per sample were 29 of the analytic iteration's 32 us).  The
// certainty increment (np.py:714) rides in column 8 and is merged by k_adam_all.
constexpr int kPairsPerRound = 4 * CLID_K;  // (query, neighbour) pairs of one wave round
constexpr int kPairStride = 12;             // floats per pair in LDS: 8 gradients | certainty increment | pad
struct PairLds {
  float val[kPairsPerRound * kPairStride];
  int row[kPairsPerRound + 4];
};
__device__ __forceinline__ void scatter_pairs_rows16(PairLds& pl, int lane, int grp, int my_k, bool odd, int row, float d0, float d1,
                                                     float d2, float d3, float cert_inc, float* __restrict__ rows16) {
  if (my_k < CLID_K) {
    const int pair = grp * CLID_K + my_k;
    *reinterpret_cast<float4*>(&pl.val[pair * kPairStride + (odd ? 4 : 0)]) = make_float4(d0, d1, d2, d3);
    if (!odd) {
      pl.val[pair * kPairStride + CLID_F] = cert_inc;
      pl.row[pair] = row;  // -1: nothing to add for this pair
    }
  }
  wave_lds_fence();
#pragma unroll
  for (int pass = 0; pass < (kPairsPerRound + 6) / 7; ++pass) {  // 7 pairs x 9 columns per instruction
    const int pi = lane / 9, col = lane - pi * 9, pair = pass * 7 + pi;
    if (lane < 63 && pair < kPairsPerRound) {
      const int r = pl.row[pair];
      const float val = pl.val[pair * kPairStride + col];
      if (r >= 0) atomicAdd(rows16 + (size_t)r * CLID_GRAD_ROW16 + col, val);
    }
  }
  wave_lds_fence();
}

}  // namespace clid

// ---- launches under the optional per-kernel timing (clid_train_args.prof, train.hip) ------------------------------------
// With a profiler object the kernel goes out through hipExtLaunchKernelGGL with a start and a stop event that the object
// keeps: they hold the DISPATCH's begin / end time stamps (the same clock rocprofv3 --kernel-trace reads).
// tag: 0 fused / decode kernel, 1 search kernel, 2 partial reduce (+ pack), 3 adam, 5 touched-row scan
#include <hip/hip_ext.h>
bool clid_prof_open(struct clid_prof* prof, int tag, hipEvent_t* a, hipEvent_t* b);
#define CLID_KLAUNCH(prof, tag, kernel, grid, block, shmem, stream, ...)                             \
  do {                                                                                               \
    hipEvent_t ea__, eb__;                                                                           \
    if (clid_prof_open(prof, tag, &ea__, &eb__))                                                     \
      hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, ea__, eb__, 0, __VA_ARGS__);         \
    else                                                                                             \
      hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                           \
  } while (0)

// ---- touched-row workspace (clid_train_args.touch_ws; include/clid_native.h) -------------------------------------------
namespace clid {
constexpr int kTouchCountSlots = 64;  // >= kMaxChunkIters
__host__ __device__ inline long long touch_stride_of(int M) { return ((long long)M + 1 + 255) & ~255LL; }
struct TouchWs {
  uint8_t* flags;    // [chunk][stride]: set by the search launch, cleared again by the scan
  unsigned* bits;    // [chunk][stride / 32]: rows touched by iteration i of the chunk
  unsigned* cumb;    // [chunk][stride / 32]: rows touched by any iteration of the mapping() call up to and including i
  unsigned* wpre;    // [chunk][stride / 32]: exclusive prefix of popcount(bits) along the row axis
  int* counts;       // [kTouchCountSlots]: rows touched by iteration i
  unsigned* cum;     // [stride / 32]: rows touched by the call's earlier chunks
};
__host__ __device__ inline TouchWs touch_carve(uint8_t* ws, long long stride, int chunk) {
  TouchWs t;
  const size_t W = (size_t)(stride / 32);
  t.flags = ws;
  t.bits = reinterpret_cast<unsigned*>(ws + (size_t)chunk * stride);
  t.cumb = t.bits + (size_t)chunk * W;
  t.wpre = t.cumb + (size_t)chunk * W;
  t.counts = reinterpret_cast<int*>(t.wpre + (size_t)chunk * W);
  t.cum = reinterpret_cast<unsigned*>(t.counts + kTouchCountSlots);
  return t;
}
__host__ inline size_t touch_bytes(long long stride, int chunk) {
  const size_t W = (size_t)(stride / 32);
  return (size_t)chunk * stride + 3 * (size_t)chunk * W * 4 + kTouchCountSlots * 4 + W * 4;
}
}  // namespace clid

// host-side launchers of the tile (matrix-core) decode kernels (train_tile.hip); prec 0 = fp32, 1 = bf16 operands
int clid_launch_decode_tile(const clid_map_view* mv, const clid_train_args* a, float* partial, const clid::TaskMap& tmap,
                            const float* rec, int prec, hipStream_t s);
int clid_decode_tile_blocks(int n_tasks);
bool clid_tiles_prenumbered(int n_tasks, const clid_map_view* mv);  // does the decode launch read the search launch's number blocks?
// host-side launchers of the analytic-eikonal iteration (train_analytic.hip)
int clid_train_wf0_rows(int n_tasks, int n_fd);  // csrc/train_wf0.hip (`weighted_first: False`)
int clid_launch_train_wf0(const clid_map_view* mv, const clid_train_args* a, float* partial, const clid::TaskMap& tmap, float* rec, hipStream_t s);  // rec: the copies' label / weight fields are scratch
int clid_launch_train_analytic(const clid_map_view* mv, const clid_train_args* a, float* partial, const float* rec, hipStream_t s);  // rec: the hoisted search's records, or NULL (search inside)
int clid_train_analytic_blocks(int bs);
