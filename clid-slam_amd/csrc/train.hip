// The fused mapping iteration: Mapper.mapping body (utils/mapper.py:642-836) =
//   get_batch gathers (mapper.py:501-507) -> query_feature (np.py:553-769) -> Decoder.sdf
//   (decoder.py:58-82) -> numerical gradient on every `decimation`-th sample (mapper.py:697-704,
//   985-1034; one more query+decode on the 6 shifted copies) -> BCE + eikonal loss
//   (loss.py:44-62, mapper.py:746-798) -> backward -> Adam (tools.py:205-255).
//
// Launch plan per iteration (numerical-eikonal mode, the shipped default):
//   k_train_fwd   all Q = bs + 6*ceil(bs/decimation) query points: search, blend, decode; saves
//                 sdf, f, (idx,w) per query; certainty / ts side effects
//   k_train_bwd   per query: dL/dsdf (BCE for batch points, central-difference eikonal for the
//                 shifted copies), decoder backward, feature-gradient scatter (atomics), per-block
//                 partials of the 833 decoder gradients and the loss sums
//   k_reduce      partials -> grad[0:833], loss_out
//   (optional RCCL all-reduce of `grad` by the host between these and Adam)
//   k_adam        dense Adam over the features and the decoder, zeroes `grad`
#include "common.hpp"

namespace clid {

constexpr int kPartialStride = 840;  // 833 decoder grads | bce sum | eik sum | pad
constexpr int kMaxBwdBlocks = 1024;

struct TrainWs {
  float* sdf;      // [Q]
  float* fvec;     // [Q][12]
  float* w;        // [Q][K]
  int* idx;        // [Q][K]
  float* partial;  // [kMaxBwdBlocks][kPartialStride]
};

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
  t.sdf = take(Q);
  t.fvec = take((size_t)Q * 12);
  t.w = take((size_t)Q * CLID_K);
  t.idx = reinterpret_cast<int*>(take((size_t)Q * CLID_K));
  t.partial = take((size_t)kMaxBwdBlocks * kPartialStride);
  return t;
}

// query q -> position in the local batch, shifted axis/sign for finite-difference copies
struct QueryId {
  int p;     // index into this rank's batch
  int axis;  // -1 main sample, else 0..2
  float sign;
};
__device__ __forceinline__ QueryId decode_query(int q, int bs, int n_fd, int first, int decim) {
  QueryId r;
  if (q < bs) {
    r.p = q; r.axis = -1; r.sign = 0.f;
  } else {
    const int e = q - bs;
    const int a = e / n_fd;  // 0..5 : +x,-x,+y,-y,+z,-z  (mapper.py:1001)
    r.p = first + (e - a * n_fd) * decim;
    r.axis = a >> 1;
    r.sign = (a & 1) ? -1.f : 1.f;
  }
  return r;
}

__global__ void __launch_bounds__(CLID_BLOCK)
k_train_fwd(clid_map_view mv, clid_train_args ta, TrainWs ws, int Q, int n_fd, int first) {
  __shared__ MlpLds mlp;
  stage_mlp(mlp, ta.W1, ta.b1, ta.W2, ta.b2);
  const int lane = threadIdx.x & 63, lane16 = lane & 15, gbase = lane & 48;
  const int q_raw = blockIdx.x * CLID_QPB + (threadIdx.x >> 4);
  const bool live = q_raw < Q;
  const int q = live ? q_raw : (Q - 1);
  const QueryId id = decode_query(q, ta.bs, n_fd, first, ta.decimation);
  const long long s = ta.index[id.p];
  float px = ta.pool_coord[s * 3 + 0], py = ta.pool_coord[s * 3 + 1], pz = ta.pool_coord[s * 3 + 2];
  if (id.axis == 0) px = fadd(px, id.sign * ta.fd_eps);  // x + [eps,0,0] in fp32 (mapper.py:988-999)
  if (id.axis == 1) py = fadd(py, id.sign * ta.fd_eps);
  if (id.axis == 2) pz = fadd(pz, id.sign * ta.fd_eps);

  TopK t;
  search_topk(mv, px, py, pz, lane16, gbase, t);
  float w[CLID_K], omega[CLID_K];
  idw_weights(t, w, omega);
  const float4* pos4 = reinterpret_cast<const float4*>(mv.pos4);
  float f[CLID_D];
#pragma unroll
  for (int c = 0; c < CLID_D; ++c) f[c] = 0.f;
#pragma unroll
  for (int k = 0; k < CLID_K; ++k) {
    if (t.j[k] >= 0) {
      float fe[CLID_F];
      load_feat(mv.feat, t.j[k], fe);
      if (mv.layer_norm) {
        float rstd;
        layer_norm8(fe, rstd);
      }
      const float4 p = pos4[t.j[k]];
#pragma unroll
      for (int c = 0; c < CLID_F; ++c) f[c] = fadd(f[c], fmul(fe[c], w[k]));
      f[CLID_F + 0] = fadd(f[CLID_F + 0], fmul(fsub(px, p.x), w[k]));
      f[CLID_F + 1] = fadd(f[CLID_F + 1], fmul(fsub(py, p.y), w[k]));
      f[CLID_F + 2] = fadd(f[CLID_F + 2], fmul(fsub(pz, p.z), w[k]));
    }
  }
  float pre[CLID_HPL];
  const float sdf = mlp_forward(mlp, f, lane16, ta.sdf_scale, pre);
  if (!live) return;
  {
    float mine = 0.f;
#pragma unroll
    for (int c = 0; c < CLID_D; ++c) mine = (lane16 == c) ? f[c] : mine;
    if (lane16 < 12) ws.fvec[(size_t)q * 12 + lane16] = mine;
    float mw = 0.f;
    int mj = -1;
#pragma unroll
    for (int k = 0; k < CLID_K; ++k) {
      mw = (lane16 == k) ? w[k] : mw;
      mj = (lane16 == k) ? t.j[k] : mj;
    }
    if (lane16 < CLID_K) {
      ws.w[(size_t)q * CLID_K + lane16] = mw;
      ws.idx[(size_t)q * CLID_K + lane16] = mj;
      if (mj >= 0) {  // training_mode side effects (np.py:708-733)
        atomicAdd(&mv.cert[mj], mw);
        if (id.axis < 0 && mv.ts_update) atomicMax(&mv.ts_update[mj], ta.pool_ts[s]);
      }
    }
    if (lane16 == 0) ws.sdf[q] = sdf;
  }
}

// block-level reduction of the per-lane decoder-gradient accumulators into partial[blockIdx.x]
struct MlpAcc {
  float dW1[CLID_HPL][CLID_D];
  float db1[CLID_HPL];
  float dW2[CLID_HPL];
  float db2;
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int u = 0; u < CLID_HPL; ++u) {
#pragma unroll
      for (int c = 0; c < CLID_D; ++c) dW1[u][c] = 0.f;
      db1[u] = 0.f;
      dW2[u] = 0.f;
    }
    db2 = 0.f;
  }
};

__device__ __forceinline__ void flush_mlp_acc(const MlpAcc& acc, float bce, float eik, float* red /*LDS*/,
                                              float* __restrict__ out /* [kPartialStride] */) {
  // red: [waves][16][56]
  constexpr int kPer = 56;
  const int lane = threadIdx.x & 63, lane16 = lane & 15, wave = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  float* mine = red + ((size_t)wave * CLID_G + lane16) * kPer;
  int n = 0;
#pragma unroll
  for (int u = 0; u < CLID_HPL; ++u) {
#pragma unroll
    for (int c = 0; c < CLID_D; ++c) {
      const float v = cross_group_sum(acc.dW1[u][c]);
      if (lane < CLID_G) mine[n] = v;
      ++n;
    }
  }
#pragma unroll
  for (int u = 0; u < CLID_HPL; ++u) {
    const float v = cross_group_sum(acc.db1[u]);
    if (lane < CLID_G) mine[n] = v;
    ++n;
  }
#pragma unroll
  for (int u = 0; u < CLID_HPL; ++u) {
    const float v = cross_group_sum(acc.dW2[u]);
    if (lane < CLID_G) mine[n] = v;
    ++n;
  }
  {
    const float v = cross_group_sum(acc.db2);  // db2/bce/eik are carried by lane16 == 0 only
    if (lane < CLID_G) mine[n] = v;
    ++n;
    const float vb = cross_group_sum(bce);
    if (lane < CLID_G) mine[n] = vb;
    ++n;
    const float ve = cross_group_sum(eik);
    if (lane < CLID_G) mine[n] = ve;
    ++n;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < CLID_MLP_PARAMS + 2; p += blockDim.x) {
    int l16, slot;
    if (p < CLID_H * CLID_D) {
      const int h = p / CLID_D, c = p - h * CLID_D;
      l16 = h & 15;
      slot = (h >> 4) * CLID_D + c;
    } else if (p < CLID_H * CLID_D + CLID_H) {
      const int h = p - CLID_H * CLID_D;
      l16 = h & 15;
      slot = CLID_HPL * CLID_D + (h >> 4);
    } else if (p < CLID_H * CLID_D + 2 * CLID_H) {
      const int h = p - CLID_H * CLID_D - CLID_H;
      l16 = h & 15;
      slot = CLID_HPL * CLID_D + CLID_HPL + (h >> 4);
    } else {
      l16 = 0;
      slot = CLID_HPL * CLID_D + 2 * CLID_HPL + (p - (CLID_MLP_PARAMS - 1));
    }
    float s = 0.f;
    for (int wv = 0; wv < nw; ++wv) s += red[((size_t)wv * CLID_G + l16) * kPer + slot];
    out[p] = s;
  }
}

// decoder backward for one query given dz = scale * dL/dsdf; returns df (replicated)
__device__ __forceinline__ void mlp_backward(const MlpLds& s, const float (&f)[CLID_D],
                                             const float (&pre)[CLID_HPL], float dz, int lane16,
                                             bool train_decoder, MlpAcc& acc, float (&df)[CLID_D]) {
  float dh[CLID_HPL];
#pragma unroll
  for (int u = 0; u < CLID_HPL; ++u) {
    const int h = lane16 + CLID_G * u;
    const bool on = pre[u] > 0.f;
    dh[u] = on ? dz * s.w[CLID_H * CLID_D + CLID_H + h] : 0.f;
    if (train_decoder) {
      acc.dW2[u] += on ? dz * pre[u] : 0.f;
      acc.db1[u] += dh[u];
#pragma unroll
      for (int c = 0; c < CLID_D; ++c) acc.dW1[u][c] = fmaf(dh[u], f[c], acc.dW1[u][c]);
    }
  }
  if (train_decoder && lane16 == 0) acc.db2 += dz;
#pragma unroll
  for (int c = 0; c < CLID_D; ++c) {
    float part = 0.f;
#pragma unroll
    for (int u = 0; u < CLID_HPL; ++u) part = fmaf(s.w[(lane16 + CLID_G * u) * CLID_D + c], dh[u], part);
    df[c] = group_sum(part);
  }
}

__global__ void __launch_bounds__(CLID_BLOCK)
k_train_bwd(clid_map_view mv, clid_train_args ta, TrainWs ws, int Q, int n_fd, int first, int n_groups_total) {
  __shared__ MlpLds mlp;
  __shared__ float red[(CLID_BLOCK / 64) * CLID_G * 56];
  stage_mlp(mlp, ta.W1, ta.b1, ta.W2, ta.b2);
  const int lane = threadIdx.x & 63, lane16 = lane & 15;
  MlpAcc acc;
  acc.zero();
  float bce_acc = 0.f, eik_acc = 0.f;
  float* g_theta = ta.grad + CLID_GRAD_FEAT_OFFSET;
  const float inv_sigma = fdiv(1.0f, ta.sigma);
  const float two_eps = 2.0f * ta.fd_eps;

  for (int g0 = blockIdx.x * CLID_QPB; g0 < n_groups_total; g0 += gridDim.x * CLID_QPB) {
    const int q_raw = g0 + (threadIdx.x >> 4);
    const bool live = q_raw < Q;
    const int q = live ? q_raw : (Q - 1);
    const QueryId id = decode_query(q, ta.bs, n_fd, first, ta.decimation);
    float delta = 0.f;  // dL/dsdf of this query
    if (id.axis < 0) {
      const long long s = ta.index[id.p];
      const float label = ta.pool_label[s];
      const float wt = ta.loss_weight_on ? fabsf(ta.pool_weight[s]) : 1.0f;  // mapper.py:747-749
      const float z = ws.sdf[q] * inv_sigma;
      const float tgt = 1.0f / (1.0f + expf(-label * inv_sigma));            // loss.py:60
      const float sg = 1.0f / (1.0f + expf(-z));
      const float li = fmaxf(z, 0.f) - z * tgt + log1pf(expf(-fabsf(z)));    // BCEWithLogits
      if (live && lane16 == 0) bce_acc += wt * li;
      delta = wt * (sg - tgt) * inv_sigma * ta.inv_n_main;
    } else {
      const int e = q - ta.bs;
      const int a = e / n_fd, jj = e - a * n_fd;
      float sv[6];
#pragma unroll
      for (int b = 0; b < 6; ++b) sv[b] = ws.sdf[ta.bs + b * n_fd + jj];
      const float gx = fdiv(sv[0] - sv[1], two_eps), gy = fdiv(sv[2] - sv[3], two_eps),
                  gz = fdiv(sv[4] - sv[5], two_eps);                          // mapper.py:1011-1013
      const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
      if (live && lane16 == 0 && a == 0) eik_acc += (nrm - 1.f) * (nrm - 1.f);
      const float gc = (id.axis == 0) ? gx : (id.axis == 1 ? gy : gz);
      // d/dg of weight_e * mean((|g|-1)^2); 0 at |g| == 0 (torch norm subgradient)
      const float dLdg = nrm > 0.f ? ta.weight_e * 2.f * (nrm - 1.f) * ta.inv_n_eik * (gc / nrm) : 0.f;
      delta = id.sign * dLdg / two_eps;
    }
    if (!live) delta = 0.f;

    float f[CLID_D];
#pragma unroll
    for (int c = 0; c < CLID_D; ++c) f[c] = ws.fvec[(size_t)q * 12 + c];
    float pre[CLID_HPL];
    (void)mlp_forward(mlp, f, lane16, ta.sdf_scale, pre);
    float df[CLID_D];
    mlp_backward(mlp, f, pre, ta.sdf_scale * delta, lane16, ta.train_decoder != 0, acc, df);

    // d theta[j_k] += w_k * df[0:F]  (through the layer-norm backward when on)
    if (live && delta != 0.f) {
      if (!mv.layer_norm) {
        float dfc = 0.f;
#pragma unroll
        for (int c = 0; c < CLID_F; ++c) dfc = ((lane16 & 7) == c) ? df[c] : dfc;
#pragma unroll
        for (int r = 0; r < (CLID_K * CLID_F + CLID_G - 1) / CLID_G; ++r) {
          const int e = lane16 + CLID_G * r;
          const int k = e >> 3;
          if (k < CLID_K) {
            const int j = ws.idx[(size_t)q * CLID_K + k];
            if (j >= 0) {
              const float wk = ws.w[(size_t)q * CLID_K + k];
              atomicAdd(&g_theta[(size_t)j * CLID_F + (e & 7)], wk * dfc);
            }
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < CLID_K; ++k) {
          const int j = ws.idx[(size_t)q * CLID_K + k];
          if (j < 0) continue;
          const float wk = ws.w[(size_t)q * CLID_K + k];
          float fe[CLID_F], rstd, dth[CLID_F];
          load_feat(mv.feat, j, fe);
          layer_norm8(fe, rstd);
#pragma unroll
          for (int c = 0; c < CLID_F; ++c) dth[c] = wk * df[c];
          layer_norm8_bwd(fe, rstd, dth);
          float mine = 0.f;
#pragma unroll
          for (int c = 0; c < CLID_F; ++c) mine = (lane16 == c) ? dth[c] : mine;
          if (lane16 < CLID_F) atomicAdd(&g_theta[(size_t)j * CLID_F + lane16], mine);
        }
      }
    }
  }
  flush_mlp_acc(acc, bce_acc, eik_acc, red, ws.partial + (size_t)blockIdx.x * kPartialStride);
}

// partial[nb][840] -> grad[0:833] (=), loss_out[0..2] (+=)
__global__ void k_reduce_partials(const float* __restrict__ partial, int nb, float* __restrict__ grad,
                                  float* __restrict__ loss_out, float inv_n_main, float inv_n_eik,
                                  float weight_e, int train_decoder) {
  __shared__ float sm[4][64];
  const int px = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + px;
  float s = 0.f;
  if (p < CLID_MLP_PARAMS + 2)
    for (int b = py; b < nb; b += 4) s += partial[(size_t)b * kPartialStride + p];
  sm[py][px] = s;
  __syncthreads();
  if (py == 0 && p < CLID_MLP_PARAMS + 2) {
    const float tot = sm[0][px] + sm[1][px] + sm[2][px] + sm[3][px];
    if (p < CLID_MLP_PARAMS) {
      if (train_decoder) grad[p] = tot;
    } else if (p == CLID_MLP_PARAMS) {
      const float bce = tot * inv_n_main;
      atomicAdd(&loss_out[1], bce);
      atomicAdd(&loss_out[0], bce);
    } else {
      const float eik = tot * inv_n_eik;
      atomicAdd(&loss_out[2], eik);
      atomicAdd(&loss_out[0], weight_e * eik);
    }
  }
}

// torch.optim.Adam._single_tensor_adam, op order as ATen's:
//   m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, 1-b2); denom = sqrt(v)/sqrt(bc2) + eps;
//   p.addcdiv_(m, denom, -lr/bc1)
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float one_m_b1, float b2,
                                            float one_m_b2, float bc2_sqrt, float eps, float neg_step,
                                            float wd) {
  if (wd != 0.f) g = fmaf(wd, p, g);
  m = fadd(m, fmul(one_m_b1, fsub(g, m)));
  v = fadd(fmul(v, b2), fmul(fmul(one_m_b2, g), g));
  const float denom = fadd(fdiv(sqrtf(v), bc2_sqrt), eps);
  p = fadd(p, fdiv(fmul(neg_step, m), denom));
}

__global__ void k_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, long long n, float one_m_b1, float b2, float one_m_b2,
                       float bc2_sqrt, float eps, float neg_step, float wd, int zero_grad) {
  const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 + 3 < n) {
    float4 P = *reinterpret_cast<float4*>(p + i4), G = *reinterpret_cast<float4*>(g + i4);
    float4 M = *reinterpret_cast<float4*>(m + i4), V = *reinterpret_cast<float4*>(v + i4);
    adam_update(P.x, G.x, M.x, V.x, one_m_b1, b2, one_m_b2, bc2_sqrt, eps, neg_step, wd);
    adam_update(P.y, G.y, M.y, V.y, one_m_b1, b2, one_m_b2, bc2_sqrt, eps, neg_step, wd);
    adam_update(P.z, G.z, M.z, V.z, one_m_b1, b2, one_m_b2, bc2_sqrt, eps, neg_step, wd);
    adam_update(P.w, G.w, M.w, V.w, one_m_b1, b2, one_m_b2, bc2_sqrt, eps, neg_step, wd);
    *reinterpret_cast<float4*>(p + i4) = P;
    *reinterpret_cast<float4*>(m + i4) = M;
    *reinterpret_cast<float4*>(v + i4) = V;
    if (zero_grad) *reinterpret_cast<float4*>(g + i4) = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    for (long long i = i4; i < n; ++i) {
      float P = p[i], M = m[i], V = v[i];
      adam_update(P, g[i], M, V, one_m_b1, b2, one_m_b2, bc2_sqrt, eps, neg_step, wd);
      p[i] = P; m[i] = M; v[i] = V;
      if (zero_grad) g[i] = 0.f;
    }
  }
}

// the 833 decoder parameters live in four separate tensors (nn.Linear weights/biases)
__global__ void k_adam_mlp(float* W1, float* b1, float* W2, float* b2, float* __restrict__ g,
                           float* __restrict__ m, float* __restrict__ v, float one_m_b1, float b2c,
                           float one_m_b2, float bc2_sqrt, float eps, float neg_step) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= CLID_MLP_PARAMS) return;
  float* dst;
  if (i < CLID_H * CLID_D) dst = W1 + i;
  else if (i < CLID_H * CLID_D + CLID_H) dst = b1 + (i - CLID_H * CLID_D);
  else if (i < CLID_H * CLID_D + 2 * CLID_H) dst = W2 + (i - CLID_H * CLID_D - CLID_H);
  else dst = b2;
  float P = *dst, M = m[i], V = v[i];
  adam_update(P, g[i], M, V, one_m_b1, b2c, one_m_b2, bc2_sqrt, eps, neg_step, 0.f);
  *dst = P; m[i] = M; v[i] = V;
  g[i] = 0.f;
}

}  // namespace clid

using namespace clid;

static int n_queries(const clid_train_args* a, int* n_fd, int* first) {
  *first = fd_first(a->batch_offset, a->decimation);
  *n_fd = (a->eikonal_mode == 1) ? fd_count(a->bs, a->batch_offset, a->decimation) : 0;
  return a->bs + 6 * (*n_fd);
}

extern "C" int64_t clid_train_workspace_floats(int32_t bs, int32_t decimation, int32_t eikonal_mode) {
  if (bs <= 0 || decimation <= 0) return -1;
  const long long nfd = eikonal_mode == 1 ? (bs + decimation - 1) / decimation : 0;
  const long long Q = bs + 6 * nfd;
  return (Q + 4) + (Q * 12 + 4) + 2 * (Q * CLID_K + 4) + (long long)kMaxBwdBlocks * kPartialStride + 64;
}

extern "C" int clid_train_fwd_bwd(const clid_map_view* mv, const clid_train_args* a, void* stream) {
  if (!mv || !a || !mv->tab || !mv->feat || !mv->cert || !a->index || !a->grad || !a->ws || !a->loss_out) {
    clid_set_error("clid_train_fwd_bwd: null argument");
    return CLID_E_ARG;
  }
  if (a->bs <= 0 || a->decimation <= 0) {
    clid_set_error("clid_train_fwd_bwd: bs=%d decimation=%d", a->bs, a->decimation);
    return CLID_E_ARG;
  }
  if (a->eikonal_mode == 2) {
    clid_set_error("clid_train_fwd_bwd: analytic eikonal mode is served by clid_train_fwd_bwd_analytic");
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  int n_fd, first;
  const int Q = n_queries(a, &n_fd, &first);
  TrainWs ws = carve(a->ws, Q);
  const int n_groups = (Q + CLID_QPB - 1) / CLID_QPB * CLID_QPB;  // padded to whole blocks
  hipLaunchKernelGGL(k_train_fwd, dim3(n_groups / CLID_QPB), dim3(CLID_BLOCK), 0, s, *mv, *a, ws, Q,
                     n_fd > 0 ? n_fd : 1, first);
  CLID_CHECK_LAUNCH();
  int nb = n_groups / CLID_QPB;
  if (nb > kMaxBwdBlocks) nb = kMaxBwdBlocks;
  hipLaunchKernelGGL(k_train_bwd, dim3(nb), dim3(CLID_BLOCK), 0, s, *mv, *a, ws, Q, n_fd > 0 ? n_fd : 1,
                     first, n_groups);
  CLID_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_reduce_partials, dim3((CLID_MLP_PARAMS + 2 + 63) / 64), dim3(256), 0, s, ws.partial,
                     nb, a->grad, a->loss_out, a->inv_n_main, a->inv_n_eik,
                     (a->eikonal_mode && n_fd > 0) ? a->weight_e : 0.f, a->train_decoder);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

static void adam_scalars(float lr, float b1, float b2, int step, float* neg_step, float* bc2_sqrt) {
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  *neg_step = (float)(-((double)lr / bc1));
  *bc2_sqrt = (float)sqrt(bc2);
}

extern "C" int clid_adam_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int32_t step, int32_t zero_grad,
                              void* stream) {
  if (!p || !g || !m || !v || n < 0 || step < 1) {
    clid_set_error("clid_adam_step: bad argument");
    return CLID_E_ARG;
  }
  if (n == 0) return CLID_OK;
  float neg_step, bc2_sqrt;
  adam_scalars(lr, beta1, beta2, step, &neg_step, &bc2_sqrt);
  const long long thr = (n + 3) / 4;
  hipLaunchKernelGGL(k_adam, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m,
                     v, (long long)n, (float)(1.0 - (double)beta1), beta2, (float)(1.0 - (double)beta2),
                     bc2_sqrt, eps, neg_step, weight_decay, zero_grad);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

extern "C" int clid_train_adam(const clid_adam_args* a, void* stream) {
  if (!a || !a->feat || !a->grad || !a->m || !a->v || a->step < 1) {
    clid_set_error("clid_train_adam: bad argument");
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  float neg_step, bc2_sqrt;
  adam_scalars(a->lr, a->beta1, a->beta2, a->step, &neg_step, &bc2_sqrt);
  const float one_m_b1 = (float)(1.0 - (double)a->beta1), one_m_b2 = (float)(1.0 - (double)a->beta2);
  if (a->train_decoder) {
    if (!a->W1 || !a->b1 || !a->W2 || !a->b2 || !a->m_mlp || !a->v_mlp) {
      clid_set_error("clid_train_adam: decoder tensors missing");
      return CLID_E_ARG;
    }
    hipLaunchKernelGGL(k_adam_mlp, dim3((CLID_MLP_PARAMS + 255) / 256), dim3(256), 0, s, a->W1, a->b1, a->W2,
                       a->b2, a->grad, a->m_mlp, a->v_mlp, one_m_b1, a->beta2, one_m_b2, bc2_sqrt, a->eps,
                       neg_step);
    CLID_CHECK_LAUNCH();
  }
  const long long thr = (a->n_feat + 3) / 4;
  hipLaunchKernelGGL(k_adam, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, s, a->feat,
                     a->grad + CLID_GRAD_FEAT_OFFSET, a->m, a->v, (long long)a->n_feat, one_m_b1, a->beta2, one_m_b2,
                     bc2_sqrt, a->eps, neg_step, a->weight_decay, 1);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}
