// Neighbour search + feature interpolation kernels (NeuralPoints.radius_neighborhood_search,
// NeuralPoints.query_feature and its autograd backward; model/neural_points.py:553-769, 971-1030)
// and the fused inference kernel query -> Decoder.sdf -> analytic d sdf/d x
// (utils/tools.py:298-311 as used by utils/error_state_iekf.py:209-227).
#include <string.h>

#include "common.hpp"

namespace clid {

// ---- a2: raw probe results in neighbour-offset order ------------------------------------------------
__global__ void k_radius_search(clid_map_view mv, const float* __restrict__ x, int N,
                                float* __restrict__ dist2_out, int* __restrict__ idx_out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)N * mv.P) return;
  const int n = (int)(t / mv.P), o = (int)(t % mv.P);
  const float px = x[n * 3 + 0], py = x[n * 3 + 1], pz = x[n * 3 + 2];
  int slot = base_slot(px, py, pz, mv.resolution, mv.buffer_size) + mv.delta[o];
  if (slot >= mv.buffer_size) slot -= mv.buffer_size;
  const int4* tab = reinterpret_cast<const int4*>(mv.tab);
  const unsigned home = tab_home(slot, mv.log2cap);
  const int cell = tab_find(tab, mv.log2cap, slot, home, tab[home]);
  int j = -1;
  float d2 = mv.max_valid_dist2;  // np.py:1013
  if (cell >= 0) {
    const float4 p = reinterpret_cast<const float4*>(mv.tab_pos)[cell];
    j = __float_as_int(p.w);
    const float ax = fsub(p.x, px), ay = fsub(p.y, py), az = fsub(p.z, pz);
    d2 = fadd(fadd(fmul(ax, ax), fmul(ay, ay)), fmul(az, az));
    if (d2 > mv.max_valid_dist2) j = -1;  // np.py:1016-1020 (dist2 keeps its value)
  }
  dist2_out[t] = d2;
  idx_out[t] = j;
}

// ---- query_certainty on the GLOBAL map straight from the reference's own table (model/neural_points.py:1032-1051) ----
// buffer_pt_index[hash] -> point -> collision test -> certainty, max over the P probe cells.  Used once per frame on the
// new samples with a 1-cell neighbourhood (utils/mapper.py:409-423): one 8-byte probe per sample instead of building a
// compact mirror of the whole global map for a single pass.
__global__ void __launch_bounds__(256)
k_query_certainty_direct(const long long* __restrict__ table, int buffer_size, const float* __restrict__ points,
                         const float* __restrict__ cert, const int* __restrict__ delta, int P, float resolution,
                         float max_valid_dist2, const float* __restrict__ x, int N, float* __restrict__ out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float px = x[n * 3 + 0], py = x[n * 3 + 1], pz = x[n * 3 + 2];
  const int r0 = base_slot(px, py, pz, resolution, buffer_size);
  float best = 0.f;  // certainties are >= 0; cells without a valid point contribute 0 (np.py:1043-1049)
  for (int o = 0; o < P; ++o) {
    int slot = r0 + delta[o];
    if (slot >= buffer_size) slot -= buffer_size;
    const long long j = table[slot];
    if (j < 0) continue;
    const float ax = fsub(points[j * 3 + 0], px), ay = fsub(points[j * 3 + 1], py), az = fsub(points[j * 3 + 2], pz);
    const float d2 = fadd(fadd(fmul(ax, ax), fmul(ay, ay)), fmul(az, az));
    if (d2 > max_valid_dist2) continue;  // hash collision: a foreign point (np.py:1016-1020)
    best = fmaxf(best, cert[j]);
  }
  out[n] = best;
}

// ---- a3: query_feature forward ------------------------------------------------------------------------
// One 16-lane group per query.  Writes f (weighted) or v_k (per neighbour), w, idx, nn, certainty.
__global__ void __launch_bounds__(CLID_BLOCK)
k_query_fwd(clid_map_view mv, const float* __restrict__ x, int N, int weighted_first,
            float* __restrict__ feat_out, float* __restrict__ w_out, int* __restrict__ idx_out,
            int* __restrict__ nn_out, float* __restrict__ cert_out) {
  __shared__ SearchLds dl;
  stage_delta(dl, mv);
  __syncthreads();
  const int lane = threadIdx.x & 63, lane16 = lane & 15, gbase = lane & 48;
  const int q_raw = blockIdx.x * CLID_QPB + (threadIdx.x >> 4);
  const bool live = q_raw < N;
  const int q = live ? q_raw : (N - 1);
  const float px = x[q * 3 + 0], py = x[q * 3 + 1], pz = x[q * 3 + 2];
  TopK t;
  search_topk(mv, dl, px, py, pz, lane16, gbase, t);
  float w[CLID_K], omega[CLID_K];
  idw_weights(t, w, omega);
  const float4* pos4 = reinterpret_cast<const float4*>(mv.pos4);
  float f[CLID_D];
#pragma unroll
  for (int c = 0; c < CLID_D; ++c) f[c] = 0.f;
  float cert = 0.f;
#pragma unroll
  for (int k = 0; k < CLID_K; ++k) {
    float v[CLID_D];
#pragma unroll
    for (int c = 0; c < CLID_D; ++c) v[c] = 0.f;
    if (t.j[k] >= 0) {
      float fe[CLID_F];
      load_feat(mv.feat, t.j[k], fe);
      if (mv.layer_norm) {
        float rstd;
        layer_norm8(fe, rstd);
      }
      const float4 p = pos4[t.j[k]];
#pragma unroll
      for (int c = 0; c < CLID_F; ++c) v[c] = fe[c];
      v[CLID_F + 0] = fsub(px, p.x);
      v[CLID_F + 1] = fsub(py, p.y);
      v[CLID_F + 2] = fsub(pz, p.z);
      cert = fadd(cert, fmul(mv.cert[t.j[k]], w[k]));
    } else if (mv.layer_norm) {
      // F.layer_norm of an all-zero row is all zero (np.py:632-633): nothing to do
    }
    if (weighted_first) {
#pragma unroll
      for (int c = 0; c < CLID_D; ++c) f[c] = fadd(f[c], fmul(v[c], w[k]));
    } else if (live) {
      // [N][K][D]: lane c writes component c
      float mine = 0.f;
#pragma unroll
      for (int c = 0; c < CLID_D; ++c) mine = (lane16 == c) ? v[c] : mine;
      if (lane16 < CLID_D) feat_out[((size_t)q * CLID_K + k) * CLID_D + lane16] = mine;
    }
  }
  if (!live) return;
  if (weighted_first) {
    float mine = 0.f;
#pragma unroll
    for (int c = 0; c < CLID_D; ++c) mine = (lane16 == c) ? f[c] : mine;
    if (lane16 < CLID_D) feat_out[(size_t)q * CLID_D + lane16] = mine;
  }
  {
    float mw = 0.f;
    int mj = -1;
#pragma unroll
    for (int k = 0; k < CLID_K; ++k) {
      mw = (lane16 == k) ? w[k] : mw;
      mj = (lane16 == k) ? t.j[k] : mj;
    }
    if (lane16 < CLID_K) {
      w_out[(size_t)q * CLID_K + lane16] = mw;
      idx_out[(size_t)q * CLID_K + lane16] = mj;
    }
  }
  if (lane16 == 0) {
    nn_out[q] = t.nn;
    cert_out[q] = cert;
  }
}

// training-mode side effects (np.py:708-733), run AFTER k_query_fwd so the certainties it returned
// are the pre-update ones, as in the reference (gather at np.py:654 precedes scatter_add_ at :714).
__global__ void k_query_side_effects(clid_map_view mv, const int* __restrict__ idx,
                                     const float* __restrict__ w, const int* __restrict__ query_ts,
                                     int NK) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= NK) return;
  const int j = idx[t];
  if (j < 0) return;
  atomicAdd(&mv.cert[j], w[t]);
  if (query_ts && mv.ts_update) atomicMax(&mv.ts_update[j], query_ts[t / CLID_K]);
}

// ---- autograd backward of query_feature ------------------------------------------------------------------
// Per neighbour: ge_k = dL/d v_k (D values) and gs_k = dL/d w_k (scalar), then
//   d theta[j_k] += LN^T ge_k[0:F];   d x = sum_k valid ge_k[F:F+3] + sum_k gs_k * dw_k/dx,
//   dw_k/dx = w_k (abar - alpha_k), alpha_k = 2 r_k omega_k  
__global__ void __launch_bounds__(CLID_BLOCK)
k_query_bwd(clid_map_view mv, const float* __restrict__ x, const int* __restrict__ idx,
            const float* __restrict__ w_in, int N, int weighted_first,
            const float* __restrict__ g_feat, const float* __restrict__ g_w,
            float* __restrict__ g_theta, float* __restrict__ g_x) {
  const int lane16 = threadIdx.x & 15;
  const int q = blockIdx.x * CLID_QPB + (threadIdx.x >> 4);
  if (q >= N) return;  // whole groups exit together; no cross-group ops below
  const float px = x[q * 3 + 0], py = x[q * 3 + 1], pz = x[q * 3 + 2];
  const float4* pos4 = reinterpret_cast<const float4*>(mv.pos4);
  float gf[CLID_D];
  if (weighted_first) {
#pragma unroll
    for (int c = 0; c < CLID_D; ++c) gf[c] = g_feat[(size_t)q * CLID_D + c];
  }
  float gx = 0.f, gy = 0.f, gz = 0.f;
  float abx = 0.f, aby = 0.f, abz = 0.f;  // abar
  float gs[CLID_K], wk[CLID_K], alx[CLID_K], aly[CLID_K], alz[CLID_K];
  float wsum_valid = 0.f;
#pragma unroll
  for (int k = 0; k < CLID_K; ++k) {
    const int j = idx[(size_t)q * CLID_K + k];
    wk[k] = w_in[(size_t)q * CLID_K + k];
    gs[k] = g_w ? g_w[(size_t)q * CLID_K + k] : 0.f;
    alx[k] = aly[k] = alz[k] = 0.f;
    if (j < 0) continue;
    const float4 p = pos4[j];
    const float rx = fsub(px, p.x), ry = fsub(py, p.y), rz = fsub(pz, p.z);
    const float d2 = fadd(fadd(fmul(rx, rx), fmul(ry, ry)), fmul(rz, rz));
    const float om = fdiv(1.0f, fadd(d2, 1e-15f));
    alx[k] = 2.f * rx * om; aly[k] = 2.f * ry * om; alz[k] = 2.f * rz * om;
    abx += wk[k] * alx[k]; aby += wk[k] * aly[k]; abz += wk[k] * alz[k];
    float fe[CLID_F], rstd = 1.f;
    const bool need_feat = weighted_first || mv.layer_norm;
    if (need_feat) {
      load_feat(mv.feat, j, fe);
      if (mv.layer_norm) layer_norm8(fe, rstd);
    }
    float ge[CLID_D];
    if (weighted_first) {
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < CLID_F; ++c) dot += gf[c] * fe[c];
      dot += gf[CLID_F] * rx + gf[CLID_F + 1] * ry + gf[CLID_F + 2] * rz;
      gs[k] += dot;
#pragma unroll
      for (int c = 0; c < CLID_D; ++c) ge[c] = wk[k] * gf[c];
    } else {
#pragma unroll
      for (int c = 0; c < CLID_D; ++c) ge[c] = g_feat[((size_t)q * CLID_K + k) * CLID_D + c];
    }
    gx += ge[CLID_F]; gy += ge[CLID_F + 1]; gz += ge[CLID_F + 2];
    if (g_theta) {
      float dth[CLID_F];
#pragma unroll
      for (int c = 0; c < CLID_F; ++c) dth[c] = ge[c];
      if (mv.layer_norm) layer_norm8_bwd(fe, rstd, dth);
      float mine = 0.f;
#pragma unroll
      for (int c = 0; c < CLID_F; ++c) mine = (lane16 == c) ? dth[c] : mine;
      if (lane16 < CLID_F) atomicAdd(&g_theta[(size_t)j * CLID_F + lane16], mine);
    }
    wsum_valid += wk[k];
  }
  (void)wsum_valid;
  if (g_x && lane16 == 0) {
#pragma unroll
    for (int k = 0; k < CLID_K; ++k) {
      const float c = gs[k] * wk[k];
      gx += c * (abx - alx[k]); gy += c * (aby - aly[k]); gz += c * (abz - alz[k]);
    }
    g_x[q * 3 + 0] = gx; g_x[q * 3 + 1] = gy; g_x[q * 3 + 2] = gz;
  }
}

// ---- fused inference: sdf + analytic gradient ------------------------------------------------------------
struct PointEval {  // replicated across the 16 lanes of the group
  float sdf, gx, gy, gz, cert;
  int nn;
};
// query (training_mode=False) -> Decoder.sdf -> analytic d sdf / d x of ONE point per 16-lane group
__device__ __forceinline__ PointEval eval_point(const clid_map_view& mv, const MlpLds& mlp, SearchLds& dl,
                                                float scale, float px, float py, float pz, int lane16, int gbase) {
  TopK t;
  search_topk(mv, dl, px, py, pz, lane16, gbase, t);
  float w[CLID_K], omega[CLID_K];
  idw_weights(t, w, omega);
  const float4* pos4 = reinterpret_cast<const float4*>(mv.pos4);
  float f[CLID_D];
#pragma unroll
  for (int c = 0; c < CLID_D; ++c) f[c] = 0.f;
  float fe[CLID_K][CLID_F];
  float rx[CLID_K], ry[CLID_K], rz[CLID_K];
  float cert = 0.f, wsum = 0.f;
  float abx = 0.f, aby = 0.f, abz = 0.f;
#pragma unroll
  for (int k = 0; k < CLID_K; ++k) {
    rx[k] = ry[k] = rz[k] = 0.f;
#pragma unroll
    for (int c = 0; c < CLID_F; ++c) fe[k][c] = 0.f;
    if (t.j[k] >= 0) {
      load_feat(mv.feat, t.j[k], fe[k]);
      if (mv.layer_norm) {
        float rstd;
        layer_norm8(fe[k], rstd);
      }
      const float4 p = pos4[t.j[k]];
      rx[k] = fsub(px, p.x); ry[k] = fsub(py, p.y); rz[k] = fsub(pz, p.z);
      cert = fadd(cert, fmul(mv.cert[t.j[k]], w[k]));
      wsum += w[k];
      abx += w[k] * 2.f * rx[k] * omega[k];
      aby += w[k] * 2.f * ry[k] * omega[k];
      abz += w[k] * 2.f * rz[k] * omega[k];
    }
#pragma unroll
    for (int c = 0; c < CLID_F; ++c) f[c] = fadd(f[c], fmul(fe[k][c], w[k]));
    f[CLID_F + 0] = fadd(f[CLID_F + 0], fmul(rx[k], w[k]));
    f[CLID_F + 1] = fadd(f[CLID_F + 1], fmul(ry[k], w[k]));
    f[CLID_F + 2] = fadd(f[CLID_F + 2], fmul(rz[k], w[k]));
  }
  float pre[CLID_HPL];
  PointEval r;
  r.sdf = mlp_forward(mlp, f, lane16, scale, pre);
  // u = scale * (W2 .* act) W1   (D values, replicated after the group reduction)
  float u[CLID_D];
#pragma unroll
  for (int c = 0; c < CLID_D; ++c) {
    float part = 0.f;
#pragma unroll
    for (int uu = 0; uu < CLID_HPL; ++uu) {
      const int h = lane16 + CLID_G * uu;
      const float a = pre[uu] > 0.f ? mlp.w[CLID_H * CLID_D + CLID_H + h] : 0.f;
      part = fmaf(a, mlp.w[h * CLID_D + c], part);
    }
    u[c] = scale * group_sum(part);
  }
  // g = sum_k (u . v_k) dw_k/dx + (sum_k w_k) u[F:F+3]
  float gx = wsum * u[CLID_F], gy = wsum * u[CLID_F + 1], gz = wsum * u[CLID_F + 2];
#pragma unroll
  for (int k = 0; k < CLID_K; ++k) {
    float dot = u[CLID_F] * rx[k] + u[CLID_F + 1] * ry[k] + u[CLID_F + 2] * rz[k];
#pragma unroll
    for (int c = 0; c < CLID_F; ++c) dot += u[c] * fe[k][c];
    const float cw = dot * w[k];
    gx += cw * (abx - 2.f * rx[k] * omega[k]);
    gy += cw * (aby - 2.f * ry[k] * omega[k]);
    gz += cw * (abz - 2.f * rz[k] * omega[k]);
  }
  r.gx = gx; r.gy = gy; r.gz = gz;
  r.cert = cert;
  r.nn = t.nn;
  return r;
}

// weighted_first = False (utils/mapper.py:107-112, utils/error_state_iekf.py:217-225, utils/mesher.py:130-138): every
// neighbour's own decoder input [feat_k | x - p_k] is decoded, the K SDFs are blended with the IDW weights:
//   sdf = sum_k w_k sdf_k,  std = sqrt(sum_k w_k (sdf_k - sdf)^2)
//   d sdf / d x = sum_k [ w_k d sdf_k / d x + sdf_k d w_k / d x ],  d sdf_k / d x = u_k[F:F+3] (the input's last 3 columns are
//   x - p_k), u_k = scale (W2 .* act_k) W1,  d w_k / d x = w_k (abar - alpha_k) with alpha_k = 2 r_k omega_k.
// Six decoder evaluations per point instead of one; no shipped config uses it.
__device__ __forceinline__ PointEval eval_point_nf(const clid_map_view& mv, const MlpLds& mlp, SearchLds& dl,
                                                   float scale, float px, float py, float pz, int lane16, int gbase,
                                                   float* sdf_std) {
  TopK t;
  search_topk(mv, dl, px, py, pz, lane16, gbase, t);
  float w[CLID_K], omega[CLID_K];
  idw_weights(t, w, omega);
  const float4* pos4 = reinterpret_cast<const float4*>(mv.pos4);
  float sk[CLID_K], ux[CLID_K], uy[CLID_K], uz[CLID_K], rx[CLID_K], ry[CLID_K], rz[CLID_K];
  float cert = 0.f, abx = 0.f, aby = 0.f, abz = 0.f, mean = 0.f;
#pragma unroll
  for (int k = 0; k < CLID_K; ++k) {
    sk[k] = ux[k] = uy[k] = uz[k] = rx[k] = ry[k] = rz[k] = 0.f;
    if (t.j[k] >= 0) {  // (group-uniform: t is replicated across the 16 lanes)
      float f[CLID_D];
      {
        float fe[CLID_F];
        load_feat(mv.feat, t.j[k], fe);
        if (mv.layer_norm) {
          float rstd;
          layer_norm8(fe, rstd);
        }
#pragma unroll
        for (int c = 0; c < CLID_F; ++c) f[c] = fe[c];
      }
      const float4 p = pos4[t.j[k]];
      rx[k] = fsub(px, p.x); ry[k] = fsub(py, p.y); rz[k] = fsub(pz, p.z);
      f[CLID_F] = rx[k]; f[CLID_F + 1] = ry[k]; f[CLID_F + 2] = rz[k];
      float pre[CLID_HPL];
      sk[k] = mlp_forward(mlp, f, lane16, scale, pre);
      float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
      for (int uu = 0; uu < CLID_HPL; ++uu) {
        const int h = lane16 + CLID_G * uu;
        const float a = pre[uu] > 0.f ? mlp.w[CLID_H * CLID_D + CLID_H + h] : 0.f;
        p0 = fmaf(a, mlp.w[h * CLID_D + CLID_F], p0);
        p1 = fmaf(a, mlp.w[h * CLID_D + CLID_F + 1], p1);
        p2 = fmaf(a, mlp.w[h * CLID_D + CLID_F + 2], p2);
      }
      ux[k] = scale * group_sum(p0); uy[k] = scale * group_sum(p1); uz[k] = scale * group_sum(p2);
      cert = fadd(cert, fmul(mv.cert[t.j[k]], w[k]));
      abx += w[k] * 2.f * rx[k] * omega[k];
      aby += w[k] * 2.f * ry[k] * omega[k];
      abz += w[k] * 2.f * rz[k] * omega[k];
      mean = fadd(mean, fmul(sk[k], w[k]));
    }
  }
  float var = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
  for (int k = 0; k < CLID_K; ++k) {
    const float d = sk[k] - mean;
    var += w[k] * d * d;
    gx += w[k] * (ux[k] + sk[k] * (abx - 2.f * rx[k] * omega[k]));
    gy += w[k] * (uy[k] + sk[k] * (aby - 2.f * ry[k] * omega[k]));
    gz += w[k] * (uz[k] + sk[k] * (abz - 2.f * rz[k] * omega[k]));
  }
  PointEval r;
  r.sdf = mean; r.gx = gx; r.gy = gy; r.gz = gz; r.cert = cert; r.nn = t.nn;
  if (sdf_std) *sdf_std = sqrtf(var);
  return r;
}

__global__ void __launch_bounds__(CLID_BLOCK)
k_sdf_grad_x(clid_map_view mv, const float* W1, const float* b1, const float* W2, const float* b2,
             float scale, const float* __restrict__ x, int N, float* __restrict__ sdf_out,
             float* __restrict__ grad_out, int* __restrict__ nn_out, float* __restrict__ cert_out) {
  __shared__ MlpLds mlp;
  __shared__ SearchLds dl;
  stage_mlp_and_delta(mlp, dl, mv, W1, b1, W2, b2);
  const int lane = threadIdx.x & 63, lane16 = lane & 15, gbase = lane & 48;
  const int q_raw = blockIdx.x * CLID_QPB + (threadIdx.x >> 4);
  const bool live = q_raw < N;
  const int q = live ? q_raw : (N - 1);
  const PointEval r = mv.weighted_first ? eval_point(mv, mlp, dl, scale, x[q * 3 + 0], x[q * 3 + 1], x[q * 3 + 2], lane16, gbase)
                                        : eval_point_nf(mv, mlp, dl, scale, x[q * 3 + 0], x[q * 3 + 1], x[q * 3 + 2], lane16, gbase, nullptr);
  if (live && lane16 == 0) {
    sdf_out[q] = r.sdf;
    grad_out[q * 3 + 0] = r.gx; grad_out[q * 3 + 1] = r.gy; grad_out[q * 3 + 2] = r.gz;
    if (nn_out) nn_out[q] = r.nn;
    if (cert_out) cert_out[q] = r.cert;
  }
}

// ---- dense inference for meshing ------------------------------------------------
// Mesher.query_points (utils/mesher.py:38-163), SDF part: query_feature(training_mode=False) -> sdf where at
// least one neighbour exists (else 0, :122-128), plus the neighbour count for the marching-cubes mask (:156-161).
__global__ void __launch_bounds__(CLID_BLOCK)
k_sdf_query(clid_map_view mv, const float* W1, const float* b1, const float* W2, const float* b2, float scale,
            const float* __restrict__ x, int N, float* __restrict__ sdf_out, int* __restrict__ nn_out) {
  __shared__ MlpLds mlp;
  __shared__ SearchLds dl;
  stage_mlp_and_delta(mlp, dl, mv, W1, b1, W2, b2);
  const int lane = threadIdx.x & 63, lane16 = lane & 15, gbase = lane & 48;
  const int my_k = lane16 >> 1;
  const bool odd = lane16 & 1;
  const float4* pos4 = reinterpret_cast<const float4*>(mv.pos4);
  const int n_groups = (N + CLID_QPB - 1) / CLID_QPB * CLID_QPB;
  for (int g0 = blockIdx.x * CLID_QPB; g0 < n_groups; g0 += gridDim.x * CLID_QPB) {
    const int q_raw = g0 + (threadIdx.x >> 4);
    const bool live = q_raw < N;
    const int q = live ? q_raw : (N - 1);
    const float px = x[(size_t)q * 3 + 0], py = x[(size_t)q * 3 + 1], pz = x[(size_t)q * 3 + 2];
    if (!mv.weighted_first) {  // decode every neighbour, blend the SDFs (utils/mesher.py:130-138)
      const PointEval r = eval_point_nf(mv, mlp, dl, scale, px, py, pz, lane16, gbase, nullptr);
      if (live && lane16 == 0) {
        sdf_out[q] = r.nn >= 1 ? r.sdf : 0.f;
        nn_out[q] = r.nn;
      }
      continue;
    }
    TopK t;
    search_topk(mv, dl, px, py, pz, lane16, gbase, t);
    float w[CLID_K], omega[CLID_K];
    idw_weights(t, w, omega);
    int my_j = -1;
    float my_w = 0.f;
#pragma unroll
    for (int k = 0; k < CLID_K; ++k) {
      my_j = (my_k == k) ? t.j[k] : my_j;
      my_w = (my_k == k) ? w[k] : my_w;
    }
    const int jc = my_j >= 0 ? my_j : 0;
    float4 v = reinterpret_cast<const float4*>(mv.feat)[(size_t)jc * 2 + (odd ? 1 : 0)];
    const float4 pj = pos4[jc];
    if (mv.layer_norm) {
      float s1 = (v.x + v.y) + (v.z + v.w);
      s1 += dpp_mov<0xB1>(s1);
      const float mu = s1 * (1.0f / CLID_F);
      v.x -= mu; v.y -= mu; v.z -= mu; v.w -= mu;
      float s2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      s2 += dpp_mov<0xB1>(s2);
      const float rstd = 1.0f / sqrtf(s2 * (1.0f / CLID_F) + 1e-5f);
      v.x *= rstd; v.y *= rstd; v.z *= rstd; v.w *= rstd;
    }
    float f[CLID_D];
    {
      float a0 = v.x * my_w, a1 = v.y * my_w, a2 = v.z * my_w, a3 = v.w * my_w;
      float r0 = fsub(px, pj.x) * my_w, r1 = fsub(py, pj.y) * my_w, r2 = fsub(pz, pj.z) * my_w;
#define CLID_BFLY(x) x += dpp_mov<0x128>(x); x += dpp_mov<0x124>(x); x += dpp_mov<0x122>(x);
      CLID_BFLY(a0) CLID_BFLY(a1) CLID_BFLY(a2) CLID_BFLY(a3) CLID_BFLY(r0) CLID_BFLY(r1) CLID_BFLY(r2)
#undef CLID_BFLY
      const float b0 = dpp_mov<0xB1>(a0), b1v = dpp_mov<0xB1>(a1), b2v = dpp_mov<0xB1>(a2), b3 = dpp_mov<0xB1>(a3);
      f[0] = odd ? b0 : a0; f[1] = odd ? b1v : a1; f[2] = odd ? b2v : a2; f[3] = odd ? b3 : a3;
      f[4] = odd ? a0 : b0; f[5] = odd ? a1 : b1v; f[6] = odd ? a2 : b2v; f[7] = odd ? a3 : b3;
      f[8] = r0; f[9] = r1; f[10] = r2;
    }
    float pre[CLID_HPL];
    const float sdf = mlp_forward(mlp, f, lane16, scale, pre);
    if (live && lane16 == 0) {
      sdf_out[q] = t.nn >= 1 ? sdf : 0.f;
      nn_out[q] = t.nn;
    }
  }
}

// ---- tracking measurement model -----------------------------------------------
// IEKFOM.h_model (utils/error_state_iekf.py:176-264) in one launch: p_map = R p + t (fp32, as transform_torch
// with the fp32 T of :182-186), sdf + analytic gradient at p_map, validity mask (:233-241), Jacobian rows
// H[0:3] = -g^T R [p]x = p x (R^T g), H[3:6] = g (:247-252), weights R_inv = 1000 / (1 + (|g|-1)^2) * 0.4 /
// (0.4 + sdf^2) (:255-259, float64), and the ONLY things update_iterated (:299-305) needs from the N x 18 H:
// S = H^T R_inv H (non-zero 6 x 6 block) and H^T R_inv z, accumulated in float64.
__global__ void __launch_bounds__(CLID_BLOCK)
k_track_model(const float* __restrict__ pc_imu, const float* __restrict__ rot_dev, const float* __restrict__ pos_dev, const float* W1,
              const float* b1, const float* W2, const float* b2, int N, clid_map_view mv, TrackParams tp, float* __restrict__ sdf_out,
              float* __restrict__ grad_out, float* __restrict__ pmap_out, int* __restrict__ valid_out,
              double* __restrict__ normal_eq /* 28 */, double* __restrict__ zero_next = nullptr) {
  // (argument order: the pointers of the wave's first loads -- the scan, the pose, the decoder -- lead the kernel-argument segment
  // and arrive in SGPRs with the wave, -amdgpu-kernarg-preload-count; the by-value structs follow)
  // (clid_track_model_call) block 0 clears the NEXT call's reduction buffer: 16 x 32 doubles, two per thread
  if (zero_next && blockIdx.x == 0)
    for (int i = threadIdx.x; i < CLID_TRACK_COPIES * 32; i += CLID_BLOCK) zero_next[i] = 0.0;
  __shared__ MlpLds mlp;
  __shared__ SearchLds dl;
  __shared__ double red[CLID_QPB][28];
  stage_mlp_and_delta(mlp, dl, mv, W1, b1, W2, b2);
  if (rot_dev) {  // the pose read on the device (uniform loads): no host round trip in front of the launch
#pragma unroll
    for (int i = 0; i < 9; ++i) tp.R[i] = rot_dev[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) tp.t[i] = pos_dev[i];
  }
  const int lane = threadIdx.x & 63, lane16 = lane & 15, gbase = lane & 48, grp_in_block = threadIdx.x >> 4;
  const int q_raw = blockIdx.x * CLID_QPB + grp_in_block;
  const bool live = q_raw < N;
  const int q = live ? q_raw : (N - 1);
  const float ix = pc_imu[q * 3 + 0], iy = pc_imu[q * 3 + 1], iz = pc_imu[q * 3 + 2];
  const float px = tp.R[0] * ix + tp.R[1] * iy + tp.R[2] * iz + tp.t[0];
  const float py = tp.R[3] * ix + tp.R[4] * iy + tp.R[5] * iz + tp.t[1];
  const float pz = tp.R[6] * ix + tp.R[7] * iy + tp.R[8] * iz + tp.t[2];
  float sdf_std = 0.f;  // stays 0 for weighted_first configs (error_state_iekf.py:188)
  const PointEval r = mv.weighted_first ? eval_point(mv, mlp, dl, tp.scale, px, py, pz, lane16, gbase)
                                        : eval_point_nf(mv, mlp, dl, tp.scale, px, py, pz, lane16, gbase, &sdf_std);
  const float gn = sqrtf(r.gx * r.gx + r.gy * r.gy + r.gz * r.gz);
  const bool valid = live && r.nn >= tp.min_nn && gn < tp.max_grad_norm && gn > tp.min_grad_norm && sdf_std < tp.max_sdf_std;
  if (live && lane16 == 0) {
    if (sdf_out) sdf_out[q] = r.sdf;
    if (grad_out) { grad_out[q * 3 + 0] = r.gx; grad_out[q * 3 + 1] = r.gy; grad_out[q * 3 + 2] = r.gz; }
    if (pmap_out) { pmap_out[q * 3 + 0] = px; pmap_out[q * 3 + 1] = py; pmap_out[q * 3 + 2] = pz; }
    if (valid_out) valid_out[q] = valid ? 1 : 0;
  }
  if (!normal_eq) return;
  // h = [p_imu x (R^T g), g] in fp32 (the reference builds it with fp32 bmm's), then float64 accumulation
  const float qx = tp.R[0] * r.gx + tp.R[3] * r.gy + tp.R[6] * r.gz;
  const float qy = tp.R[1] * r.gx + tp.R[4] * r.gy + tp.R[7] * r.gz;
  const float qz = tp.R[2] * r.gx + tp.R[5] * r.gy + tp.R[8] * r.gz;
  double h[6] = {(double)(iy * qz - iz * qy), (double)(iz * qx - ix * qz), (double)(ix * qy - iy * qx),
                 (double)r.gx, (double)r.gy, (double)r.gz};
  const double z = (double)r.sdf, ga = (double)gn - 1.0;
  const double wgt = valid ? (1.0 / (1.0 + ga * ga)) * (0.4 / (0.4 + z * z)) * 1000.0 : 0.0;
  if (lane16 == 0) {
    int n = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b2i = a; b2i < 6; ++b2i) red[grp_in_block][n++] = wgt * h[a] * h[b2i];  // 21 upper-triangle entries
#pragma unroll
    for (int a = 0; a < 6; ++a) red[grp_in_block][n++] = wgt * h[a] * z;               // H^T R_inv z
    red[grp_in_block][27] = valid ? 1.0 : 0.0;
  }
  __syncthreads();
  if (threadIdx.x < 28) {
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < CLID_QPB; ++g) s += red[g][threadIdx.x];
    // CLID_TRACK_COPIES line-separated copies of the 28 sums, block b adds to copy b mod 16: atomics on one cache line retire
    // one after the other (512 blocks x 28 adds on two lines were 8 of the launch's 19 us); the caller adds the copies up
    if (s != 0.0) atomicAdd(&normal_eq[(blockIdx.x % CLID_TRACK_COPIES) * 32 + threadIdx.x], s);
  }
}

// one block behind k_track_model: the 16 line-separated copies added up in a fixed order, the 28 sums written to pinned
// host-mapped memory, then -- behind a system-scope fence -- the epoch the host polls for
__global__ void __launch_bounds__(64) k_track_finish(const double* __restrict__ normal_eq, double* __restrict__ result, double epoch) {
  if (threadIdx.x < 28) {
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < CLID_TRACK_COPIES; ++c) s += normal_eq[c * 32 + threadIdx.x];
    result[threadIdx.x] = s;
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(&result[31], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---- IEKFOM.h_model's outputs from the per-point outputs (clid_track_valid_count / clid_track_rows) -----------------------
__global__ void __launch_bounds__(256) k_track_count(const int* __restrict__ valid, int N, int* __restrict__ block_count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool v = i < N && valid[i] != 0;
  const unsigned long long b = __ballot(v);
  __shared__ int wc[4];
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}
// one block: block counts -> exclusive prefix in place (+ total behind it), total and epoch to the pinned block
__global__ void __launch_bounds__(1024) k_track_scan(int* __restrict__ block_count, int nb, double* __restrict__ result, double epoch) {
  __shared__ int part[1024];
  const int per = (nb + 1023) / 1024, b0 = threadIdx.x * per;
  int s = 0;
  for (int i = 0; i < per; ++i)
    if (b0 + i < nb) s += block_count[b0 + i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // inclusive scan of the per-thread sums
    const int add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  int run = part[threadIdx.x] - s;
  for (int i = 0; i < per; ++i)
    if (b0 + i < nb) {
      const int c = block_count[b0 + i];
      block_count[b0 + i] = run;
      run += c;
    }
  if (threadIdx.x == 1023) {
    block_count[nb] = part[1023];
    result[29] = (double)part[1023];
    __threadfence_system();
    __hip_atomic_store(&result[31], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ void __launch_bounds__(256)
k_track_rows(const float* __restrict__ pc_imu, const float* __restrict__ sdf, const float* __restrict__ grad, const float* __restrict__ pmap,
             const int* __restrict__ valid, int N, TrackParams tp, const float* __restrict__ rot_dev, const int* __restrict__ block_prefix,
             double* __restrict__ z_out, double* __restrict__ H_out, float* __restrict__ vp_out, double* __restrict__ rinv_out) {
  if (rot_dev) {
#pragma unroll
    for (int i = 0; i < 9; ++i) tp.R[i] = rot_dev[i];
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool v = i < N && valid[i] != 0;
  const unsigned long long b = __ballot(v);
  __shared__ int wc[4];
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  int base = block_prefix[blockIdx.x];
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wc[w];
  if (!v) return;
  const long long r = base + __popcll(b & ((1ull << (threadIdx.x & 63)) - 1ull));
  const float gx = grad[i * 3 + 0], gy = grad[i * 3 + 1], gz = grad[i * 3 + 2];
  const float ix = pc_imu[i * 3 + 0], iy = pc_imu[i * 3 + 1], iz = pc_imu[i * 3 + 2];
  // q = R^T g, h = [p_imu x q | g] in fp32 (utils/error_state_iekf.py:246-251 builds them with fp32 bmm's), stored as float64
  const float qx = tp.R[0] * gx + tp.R[3] * gy + tp.R[6] * gz;
  const float qy = tp.R[1] * gx + tp.R[4] * gy + tp.R[7] * gz;
  const float qz = tp.R[2] * gx + tp.R[5] * gy + tp.R[8] * gz;
  double* H = H_out + r * 18;
  H[0] = (double)(iy * qz - iz * qy);
  H[1] = (double)(iz * qx - ix * qz);
  H[2] = (double)(ix * qy - iy * qx);
  H[3] = (double)gx;
  H[4] = (double)gy;
  H[5] = (double)gz;
#pragma unroll
  for (int k = 6; k < 18; ++k) H[k] = 0.0;
  const double z = (double)sdf[i];
  const float gn = sqrtf(gx * gx + gy * gy + gz * gz);
  const double ga = (double)(gn - 1.0f);   // (grad_norm - 1.0 in fp32, then widened: :256)
  z_out[r] = z;
  rinv_out[r] = 1.0 / (1.0 + ga * ga) * (0.4 / (0.4 + z * z)) * 1000.0;
  vp_out[r * 3 + 0] = pmap[i * 3 + 0];
  vp_out[r * 3 + 1] = pmap[i * 3 + 1];
  vp_out[r * 3 + 2] = pmap[i * 3 + 2];
}

}  // namespace clid

static int check_view(const clid_map_view* mv, const char* who) {
  if (!mv || !mv->tab || !mv->tab_pos || !mv->pos4 || !mv->delta || mv->P <= 0 || mv->P > clid::kMaxProbes ||
      mv->log2cap < 4 || mv->buffer_size <= 0) {
    clid_set_error("%s: incomplete map view", who);
    return CLID_E_ARG;
  }
  if (mv->M >= (1 << clid::probe_shift_of(mv->P))) {
    clid_set_error("%s: a table of %d points exceeds the 2^%d the searches' candidates address with %d-cell neighbourhoods", who, mv->M,
                   clid::probe_shift_of(mv->P), mv->P);
    return CLID_E_SHAPE;
  }
  return CLID_OK;
}

extern "C" int clid_radius_search(const clid_map_view* mv, const float* x, int32_t N, float* dist2_out,
                                  int32_t* idx_out, void* stream) {
  if (int e = check_view(mv, "clid_radius_search")) return e;
  if (N <= 0) return CLID_OK;
  const long long total = (long long)N * mv->P;
  hipLaunchKernelGGL(clid::k_radius_search, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, *mv, x, N, dist2_out, idx_out);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

extern "C" int clid_query_certainty(const int64_t* buffer_pt_index, int64_t buffer_size, const float* neural_points,
                                    const float* point_certainties, const int32_t* delta, int32_t P, float resolution,
                                    float max_valid_dist2, const float* x, int32_t N, float* cert_out, void* stream) {
  if (!buffer_pt_index || !neural_points || !point_certainties || !delta || !x || !cert_out || P <= 0 ||
      buffer_size <= 0 || buffer_size >= (1LL << 30)) {
    clid_set_error("clid_query_certainty: bad argument");
    return CLID_E_ARG;
  }
  if (N <= 0) return CLID_OK;
  hipLaunchKernelGGL(clid::k_query_certainty_direct, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(buffer_pt_index), (int)buffer_size, neural_points, point_certainties,
                     delta, P, resolution, max_valid_dist2, x, N, cert_out);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

extern "C" int clid_query_fwd(const clid_map_view* mv, const float* x, const int32_t* query_ts, int32_t N,
                              int32_t training_mode, int32_t weighted_first, float* feat_out, float* w_out,
                              int32_t* idx_out, int32_t* nn_out, float* cert_out, void* stream) {
  if (int e = check_view(mv, "clid_query_fwd")) return e;
  if (!mv->feat || !mv->cert) {
    clid_set_error("clid_query_fwd: view has no feature/certainty arrays");
    return CLID_E_ARG;
  }
  if (N <= 0) return CLID_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(clid::k_query_fwd, dim3((N + CLID_QPB - 1) / CLID_QPB), dim3(CLID_BLOCK), 0, s, *mv, x,
                     N, weighted_first, feat_out, w_out, idx_out, nn_out, cert_out);
  CLID_CHECK_LAUNCH();
  if (training_mode) {
    const int NK = N * CLID_K;
    hipLaunchKernelGGL(clid::k_query_side_effects, dim3((NK + 255) / 256), dim3(256), 0, s, *mv, idx_out,
                       w_out, query_ts, NK);
    CLID_CHECK_LAUNCH();
  }
  return CLID_OK;
}

extern "C" int clid_query_bwd(const clid_map_view* mv, const float* x, const int32_t* idx, const float* w,
                              int32_t N, int32_t weighted_first, const float* g_feat, const float* g_w,
                              float* g_theta, float* g_x_out, void* stream) {
  if (int e = check_view(mv, "clid_query_bwd")) return e;
  if (N <= 0) return CLID_OK;
  hipLaunchKernelGGL(clid::k_query_bwd, dim3((N + CLID_QPB - 1) / CLID_QPB), dim3(CLID_BLOCK), 0,
                     (hipStream_t)stream, *mv, x, idx, w, N, weighted_first, g_feat, g_w, g_theta, g_x_out);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

extern "C" int clid_sdf_grad_x(const clid_map_view* mv, const float* W1, const float* b1, const float* W2,
                               const float* b2, float sdf_scale, const float* x, int32_t N, float* sdf_out,
                               float* grad_out, int32_t* nn_out, float* cert_out, void* stream) {
  if (int e = check_view(mv, "clid_sdf_grad_x")) return e;
  if (N <= 0) return CLID_OK;
  hipLaunchKernelGGL(clid::k_sdf_grad_x, dim3((N + CLID_QPB - 1) / CLID_QPB), dim3(CLID_BLOCK), 0,
                     (hipStream_t)stream, *mv, W1, b1, W2, b2, sdf_scale, x, N, sdf_out, grad_out, nn_out,
                     cert_out);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

// csrc/track_tile.hip: the measurement model on the matrix cores (weighted_first: True)
bool clid_track_tile_ok(const clid_map_view* mv);
int clid_launch_track_tile(const clid_map_view* mv, const float* W1, const float* b1, const float* W2, const float* b2,
                           const clid::TrackParams& tp, const float* rot_dev, const float* pos_dev, const float* pc_imu, int N,
                           float* sdf_out, float* grad_out, float* pmap_out, int* valid_out, double* normal_eq, double* zero_next,
                           hipStream_t s);
// Which kernel: the 16-lane kernel is one chain of ~11 us while ONE round of its blocks covers the scan (3 waves per SIMD x 4 points
// per wave = 12 288 points on this part) and two rounds beyond (29 us at 22 k points); the tile kernel is one chain of ~13 us up to
// 24 576 points (22 us with the finish launch at 22 k points against 34: profiles/r06_track_tile_ab.jsonl).  CLID_TRACK_TILE=0 / 1:
// never / always (A/B; tests run both).
static bool track_tile_on(const clid_map_view* mv, int N) {
  const char* e = getenv("CLID_TRACK_TILE");
  if (e && e[0] == '0') return false;
  if (!clid_track_tile_ok(mv)) return false;
  return (e && e[0] == '1') || N > 12288;
}

static int track_model_launch(const clid_map_view* mv, const float* W1, const float* b1, const float* W2, const float* b2,
                              float sdf_scale, const float* rot_host, const float* pos_host, const float* rot_dev,
                              const float* pos_dev, int32_t min_nn, float min_grad_norm, float max_grad_norm, float max_sdf_std,
                              const float* pc_imu, int32_t N, float* sdf_out, float* grad_out, float* pmap_out, int32_t* valid_out,
                              double* normal_eq, void* stream) {
  if (int e = check_view(mv, "clid_track_model")) return e;
  if (!((rot_host && pos_host) || (rot_dev && pos_dev)) || !pc_imu || N < 0) {
    clid_set_error("clid_track_model: bad argument");
    return CLID_E_ARG;
  }
  if (N == 0) return CLID_OK;
  clid::TrackParams tp;
  for (int i = 0; i < 9; ++i) tp.R[i] = rot_host ? rot_host[i] : 0.f;
  for (int i = 0; i < 3; ++i) tp.t[i] = pos_host ? pos_host[i] : 0.f;
  tp.scale = sdf_scale;
  tp.min_grad_norm = min_grad_norm;
  tp.max_grad_norm = max_grad_norm;
  tp.min_nn = min_nn;
  tp.max_sdf_std = max_sdf_std;
  if (track_tile_on(mv, N))
    return clid_launch_track_tile(mv, W1, b1, W2, b2, tp, rot_dev, pos_dev, pc_imu, N, sdf_out, grad_out, pmap_out, valid_out, normal_eq,
                                  nullptr, (hipStream_t)stream);
  hipLaunchKernelGGL(clid::k_track_model, dim3((N + CLID_QPB - 1) / CLID_QPB), dim3(CLID_BLOCK), 0,
                     (hipStream_t)stream, pc_imu, rot_dev, pos_dev, W1, b1, W2, b2, N, *mv, tp, sdf_out, grad_out, pmap_out, valid_out,
                     normal_eq);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

extern "C" int clid_track_model(const clid_map_view* mv, const float* W1, const float* b1, const float* W2,
                                const float* b2, float sdf_scale, const float* rot_host, const float* pos_host,
                                int32_t min_nn, float min_grad_norm, float max_grad_norm, float max_sdf_std,
                                const float* pc_imu, int32_t N, float* sdf_out, float* grad_out, float* pmap_out, int32_t* valid_out,
                                double* normal_eq, void* stream) {
  return track_model_launch(mv, W1, b1, W2, b2, sdf_scale, rot_host, pos_host, nullptr, nullptr, min_nn, min_grad_norm, max_grad_norm,
                            max_sdf_std, pc_imu, N, sdf_out, grad_out, pmap_out, valid_out, normal_eq, stream);
}

// the same with the pose in DEVICE memory (rot_dev [9] row-major fp32, pos_dev [3]): the reference's filter keeps its state
// in device tensors (utils/error_state_iekf.py:176-186), so iterating the update needs no host round trip per evaluation
extern "C" int clid_track_model_dev(const clid_map_view* mv, const float* W1, const float* b1, const float* W2,
                                    const float* b2, float sdf_scale, const float* rot_dev, const float* pos_dev,
                                    int32_t min_nn, float min_grad_norm, float max_grad_norm, float max_sdf_std,
                                    const float* pc_imu, int32_t N, float* sdf_out, float* grad_out, float* pmap_out,
                                    int32_t* valid_out, double* normal_eq, void* stream) {
  return track_model_launch(mv, W1, b1, W2, b2, sdf_scale, nullptr, nullptr, rot_dev, pos_dev, min_nn, min_grad_norm, max_grad_norm,
                            max_sdf_std, pc_imu, N, sdf_out, grad_out, pmap_out, valid_out, normal_eq, stream);
}

extern "C" int clid_track_model_call(const clid_track_call* c, const float* rot, const float* pos, int32_t pose_on_device,
                                     double* normal_eq, double* zero_next, double* result, double epoch, void* stream) {
  if (!c || !rot || !pos || (c->N > 0 && !c->pc_imu) || c->N < 0 || (result && !normal_eq)) {  // (an empty tensor has no storage)
    clid_set_error("clid_track_model_call: bad argument");
    return CLID_E_ARG;
  }
  if (int e = check_view(&c->mv, "clid_track_model_call")) return e;
  if (c->N == 0) {
    // an empty scan evaluates nothing, but the caller's ring moved on: the buffer the NEXT evaluation accumulates into must still
    // be cleared (k_track_model does that) and the epoch the host waits for must still be published (k_track_finish over the
    // all-zero buffer of this evaluation: 28 zero sums, n_valid 0)
    hipStream_t s0 = (hipStream_t)stream;
    if (zero_next && hipMemsetAsync(zero_next, 0, sizeof(double) * CLID_TRACK_COPIES * 32, s0) != hipSuccess) {
      clid_set_error("clid_track_model_call: %s", hipGetErrorString(hipGetLastError()));
      return CLID_E_HIP;
    }
    if (result) hipLaunchKernelGGL(clid::k_track_finish, dim3(1), dim3(64), 0, s0, normal_eq, result, epoch);
    CLID_CHECK_LAUNCH();
    return CLID_OK;
  }
  clid::TrackParams tp;
  for (int i = 0; i < 9; ++i) tp.R[i] = pose_on_device ? 0.f : rot[i];
  for (int i = 0; i < 3; ++i) tp.t[i] = pose_on_device ? 0.f : pos[i];
  tp.scale = c->sdf_scale;
  tp.min_grad_norm = c->min_grad_norm;
  tp.max_grad_norm = c->max_grad_norm;
  tp.min_nn = c->min_nn;
  tp.max_sdf_std = c->max_sdf_std;
  hipStream_t s = (hipStream_t)stream;
  if (track_tile_on(&c->mv, c->N)) {
    if (int e = clid_launch_track_tile(&c->mv, c->W1, c->b1, c->W2, c->b2, tp, pose_on_device ? rot : nullptr,
                                       pose_on_device ? pos : nullptr, c->pc_imu, c->N, c->sdf_out, c->grad_out, c->pmap_out,
                                       c->valid_out, normal_eq, zero_next, s))
      return e;
  } else
  hipLaunchKernelGGL(clid::k_track_model, dim3((c->N + CLID_QPB - 1) / CLID_QPB), dim3(CLID_BLOCK), 0, s, c->pc_imu,
                     pose_on_device ? rot : nullptr, pose_on_device ? pos : nullptr, c->W1, c->b1, c->W2, c->b2, c->N, c->mv, tp,
                     c->sdf_out, c->grad_out, c->pmap_out, c->valid_out, normal_eq, zero_next);
  if (result) hipLaunchKernelGGL(clid::k_track_finish, dim3(1), dim3(64), 0, s, normal_eq, result, epoch);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

extern "C" int clid_track_valid_count(const int32_t* valid, int32_t N, int32_t* block_prefix, double* result, double epoch,
                                      void* stream) {
  if (!valid || N <= 0 || !block_prefix || !result || N > (1 << 24)) {
    clid_set_error("clid_track_valid_count: bad argument");
    return CLID_E_ARG;
  }
  const int nb = (N + 255) / 256;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(clid::k_track_count, dim3(nb), dim3(256), 0, s, valid, N, block_prefix);
  hipLaunchKernelGGL(clid::k_track_scan, dim3(1), dim3(1024), 0, s, block_prefix, nb, result, epoch);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

extern "C" int clid_track_rows(const clid_track_call* c, const float* rot, const float* pos, int32_t pose_on_device,
                               const int32_t* block_prefix, double* z_out, double* H_out, float* valid_points_out, double* r_inv_out,
                               void* stream) {
  if (!c || !rot || !c->pc_imu || !c->sdf_out || !c->grad_out || !c->pmap_out || !c->valid_out || c->N <= 0 || !block_prefix ||
      !z_out || !H_out || !valid_points_out || !r_inv_out) {
    clid_set_error("clid_track_rows: bad argument (needs the per-point outputs of the bound call)");
    return CLID_E_ARG;
  }
  (void)pos;
  clid::TrackParams tp;
  for (int i = 0; i < 9; ++i) tp.R[i] = pose_on_device ? 0.f : rot[i];
  for (int i = 0; i < 3; ++i) tp.t[i] = 0.f;
  tp.scale = c->sdf_scale;
  tp.min_grad_norm = c->min_grad_norm;
  tp.max_grad_norm = c->max_grad_norm;
  tp.min_nn = c->min_nn;
  tp.max_sdf_std = c->max_sdf_std;
  hipLaunchKernelGGL(clid::k_track_rows, dim3((c->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, c->pc_imu, c->sdf_out, c->grad_out,
                     c->pmap_out, c->valid_out, c->N, tp, pose_on_device ? rot : nullptr, block_prefix, z_out, H_out, valid_points_out,
                     r_inv_out);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

extern "C" int clid_pinned_alloc(int64_t bytes, void** host_out, void** dev_out) {
  if (bytes <= 0 || !host_out || !dev_out) {
    clid_set_error("clid_pinned_alloc: bad argument");
    return CLID_E_ARG;
  }
  void* h = nullptr;
  void* d = nullptr;
  if (hipHostMalloc(&h, (size_t)bytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
    clid_set_error("clid_pinned_alloc: %s", hipGetErrorString(hipGetLastError()));
    if (h) (void)hipHostFree(h);
    return CLID_E_HIP;
  }
  memset(h, 0, (size_t)bytes);
  *host_out = h;
  *dev_out = d;
  return CLID_OK;
}
extern "C" void clid_pinned_free(void* host_ptr) {
  if (host_ptr) (void)hipHostFree(host_ptr);
}

bool clid_sdf_query_tile_ok(const clid_map_view* mv);
int clid_launch_sdf_query_tile(const clid_map_view* mv, const float* W1, const float* b1, const float* W2, const float* b2,
                               float sdf_scale, const float* x, int N, float* sdf_out, int* nn_out, hipStream_t s);

extern "C" int clid_sdf_query(const clid_map_view* mv, const float* W1, const float* b1, const float* W2,
                              const float* b2, float sdf_scale, const float* x, int32_t N, float* sdf_out,
                              int32_t* nn_out, void* stream) {
  if (int e = check_view(mv, "clid_sdf_query")) return e;
  if (!x || !sdf_out || !nn_out || N < 0) {
    clid_set_error("clid_sdf_query: bad argument");
    return CLID_E_ARG;
  }
  if (N == 0) return CLID_OK;
  // weighted_first: True: the tile kernel (csrc/query_tile.hip: 8-lane search + decoder on the matrix cores);
  // CLID_SDF_TILE=0 keeps the 16-lane kernel (A/B, tests compare the two)
  const char* tile_env = getenv("CLID_SDF_TILE");
  const bool tile_on = !(tile_env && tile_env[0] == '0');
  if (tile_on && clid_sdf_query_tile_ok(mv))
    return clid_launch_sdf_query_tile(mv, W1, b1, W2, b2, sdf_scale, x, N, sdf_out, nn_out, (hipStream_t)stream);
  int nb = (N + CLID_QPB - 1) / CLID_QPB;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(clid::k_sdf_query, dim3(nb), dim3(CLID_BLOCK), 0, (hipStream_t)stream, *mv, W1, b1, W2, b2,
                     sdf_scale, x, N, sdf_out, nn_out);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}
