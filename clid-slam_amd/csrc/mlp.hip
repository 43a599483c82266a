// Stand-alone decoder and loss kernels for callers that use the reference's un-fused call sequence
// (Decoder.sdf, model/decoder.py:58-82; sdf_bce_loss, utils/loss.py:44-62; eikonal term,
// utils/mapper.py:779-798).  The fused mapping iteration (train.hip) does not go through these.
#include "common.hpp"

namespace clid {

__global__ void __launch_bounds__(CLID_BLOCK)
k_mlp_fwd(const float* W1, const float* b1, const float* W2, const float* b2, float scale,
          const float* __restrict__ feat, int rows, float* __restrict__ sdf_out) {
  __shared__ MlpLds mlp;
  stage_mlp(mlp, W1, b1, W2, b2);
  const int lane16 = threadIdx.x & 15;
  for (int r0 = blockIdx.x * CLID_QPB; r0 < rows; r0 += gridDim.x * CLID_QPB) {
    const int r_raw = r0 + (threadIdx.x >> 4);
    const int r = r_raw < rows ? r_raw : rows - 1;
    float f[CLID_D];
#pragma unroll
    for (int c = 0; c < CLID_D; ++c) f[c] = feat[(size_t)r * CLID_D + c];
    float pre[CLID_HPL];
    const float s = mlp_forward(mlp, f, lane16, scale, pre);
    if (r_raw < rows && lane16 == 0) sdf_out[r] = s;
  }
}

__global__ void __launch_bounds__(CLID_BLOCK)
k_mlp_bwd(const float* W1, const float* b1, const float* W2, const float* b2, float scale,
          const float* __restrict__ feat, const float* __restrict__ g_sdf, int rows,
          float* __restrict__ g_feat, float* __restrict__ g_mlp) {
  __shared__ MlpLds mlp;
  stage_mlp(mlp, W1, b1, W2, b2);
  const int lane = threadIdx.x & 63, lane16 = lane & 15;
  float dW1[CLID_HPL][CLID_D], db1[CLID_HPL], dW2[CLID_HPL], db2 = 0.f;
#pragma unroll
  for (int u = 0; u < CLID_HPL; ++u) {
    db1[u] = dW2[u] = 0.f;
#pragma unroll
    for (int c = 0; c < CLID_D; ++c) dW1[u][c] = 0.f;
  }
  for (int r0 = blockIdx.x * CLID_QPB; r0 < rows; r0 += gridDim.x * CLID_QPB) {
    const int r_raw = r0 + (threadIdx.x >> 4);
    const bool live = r_raw < rows;
    const int r = live ? r_raw : rows - 1;
    float f[CLID_D];
#pragma unroll
    for (int c = 0; c < CLID_D; ++c) f[c] = feat[(size_t)r * CLID_D + c];
    float pre[CLID_HPL];
    (void)mlp_forward(mlp, f, lane16, scale, pre);
    const float dz = live ? scale * g_sdf[r] : 0.f;
    float dh[CLID_HPL];
#pragma unroll
    for (int u = 0; u < CLID_HPL; ++u) {
      const int h = lane16 + CLID_G * u;
      const bool on = pre[u] > 0.f;
      dh[u] = on ? dz * mlp.w[CLID_H * CLID_D + CLID_H + h] : 0.f;
      dW2[u] += on ? dz * pre[u] : 0.f;
      db1[u] += dh[u];
#pragma unroll
      for (int c = 0; c < CLID_D; ++c) dW1[u][c] = fmaf(dh[u], f[c], dW1[u][c]);
    }
    if (lane16 == 0) db2 += dz;
    if (g_feat) {
      float mine = 0.f;
#pragma unroll
      for (int c = 0; c < CLID_D; ++c) {
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < CLID_HPL; ++u) part = fmaf(mlp.w[(lane16 + CLID_G * u) * CLID_D + c], dh[u], part);
        const float tot = group_sum(part);
        mine = (lane16 == c) ? tot : mine;
      }
      if (live && lane16 < CLID_D) g_feat[(size_t)r * CLID_D + lane16] = mine;
    }
  }
  if (!g_mlp) return;
  // wave-level reduce, then one atomic per wave per parameter (grid is capped by the host)
#pragma unroll
  for (int u = 0; u < CLID_HPL; ++u) {
    const int h = lane16 + CLID_G * u;
#pragma unroll
    for (int c = 0; c < CLID_D; ++c) {
      const float v = cross_group_sum(dW1[u][c]);
      if (lane < CLID_G) atomicAdd(&g_mlp[h * CLID_D + c], v);
    }
    const float vb = cross_group_sum(db1[u]);
    if (lane < CLID_G) atomicAdd(&g_mlp[CLID_H * CLID_D + h], vb);
    const float vw = cross_group_sum(dW2[u]);
    if (lane < CLID_G) atomicAdd(&g_mlp[CLID_H * CLID_D + CLID_H + h], vw);
  }
  const float v2 = cross_group_sum(db2);
  if (lane == 0) atomicAdd(&g_mlp[CLID_MLP_PARAMS - 1], v2);
}

// BCE-with-logits (weighted mean) + eikonal; gradients w.r.t. pred and g
__global__ void k_loss(const float* __restrict__ pred, const float* __restrict__ label,
                       const float* __restrict__ weight, int N, float sigma, int weighted,
                       const float* __restrict__ g, int Ng, float weight_e, float* __restrict__ loss_out,
                       float* __restrict__ d_pred, float* __restrict__ d_g) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float bce = 0.f, eik = 0.f;
  const float inv_sigma = fdiv(1.0f, sigma);
  if (i < N) {
    const float z = pred[i] * inv_sigma;
    const float tgt = 1.0f / (1.0f + expf(-label[i] * inv_sigma));
    const float wt = weighted ? weight[i] : 1.0f;
    bce = wt * (fmaxf(z, 0.f) - z * tgt + log1pf(expf(-fabsf(z))));
    if (d_pred) d_pred[i] = wt * (1.0f / (1.0f + expf(-z)) - tgt) * inv_sigma / (float)N;
  }
  if (i < Ng) {
    const float gx = g[i * 3], gy = g[i * 3 + 1], gz = g[i * 3 + 2];
    const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
    eik = (nrm - 1.f) * (nrm - 1.f);
    if (d_g) {
      const float c = nrm > 0.f ? weight_e * 2.f * (nrm - 1.f) / ((float)Ng * nrm) : 0.f;
      d_g[i * 3] = c * gx; d_g[i * 3 + 1] = c * gy; d_g[i * 3 + 2] = c * gz;
    }
  }
  bce = wave_sum(bce);
  eik = wave_sum(eik);
  __shared__ float sb[CLID_BLOCK / 64], se[CLID_BLOCK / 64];
  if ((threadIdx.x & 63) == 0) {
    sb[threadIdx.x >> 6] = bce;
    se[threadIdx.x >> 6] = eik;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tb = 0.f, te = 0.f;
    for (int w = 0; w < CLID_BLOCK / 64; ++w) {
      tb += sb[w];
      te += se[w];
    }
    tb = N > 0 ? tb / (float)N : 0.f;
    te = Ng > 0 ? te / (float)Ng : 0.f;
    atomicAdd(&loss_out[1], tb);
    atomicAdd(&loss_out[2], te);
    atomicAdd(&loss_out[0], tb + weight_e * te);
  }
}

}  // namespace clid

extern "C" int clid_mlp_sdf_fwd(const float* W1, const float* b1, const float* W2, const float* b2,
                                float sdf_scale, const float* feat, int32_t rows, float* sdf_out,
                                void* stream) {
  if (!W1 || !b1 || !W2 || !b2 || !feat || !sdf_out || rows < 0) {
    clid_set_error("clid_mlp_sdf_fwd: bad argument");
    return CLID_E_ARG;
  }
  if (rows == 0) return CLID_OK;
  int nb = (rows + CLID_QPB - 1) / CLID_QPB;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(clid::k_mlp_fwd, dim3(nb), dim3(CLID_BLOCK), 0, (hipStream_t)stream, W1, b1, W2, b2,
                     sdf_scale, feat, rows, sdf_out);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

extern "C" int clid_mlp_sdf_bwd(const float* W1, const float* b1, const float* W2, const float* b2,
                                float sdf_scale, const float* feat, const float* g_sdf, int32_t rows,
                                float* g_feat_out, float* g_mlp, void* stream) {
  if (!W1 || !b1 || !W2 || !b2 || !feat || !g_sdf || rows < 0) {
    clid_set_error("clid_mlp_sdf_bwd: bad argument");
    return CLID_E_ARG;
  }
  if (rows == 0) return CLID_OK;
  int nb = (rows + CLID_QPB - 1) / CLID_QPB;
  if (nb > 256) nb = 256;  // bounds the same-address atomics on g_mlp
  hipLaunchKernelGGL(clid::k_mlp_bwd, dim3(nb), dim3(CLID_BLOCK), 0, (hipStream_t)stream, W1, b1, W2, b2,
                     sdf_scale, feat, g_sdf, rows, g_feat_out, g_mlp);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

extern "C" int clid_loss_fwd_bwd(const float* pred, const float* label, const float* weight, int32_t N,
                                 float sigma, int32_t weighted, const float* g, int32_t Ng, float weight_e,
                                 float* loss_out, float* d_pred_out, float* d_g_out, void* stream) {
  if (!pred || !label || (weighted && !weight) || !loss_out || N < 0 || Ng < 0 || (Ng > 0 && !g)) {
    clid_set_error("clid_loss_fwd_bwd: bad argument");
    return CLID_E_ARG;
  }
  const int n = N > Ng ? N : Ng;
  if (n == 0) return CLID_OK;
  hipLaunchKernelGGL(clid::k_loss, dim3((n + CLID_BLOCK - 1) / CLID_BLOCK), dim3(CLID_BLOCK), 0,
                     (hipStream_t)stream, pred, label, weight, N, sigma, weighted, g, Ng, weight_e, loss_out,
                     d_pred_out, d_g_out);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}
