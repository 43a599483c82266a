// Dense SDF query on the matrix cores: Mesher.query_points (utils/mesher.py:38-163), SDF part, for `weighted_first: True`
// (every shipped config) -- the throughput form of k_sdf_query (csrc/query.hip, 16 lanes per point, decoder on the VALU).
//
// One wave per TILE of 16 query points, the two halves of the training path back to back without a trip through HBM:
//   search   two rounds of the 8-lane search of the training launches (csrc/search8.hpp: walk of the map's cell directory,
//            probing where a point lies outside its box) -- 8 points per round, winners, IDW weights and the blended offset
//            sum_k w_k (x - p_k) (model/neural_points.py:653-706) into LDS in the layout of a search record;
//   decode   the forward half of k_decode_tile (csrc/train_tile.hip): lane (q = lane & 15, g = lane >> 4) gathers ITS four
//            columns of the six neighbours' rows, blends them in registers, layer 1 is 16 x v_mfma_f32_16x16x4_f32 against
//            the staged W1 | b1 operand, layer 2 sixteen in-lane FMAs and two lane exchanges (model/decoder.py:58-82).
// SDF where at least one probe found a point within range (else 0, utils/mesher.py:122-128) and that count for the
// marching-cubes mask (:156-161).  Per point ~90 + ~25 wave instructions instead of ~125 + ~40 with 16 lanes per point.
#include "search8.hpp"

namespace clid {

constexpr int kQtBlock = 256, kQtWaves = kQtBlock / 64;
#ifndef CLID_QT_WAVES
#define CLID_QT_WAVES 4  // waves per SIMD the kernel is compiled for (128 VGPRs; measured 2 / 3 / 4 / 5 / 6 / 8: 1.20 / 0.96 / 0.88 / 1.04 / 1.15 / 1.00 ms per 4.2 M points)
#endif

struct QtHead {
  float2 win[8][8];  // per point of the round: k < 6: (w_k, id bits; -1 none) | [6] = (fx, fy) | [7] = (fz, valid-probe count bits)
};

__device__ __forceinline__ float qt_xsum16(float v) {  // v[lane] + v[lane ^ 16]
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float qt_xsum32(float v) {  // v[lane] + v[lane ^ 32]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// PSH: the id's width in the search's candidates (search8.hpp): 22 bits, or 24 for tables beyond 2^22 points with <= 128-cell stencils
template <int PSH>
__global__ void __launch_bounds__(kQtBlock, CLID_QT_WAVES)
k_sdf_query_tile(clid_map_view mv, const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                 const float* __restrict__ b2p, float sc, const float* __restrict__ x, int N, float* __restrict__ sdf_out,
                 int* __restrict__ nn_out) {
  __shared__ DeltaLds dl;
  __shared__ CellLds cl;
  __shared__ QtHead heads[kQtWaves][2];
  __shared__ int lists[kQtWaves * 8 * kCdHits];  // one hit list per point of a round
  __shared__ float wl[CLID_MLP_PARAMS + 3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lane8 = lane & 7, slot8 = lane >> 3;
  const int q = lane & 15, g = lane >> 4;
  stage_delta(dl, mv);
  stage_cells(cl, mv, true);
  for (int i = threadIdx.x; i < CLID_H * CLID_D; i += kQtBlock) wl[i] = W1[i];
  if (threadIdx.x < CLID_H) {
    wl[CLID_H * CLID_D + threadIdx.x] = b1[threadIdx.x];
    wl[CLID_H * CLID_D + CLID_H + threadIdx.x] = W2[threadIdx.x];
  }
  if (threadIdx.x == 0) wl[CLID_MLP_PARAMS - 1] = b2p[0];
  __syncthreads();
  // constant MFMA operands (train_tile.hip): A1[u][s] = W1e[16u + q][4g + s], W1e = [W1 | b1 | 0 0 0 0]; W2r[u][r] = W2[16u + 4g + r]
  float A1[4][4], W2r[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int h = 16 * u + q, c = 4 * g + s;
      float a = 0.f;
      if (c < CLID_D) a = wl[h * CLID_D + c];
      else if (c == CLID_D) a = wl[CLID_H * CLID_D + h];
      A1[u][s] = a;
      W2r[u][s] = wl[CLID_H * CLID_D + CLID_H + 16 * u + 4 * g + s];
    }
  const float b2 = wl[CLID_MLP_PARAMS - 1];
  const float4* __restrict__ feat4 = reinterpret_cast<const float4*>(mv.feat);
  const float4* __restrict__ pos4 = reinterpret_cast<const float4*>(mv.pos4);
  const int nc = cl.nc;
  const int n_tiles = (N + 15) >> 4;

  for (int tile = blockIdx.x * kQtWaves + wave; tile < n_tiles; tile += gridDim.x * kQtWaves) {
    // ================= search: two rounds of 8 points, 8 lanes per point
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      const int p_raw = tile * 16 + half * 8 + slot8;
      const int p = p_raw < N ? p_raw : N - 1;  // (padding lanes search the last point again: nobody reads them)
      const float px = x[(size_t)p * 3 + 0], py = x[(size_t)p * 3 + 1], pz = x[(size_t)p * 3 + 2];
      float2* win = heads[wave][half].win[slot8];
      const int rx = (int)floorf(fdiv(px, mv.resolution)) - cl.ox, ry = (int)floorf(fdiv(py, mv.resolution)) - cl.oy;
      const int rz0 = (int)floorf(fdiv(pz, mv.resolution)) - cl.oz - nc;
      // all 2 nc + 1 cells per axis inside the directory's box?  Outside it a probe can only meet a foreign collision: the
      // probing search answers that exactly (as search_task, csrc/train.hip)
      const bool inside = (unsigned)(rx - nc) < (unsigned)(cl.nx - 2 * nc) && (unsigned)(ry - nc) < (unsigned)(cl.ny - 2 * nc) &&
                          (unsigned)rz0 < (unsigned)(cl.nz - 2 * nc);
      int nvalid = 0;
      if (cl.valid && !__any(!inside))
        search_cells<true, PSH>(mv, cl, lists + (wave * 8 + slot8) * kCdHits, px, py, pz, rx, ry, rz0, lane8, lane & 56, win, false, &nvalid);
      else
        search8<false, CLID_K, true, PSH>(mv, dl, px, py, pz, lane8, lane & 56, win, nullptr, &nvalid);
      nvalid = group8_sum_i(nvalid);
      // (d2, id) -> (IDW weight, id) + blended offset (np.py:653-706), lane8 = k
      wave_lds_fence();
      const float2 wn = win[lane8 < CLID_K ? lane8 : 0];
      const int id = __float_as_int(wn.y);
      const bool valid = lane8 < CLID_K && id >= 0;
      const float om = valid ? fdiv(1.0f, fadd(wn.x, 1e-15f)) : 0.f;  // np.py:688-693
      const float osum = group8_sum(om);
      const float w = valid ? fmul(om, fdiv(1.0f, osum)) : 0.f;       // np.py:699-706
      const float4 pk = pos4[valid ? id : 0];
      const float fx = group8_sum(fsub(px, pk.x) * w), fy = group8_sum(fsub(py, pk.y) * w), fz = group8_sum(fsub(pz, pk.z) * w);
      wave_lds_fence();
      win[lane8] = lane8 < CLID_K ? make_float2(w, wn.y) : (lane8 == CLID_K ? make_float2(fx, fy) : make_float2(fz, __int_as_float(nvalid)));
    }
    wave_lds_fence();
    // ================= decode: lane (q, g), point q of the tile
    const float2* wq = heads[wave][q >> 3].win[q & 7];
    float w[CLID_K];
    int j[CLID_K];
#pragma unroll
    for (int k = 0; k < CLID_K; ++k) {
      const float2 e = wq[k];
      j[k] = __float_as_int(e.y);
      w[k] = j[k] >= 0 ? e.x : 0.f;
    }
    const float2 f01 = wq[CLID_K], f2n = wq[CLID_K + 1];
    float4 v[CLID_K];
#pragma unroll
    for (int k = 0; k < CLID_K; ++k) {
      v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g < 2) v[k] = feat4[(size_t)(j[k] >= 0 ? j[k] : 0) * 2 + g];
    }
    if (mv.layer_norm) {  // F.layer_norm over the 8 features of every neighbour row (np.py:632-633); lanes g = 0, 1 hold the halves
#pragma unroll
      for (int k = 0; k < CLID_K; ++k) {
        const float mu = qt_xsum16((v[k].x + v[k].y) + (v[k].z + v[k].w)) * (1.0f / CLID_F);
        const float4 c = make_float4(v[k].x - mu, v[k].y - mu, v[k].z - mu, v[k].w - mu);
        const float var = qt_xsum16((c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w)) * (1.0f / CLID_F);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        v[k] = make_float4(c.x * rstd, c.y * rstd, c.z * rstd, c.w * rstd);
      }
    }
    float pc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CLID_K; ++k) {
      pc[0] = fmaf(v[k].x, w[k], pc[0]);
      pc[1] = fmaf(v[k].y, w[k], pc[1]);
      pc[2] = fmaf(v[k].z, w[k], pc[2]);
      pc[3] = fmaf(v[k].w, w[k], pc[3]);
    }
    if (g == 2) {  // the blended offset and the bias input
      pc[0] = f01.x; pc[1] = f01.y; pc[2] = f2n.x; pc[3] = 1.0f;
    } else if (g == 3) {
      pc[0] = pc[1] = pc[2] = pc[3] = 0.f;
    }
    f32x4 D[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      D[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) D[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[u][s], pc[s], D[u], 0, 0, 0);
    }
    float part = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) part = fmaf(W2r[u][rr], fmaxf(D[u][rr], 0.f), part);
    const float sdf = sc * (qt_xsum32(qt_xsum16(part)) + b2);
    const int p_out = tile * 16 + q;
    if (g == 0 && p_out < N) {
      const int nn = __float_as_int(f2n.y);
      sdf_out[p_out] = nn >= 1 ? sdf : 0.f;
      nn_out[p_out] = nn;
    }
    wave_lds_fence();  // (the heads are free again)
  }
}

}  // namespace clid

// launcher for clid_sdf_query (csrc/query.hip): the configurations this kernel covers
bool clid_sdf_query_tile_ok(const clid_map_view* mv) {
  return mv->weighted_first != 0 && mv->P <= clid::kMaxProbes && mv->M < (1 << clid::probe_shift_of(mv->P));
}

static int qt_resident_blocks() {
  static thread_local int dev_cached = -1, cus_cached = 0;  // (a cache of a device attribute)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256 * CLID_QT_WAVES;
  if (dev != dev_cached) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    cus_cached = cus;
    dev_cached = dev;
  }
  return cus_cached * 4 * CLID_QT_WAVES / clid::kQtWaves;
}

int clid_launch_sdf_query_tile(const clid_map_view* mv, const float* W1, const float* b1, const float* W2, const float* b2,
                               float sdf_scale, const float* x, int N, float* sdf_out, int* nn_out, hipStream_t s) {
  const int n_tiles = (N + 15) / 16;
  int nb = (n_tiles + clid::kQtWaves - 1) / clid::kQtWaves;
  const int resident = qt_resident_blocks();
  if (nb > resident) nb = resident;
  if (mv->M >= (1 << clid::kProbeShift))  // (only reached with <= 128-cell stencils: clid_sdf_query_tile_ok)
    hipLaunchKernelGGL(clid::k_sdf_query_tile<clid::kProbeShiftWide>, dim3(nb), dim3(clid::kQtBlock), 0, s, *mv, W1, b1, W2, b2,
                       sdf_scale, x, N, sdf_out, nn_out);
  else
    hipLaunchKernelGGL(clid::k_sdf_query_tile<clid::kProbeShift>, dim3(nb), dim3(clid::kQtBlock), 0, s, *mv, W1, b1, W2, b2, sdf_scale,
                       x, N, sdf_out, nn_out);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}
