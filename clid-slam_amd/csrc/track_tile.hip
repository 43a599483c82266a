// Tracking measurement model on the matrix cores: IEKFOM.h_model (utils/error_state_iekf.py:176-264) for `weighted_first: True`
// (every shipped config) -- the tile form of k_track_model (csrc/query.hip: 16 lanes per point, every lane repeating the six
// neighbours' gathers and the blend, 145 registers, 3 waves per SIMD: 29 us for a 22 k-point scan, two rounds of blocks).
//
// One wave per tile of 8 (or, for scans beyond one round of the chip, 16) scan points, built from the two halves the training
// path and the dense SDF query already run (csrc/search8.hpp, csrc/query_tile.hip, csrc/train_tile.hip):
//   pose     p_map = R p_imu + t in fp32 (transform_torch with the fp32 T of :182-186), per point by its 8 search lanes;
//   search   the 8-lane walk of the window's cell directory (probing where a point lies outside its box), winners, IDW weights and
//            the blended offset (model/neural_points.py:653-706) into LDS -- one round of 8 points, so the launch is ONE search
//            task + one decode long whatever the scan's size up to 8 x the resident waves;
//   forward  lane (q = lane & 15, g = lane >> 4) gathers its four columns of the six neighbour rows, blends, layer 1 on
//            v_mfma_f32_16x16x4_f32, layer 2 in-lane (model/decoder.py:58-82);
//   backward u = d sdf / d f = scale (W2 .* relu') W1 with the operands swapped (A = the gate x W2 per hidden unit straight from
//            the accumulator registers, B = W1 rows): column c of point 4G + r lands on lane (c, G), register r; one trip through
//            LDS puts u back on the point's lanes;
//   gradient d sdf / d x = sum_k (u . v_k) d w_k / d x + (sum_k w_k) u[8:11],  v_k = [row_k | x - p_k],
//            d w_k / d x = w_k (abar - alpha_k),  alpha_k = 2 (x - p_k) omega_k,  abar = sum_k w_k alpha_k  (SURVEY.md A.4): the
//            dot products are split over the point's lanes (g = 0, 1: the feature halves; g = 2: the offset part), summed by
//            two lane exchanges, the rest runs on the point's g = 2 lane;
//   outputs  sdf, gradient, p_map, validity mask (:233-241) per point (optional), and the float64 sums update_iterated (:299-305)
//            needs -- S = H^T R_inv H (upper triangle of the 6 x 6 block), H^T R_inv z, the valid count -- added up over the tile
//            in LDS and to one of CLID_TRACK_COPIES line-separated copies.
#include "search8.hpp"

namespace clid {

constexpr int kTtBlock = 256, kTtWaves = kTtBlock / 64;

struct TtHead {
  float2 win[8][8];  // per point of the round: k < 6: (w_k, id bits; -1 none) | [6] = (fx, fy) | [7] = (fz, valid-probe count bits)
};
struct TtWave {
  TtHead heads[2];
  float4 pmap[16];   // p_map of the tile's points (w: unused)
  float4 pimu[16];   // p_imu
  float ut[16][12];  // u = d sdf / d f per point (11 columns)
  double red[16][28];
};

__device__ __forceinline__ float tt_xsum16(float v) {  // v[lane] + v[lane ^ 16]
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float tt_xsum32(float v) {  // v[lane] + v[lane ^ 32]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// PSH: the id's width in the search's candidates (search8.hpp); HALVES: 8-point rounds per tile (1: 8 points per wave, 2: 16)
template <int PSH>
__global__ void __launch_bounds__(kTtBlock, 3)
k_track_tile(const float* __restrict__ pc_imu, const float* __restrict__ rot_dev, const float* __restrict__ pos_dev,
             const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2p, int N,
             int halves, clid_map_view mv, TrackParams tp, float* __restrict__ sdf_out, float* __restrict__ grad_out,
             float* __restrict__ pmap_out, int* __restrict__ valid_out, double* __restrict__ normal_eq, double* __restrict__ zero_next) {
  if (zero_next && blockIdx.x == 0)  // (clid_track_model_call) block 0 clears the NEXT call's reduction buffer
    for (int i = threadIdx.x; i < CLID_TRACK_COPIES * 32; i += kTtBlock) zero_next[i] = 0.0;
  __shared__ DeltaLds dl;
  __shared__ CellLds cl;
  __shared__ TtWave wv_lds[kTtWaves];
  __shared__ int lists[kTtWaves * 8 * kCdHits];  // one hit list per point of a round
  __shared__ float wl[CLID_MLP_PARAMS + 3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lane8 = lane & 7, slot8 = lane >> 3;
  const int q = lane & 15, g = lane >> 4;
  TtWave& tw = wv_lds[wave];
  stage_delta(dl, mv);
  stage_cells(cl, mv, true);
  for (int i = threadIdx.x; i < CLID_H * CLID_D; i += kTtBlock) wl[i] = W1[i];
  if (threadIdx.x < CLID_H) {
    wl[CLID_H * CLID_D + threadIdx.x] = b1[threadIdx.x];
    wl[CLID_H * CLID_D + CLID_H + threadIdx.x] = W2[threadIdx.x];
  }
  if (threadIdx.x == 0) wl[CLID_MLP_PARAMS - 1] = b2p[0];
  if (rot_dev) {  // the pose read on the device (uniform loads): no host round trip in front of the launch
#pragma unroll
    for (int i = 0; i < 9; ++i) tp.R[i] = rot_dev[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) tp.t[i] = pos_dev[i];
  }
  __syncthreads();
  // constant MFMA operands (query_tile.hip / train_tile.hip): A1[u][s] = W1e[16u + q][4g + s], W1e = [W1 | b1 | 0 0 0 0];
  // W2r[u][r] = W2[16u + 4g + r]; A2[u][r] = W1[16u + 4g + r][q] (q < 11), the B operand of the swapped backward contraction
  float A1[4][4], W2r[4][4], A2[4][4];
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
      A2[u][s] = q < CLID_D ? wl[(16 * u + 4 * g + s) * CLID_D + q] : 0.f;
    }
  const float b2 = wl[CLID_MLP_PARAMS - 1];
  const float sc = tp.scale;
  const float4* __restrict__ feat4 = reinterpret_cast<const float4*>(mv.feat);
  const float4* __restrict__ pos4 = reinterpret_cast<const float4*>(mv.pos4);
  const int nc = cl.nc;
  const int ppt = 8 * halves;  // points per tile
  const int n_tiles = (N + ppt - 1) / ppt;

  for (int tile = blockIdx.x * kTtWaves + wave; tile < n_tiles; tile += gridDim.x * kTtWaves) {
    // ================= pose + search: `halves` rounds of 8 points, 8 lanes per point
#pragma unroll 1
    for (int half = 0; half < halves; ++half) {
      const int p_raw = tile * ppt + half * 8 + slot8;
      const int p = p_raw < N ? p_raw : N - 1;  // (padding lanes search the last point again: nobody reads them)
      const float ix = pc_imu[(size_t)p * 3 + 0], iy = pc_imu[(size_t)p * 3 + 1], iz = pc_imu[(size_t)p * 3 + 2];
      const float px = tp.R[0] * ix + tp.R[1] * iy + tp.R[2] * iz + tp.t[0];
      const float py = tp.R[3] * ix + tp.R[4] * iy + tp.R[5] * iz + tp.t[1];
      const float pz = tp.R[6] * ix + tp.R[7] * iy + tp.R[8] * iz + tp.t[2];
      if (lane8 == 0) {
        tw.pmap[half * 8 + slot8] = make_float4(px, py, pz, 0.f);
        tw.pimu[half * 8 + slot8] = make_float4(ix, iy, iz, 0.f);
      }
      float2* win = tw.heads[half].win[slot8];
      const int rx = (int)floorf(fdiv(px, mv.resolution)) - cl.ox, ry = (int)floorf(fdiv(py, mv.resolution)) - cl.oy;
      const int rz0 = (int)floorf(fdiv(pz, mv.resolution)) - cl.oz - nc;
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
    // ================= forward: lane (q, g), point q of the tile (q >= ppt: no point)
    const bool has = q < ppt;
    const float2* wq = tw.heads[has ? (q >> 3) : 0].win[q & 7];
    float w[CLID_K];
    int j[CLID_K];
#pragma unroll
    for (int k = 0; k < CLID_K; ++k) {
      const float2 e = wq[k];
      j[k] = has ? __float_as_int(e.y) : -1;
      w[k] = j[k] >= 0 ? e.x : 0.f;
    }
    const float2 f01 = wq[CLID_K], f2n = wq[CLID_K + 1];
    float4 v[CLID_K];
#pragma unroll
    for (int k = 0; k < CLID_K; ++k) {
      v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g < 2) v[k] = feat4[(size_t)(j[k] >= 0 ? j[k] : 0) * 2 + g];
      else if (g == 2) v[k] = pos4[j[k] >= 0 ? j[k] : 0];  // the neighbour's position: x - p_k for the gradient below
    }
    if (mv.layer_norm) {  // F.layer_norm over the 8 features of every neighbour row (np.py:632-633); lanes g = 0, 1 hold the halves
#pragma unroll
      for (int k = 0; k < CLID_K; ++k) {
        const float4 raw = g < 2 ? v[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float mu = tt_xsum16((raw.x + raw.y) + (raw.z + raw.w)) * (1.0f / CLID_F);
        const float4 c = make_float4(raw.x - mu, raw.y - mu, raw.z - mu, raw.w - mu);
        const float cs = g < 2 ? (c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w) : 0.f;
        const float var = tt_xsum16(cs) * (1.0f / CLID_F);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        if (g < 2) v[k] = make_float4(c.x * rstd, c.y * rstd, c.z * rstd, c.w * rstd);
      }
    }
    float pc[4] = {0.f, 0.f, 0.f, 0.f};
    if (g < 2) {
#pragma unroll
      for (int k = 0; k < CLID_K; ++k) {
        pc[0] = fmaf(v[k].x, w[k], pc[0]);
        pc[1] = fmaf(v[k].y, w[k], pc[1]);
        pc[2] = fmaf(v[k].z, w[k], pc[2]);
        pc[3] = fmaf(v[k].w, w[k], pc[3]);
      }
    } else if (g == 2) {  // the blended offset and the bias input
      pc[0] = f01.x; pc[1] = f01.y; pc[2] = f2n.x; pc[3] = 1.0f;
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
    const float sdf = sc * (tt_xsum32(tt_xsum16(part)) + b2);
    // ================= backward: u = d sdf / d f, operands swapped: Df[r] of lane (c = lane & 15, G = lane >> 4) = u[c] of point 4G + r
    f32x4 Df = {0.f, 0.f, 0.f, 0.f}, Df2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; u += 2)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        Df = __builtin_amdgcn_mfma_f32_16x16x4f32(D[u][rr] > 0.f ? sc * W2r[u][rr] : 0.f, A2[u][rr], Df, 0, 0, 0);
        Df2 = __builtin_amdgcn_mfma_f32_16x16x4f32(D[u + 1][rr] > 0.f ? sc * W2r[u + 1][rr] : 0.f, A2[u + 1][rr], Df2, 0, 0, 0);
      }
    Df += Df2;
    if (q < CLID_D) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) tw.ut[4 * g + rr][q] = Df[rr];
    }
    wave_lds_fence();
    // ================= gradient: the dot products u . v_k over the point's lanes, the rest on its g = 2 lane
    const float4 pm = tw.pmap[has ? q : 0];
    float u4[4] = {0.f, 0.f, 0.f, 0.f};
    if (g < 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) u4[i] = (4 * g + i < CLID_D) ? tw.ut[q][4 * g + i] : 0.f;
    }
    float rxk[CLID_K], ryk[CLID_K], rzk[CLID_K], sk[CLID_K];
#pragma unroll
    for (int k = 0; k < CLID_K; ++k) {
      float d = 0.f;
      rxk[k] = ryk[k] = rzk[k] = 0.f;
      if (g < 2) {
        d = (u4[0] * v[k].x + u4[1] * v[k].y) + (u4[2] * v[k].z + u4[3] * v[k].w);
        if (j[k] < 0) d = 0.f;
      } else if (g == 2 && j[k] >= 0) {
        rxk[k] = fsub(pm.x, v[k].x);
        ryk[k] = fsub(pm.y, v[k].y);
        rzk[k] = fsub(pm.z, v[k].z);
        d = u4[0] * rxk[k] + u4[1] * ryk[k] + u4[2] * rzk[k];
      }
      sk[k] = tt_xsum32(tt_xsum16(d));
    }
    const int p_out = tile * ppt + q;
    const bool live = has && p_out < N;
    if (g == 2) {
      float wsum = 0.f, abx = 0.f, aby = 0.f, abz = 0.f, om[CLID_K];
#pragma unroll
      for (int k = 0; k < CLID_K; ++k) {
        om[k] = 0.f;
        if (j[k] >= 0) {
          const float d2 = fadd(fadd(fmul(rxk[k], rxk[k]), fmul(ryk[k], ryk[k])), fmul(rzk[k], rzk[k]));
          om[k] = fdiv(1.0f, fadd(d2, 1e-15f));
          wsum += w[k];
          abx += w[k] * 2.f * rxk[k] * om[k];
          aby += w[k] * 2.f * ryk[k] * om[k];
          abz += w[k] * 2.f * rzk[k] * om[k];
        }
      }
      float gx = wsum * u4[0], gy = wsum * u4[1], gz = wsum * u4[2];
#pragma unroll
      for (int k = 0; k < CLID_K; ++k) {
        const float cw = sk[k] * w[k];
        gx += cw * (abx - 2.f * rxk[k] * om[k]);
        gy += cw * (aby - 2.f * ryk[k] * om[k]);
        gz += cw * (abz - 2.f * rzk[k] * om[k]);
      }
      const int nn = __float_as_int(f2n.y);
      const float gn = sqrtf(gx * gx + gy * gy + gz * gz);
      // (sdf_std stays 0 for weighted_first configs, error_state_iekf.py:188: the std mask passes whenever its threshold is positive)
      const bool valid = live && nn >= tp.min_nn && gn < tp.max_grad_norm && gn > tp.min_grad_norm && 0.f < tp.max_sdf_std;
      if (live) {
        if (sdf_out) sdf_out[p_out] = sdf;
        if (grad_out) { grad_out[p_out * 3 + 0] = gx; grad_out[p_out * 3 + 1] = gy; grad_out[p_out * 3 + 2] = gz; }
        if (pmap_out) { pmap_out[p_out * 3 + 0] = pm.x; pmap_out[p_out * 3 + 1] = pm.y; pmap_out[p_out * 3 + 2] = pm.z; }
        if (valid_out) valid_out[p_out] = valid ? 1 : 0;
      }
      if (normal_eq) {
        // h = [p_imu x (R^T g), g] in fp32 (the reference builds it with fp32 bmm's), then float64 products
        const float4 pi = tw.pimu[has ? q : 0];
        const float qx = tp.R[0] * gx + tp.R[3] * gy + tp.R[6] * gz;
        const float qy = tp.R[1] * gx + tp.R[4] * gy + tp.R[7] * gz;
        const float qz = tp.R[2] * gx + tp.R[5] * gy + tp.R[8] * gz;
        const double h[6] = {(double)(pi.y * qz - pi.z * qy), (double)(pi.z * qx - pi.x * qz), (double)(pi.x * qy - pi.y * qx),
                             (double)gx, (double)gy, (double)gz};
        const double z = (double)sdf, ga = (double)gn - 1.0;
        const double wgt = valid ? (1.0 / (1.0 + ga * ga)) * (0.4 / (0.4 + z * z)) * 1000.0 : 0.0;
        double* red = tw.red[q];
        int n = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b2i = a; b2i < 6; ++b2i) red[n++] = wgt * h[a] * h[b2i];  // 21 upper-triangle entries
#pragma unroll
        for (int a = 0; a < 6; ++a) red[n++] = wgt * h[a] * z;               // H^T R_inv z
        red[27] = valid ? 1.0 : 0.0;
      }
    }
    if (normal_eq) {
      wave_lds_fence();
      if (lane < 28) {
        double s = 0.0;
        for (int pnt = 0; pnt < ppt; ++pnt) s += tw.red[pnt][lane];
        // CLID_TRACK_COPIES line-separated copies of the 28 sums: atomics on one cache line retire one after the other
        if (s != 0.0) atomicAdd(&normal_eq[(tile % CLID_TRACK_COPIES) * 32 + lane], s);
      }
    }
    wave_lds_fence();  // (the wave's LDS is free again)
  }
}

}  // namespace clid

// does the tile kernel cover this view?  (as the dense SDF query's: csrc/query_tile.hip)
bool clid_track_tile_ok(const clid_map_view* mv) {
  return mv->weighted_first != 0 && mv->P <= clid::kMaxProbes && mv->M < (1 << clid::probe_shift_of(mv->P));
}

static int tt_device_cus() {
  static thread_local int dev_cached = -1, cus_cached = 0;  // (a cache of a device attribute)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev != dev_cached) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    cus_cached = cus;
    dev_cached = dev;
  }
  return cus_cached;
}

int clid_launch_track_tile(const clid_map_view* mv, const float* W1, const float* b1, const float* W2, const float* b2,
                           const clid::TrackParams& tp, const float* rot_dev, const float* pos_dev, const float* pc_imu, int N,
                           float* sdf_out, float* grad_out, float* pmap_out, int* valid_out, double* normal_eq, double* zero_next,
                           hipStream_t s) {
  // 8 points per wave while one round of waves (3 per SIMD: 168 registers) covers the scan -- the launch is then ONE search task + one decode long --,
  // 16 beyond that
  const int resident_waves = tt_device_cus() * 4 * 3;
  const int halves = (N + 7) / 8 <= resident_waves ? 1 : 2;
  const int n_tiles = (N + 8 * halves - 1) / (8 * halves);
  int nb = (n_tiles + clid::kTtWaves - 1) / clid::kTtWaves;
  const int resident_blocks = resident_waves / clid::kTtWaves;
  if (nb > resident_blocks) nb = resident_blocks;
  if (mv->M >= (1 << clid::kProbeShift))  // (only reached with <= 128-cell stencils: clid_track_tile_ok)
    hipLaunchKernelGGL(clid::k_track_tile<clid::kProbeShiftWide>, dim3(nb), dim3(clid::kTtBlock), 0, s, pc_imu, rot_dev, pos_dev, W1, b1,
                       W2, b2, N, halves, *mv, tp, sdf_out, grad_out, pmap_out, valid_out, normal_eq, zero_next);
  else
    hipLaunchKernelGGL(clid::k_track_tile<clid::kProbeShift>, dim3(nb), dim3(clid::kTtBlock), 0, s, pc_imu, rot_dev, pos_dev, W1, b1, W2,
                       b2, N, halves, *mv, tp, sdf_out, grad_out, pmap_out, valid_out, normal_eq, zero_next);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}
