// The fused mapping iteration: Mapper.mapping body (utils/mapper.py:642-836) =
//   get_batch gathers (mapper.py:501-507) -> query_feature (np.py:553-769) -> Decoder.sdf
//   (decoder.py:58-82) -> numerical gradient on every `decimation`-th sample (mapper.py:697-704,
//   985-1034; one more query+decode on the 6 shifted copies) -> BCE + eikonal loss
//   (loss.py:44-62, mapper.py:746-798) -> backward -> Adam (tools.py:205-255).
//
// Launch plan per iteration (numerical-eikonal mode, the shipped default):
//   k_train_fused8  all Q = bs + 6*ceil(bs/decimation) query points: search, blend, decode, loss, backward,
//                   feature-gradient scatter (atomics), per-block partials of the 833 decoder gradients
//   [k_reduce_partials + RCCL all-reduce of `grad` by the host when world > 1]
//   k_adam_all      partial reduction (single GPU), dense Adam over features and decoder, zeroes `grad`
#include "train_common.hpp"
#include "search8.hpp"

namespace clid {

// ---- the fused iteration kernel ---------------------------------------------------------------------------
// A wave (4 query groups) executes TASKS of 2 rounds x 4 queries:
//   bundle task j (j < n_fd):  round A = the x+,x-,y+,y- shifted copies of decimated sample j,
//                              round B = z+, z-, the sample itself, and one non-decimated sample;
//   plain task:                8 non-decimated samples.
// All six finite-difference SDFs of a bundle therefore live in one wave, so the eikonal term, its
// backward, the BCE term and the decoder / feature gradients are produced without a grid-wide hand-off
// and without writing per-query state to HBM.

#ifndef CLID_FUSED_WAVES
#define CLID_FUSED_WAVES 4
#endif
#ifndef CLID_SEARCH_WAVES
#define CLID_SEARCH_WAVES 6  // waves per SIMD the search-only instantiation is compiled for (tools/variant_bench.py sweeps it)
#endif

// ---- the kernel: 8-lane search groups, 16-lane decode groups ----------------------------------
// The probe/selection phase is replicated work per lane-slot, so it runs with 8 lanes per query: one pass
// serves all 8 queries of a task (both "rounds" at once: half the instructions per query for the selection,
// 11 probe rows instead of 2 x 6, and the two rounds' dependent loads overlap).  Winners go to LDS; the
// decode/backward phase keeps 16 lanes per query (4 hidden units per lane, MFMA operand layout) in the
// lane16 = 2k + half arrangement: every lane owns half a feature row of ONE neighbour, weights and blends
// are DPP butterflies, the feature-gradient scatter is 4 atomics per lane.

struct WaveHead {       // what the search phase produces (one record of kRecFloat4 float4 per task)
  float4 qinfo[8];      // per query slot: x, y, z, time stamp of the sample (int bits; -1 = padding slot, 0 for shifted copies)
  float4 qdesc[8];      // per query slot: batch position (int bits, -1 = padding), axis/sign code (int bits: -1 = the
                        // sample itself, else 2*axis + (sign > 0)), SDF label, loss weight -- everything the decode
                        // phase needs of the pool, gathered while the search's own loads are in flight
  float2 win[8][8];     // per slot: the K nearest neighbours, ascending distance: (IDW weight w_k, local id bits; -1 = none);
                        // [6] = (fx, fy), [7] = (fz, -): the blended offset sum_k w_k (x - p_k) = decoder inputs 8..10.
                        // Weights and offsets depend on positions only, never on the training state, so the hoisted
                        // search resolves them once (np.py:653-706) and the decode kernels gather features only.
};
struct WaveLds : WaveHead {
  float4 st[2][64][2];  // per decode round, per lane: {f[lane16], w_k, j_k bits, sdf}, {pre[0..3]}
  float cacc[CLID_K][CLID_F];  // bundle tasks: gradient rows of the decimated sample's neighbours, summed in-wave
  float ccert[8];              // ... and their certainty increments
};

// Bundle tasks: the 7 queries around one decimated sample (itself + 6 copies shifted by 0.08 m) share almost
// all of their 6 neighbours, so their row updates are summed in LDS and leave the wave as ONE set of atomics
// (global fp32 atomics are the second-largest cost of the kernel: ~110 G lane-atomics/s when 8 lanes hit one
// row, 18 G/s scattered -- tools/ubench_gather.hip).  Returns the position of row j among the sample's own
// neighbours (slot 6 of the task), or -1.
__device__ __forceinline__ int match_base(const WaveLds& wl, int j) {
  int m = -1;
#pragma unroll
  for (int k = CLID_K - 1; k >= 0; --k) m = (__float_as_int(wl.win[6][k].y) == j) ? k : m;
  return j >= 0 ? m : -1;
}


constexpr int kFusedBlock = 512;  // 8 waves share one partial row: half as many rows for k_adam_all to reduce
// MODE 0: search + decode in one launch (clid_train_fwd_bwd).  MODE 2: the decode phase alone, from the records the
// hoisted search launch (k_search_tiles below) parked in HBM: kRecFloat4 float4 per task = qinfo | qdesc | win, the head of
// WaveLds.  The search reads nothing that training writes (positions, table and sample indices only), so one search launch
// resolves a whole chunk of iterations ahead of the dependent decode -> Adam chain (clid_train_search).
constexpr int kRecFloat4 = 48;
#ifndef CLID_XCD_MAP
#define CLID_XCD_MAP 1
#endif
template <int MODE>
__global__ void __launch_bounds__(kFusedBlock, CLID_FUSED_WAVES)
k_train_fused8(clid_map_view mv, clid_train_args ta, float* __restrict__ partial, TaskMap tmap,
               float4* __restrict__ rec) {
  static_assert(MODE == 0 || MODE == 2, "MODE 1 (search only) is k_search_tiles");
  __shared__ MlpLds mlp;
  __shared__ DeltaLds dl;
  __shared__ WaveLds wlds[kFusedBlock / 64];
  __shared__ float red[(kFusedBlock / 64) * kRedFloats];
  static_assert(sizeof(WaveHead) == kRecFloat4 * sizeof(float4), "record layout");
  if constexpr (MODE == 2) {
    stage_mlp(mlp, ta.W1, ta.b1, ta.W2, ta.b2);
  } else {
    stage_mlp_and_delta(mlp, dl, mv, ta.W1, ta.b1, ta.W2, ta.b2);
  }
  const int lane = threadIdx.x & 63, lane16 = lane & 15, grp = lane >> 4;
  const int lane8 = lane & 7, slot8 = lane >> 3;
  const int wave = threadIdx.x >> 6, waves_per_block = kFusedBlock / 64;
  const int my_k = lane16 >> 1;
  const bool odd = lane16 & 1;
  WaveLds& wl = wlds[wave];
  WaveHead& hd = static_cast<WaveHead&>(wl);
  MlpAcc acc;
  acc.zero();
  float bce_acc = 0.f, eik_acc = 0.f;
  const int gstride = ta.grad_stride == CLID_GRAD_ROW16 ? CLID_GRAD_ROW16 : CLID_F;  // floats per accumulation row
  float* g_theta = ta.grad + CLID_GRAD_OFFSET(gstride);
  const float inv_sigma = fdiv(1.0f, ta.sigma);
  const float inv_two_eps = fdiv(1.0f, 2.0f * ta.fd_eps);
  const float4* pos4 = reinterpret_cast<const float4*>(mv.pos4);
  const float sc = ta.sdf_scale;

  for (int task = blockIdx.x * waves_per_block + wave; task < tmap.n_tasks; task += gridDim.x * waves_per_block) {
    const long long* index = reinterpret_cast<const long long*>(ta.index);
    const bool bundle = task < tmap.n_fd;
    CLID_STAMP(0);
    // ================= search: 8 slots x 8 lanes
    if constexpr (MODE == 2) {
      if (lane < kRecFloat4) reinterpret_cast<float4*>(&hd)[lane] = rec[(size_t)task * kRecFloat4 + lane];
    } else {
      const QDesc qd = task_query(tmap, task, slot8 >> 2, slot8 & 3);
      const bool live = qd.p >= 0;
      const long long s = index[live ? qd.p : 0];
      float px = ta.pool_coord[s * 3 + 0], py = ta.pool_coord[s * 3 + 1], pz = ta.pool_coord[s * 3 + 2];
      if (qd.axis == 0) px = fadd(px, qd.sign * ta.fd_eps);  // x + [eps,0,0] in fp32 (mapper.py:988-999)
      if (qd.axis == 1) py = fadd(py, qd.sign * ta.fd_eps);
      if (qd.axis == 2) pz = fadd(pz, qd.sign * ta.fd_eps);
      if (lane8 == 0) {
        float label = 0.f, wt = 1.f;
        int ts = live ? 0 : -1;
        if (live && qd.axis < 0) {  // the sample itself: its label, weight (mapper.py:747-749) and time stamp
          label = ta.pool_label[s];
          if (ta.loss_weight_on) wt = fabsf(ta.pool_weight[s]);
          if (ta.pool_ts) ts = ta.pool_ts[s];
        }
        const int code = qd.axis < 0 ? -1 : 2 * qd.axis + (qd.sign > 0.f ? 1 : 0);
        hd.qinfo[slot8] = make_float4(px, py, pz, __int_as_float(ts));
        hd.qdesc[slot8] = make_float4(__int_as_float(qd.p), __int_as_float(code), label, wt);
      }
      asm volatile("" ::"v"(px), "v"(py), "v"(pz));
      CLID_STAMP(1);
      search8<false, CLID_K>(mv, dl, px, py, pz, lane8, lane & 56, hd.win[slot8]);
      // (d2, id) -> (IDW weight, id) + blended offset (np.py:653-706), lane8 = k
      wave_lds_fence();
      {
        const float2 wn = hd.win[slot8][lane8 < CLID_K ? lane8 : 0];
        const int id = __float_as_int(wn.y);
        const bool valid = lane8 < CLID_K && id >= 0;
        const float om = valid ? fdiv(1.0f, fadd(wn.x, 1e-15f)) : 0.f;   // np.py:688-693
        const float osum = group8_sum(om);
        const float w = valid ? fmul(om, fdiv(1.0f, osum)) : 0.f;        // np.py:699-706
        const float4 pk = pos4[valid ? id : 0];
        const float rx = group8_sum(fsub(px, pk.x) * w), ry = group8_sum(fsub(py, pk.y) * w),
                    rz = group8_sum(fsub(pz, pk.z) * w);
        wave_lds_fence();
        hd.win[slot8][lane8] = lane8 < CLID_K ? make_float2(w, wn.y)
                                              : (lane8 == CLID_K ? make_float2(rx, ry) : make_float2(rz, 0.f));
      }
    }
    CLID_STAMP(3);
    if (bundle) {
      if (lane < CLID_K * CLID_F) (&wl.cacc[0][0])[lane] = 0.f;
      if (lane < 8) wl.ccert[lane] = 0.f;
    }
    wave_lds_fence();
    CLID_STAMP(4);
    // ================= decode forward: 2 rounds x 4 queries x 16 lanes
#pragma unroll 1
    for (int round = 0; round < 2; ++round) {
      const int s16 = round * 4 + grp;
      const float4 qi = wl.qinfo[s16];
      const int sidx = __float_as_int(qi.w);
      const float2 wn = wl.win[s16][my_k];  // my_k < 8 always in range; k = 6,7 hold stale/none -> masked
      int my_j = (lane16 < 2 * CLID_K && sidx >= 0) ? __float_as_int(wn.y) : -1;
      const bool valid = my_j >= 0;
      const float my_w = valid ? wn.x : 0.f;  // IDW weight from the search record (np.py:688-706)
      const int jc = valid ? my_j : 0;
      float4 v = reinterpret_cast<const float4*>(mv.feat)[(size_t)jc * 2 + (odd ? 1 : 0)];
      const float4 pj = pos4[jc];
      if (mv.layer_norm) {  // np.py:632-633
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
        float r0 = fsub(qi.x, pj.x) * my_w, r1 = fsub(qi.y, pj.y) * my_w, r2 = fsub(qi.z, pj.z) * my_w;
#define CLID_BFLY(x) x += dpp_mov<0x128>(x); x += dpp_mov<0x124>(x); x += dpp_mov<0x122>(x);
        CLID_BFLY(a0) CLID_BFLY(a1) CLID_BFLY(a2) CLID_BFLY(a3) CLID_BFLY(r0) CLID_BFLY(r1) CLID_BFLY(r2)
#undef CLID_BFLY
        const float b0 = dpp_mov<0xB1>(a0), b1 = dpp_mov<0xB1>(a1), b2 = dpp_mov<0xB1>(a2), b3 = dpp_mov<0xB1>(a3);
        f[0] = odd ? b0 : a0; f[1] = odd ? b1 : a1; f[2] = odd ? b2 : a2; f[3] = odd ? b3 : a3;
        f[4] = odd ? a0 : b0; f[5] = odd ? a1 : b1; f[6] = odd ? a2 : b2; f[7] = odd ? a3 : b3;
        f[8] = r0; f[9] = r1; f[10] = r2;
      }
      float pre[CLID_HPL];
      const float sdf = mlp_forward(mlp, f, lane16, sc, pre);
      if (valid && !odd && !(ta.debug_flags & 1)) {  // training_mode side effects (np.py:708-733)
        const int q_axis = __float_as_int(wl.qdesc[s16].y);  // -1 = the sample itself
        const bool shares = bundle && !(round == 1 && grp == 3);  // every slot but the unrelated 8th sample
        const int m = shares ? match_base(wl, my_j) : -1;
        if (m >= 0) atomicAdd(&wl.ccert[m], my_w);
        else atomicAdd(&mv.cert[my_j], my_w);
        if (q_axis < 0 && mv.ts_update) {
          // amax is idempotent: only the first touch of a point by a newer stamp needs the atomic (a scattered
          // atomic costs ~10x a scattered load: tools/ubench_gather.hip)
          const int ts = sidx;  // qinfo.w carries the sample's time stamp
          if (mv.ts_update[my_j] < ts) atomicMax(&mv.ts_update[my_j], ts);
        }
      }
      float fb = (lane16 == CLID_D) ? 1.0f : 0.f;
#pragma unroll
      for (int c = 0; c < CLID_D; ++c) fb = (lane16 == c) ? f[c] : fb;
      wl.st[round][lane][0] = make_float4(fb, my_w, __int_as_float(my_j), sdf);
      wl.st[round][lane][1] = make_float4(pre[0], pre[1], pre[2], pre[3]);
      CLID_STAMP(5 + round);
    }
    wave_lds_fence();
    CLID_STAMP(7);
    // ================= losses
    float ecoef = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
    if (bundle) {
      gx = (wl.st[0][0][0].w - wl.st[0][16][0].w) * inv_two_eps;   // mapper.py:1011-1013
      gy = (wl.st[0][32][0].w - wl.st[0][48][0].w) * inv_two_eps;
      gz = (wl.st[1][0][0].w - wl.st[1][16][0].w) * inv_two_eps;
      const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
      if (lane == 0) eik_acc += (nrm - 1.f) * (nrm - 1.f);
      // d/dg of weight_e * mean((|g|-1)^2); 0 at |g| == 0 (torch norm subgradient)
      ecoef = nrm > 0.f ? ta.weight_e * 2.f * (nrm - 1.f) * ta.inv_n_eik * inv_two_eps / nrm : 0.f;
    }
    // ================= backward
#pragma unroll 1
    for (int round = 0; round < 2; ++round) {
      QDesc qd;
      float q_label, q_wt;
      {
        const float4 qq = wl.qdesc[round * 4 + grp];
        const int code = __float_as_int(qq.y);
        qd.p = __float_as_int(qq.x); qd.axis = code < 0 ? -1 : (code >> 1); qd.sign = (code & 1) ? 1.0f : -1.0f;
        q_label = qq.z; q_wt = qq.w;
      }
      const float4 s0 = wl.st[round][lane][0], s1 = wl.st[round][lane][1];
      const float fb = s0.x, my_w = s0.y, sdf = s0.w;
      const int my_j = __float_as_int(s0.z);
      float delta = 0.f;
      if (qd.p >= 0) {
        if (qd.axis < 0) {
          const float label = q_label, wt = q_wt;
          const float z = sdf * inv_sigma;
          const float tgt = __frcp_rn(1.0f + __expf(-label * inv_sigma));           // loss.py:60
          const float ez = __expf(-fabsf(z));
          const float sg = (z >= 0.f ? 1.0f : ez) * __frcp_rn(1.0f + ez);
          const float li = fmaxf(z, 0.f) - z * tgt + __logf(1.0f + ez);             // BCEWithLogits
          if (lane16 == 0) bce_acc += wt * li;
          delta = wt * (sg - tgt) * inv_sigma * ta.inv_n_main;
        } else {
          const float ga = qd.axis == 0 ? gx : (qd.axis == 1 ? gy : gz);
          delta = qd.sign * ecoef * ga;
        }
      }
      // decoder backward (dz = scale * delta)
      const float dz = sc * delta;
      const int l16 = lane16 + opaque_zero();
      const float pre[CLID_HPL] = {s1.x, s1.y, s1.z, s1.w};
      float dh[CLID_HPL];
#pragma unroll
      for (int u = 0; u < CLID_HPL; ++u) {
        const bool on = pre[u] > 0.f;
        dh[u] = on ? dz * mlp.w[CLID_H * CLID_D + CLID_H + l16 + CLID_G * u] : 0.f;
        if (ta.train_decoder) {
          acc.dW2[u] += on ? dz * pre[u] : 0.f;
          acc.dW1[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(dh[u], fb, acc.dW1[u], 0, 0, 0);
        }
      }
      if (ta.train_decoder && lane16 == 0) acc.db2 += dz;
      // d f[0:8] (replicated), then the scatter d theta[j_k][c] += w_k df[c]
      float df8[CLID_F];
#pragma unroll
      for (int c = 0; c < CLID_F; ++c) {
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < CLID_HPL; ++u) part = fmaf(mlp.w[(l16 + CLID_G * u) * CLID_D + c], dh[u], part);
        df8[c] = group_sum(part);
      }
      if (!(ta.debug_flags & 2)) {
        const int gb16 = lane & 48;
        const bool hi = lane16 >= 8;
        if (!mv.layer_norm) {
          // coalesced atomics: in pass r the low / high 8 lanes of the group write the 8 consecutive floats
          // of neighbour 2r / 2r+1 (one 32-byte segment each); its (j, w) come from the owning lane pair
          float dfc = 0.f;
#pragma unroll
          for (int c = 0; c < CLID_F; ++c) dfc = ((lane16 & 7) == c) ? df8[c] : dfc;
#pragma unroll
          for (int r = 0; r < CLID_K / 2; ++r) {
            const int src = gb16 + 4 * r + (hi ? 2 : 0);
            const int jk = __shfl(my_j, src, 64);
            const float wk = __shfl(my_w, src, 64);
            if (jk >= 0 && delta != 0.f) {
              const int m = (bundle && !(round == 1 && grp == 3)) ? match_base(wl, jk) : -1;
              if (m >= 0) atomicAdd(&wl.cacc[m][lane16 & 7], wk * dfc);
              else atomicAdd(&g_theta[(size_t)jk * gstride + (lane16 & 7)], wk * dfc);
            }
          }
        } else {
          // layer-norm backward in the (neighbour, half) lane layout, then the same coalesced scatter
          const int jc = my_j >= 0 ? my_j : 0;
          float4 v = reinterpret_cast<const float4*>(mv.feat)[(size_t)jc * 2 + (odd ? 1 : 0)];
          float t1 = (v.x + v.y) + (v.z + v.w);
          t1 += dpp_mov<0xB1>(t1);
          const float mu = t1 * (1.0f / CLID_F);
          v.x -= mu; v.y -= mu; v.z -= mu; v.w -= mu;
          float t2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
          t2 += dpp_mov<0xB1>(t2);
          const float rstd = 1.0f / sqrtf(t2 * (1.0f / CLID_F) + 1e-5f);
          v.x *= rstd; v.y *= rstd; v.z *= rstd; v.w *= rstd;
          float d4[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {  // explicit select: `df8[4*odd + i]` would put df8 in scratch
            const float lo = df8[i], hi4 = df8[4 + i];
            float sel = lo;
            asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(sel) : "v"(lo), "v"(hi4), "s"(__ballot(odd)));
            d4[i] = sel * my_w;
          }
          float m1 = (d4[0] + d4[1]) + (d4[2] + d4[3]);
          float m2 = (d4[0] * v.x + d4[1] * v.y) + (d4[2] * v.z + d4[3] * v.w);
          m1 += dpp_mov<0xB1>(m1);
          m2 += dpp_mov<0xB1>(m2);
          m1 *= (1.0f / CLID_F);
          m2 *= (1.0f / CLID_F);
          d4[0] = rstd * (d4[0] - m1 - v.x * m2); d4[1] = rstd * (d4[1] - m1 - v.y * m2);
          d4[2] = rstd * (d4[2] - m1 - v.z * m2); d4[3] = rstd * (d4[3] - m1 - v.w * m2);
#pragma unroll
          for (int r = 0; r < CLID_K / 2; ++r) {
            const int c = lane16 & 7;
            const int src = gb16 + 4 * r + (hi ? 2 : 0) + (c >> 2);  // lane holding floats 4*(c>>2).. of neighbour k
            const int jk = __shfl(my_j, src, 64);
            const float e0 = __shfl(d4[0], src, 64), e1 = __shfl(d4[1], src, 64), e2 = __shfl(d4[2], src, 64),
                        e3 = __shfl(d4[3], src, 64);
            const float val = (c & 2) ? ((c & 1) ? e3 : e2) : ((c & 1) ? e1 : e0);
            if (jk >= 0 && delta != 0.f) {
              const int m = (bundle && !(round == 1 && grp == 3)) ? match_base(wl, jk) : -1;
              if (m >= 0) atomicAdd(&wl.cacc[m][c], val);
              else atomicAdd(&g_theta[(size_t)jk * gstride + c], val);
            }
          }
        }
      }
      CLID_STAMP(8 + round);
    }
    if (bundle) {  // the combined rows of the decimated sample's neighbours leave the wave once
      wave_lds_fence();
      if (lane < CLID_K * CLID_F) {
        const int m = lane >> 3, c = lane & 7;
        const int j = __float_as_int(wl.win[6][m].y);
        if (j >= 0) {
          if (!(ta.debug_flags & 2)) atomicAdd(&g_theta[(size_t)j * gstride + c], wl.cacc[m][c]);
          if (c == 0 && !(ta.debug_flags & 1) && wl.ccert[m] != 0.f) atomicAdd(&mv.cert[j], wl.ccert[m]);
        }
      }
    }
    wave_lds_fence();
    CLID_STAMP(10);
  }
  CLID_STAMP(24);
  flush_mlp_acc(acc, bce_acc, eik_acc, red, partial + (size_t)blockIdx.x * kPartialStride, ta.train_decoder != 0);
  CLID_STAMP(25);
}

// ---- the hoisted search launch (clid_train_search): tasks in pairs, tiles numbered in the same pass ------------------------
// The search phase of k_train_fused8 per TILE (= two consecutive tasks, the unit of the matrix-core
// decode kernels): the wave searches both tasks, writes their records, and -- with both records still in LDS -- numbers the
// tile's (query, neighbour) pairs per distinct map row into the tile's number block (train_common.hpp kTileNumWords).  The
// numbering depends on the records only, so it belongs here, once per chunk, not in every decode launch's dependent chain
// (decode 14.5 -> 12.6 us at 16 384 samples; the search launch pays 1.0 us per iteration for it -- 8.2 instead of 7.2 -- and
// the step over 200 iterations goes from 29.3 to 28.4 us); a separate pass over the records cost 1.5 us per iteration.
#ifndef CLID_SEARCH_DMA
// 1: k_search_tasks<., 1> brings a task's inputs into LDS by DMA, two tasks ahead (InStage below) -- the register-free form of
// CLID_SEARCH_PIPE.  Measured and left off: 16.9 -> 18.7 us per iteration at 65 536 samples, 66.9 -> 73.5 at 262 144 (records
// identical): at 8 waves per SIMD the index -> pool hop is already hidden, the launch is bound by what it issues, and the staging
// adds LDS traffic, 48 B of scratch and a full vmcnt wait per task (profiles/r06_search_pipeline_ab.jsonl)
#define CLID_SEARCH_DMA 0
#endif
#ifndef CLID_SEARCH_INLINE_PROBE
// 1: k_search_tiles<., 1> probes a deferred tile (a query point outside the cell directory's box) itself instead of leaving it to a
// second launch over the deferred lists -- a launch that is empty on almost every call and costs its 4.6 us of launch boundary in
// front of the first decode all the same.  Measured and left off: with the probing code in the kernel the 80-register budget of
// 6 waves per SIMD spills in the directory path too (16 -> 176 B of scratch): 5.6 -> 7.1 us per iteration, 0.0270 -> 0.0290 ms per
// step (profiles/r06_search_pipeline_ab.jsonl); the split launch stays.
#define CLID_SEARCH_INLINE_PROBE 0
#endif
#ifndef CLID_SEARCH_PIPE
// 1: k_search_tiles<., 1> requests a task's index -> pool loads one task ahead (the hop is 1.85 us of a task's 6 when taken alone,
// tools/search_stage_timing.py).  Measured and left off: the launch is bound by how many units are resident, not by one unit's
// chain -- 7 more live registers cost 64 B of scratch at 6 waves per SIMD (5.5 -> 6.0 us per iteration) and a fifth of the
// resident grid at 5 (96 registers, no scratch: 8.5 us); records bit-identical either way (profiles/r06_search_pipeline_ab.jsonl)
#define CLID_SEARCH_PIPE 0
#endif
constexpr int kNumHash = 128;  // LDS hash slots for the <= 96 distinct map rows of a tile
struct TileNumLds {
  int hkey[kNumHash];            // hash slot -> map row id, -1 empty
  unsigned char hrow[kNumHash];  // hash slot -> row number inside the tile
};

// one wave task: pool gathers, 81-cell search of its 8 query slots, IDW weights / blended offsets -> record in `hd`
// CD: the cell-directory search (returns true = a query point lies outside the directory's box: nothing was written, the task
// is left to the probing kernels); else search8 (returns false).
// What a task reads of the batch and the pool: requested by `load_task_inputs` -- for the directory-search launch one task AHEAD
// of its use (k_search_tiles: the index -> pool hop, two dependent global loads = 1.85 us of a task's 6, tools/search_stage_timing.py,
// then runs under the previous task's search instead of in front of its own).
struct TaskIn {
  float x, y, z;     // pool_coord[s] of the lane's query slot (every lane of the slot's 8)
  float label, wt;   // lane8 == 0, the sample itself: its label / |weight| (else 0 / 1)
  int ts;            // its frame stamp (0 without a stamp array, -1 padding slot)
  int fr;            // Mapper.ba_done_flag: the sample's frame (every lane)
};
__device__ __forceinline__ TaskIn load_task_inputs(const clid_train_args& ta, const TaskMap& tmap, const long long* __restrict__ index,
                                                   int task) {
  const int lane = threadIdx.x & 63, lane8 = lane & 7, slot8 = lane >> 3;
  const QDesc qd = task_query(tmap, task, slot8 >> 2, slot8 & 3);
  const bool live = qd.p >= 0;
  const long long s = index[live ? qd.p : 0];
  TaskIn in;
  in.x = ta.pool_coord[s * 3 + 0];
  in.y = ta.pool_coord[s * 3 + 1];
  in.z = ta.pool_coord[s * 3 + 2];
  in.fr = ta.pool_pose ? ta.pool_ts[s] : 0;
  in.label = 0.f;
  in.wt = 1.f;
  in.ts = live ? 0 : -1;
  if (lane8 == 0 && live && qd.axis < 0) {  // the sample itself: its label, weight (mapper.py:747-749) and time stamp
    in.label = ta.pool_label[s];
    if (ta.loss_weight_on) in.wt = fabsf(ta.pool_weight[s]);
    if (ta.pool_ts) in.ts = ta.pool_ts[s];
  }
  return in;
}

// ---- task inputs by LDS-DMA (CLID_SEARCH_DMA) --------------------------------------------------------------------------
// The index -> pool hop in front of every task (two dependent global loads, 1.85 us of a task's 6) is taken off the wave's chain
// WITHOUT holding the next task's inputs in registers (that costs scratch at 80 registers: CLID_SEARCH_PIPE): global_load_lds writes
// straight into LDS.  Two requests per task, each one task apart: A(t) = the 8 query slots' batch indices (two dwords each) ->
// idx[t & 1]; B(t) = with those indices, the slots' pool rows (x, y, z, label, weight, frame) -> dat[t & 1].  At the top of task t
// the wave waits for everything it has in flight -- B(t) and A(t + 1) went out a whole task ago --, stores the PREVIOUS task's
// record (held back so that the wait does not sit behind its own stores: vmcnt counts loads and stores in order), requests
// B(t + 1) and A(t + 2), and starts on inputs that are already in LDS.
struct InStage {
  unsigned idx[2][64];  // [t & 1][slot8 * 8 + j]: j = 0, 1: the dwords of index[p] of the slot's query
  unsigned dat[2][64];  // [t & 1][slot8 * 8 + j]: j = 0..2 pool_coord, 3 label, 4 weight, 5 frame stamp
};
#define CLID_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
__device__ __forceinline__ void dma_task_index(InStage& st, int par, const TaskMap& tmap, const long long* __restrict__ index, int task) {
  const int lane = threadIdx.x & 63, lane8 = lane & 7, slot8 = lane >> 3;
  const QDesc qd = task_query(tmap, task, slot8 >> 2, slot8 & 3);
  const unsigned* src = reinterpret_cast<const unsigned*>(index + (qd.p >= 0 ? qd.p : 0)) + (lane8 & 1);
  if (lane8 < 2) __builtin_amdgcn_global_load_lds(src, CLID_LDS_PTR(&st.idx[par][0]), 4, 0, 0);
}
__device__ __forceinline__ void dma_task_data(InStage& st, int par, const clid_train_args& ta) {
  const int lane = threadIdx.x & 63, lane8 = lane & 7, slot8 = lane >> 3;
  const long long s = (long long)(((unsigned long long)st.idx[par][slot8 * 8 + 1] << 32) | st.idx[par][slot8 * 8]);
  const void* src = nullptr;
  if (lane8 < 3) src = ta.pool_coord + s * 3 + lane8;
  else if (lane8 == 3) src = ta.pool_label + s;
  else if (lane8 == 4) src = ta.pool_weight ? ta.pool_weight + s : nullptr;
  else if (lane8 == 5) src = ta.pool_ts ? ta.pool_ts + s : nullptr;
  if (src) __builtin_amdgcn_global_load_lds(src, CLID_LDS_PTR(&st.dat[par][0]), 4, 0, 0);
}
__device__ __forceinline__ TaskIn read_task_inputs(const InStage& st, int par, const clid_train_args& ta, const TaskMap& tmap, int task) {
  const int lane = threadIdx.x & 63, lane8 = lane & 7, slot8 = lane >> 3;
  const QDesc qd = task_query(tmap, task, slot8 >> 2, slot8 & 3);
  const bool live = qd.p >= 0;
  const unsigned* d = &st.dat[par][slot8 * 8];
  TaskIn in;
  in.x = __uint_as_float(d[0]);
  in.y = __uint_as_float(d[1]);
  in.z = __uint_as_float(d[2]);
  in.fr = ta.pool_pose ? (int)d[5] : 0;
  in.label = 0.f;
  in.wt = 1.f;
  in.ts = live ? 0 : -1;
  if (lane8 == 0 && live && qd.axis < 0) {  // the sample itself: its label, weight (mapper.py:747-749) and time stamp
    in.label = __uint_as_float(d[3]);
    if (ta.loss_weight_on) in.wt = fabsf(__uint_as_float(d[4]));
    if (ta.pool_ts) in.ts = (int)d[5];
  }
  return in;
}
__device__ __forceinline__ void wait_all_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <bool CD>
__device__ __forceinline__ bool search_task_body(const clid_map_view& mv, const clid_train_args& ta, const TaskMap& tmap,
                                                 const DeltaLds& dl, int task, int it, int use_filter,
                                                 const unsigned* __restrict__ filt_lds, WaveHead& hd, const CellLds& cl, const TaskIn& in) {
  const int lane = threadIdx.x & 63, lane8 = lane & 7, slot8 = lane >> 3;
  const float4* pos4 = reinterpret_cast<const float4*>(mv.pos4);
  const QDesc qd = task_query(tmap, task, slot8 >> 2, slot8 & 3);
  const bool live = qd.p >= 0;
  float px = in.x, py = in.y, pz = in.z;
  if (ta.pool_pose) {
    // Mapper.ba_done_flag (mapper.py:646-658): the pool is in the samples' sensor frames; bmm(R, p) + t of
    // utils/tools.py:612-636 with the pose of the sample's frame, products and sums unfused in that order
    int fr = in.fr;
    fr = fr < 0 ? 0 : (fr >= ta.n_pose ? ta.n_pose - 1 : fr);
    const float4* T = reinterpret_cast<const float4*>(ta.pool_pose) + (size_t)fr * 3;
    const float4 r0 = T[0], r1 = T[1], r2 = T[2];
    const float x = px, y = py, z = pz;
    px = fadd(fadd(fadd(fmul(r0.x, x), fmul(r0.y, y)), fmul(r0.z, z)), r0.w);
    py = fadd(fadd(fadd(fmul(r1.x, x), fmul(r1.y, y)), fmul(r1.z, z)), r1.w);
    pz = fadd(fadd(fadd(fmul(r2.x, x), fmul(r2.y, y)), fmul(r2.z, z)), r2.w);
  }
  if (qd.axis == 0) px = fadd(px, qd.sign * ta.fd_eps);  // x + [eps,0,0] in fp32 (mapper.py:988-999)
  if (qd.axis == 1) py = fadd(py, qd.sign * ta.fd_eps);
  if (qd.axis == 2) pz = fadd(pz, qd.sign * ta.fd_eps);
  if (lane8 == 0) {
    const int code = qd.axis < 0 ? -1 : 2 * qd.axis + (qd.sign > 0.f ? 1 : 0);
    hd.qinfo[slot8] = make_float4(px, py, pz, __int_as_float(in.ts));
    hd.qdesc[slot8] = make_float4(__int_as_float(qd.p), __int_as_float(code), in.label, in.wt);
  }
  asm volatile("" ::"v"(px), "v"(py), "v"(pz));
  if constexpr (CD) CLID_STAMP(13);  // index -> pool coordinates (+ label / weight / stamp) arrived
  if constexpr (CD) {
    // the window's cell directory (the dynamic LDS holds the hit lists)
    const int nc = cl.nc;
    const int rx = (int)floorf(fdiv(px, mv.resolution)) - cl.ox, ry = (int)floorf(fdiv(py, mv.resolution)) - cl.oy;
    const int rz0 = (int)floorf(fdiv(pz, mv.resolution)) - cl.oz - nc;
    // all 2 nc + 1 cells per axis inside the box?  Outside it a probe can only meet a foreign collision: the probing kernels
    // answer that exactly (padding slots search the coordinates of sample 0 like they do: identical records).  A window
    // beyond the directory's capacity (cl.valid == 0) defers every task.
    const bool inside = (unsigned)(rx - nc) < (unsigned)(cl.nx - 2 * nc) && (unsigned)(ry - nc) < (unsigned)(cl.ny - 2 * nc) &&
                        (unsigned)rz0 < (unsigned)(cl.nz - 2 * nc);
    if (__any(!inside) || !cl.valid) return true;
    int* list = const_cast<int*>(reinterpret_cast<const int*>(filt_lds)) + ((threadIdx.x >> 6) * 8 + slot8) * kCdHits;
    search_cells(mv, cl, list, px, py, pz, rx, ry, rz0, lane8, lane & 56, hd.win[slot8], (ta.debug_flags & 4) != 0);
    CLID_STAMP(15);  // hits' positions loaded, distances, the K winners selected
  } else {
    bool redo;
    if (use_filter == 1) redo = search8<true, 3>(mv, dl, px, py, pz, lane8, lane & 56, hd.win[slot8], filt_lds);
    else if (use_filter == 2) redo = search8<true, 3>(mv, dl, px, py, pz, lane8, lane & 56, hd.win[slot8], mv.filter);
    else redo = search8<false, 3>(mv, dl, px, py, pz, lane8, lane & 56, hd.win[slot8]);
    if (__any(redo) || (ta.debug_flags & 4))  // rare (debug bit 2 forces it: tests compare the two paths)
      search8<false, CLID_K>(mv, dl, px, py, pz, lane8, lane & 56, hd.win[slot8]);
  }
  // (d2, id) -> (IDW weight, id) + blended offset (np.py:653-706), lane8 = k
  wave_lds_fence();
  const float2 wn = hd.win[slot8][lane8 < CLID_K ? lane8 : 0];
  const int id = __float_as_int(wn.y);
  const bool valid = lane8 < CLID_K && id >= 0;
  const float om = valid ? fdiv(1.0f, fadd(wn.x, 1e-15f)) : 0.f;   // np.py:688-693
  const float osum = group8_sum(om);
  const float w = valid ? fmul(om, fdiv(1.0f, osum)) : 0.f;        // np.py:699-706
  const float4 pk = pos4[valid ? id : 0];
  // touched-row flag of (iteration of the chunk, map row): a plain byte store -- racing writers store the same value --
  // that the chunk's scan turns into the iteration's row set (sparse exchange / Adam, train_common.hpp)
  if (ta.touch_ws && valid && live) ta.touch_ws[(size_t)it * ta.touch_stride + id] = 1;
  const float rx = group8_sum(fsub(px, pk.x) * w), ry = group8_sum(fsub(py, pk.y) * w), rz = group8_sum(fsub(pz, pk.z) * w);
  wave_lds_fence();
  hd.win[slot8][lane8] = lane8 < CLID_K ? make_float2(w, wn.y) : (lane8 == CLID_K ? make_float2(rx, ry) : make_float2(rz, 0.f));
  wave_lds_fence();
  if constexpr (CD) CLID_STAMP(18);  // winners' pos4 rows gathered, IDW weights, blended offset
  return false;
}
// one wave task: pool gathers, 81-cell search of its 8 query slots, IDW weights / blended offsets -> record in `hd`
template <bool CD>
__device__ __forceinline__ bool search_task(const clid_map_view& mv, const clid_train_args& ta, const TaskMap& tmap,
                                            const DeltaLds& dl, const long long* __restrict__ index, int task, int it, int use_filter,
                                            const unsigned* __restrict__ filt_lds, WaveHead& hd, const CellLds& cl) {
  if constexpr (CD) CLID_STAMP(12);  // (tools/search_stage_timing.py: the stages of one directory-search task)
  const TaskIn in = load_task_inputs(ta, tmap, index, task);
  return search_task_body<CD>(mv, ta, tmap, dl, task, it, use_filter, filt_lds, hd, cl, in);
}

// The tile's pairs numbered per distinct map row: lane = (q = lane & 15, g = lane >> 4) as in k_decode_tile, lane (q, g)
// owns neighbours k = g and (g < 2) k = g + 4 of query slot q of the tile.  The ids go through a 128-slot LDS hash (integer
// ds_cmpst; the first pair of a row wins its slot), the winners take consecutive row numbers from two ballots.  Measured
// against a variant without LDS atomics (plain-store arbitration in rounds, linear or double hashing): 8.16 vs 8.42 / 8.43 us
// per iteration for the search launch (7.3 without any numbering); in the decode kernel itself the round-based variant was
// slower than the CAS loop it replaced (16.2 vs 14.5 us).  (Its own function: it keeps the search loop's registers apart.)
__device__ __noinline__ void number_tile(const WaveHead& h0, const WaveHead& h1, TileNumLds& nl, int* __restrict__ tn,
                                         bool second_live) {
  const int lane = threadIdx.x & 63;
  const int q = lane & 15, g = lane >> 4;
  const bool tlive = (q >> 3) == 0 || second_live;
  const WaveHead& hq = (q >> 3) ? h1 : h0;
  const bool qlive = tlive && __float_as_int(hq.qinfo[q & 7].w) >= 0;  // (padding slots carry a search nobody reads)
  nl.hkey[lane] = -1;
  nl.hkey[lane + 64] = -1;
  int jk[2], hs[2];
  bool won[2] = {false, false};
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int k = g + 4 * t;
    jk[t] = (k < CLID_K && qlive) ? __float_as_int(hq.win[q & 7][k < CLID_K ? k : 0].y) : -1;
    hs[t] = (int)(((unsigned)jk[t] * 2654435761u) >> 25);  // 7 bits
  }
  wave_lds_fence();
#pragma unroll
  for (int t = 0; t < 2; ++t)
    if (jk[t] >= 0) {
      for (;;) {
        const int old = atomicCAS(&nl.hkey[hs[t]], -1, jk[t]);
        if (old == -1) {  // first pair of this row in the tile
          won[t] = true;
          break;
        }
        if (old == jk[t]) break;
        hs[t] = (hs[t] + 1) & (kNumHash - 1);
      }
    }
  // the winners take consecutive row numbers from two ballots (no counter to contend for)
  const unsigned long long b0 = __ballot(won[0]), b1 = __ballot(won[1]);
  const unsigned long long below = (1ull << lane) - 1ull;
  const int n0 = __popcll(b0), count = n0 + __popcll(b1);
  if (won[0]) {
    const int d = __popcll(b0 & below);
    nl.hrow[hs[0]] = (unsigned char)d;
    tn[d] = jk[0];
  }
  if (won[1]) {
    const int d = n0 + __popcll(b1 & below);
    nl.hrow[hs[1]] = (unsigned char)d;
    tn[d] = jk[1];
  }
  wave_lds_fence();
  unsigned char* rbytes = reinterpret_cast<unsigned char*>(tn + kTileNumBytes);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int k = g + 4 * t;
    if (k < CLID_K) rbytes[q * CLID_K + k] = jk[t] >= 0 ? nl.hrow[hs[t]] : (unsigned char)255;
  }
  if (lane == 0) tn[kTileNumCount] = count;
  wave_lds_fence();
}

// MODE 2: the deferred FLAGS of an iteration (one word per task / tile, written by every unit of the directory launch: no reset
// needed) are shared out in slices of kDeferSlice units per block (grid.x = ceil(units / kDeferSlice)); a block compacts the
// set flags of its slice into `found` and returns whether there is anything to do.  Block-uniform.
constexpr int kDeferSlice = 512;
static_assert(kDeferSlice == kFusedBlock, "one unit of the slice per thread");
__device__ __forceinline__ bool gather_deferred(const int* __restrict__ flags, int n_units, int* found, int& count) {
  if (threadIdx.x == 0) count = 0;
  __syncthreads();
  const int u = blockIdx.x * kDeferSlice + threadIdx.x;  // (kDeferSlice == kFusedBlock: one unit per thread)
  const bool set = u < n_units && flags[u] != 0;
  const unsigned long long b = __ballot(set);
  int base = 0;
  if ((threadIdx.x & 63) == 0 && b) base = atomicAdd(&count, __popcll(b));
  base = __shfl(base, 0, 64);
  if (set) found[base + __popcll(b & ((1ull << (threadIdx.x & 63)) - 1ull))] = u;
  __syncthreads();
  return count > 0;
}

// MODE 0: probing (search8) over every tile of the launch.  MODE 1: the cell-directory search over every tile; a tile with a query
// point outside the directory's box goes on its iteration's deferred list (train_common.hpp).  MODE 2: probing over the
// deferred lists (grid.y = iteration; almost always empty: the block leaves before it stages anything).
template <bool XMAP, int MODE>
__global__ void __launch_bounds__(kFusedBlock, MODE == 1 ? CLID_CD_WAVES_TILES : CLID_SEARCH_WAVES)
k_search_tiles(clid_map_view mv, clid_train_args ta, TaskMap tmap, float4* __restrict__ rec, int n_iter, long long index_stride,
               int use_filter) {
  __shared__ DeltaLds dl;
  __shared__ CellLds cl;
  __shared__ WaveHead heads[kFusedBlock / 64][2];
  __shared__ TileNumLds nums[kFusedBlock / 64];
  extern __shared__ unsigned filt_lds[];  // MODE 0 / 2 with a prefilter of <= 32 KB: 2^log2filter bits; MODE 1: the hit lists
  const size_t iter_f4 = rec_floats_per_iter(tmap.n_tasks) / 4;
  const size_t def_off = rec_deferred_offset(tmap.n_tasks);
  const int n_tiles = (tmap.n_tasks + 1) / 2;
  __shared__ int dfound[kDeferSlice], dcount;
  if (MODE == 2) {  // this block's slice of the iteration's deferred flags -> its work list (usually empty: leave at once)
    if (!gather_deferred(reinterpret_cast<const int*>(rec + (size_t)blockIdx.y * iter_f4) + def_off, n_tiles, dfound, dcount)) return;
  }
  constexpr bool INLINE = MODE == 1 && CLID_SEARCH_INLINE_PROBE;  // a deferred tile is probed by its own wave, right here
  if (MODE != 1 || INLINE) stage_delta(dl, mv);
  stage_cells(cl, mv, MODE == 1);
  if (MODE != 1 && use_filter == 1)
    for (int i = threadIdx.x; i < (1 << mv.log2filter) / 32; i += kFusedBlock) filt_lds[i] = mv.filter[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves_per_block = kFusedBlock / 64;
  // XCD-aware tile mapping for maps beyond one L2: block b runs on XCD b % 8 (observed dispatch order; a speed matter only)
  // and every XCD has its own 4 MB L2.  On batches in Morton order consecutive tasks are neighbours in space, so XCD x takes
  // the x-th eighth of the bundle tiles and of the plain tiles of EVERY iteration: its L2 then serves one eighth of the map
  // (M = 243 k: search 18.9 -> 17.5 us per iteration; at M = 23 k it costs 0.2 us)
  const int xcd = blockIdx.x & 7, xb = blockIdx.x >> 3, xnb = ((int)gridDim.x + 7 - xcd) >> 3;
  const int nb_t = (tmap.n_fd + 1) / 2, n_rest = n_tiles - nb_t;
  const int xb0 = (int)((long long)nb_t * xcd / 8), xb1 = (int)((long long)nb_t * (xcd + 1) / 8);
  const int xr0 = (int)((long long)n_rest * xcd / 8), xr1 = (int)((long long)n_rest * (xcd + 1) / 8);
  const int xlen = (xb1 - xb0) + (xr1 - xr0);
  constexpr bool xmap = XMAP && MODE != 2;  // (its own instantiation: the mapping's scalars cost the small-map kernel registers it has to spill)
  const int w_first = MODE == 2 ? wave : (xmap ? xb * waves_per_block + wave : blockIdx.x * waves_per_block + wave);
  const int w_step = MODE == 2 ? waves_per_block : (xmap ? xnb * waves_per_block : gridDim.x * waves_per_block);
  const int w_total = MODE == 2 ? dcount : (xmap ? xlen * n_iter : n_tiles * n_iter);
  auto unit_of = [&](int w, int& it, int& tile) {
    if (MODE == 2) {
      it = blockIdx.y;
      tile = dfound[w];
    } else if (xmap) {
      it = w / xlen;
      const int u = w - it * xlen;
      tile = u < xb1 - xb0 ? xb0 + u : nb_t + xr0 + (u - (xb1 - xb0));
    } else {
      it = w / n_tiles;
      tile = w - it * n_tiles;
    }
  };
  const long long* index0 = reinterpret_cast<const long long*>(ta.index);
  // MODE 1: the inputs of a task are requested one task ahead (CLID_SEARCH_PIPE, default on)
  constexpr bool PIPE = MODE == 1 && CLID_SEARCH_PIPE;
  TaskIn in0 = {}, in1 = {};
  if (PIPE && w_first < w_total) {
    int it, tile;
    unit_of(w_first, it, tile);
    in0 = load_task_inputs(ta, tmap, index0 + (long long)it * index_stride, 2 * tile);
  }
  for (int w = w_first; w < w_total; w += w_step) {
    int it, tile;
    unit_of(w, it, tile);
    const long long* index = index0 + (long long)it * index_stride;
    float4* __restrict__ out = rec + (size_t)it * iter_f4;
    bool deferred = false;
    if constexpr (PIPE) {
      const bool has1 = 2 * tile + 1 < tmap.n_tasks;
      if (has1) in1 = load_task_inputs(ta, tmap, index, 2 * tile + 1);
      CLID_STAMP(12);
      deferred = search_task_body<true>(mv, ta, tmap, dl, 2 * tile, it, use_filter, filt_lds, heads[wave][0], cl, in0);
      if (!deferred && lane < kRecFloat4) out[(size_t)(2 * tile) * kRecFloat4 + lane] = reinterpret_cast<const float4*>(&heads[wave][0])[lane];
      if (w + w_step < w_total) {  // the next unit's first task
        int itn, tilen;
        unit_of(w + w_step, itn, tilen);
        in0 = load_task_inputs(ta, tmap, index0 + (long long)itn * index_stride, 2 * tilen);
      }
      if (!deferred && has1) {
        deferred = search_task_body<true>(mv, ta, tmap, dl, 2 * tile + 1, it, use_filter, filt_lds, heads[wave][1], cl, in1);
        if (!deferred && lane < kRecFloat4)
          out[(size_t)(2 * tile + 1) * kRecFloat4 + lane] = reinterpret_cast<const float4*>(&heads[wave][1])[lane];
      }
    } else {
#pragma unroll 1
      for (int half = 0; half < 2; ++half) {
        const int task = 2 * tile + half;
        if (task >= tmap.n_tasks) break;
        deferred = search_task<MODE == 1>(mv, ta, tmap, dl, index, task, it, use_filter, filt_lds, heads[wave][half], cl);
        if (deferred) break;
        if (lane < kRecFloat4) out[(size_t)task * kRecFloat4 + lane] = reinterpret_cast<const float4*>(&heads[wave][half])[lane];
      }
    }
    if (MODE == 1 && lane == 0) reinterpret_cast<int*>(out)[def_off + tile] = deferred ? 1 : 0;  // (every tile writes its flag)
    if (deferred) {
      wave_lds_fence();
      if constexpr (INLINE) {
        // a query point outside the directory's box: the tile's two tasks through the probing search (what the separate launch over
        // the deferred lists did: identical records), without the LDS prefilter (the dynamic LDS holds the hit lists here)
        const int uf = use_filter == 1 ? 0 : use_filter;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
          const int task = 2 * tile + half;
          if (task >= tmap.n_tasks) break;
          (void)search_task<false>(mv, ta, tmap, dl, index, task, it, uf, mv.filter, heads[wave][half], cl);
          if (lane < kRecFloat4) out[(size_t)task * kRecFloat4 + lane] = reinterpret_cast<const float4*>(&heads[wave][half])[lane];
        }
      } else {
        continue;
      }
    }
    if (MODE == 1) CLID_STAMP(19);  // both tasks' records stored
    number_tile(heads[wave][0], heads[wave][1], nums[wave],
                reinterpret_cast<int*>(out + (size_t)tmap.n_tasks * kRecFloat4) + (size_t)tile * kTileNumWords,
                2 * tile + 1 < tmap.n_tasks);
    if (MODE == 1) CLID_STAMP(20);  // tile numbered
  }
}

// The same search per TASK, without numbering: iterations of more than kTileLargeFrom tiles (65 536 samples and up), whose
// decode launch numbers its tiles in place.  (Pairing the tasks costs the search 5-8 % by itself, the numbering another
// 13 %: 105 -> 113 -> 126 us per iteration at 262 144 samples.)  MODE as in k_search_tiles.
template <bool XMAP, int MODE>
__global__ void __launch_bounds__(kFusedBlock, MODE == 1 ? CLID_CD_WAVES_TASKS : CLID_SEARCH_WAVES)
k_search_tasks(clid_map_view mv, clid_train_args ta, TaskMap tmap, float4* __restrict__ rec, int n_iter, long long index_stride,
               int use_filter) {
  __shared__ DeltaLds dl;
  __shared__ CellLds cl;
  __shared__ WaveHead heads[kFusedBlock / 64];
  extern __shared__ unsigned filt_lds[];
  const size_t iter_f4 = rec_floats_per_iter(tmap.n_tasks) / 4;
  const size_t def_off = rec_deferred_offset(tmap.n_tasks);
  __shared__ int dfound[kDeferSlice], dcount;
  if (MODE == 2) {
    if (!gather_deferred(reinterpret_cast<const int*>(rec + (size_t)blockIdx.y * iter_f4) + def_off, tmap.n_tasks, dfound, dcount)) return;
  }
  if (MODE != 1) stage_delta(dl, mv);
  stage_cells(cl, mv, MODE == 1);
  if (MODE != 1 && use_filter == 1)
    for (int i = threadIdx.x; i < (1 << mv.log2filter) / 32; i += kFusedBlock) filt_lds[i] = mv.filter[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves_per_block = kFusedBlock / 64;
  // XCD x takes the x-th eighth of the bundle tasks and of the plain tasks of every iteration (see k_search_tiles)
  const int xcd = blockIdx.x & 7, xb = blockIdx.x >> 3, xnb = ((int)gridDim.x + 7 - xcd) >> 3;
  const int n_rest = tmap.n_tasks - tmap.n_fd;
  const int xb0 = (int)((long long)tmap.n_fd * xcd / 8), xb1 = (int)((long long)tmap.n_fd * (xcd + 1) / 8);
  const int xr0 = (int)((long long)n_rest * xcd / 8), xr1 = (int)((long long)n_rest * (xcd + 1) / 8);
  const int xlen = (xb1 - xb0) + (xr1 - xr0);
  constexpr bool xmap = XMAP && MODE != 2;
  const int w_first = MODE == 2 ? wave : (xmap ? xb * waves_per_block + wave : blockIdx.x * waves_per_block + wave);
  const int w_step = MODE == 2 ? waves_per_block : (xmap ? xnb * waves_per_block : gridDim.x * waves_per_block);
  const int w_total = MODE == 2 ? dcount : (xmap ? xlen * n_iter : tmap.n_tasks * n_iter);
  auto unit_of = [&](int w, int& it, int& task) {
    if (MODE == 2) {
      it = blockIdx.y;
      task = dfound[w];
    } else if (xmap) {
      it = w / xlen;
      const int u = w - it * xlen;
      task = u < xb1 - xb0 ? xb0 + u : tmap.n_fd + xr0 + (u - (xb1 - xb0));
    } else {
      it = w / tmap.n_tasks;
      task = w - it * tmap.n_tasks;
    }
  };
  const long long* index0 = reinterpret_cast<const long long*>(ta.index);
  constexpr bool DMA = MODE == 1 && CLID_SEARCH_DMA;
  __shared__ InStage stages[DMA ? kFusedBlock / 64 : 1];
  if constexpr (DMA) {
    InStage& st = stages[wave];
    auto req_index = [&](int w, int par) {  // A: the task's batch indices
      if (w >= w_total) return;
      int it, task;
      unit_of(w, it, task);
      dma_task_index(st, par, tmap, index0 + (long long)it * index_stride, task);
    };
    req_index(w_first, 0);
    req_index(w_first + w_step, 1);
    wait_all_vmem();
    if (w_first < w_total) dma_task_data(st, 0, ta);
    // the previous task's record and deferred flag: stored BEHIND the next wait (a store in flight at the wait costs its whole
    // write latency there)
    float4* pend = nullptr;
    int* pend_flag = nullptr;
    bool pend_deferred = false;
    auto flush = [&]() {
      if (!pend_flag) return;
      if (!pend_deferred && lane < kRecFloat4) pend[lane] = reinterpret_cast<const float4*>(&heads[wave])[lane];
      if (lane == 0) *pend_flag = pend_deferred ? 1 : 0;
      pend_flag = nullptr;
    };
    int par = 0;
    for (int w = w_first; w < w_total; w += w_step, par ^= 1) {
      int it, task;
      unit_of(w, it, task);
      wait_all_vmem();  // B(this task) and A(next task) arrived (requested a task ago)
      flush();
      if (w + w_step < w_total) dma_task_data(st, par ^ 1, ta);
      req_index(w + 2 * w_step, par);
      const TaskIn in = read_task_inputs(st, par, ta, tmap, task);
      pend_deferred = search_task_body<true>(mv, ta, tmap, dl, task, it, use_filter, filt_lds, heads[wave], cl, in);
      pend = rec + (size_t)it * iter_f4 + (size_t)task * kRecFloat4;
      pend_flag = reinterpret_cast<int*>(rec + (size_t)it * iter_f4) + def_off + task;
      wave_lds_fence();
    }
    flush();
    return;
  }
  for (int w = w_first; w < w_total; w += w_step) {
    int it, task;
    unit_of(w, it, task);
    const long long* index = index0 + (long long)it * index_stride;
    const bool deferred = search_task<MODE == 1>(mv, ta, tmap, dl, index, task, it, use_filter, filt_lds, heads[wave], cl);
    if (MODE == 1 && lane == 0) reinterpret_cast<int*>(rec + (size_t)it * iter_f4)[def_off + task] = deferred ? 1 : 0;
    if (deferred) {
    } else if (lane < kRecFloat4) {
      rec[(size_t)it * iter_f4 + (size_t)task * kRecFloat4 + lane] = reinterpret_cast<const float4*>(&heads[wave])[lane];
    }
    wave_lds_fence();
  }
}

// ---- partial reduction + Adam ---------------------------------------------------------------------------
// torch.optim.Adam._single_tensor_adam, op order as ATen's:
//   m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, 1-b2); denom = sqrt(v)/sqrt(bc2) + eps;
//   p.addcdiv_(m, denom, -lr/bc1)
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

__device__ __forceinline__ float* mlp_param_ptr(float* W1, float* b1, float* W2, float* b2, int i) {
  if (i < CLID_H * CLID_D) return W1 + i;
  if (i < CLID_H * CLID_D + CLID_H) return b1 + (i - CLID_H * CLID_D);
  if (i < CLID_H * CLID_D + 2 * CLID_H) return W2 + (i - CLID_H * CLID_D - CLID_H);
  return b2;
}

// sums of the 16 columns p0 .. p0+15 over the nb partial rows by ONE BLOCK of 256 threads: thread (c = t & 15, rg = t >> 4)
// takes column p0 + c of rows rg, rg + 16, ... with up to 32 loads in flight, so a load instruction of a wave reads 4 rows x
// 64 contiguous bytes (one wave per column read 64 different lines per instruction: 835 x nb lines = 22 MB through L2 at
// nb = 410 for 1.4 MB of partial rows, and k_adam_all's time followed nb: 4.35 / 5.2 / 7.8 us at 256 / 512 / 1024 rows).
// The row groups are folded by two lane exchanges and one trip through LDS; the sums arrive on threads 0..15.
constexpr int kColsPerBlock = 16;
constexpr int kColBlocks = (CLID_MLP_PARAMS + 2 + kColsPerBlock - 1) / kColsPerBlock;
__device__ __forceinline__ float column_sum_block16(const float* __restrict__ partial, int nb, int p0, float* red /* LDS [64] */) {
  const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (p0 + c < kPartialStride) {  // (the last block's columns beyond the row's pad would run into the next row)
    const float* __restrict__ src = partial + p0 + c;
    for (int b0 = rg; b0 < nb; b0 += 16 * 32) {
      float a[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const int b = b0 + 16 * i;
        a[i] = b < nb ? src[(size_t)b * kPartialStride] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i & 3] += a[i];
    }
  }
  float s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  if ((threadIdx.x & 63) < 16) red[(threadIdx.x >> 6) * 16 + c] = s;
  __syncthreads();
  return threadIdx.x < 16 ? (red[c] + red[16 + c]) + (red[32 + c] + red[48 + c]) : 0.f;
}

// eik_inv_n != NULL (config.ekional_add_to "surface" / "freespace"): the eikonal mean runs over that many decimated samples
__device__ __forceinline__ float eik_normaliser(const float* eik_inv_n, float inv_n_eik) {
  return eik_inv_n ? *eik_inv_n : inv_n_eik;
}
__device__ __forceinline__ void finish_loss(int p, float tot, float* loss_out, float inv_n_main, float inv_n_eik,
                                            float weight_e) {
  if (p == CLID_MLP_PARAMS) {
    const float bce = tot * inv_n_main;
    atomicAdd(&loss_out[1], bce);
    atomicAdd(&loss_out[0], bce);
  } else if (p == CLID_MLP_PARAMS + 1) {
    const float eik = tot * inv_n_eik;
    atomicAdd(&loss_out[2], eik);
    atomicAdd(&loss_out[0], weight_e * eik);
  }
}

// ---- touched rows: flags -> bit sets / prefix sums / counts (clid_train_touch_scan) ----------------------------------
// One thread per 32-row word column, over the chunk's iterations: 32 flag bytes -> one bit word, the running union of the
// call, the word's population count (scanned along the row axis by k_touch_scan); the flags are cleared for the next chunk.
__device__ __forceinline__ unsigned flags_to_nibble(unsigned d) {  // 4 flag bytes (0 / 1) -> 4 bits, byte i -> bit i
  return (((d & 0x01010101u) * 0x01020408u) >> 24) & 0xFu;
}
__global__ void __launch_bounds__(64) k_touch_bits(TouchWs tw, long long stride, int n_it, int it0) {
  const long long W = stride / 32;
  const long long w = (long long)blockIdx.x * 64 + threadIdx.x;
  if (w >= W) return;
  unsigned seen = it0 > 0 ? tw.cum[w] : 0u;
  uint8_t* __restrict__ flags = tw.flags;
  for (int b0 = 0; b0 < n_it; b0 += 8) {  // 8 iterations' flag words requested together (16 loads in flight per thread:
    uint4 a[8], b[8];                     // the kernel is a handful of waves deep, i.e. pure load latency)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int it = b0 + k < n_it ? b0 + k : n_it - 1;
      const uint4* f = reinterpret_cast<const uint4*>(flags + (size_t)it * stride + (size_t)w * 32);
      a[k] = f[0];
      b[k] = f[1];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int it = b0 + k;
      if (it >= n_it) break;
      const unsigned word = flags_to_nibble(a[k].x) | (flags_to_nibble(a[k].y) << 4) | (flags_to_nibble(a[k].z) << 8) |
                            (flags_to_nibble(a[k].w) << 12) | (flags_to_nibble(b[k].x) << 16) | (flags_to_nibble(b[k].y) << 20) |
                            (flags_to_nibble(b[k].z) << 24) | (flags_to_nibble(b[k].w) << 28);
      if (word) {
        uint4* f = reinterpret_cast<uint4*>(flags + (size_t)it * stride + (size_t)w * 32);
        f[0] = f[1] = make_uint4(0u, 0u, 0u, 0u);
      }
      seen |= word;
      tw.bits[(size_t)it * W + w] = word;
      tw.cumb[(size_t)it * W + w] = seen;
      tw.wpre[(size_t)it * W + w] = (unsigned)__popc(word);
    }
  }
  tw.cum[w] = seen;
}
// one block per iteration of the chunk: exclusive scan of the word counts in place, total -> counts[it].  A thread takes 8
// consecutive words per round (two 16-byte loads), so a local map of 262 144 rows is ONE round of the block.
__global__ void __launch_bounds__(1024) k_touch_scan(TouchWs tw, long long stride) {
  __shared__ unsigned wsum[16];
  __shared__ unsigned carry;
  const long long W = stride / 32;  // a multiple of 8 (stride is a multiple of 256)
  unsigned* x = tw.wpre + (size_t)blockIdx.x * W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0u;
  __syncthreads();
  for (long long base = 0; base < W; base += 8192) {
    const long long i = base + (long long)threadIdx.x * 8;
    uint4 p = make_uint4(0u, 0u, 0u, 0u), q = p;
    if (i < W) {
      p = *reinterpret_cast<const uint4*>(x + i);
      q = *reinterpret_cast<const uint4*>(x + i + 4);
    }
    const unsigned v = p.x + p.y + p.z + p.w + q.x + q.y + q.z + q.w;
    unsigned inc = v;  // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = (unsigned)__shfl_up((int)inc, d, 64);
      if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned off = carry;
    for (int k = 0; k < wave; ++k) off += wsum[k];
    if (i < W) {
      unsigned e = off + inc - v;
      uint4 po, qo;
      po.x = e; e += p.x; po.y = e; e += p.y; po.z = e; e += p.z; po.w = e; e += p.w;
      qo.x = e; e += q.x; qo.y = e; e += q.y; qo.z = e; e += q.z; qo.w = e;
      *reinterpret_cast<uint4*>(x + i) = po;
      *reinterpret_cast<uint4*>(x + i + 4) = qo;
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry = off + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) tw.counts[blockIdx.x] = (int)carry;
}

// what an iteration's launches read of the touched-row workspace
struct TouchIter {
  const unsigned* bits;  // [stride / 32] rows touched by this iteration (NULL: bookkeeping off)
  const unsigned* cumb;  // [stride / 32] rows touched so far in the call
  const unsigned* wpre;  // [stride / 32]
  const int* count;      // rows touched by this iteration
};
__device__ __forceinline__ int touch_pos(const TouchIter& ti, long long row) {  // position of a touched row in the iteration's list
  const unsigned word = ti.bits[row >> 5];
  return (int)(ti.wpre[row >> 5] + (unsigned)__popc(word & ((1u << (row & 31)) - 1u)));
}

// multi-GPU path: partial[nb][840] -> dst[0:833] (=), loss_out (+=), so the host can all-reduce the gradients.
// Blocks beyond the column blocks (sparse exchange, clid_train_args.cbuf): the accumulation rows this iteration touched are
// packed in ascending row order behind the decoder gradients, [848 | 8 n gradient columns | n certainty increments], and
// zeroed -- the all-reduce then moves 36 bytes per touched row instead of 64 per row of the local map.
__global__ void __launch_bounds__(256)
k_reduce_partials(const float* __restrict__ partial, int nb, float* __restrict__ dst,
                  float* __restrict__ loss_out, float inv_n_main, float inv_n_eik, float weight_e,
                  int train_decoder, TouchIter ti, float* __restrict__ rows, float* __restrict__ cbuf, long long n_rows,
                  const float* __restrict__ eik_inv_n) {
  if ((int)blockIdx.x >= kColBlocks) {
    const long long idx = (long long)((int)blockIdx.x - kColBlocks) * 256 + threadIdx.x;
    const long long row = idx >> 1;
    const int half = (int)(idx & 1);
    if (row >= n_rows) return;
    if (!((ti.bits[row >> 5] >> (row & 31)) & 1u)) return;
    const int pos = touch_pos(ti, row), n = *ti.count;
    float* gr = rows + row * CLID_GRAD_ROW16 + 4 * half;
    *reinterpret_cast<float4*>(cbuf + CLID_GRAD_FEAT_OFFSET16 + (size_t)pos * CLID_F + 4 * half) = *reinterpret_cast<float4*>(gr);
    *reinterpret_cast<float4*>(gr) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (half) {
      cbuf[CLID_GRAD_FEAT_OFFSET16 + (size_t)n * CLID_F + pos] = gr[4];
      gr[4] = 0.f;
    }
    return;
  }
  __shared__ float red[64];
  const int p0 = (int)blockIdx.x * kColsPerBlock, p = p0 + (int)(threadIdx.x & 15);
  if (!train_decoder && p0 + kColsPerBlock <= CLID_MLP_PARAMS) return;  // frozen decoder: only the loss columns carry data
  const float tot = column_sum_block16(partial, nb, p0, red);
  if (threadIdx.x >= 16 || p >= CLID_MLP_PARAMS + 2) return;
  if (p < CLID_MLP_PARAMS) {
    if (train_decoder) dst[p] = tot;
  } else {
    finish_loss(p, tot, loss_out, inv_n_main, eik_normaliser(eik_inv_n, inv_n_eik), weight_e);
  }
}

// config.ekional_add_to "surface" / "freespace" (utils/mapper.py:779-789): size of the iteration's eikonal subset = decimated
// samples (positions first, first + decimation, ...) whose |sdf_label| is below / not below the surface range.  One block per
// iteration of the chunk.
__global__ void __launch_bounds__(256)
k_eik_mask_count(const long long* __restrict__ index, long long index_stride, const float* __restrict__ pool_label, int first,
                 int decimation, int n_fd, int mode, float range, float* __restrict__ inv_count) {
  const long long* idx = index + (long long)blockIdx.x * index_stride;
  int c = 0;
  for (int k = threadIdx.x; k < n_fd; k += 256) {
    const float lab = pool_label[idx[first + (long long)k * decimation]];
    c += ((fabsf(lab) < range) == (mode == 1)) ? 1 : 0;
  }
  __shared__ int red[4];
  c = (int)wave_sum((float)c);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int n = red[0] + red[1] + red[2] + red[3];
    inv_count[blockIdx.x] = n > 0 ? fdiv(1.0f, (float)n) : 0.f;  // (an empty subset contributes nothing)
  }
}

struct AdamLaunch {
  float* feat; float* grad; float* m; float* v; long long n_feat;
  float* W1; float* b1; float* W2; float* b2; float* m_mlp; float* v_mlp;
  const float* partial; int nb;           // non-null: decoder grads / loss sums come from the partial rows
  float* loss_out; float inv_n_main, inv_n_eik, weight_e;
  int train_decoder; int n_feat_blocks;
  int gstride;             // floats per accumulation row of `grad`: CLID_F, or CLID_GRAD_ROW16 (column 8 = certainty increment)
  float* cert; int n_cert;
  AdamK k;
  TouchIter ti;            // ti.bits != NULL: only rows touched so far in the call are visited (16-float rows only)
  const float* cbuf;       // non-null (with ti): gradients / certainty increments / decoder gradients from the compact buffer
  const float* eik_inv_n;  // non-null: 1 / size of this iteration's eikonal subset (config.ekional_add_to != "all")
  int zero_partial;        // 1: `partial` holds the sharded dense exchange's decoder-gradient copies (zeroed behind the sum)
};

// blocks [0, kColBlocks): 16 decoder parameters / loss columns each (reduce partial rows or read grad), Adam
// blocks [kColBlocks, +n_feat_blocks): dense Adam over the feature table (float4), gradient zeroed in the same pass
// (the pointers a wave needs first lead the kernel-argument segment: they arrive in SGPRs with the wave, -amdgpu-kernarg-preload-count)
__global__ void __launch_bounds__(256) k_adam_all(float* feat_, float* grad_, float* m_, float* v_, const float* partial_, int nb_,
                                                  AdamLaunch a) {
  a.feat = feat_; a.grad = grad_; a.m = m_; a.v = v_; a.partial = partial_; a.nb = nb_;
  // the column blocks carry the longer chain (7 loads -> wave sum -> Adam -> store): they are dispatched first
  if ((int)blockIdx.x >= kColBlocks) {
    const long long i4 = ((long long)((int)blockIdx.x - kColBlocks) * 256 + threadIdx.x) * 4;
    float* g = a.grad + CLID_GRAD_OFFSET(a.gstride);
    if (a.gstride == CLID_GRAD_ROW16 && a.ti.bits) {
      // touched-row bookkeeping on: a row the call has not touched yet has g = m = v = 0 and its update is exactly zero --
      // it costs one cached bit test; a row touched earlier but not by this iteration carries momentum only (g = 0: no
      // gradient read, nothing to zero); a row of this iteration reads its gradient from the accumulation row (single GPU)
      // or from the all-reduced compact buffer
      if (i4 >= a.n_feat) return;
      const long long row = i4 >> 3;
      const unsigned sh = (unsigned)(row & 31);
      if (!((a.ti.cumb[row >> 5] >> sh) & 1u)) return;
      const bool now = (a.ti.bits[row >> 5] >> sh) & 1u;
      float4 G = make_float4(0.f, 0.f, 0.f, 0.f);
      float inc = 0.f;
      const bool cert_lane = (i4 & 7) == 4 && a.cert && row < a.n_cert;
      if (now) {
        if (a.cbuf) {
          const int pos = touch_pos(a.ti, row);
          G = *reinterpret_cast<const float4*>(a.cbuf + CLID_GRAD_FEAT_OFFSET16 + (size_t)pos * CLID_F + (i4 & 7));
          if (cert_lane) inc = a.cbuf[CLID_GRAD_FEAT_OFFSET16 + (size_t)(*a.ti.count) * CLID_F + pos];
        } else {
          float* gr = g + row * CLID_GRAD_ROW16 + (i4 & 7);
          G = *reinterpret_cast<float4*>(gr);
          *reinterpret_cast<float4*>(gr) = make_float4(0.f, 0.f, 0.f, 0.f);
          if (cert_lane) {
            inc = gr[4];
            gr[4] = 0.f;
          }
        }
      }
      float4 P = *reinterpret_cast<float4*>(a.feat + i4);
      float4 M = *reinterpret_cast<float4*>(a.m + i4), V = *reinterpret_cast<float4*>(a.v + i4);
      adam_update(P.x, G.x, M.x, V.x, a.k, 0.f);
      adam_update(P.y, G.y, M.y, V.y, a.k, 0.f);
      adam_update(P.z, G.z, M.z, V.z, a.k, 0.f);
      adam_update(P.w, G.w, M.w, V.w, a.k, 0.f);
      *reinterpret_cast<float4*>(a.feat + i4) = P;
      *reinterpret_cast<float4*>(a.m + i4) = M;
      *reinterpret_cast<float4*>(a.v + i4) = V;
      if (inc != 0.f) a.cert[row] += inc;  // np.py:714
      return;
    }
    if (a.gstride == CLID_GRAD_ROW16) {  // 16-float accumulation rows: gradients in columns 0..7, certainty increment in 8
      if (i4 >= a.n_feat) return;
      const long long row = i4 >> 3;
      float* gr = g + row * CLID_GRAD_ROW16 + (i4 & 7);
      float4 G = *reinterpret_cast<float4*>(gr);
      float4 M = *reinterpret_cast<float4*>(a.m + i4), V = *reinterpret_cast<float4*>(a.v + i4);
      const bool idle = a.k.wd == 0.f && G.x == 0.f && G.y == 0.f && G.z == 0.f && G.w == 0.f && M.x == 0.f && M.y == 0.f &&
                        M.z == 0.f && M.w == 0.f && V.x == 0.f && V.y == 0.f && V.z == 0.f && V.w == 0.f;
      if (idle) {
        // never touched since this mapping() call started (the optimiser state restarts per call): the update is exactly
        // zero, nothing to read of the parameters and nothing to write -- 48 instead of 144 bytes per float4 on the
        // (majority of) rows of a large local map that the call's batches do not reach
        if ((i4 & 7) == 4 && a.cert && row < a.n_cert) {
          const float inc = gr[4];
          if (inc != 0.f) {
            a.cert[row] += inc;
            gr[4] = 0.f;
          }
        }
        return;
      }
      float4 P = *reinterpret_cast<float4*>(a.feat + i4);  // (requesting it with the state, before the idle test, gains 0.07 us)
      adam_update(P.x, G.x, M.x, V.x, a.k, a.k.wd);
      adam_update(P.y, G.y, M.y, V.y, a.k, a.k.wd);
      adam_update(P.z, G.z, M.z, V.z, a.k, a.k.wd);
      adam_update(P.w, G.w, M.w, V.w, a.k, a.k.wd);
      *reinterpret_cast<float4*>(a.feat + i4) = P;
      *reinterpret_cast<float4*>(a.m + i4) = M;
      *reinterpret_cast<float4*>(a.v + i4) = V;
      *reinterpret_cast<float4*>(gr) = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((i4 & 7) == 4 && a.cert && row < a.n_cert) {  // the thread next to column 8 merges the certainty increment (np.py:714)
        const float inc = gr[4];
        if (inc != 0.f) {
          a.cert[row] += inc;
          gr[4] = 0.f;
        }
      }
      return;
    }
    if (i4 + 3 < a.n_feat) {
      float4 P = *reinterpret_cast<float4*>(a.feat + i4), G = *reinterpret_cast<float4*>(g + i4);
      float4 M = *reinterpret_cast<float4*>(a.m + i4), V = *reinterpret_cast<float4*>(a.v + i4);
      adam_update(P.x, G.x, M.x, V.x, a.k, a.k.wd);
      adam_update(P.y, G.y, M.y, V.y, a.k, a.k.wd);
      adam_update(P.z, G.z, M.z, V.z, a.k, a.k.wd);
      adam_update(P.w, G.w, M.w, V.w, a.k, a.k.wd);
      *reinterpret_cast<float4*>(a.feat + i4) = P;
      *reinterpret_cast<float4*>(a.m + i4) = M;
      *reinterpret_cast<float4*>(a.v + i4) = V;
      *reinterpret_cast<float4*>(g + i4) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      for (long long i = i4; i < a.n_feat; ++i) {
        float P = a.feat[i], M = a.m[i], V = a.v[i];
        adam_update(P, g[i], M, V, a.k, a.k.wd);
        a.feat[i] = P; a.m[i] = M; a.v[i] = V; g[i] = 0.f;
      }
    }
    return;
  }
  __shared__ float red[64];
  const int p0 = (int)blockIdx.x * kColsPerBlock, p = p0 + (int)(threadIdx.x & 15);
  if (!a.train_decoder && p0 + kColsPerBlock <= CLID_MLP_PARAMS) return;  // frozen decoder: no decoder gradients were produced
  const bool owner = threadIdx.x < 16 && p < CLID_MLP_PARAMS + 2;
  float* dst = nullptr;
  float P = 0.f, M = 0.f, V = 0.f;
  if (owner && p < CLID_MLP_PARAMS && a.train_decoder) {  // parameter and state requested before the column sum, not after it
    dst = mlp_param_ptr(a.W1, a.b1, a.W2, a.b2, p);
    P = *dst; M = a.m_mlp[p]; V = a.v_mlp[p];
  }
  float gsum;
  if (a.partial) {
    gsum = column_sum_block16(a.partial, a.nb, p0, red);
    if (a.zero_partial) {  // (each (column, row) pair was read by the one thread that clears it here; nb <= 16 x 32)
      const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
      if (p0 + c < CLID_MLP_PARAMS + 2)
        for (int b = rg; b < a.nb; b += 16) const_cast<float*>(a.partial)[(size_t)b * kPartialStride + p0 + c] = 0.f;
    }
  }
  else gsum = (owner && p < CLID_MLP_PARAMS) ? (a.cbuf ? a.cbuf[p] : a.grad[p]) : 0.f;
  if (!owner) return;
  if (p < CLID_MLP_PARAMS) {
    if (a.train_decoder) {
      adam_update(P, gsum, M, V, a.k, 0.f);
      *dst = P; a.m_mlp[p] = M; a.v_mlp[p] = V;
    }
    if (!a.cbuf) a.grad[p] = 0.f;  // (the compact buffer is rewritten by the next iteration's pack)
  } else if (a.partial && a.loss_out) {
    finish_loss(p, gsum, a.loss_out, a.inv_n_main, eik_normaliser(a.eik_inv_n, a.inv_n_eik), a.weight_e);
  }
}

// stand-alone elementwise Adam (clid_adam_step)
__global__ void k_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, long long n, AdamK k, int zero_grad) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float P = p[i], M = m[i], V = v[i];
    adam_update(P, g[i], M, V, k, k.wd);
    p[i] = P; m[i] = M; v[i] = V;
    if (zero_grad) g[i] = 0.f;
  }
}

}  // namespace clid

using namespace clid;

// ---- optional per-kernel timing (bench.py roofline leg): a profiler object handed in through clid_train_args.prof ----
#include <cstdlib>
#include <vector>
struct ProfSpan {
  int tag;  // 0 fused / decode kernel, 1 search kernel (hoisted-search loop), 2 partial reduce (+ pack), 3 adam, 5 touch scan
  hipEvent_t a, b;
};
struct clid_prof {
  std::vector<ProfSpan> spans;
};
// CLID_KLAUNCH asks for a (start, stop) event pair: the launch then goes out through hipExtLaunchKernelGGL and the pair
// carries the dispatch's own begin / end time stamps (what rocprofv3 --kernel-trace reports)
bool clid_prof_open(clid_prof* prof, int tag, hipEvent_t* a, hipEvent_t* b) {
  if (!prof) return false;
  ProfSpan sp;
  sp.tag = tag;
  if (hipEventCreate(&sp.a) != hipSuccess) return false;
  if (hipEventCreate(&sp.b) != hipSuccess) {
    hipEventDestroy(sp.a);
    return false;
  }
  prof->spans.push_back(sp);
  *a = sp.a;
  *b = sp.b;
  return true;
}

#ifdef CLID_TIMING
extern "C" int clid_debug_read_stamps(long long* out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(clid::clid_stamps), sizeof(long long) * 256 * 32) == hipSuccess ? 0 : -3;
}
#endif

extern "C" clid_prof* clid_profile_create(void) { return new clid_prof(); }
static void prof_clear(clid_prof* p) {
  for (ProfSpan& sp : p->spans) {
    hipEventDestroy(sp.a);
    hipEventDestroy(sp.b);
  }
  p->spans.clear();
}
extern "C" void clid_profile_destroy(clid_prof* prof) {
  if (!prof) return;
  prof_clear(prof);
  delete prof;
}

// sums of elapsed ms per kernel over all recorded launches: out[0] = fused (or decode) kernel, out[1] = search kernel of
// the hoisted-search loop, out[2] = partial reduce (+ pack), out[3] = adam, out[5] = touched-row scan; out[4] = back-to-back
// event-pair overhead (ms, mean) measured now; *iters = decode launches recorded.  The spans are dropped.
extern "C" int clid_profile_read(clid_prof* prof, double* out, int* iters, void* stream) {
  if (!prof || !out || !iters) {
    clid_set_error("clid_profile_read: null argument");
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  if (hipDeviceSynchronize() != hipSuccess) return CLID_E_HIP;
  for (int i = 0; i < 6; ++i) out[i] = 0.0;
  int n = 0;
  for (const ProfSpan& sp : prof->spans) {
    float ms = 0.f;
    if (sp.tag < 0 || sp.tag > 5 || sp.tag == 4) continue;
    if (hipEventElapsedTime(&ms, sp.a, sp.b) != hipSuccess) continue;
    out[sp.tag] += ms;
    n += sp.tag == 0;
  }
  *iters = n;
  prof_clear(prof);
  // marginal cost of an (event, event) bracket in a BUSY stream: 64 consecutive records, one sync
  {
    hipEvent_t ev[64];
    for (int r = 0; r < 64; ++r) hipEventCreate(&ev[r]);
    for (int r = 0; r < 64; ++r) hipEventRecord(ev[r], s);
    hipEventSynchronize(ev[63]);
    double acc = 0.0;
    for (int r = 8; r + 1 < 64; r += 2) {
      float ms;
      hipEventElapsedTime(&ms, ev[r], ev[r + 1]);
      acc += ms;
    }
    out[4] = acc / 28.0;
    for (int r = 0; r < 64; ++r) hipEventDestroy(ev[r]);
  }
  return CLID_OK;
}

static int n_queries(const clid_train_args* a, int* n_fd, int* first) {
  *first = fd_first(a->batch_offset, a->decimation);
  *n_fd = (a->eikonal_mode == 1) ? fd_count(a->bs, a->batch_offset, a->decimation) : 0;
  return a->bs + 6 * (*n_fd);
}

// Grid of the hoisted search launch (grid-stride over chunk x tasks): exactly the blocks that are resident at once
// (CUs x 4 SIMDs x CLID_SEARCH_WAVES / waves per block = 768 on MI355X), so every block stages the 32 KB prefilter into
// LDS once per launch.  A 2048-block grid re-staged it 2.7 times: 42.5 -> 39.2 us per iteration in the mapping(10) regime.
static int search_blocks(int waves_per_simd = CLID_SEARCH_WAVES) {
  static thread_local int dev_cached = -1, cus_cached = 0;  // (a cache of a device attribute, not state of the loop)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 128 * waves_per_simd;
  if (dev != dev_cached) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    cus_cached = cus;
    dev_cached = dev;
  }
  return cus_cached * 4 * waves_per_simd / (kFusedBlock / 64);
}

static int fused_blocks(int n_tasks, int block = kFusedBlock) {
  int nb = (n_tasks + block / 64 - 1) / (block / 64);
  return nb > kMaxBwdBlocks ? kMaxBwdBlocks : (nb < 1 ? 1 : nb);
}

// the tile kernels cover the numerical / no-eikonal modes on 16-float accumulation rows; everything else runs on the
// 16-lanes-per-query kernel
static int decode_variant_for(const clid_train_args* a) {
  const int v = a->decode_variant < 0 ? 0 : (a->decode_variant > 2 ? 2 : a->decode_variant);
  if (v == 0 || a->eikonal_mode == 2 || a->decode_each_neighbour || a->grad_stride != CLID_GRAD_ROW16) return 0;
  return v;
}
extern "C" int clid_train_decode_kernel(const clid_map_view* mv, const clid_train_args* a) {
  return (mv && a) ? decode_variant_for(a) : CLID_E_ARG;
}
static bool hoisted(const clid_train_args* a) { return a->pipeline != 0; }  // (analytic mode too since round 4: records of plain tasks)
// Rows of per-block partials the forward/backward launch of an iteration leaves in the workspace -- a function of the
// arguments alone (ABI 2 handed it from clid_train_decode to clid_train_adam through a thread_local): the analytic kernel,
// the tile kernels (hoisted schedule) and the 16-lane kernel each have their own grid rule.
static int partial_rows(const clid_train_args* a) {
  if (a->eikonal_mode == 2) return clid_train_analytic_blocks(a->bs);
  int n_fd, first;
  n_queries(a, &n_fd, &first);
  const TaskMap tmap = make_task_map(a->bs, n_fd, first, a->decimation);
  if (a->decode_each_neighbour) return clid_train_wf0_rows(tmap.n_tasks, n_fd);
  if (hoisted(a) && decode_variant_for(a)) return clid_decode_tile_blocks(tmap.n_tasks);
  return fused_blocks(tmap.n_tasks);
}

extern "C" int32_t clid_train_partial_rows(const clid_train_args* t) { return t ? partial_rows(t) : 0; }

// ---- config.consistency_loss_on (utils/mapper.py:770-776): 1 - cos(g[near_index[j]], g_near[j]), mean over j -------------------
// F.cosine_similarity divides each vector by max(norm, eps) first; d cos / d a = (bh - cos ah) / |a| above the clamp, bh / eps below
__global__ void __launch_bounds__(256)
k_consistency_couple(const float* __restrict__ g_main, const float* __restrict__ g_near, const long long* __restrict__ near_index,
                     int n_c, int n_main, float weight_c, float* __restrict__ c_main, float* __restrict__ c_near,
                     float* __restrict__ loss_out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  float term = 0.f;
  if (j < n_c) {
    long long i = near_index[j];
    i = i < 0 ? 0 : (i >= n_main ? n_main - 1 : i);
    const float ax = g_main[i * 3 + 0], ay = g_main[i * 3 + 1], az = g_main[i * 3 + 2];
    const float bx = g_near[(size_t)j * 3 + 0], by = g_near[(size_t)j * 3 + 1], bz = g_near[(size_t)j * 3 + 2];
    const float na = sqrtf(ax * ax + ay * ay + az * az), nb = sqrtf(bx * bx + by * by + bz * bz);
    const float ia = 1.0f / fmaxf(na, 1e-8f), ib = 1.0f / fmaxf(nb, 1e-8f);
    const float hax = ax * ia, hay = ay * ia, haz = az * ia, hbx = bx * ib, hby = by * ib, hbz = bz * ib;
    const float cs = hax * hbx + hay * hby + haz * hbz;
    term = 1.0f - cs;
    const float k = -weight_c / (float)n_c;  // dL/dcos
    const float ka = na > 1e-8f ? cs : 0.f, kb = nb > 1e-8f ? cs : 0.f;
    atomicAdd(&c_main[i * 3 + 0], k * (hbx - ka * hax) * ia);
    atomicAdd(&c_main[i * 3 + 1], k * (hby - ka * hay) * ia);
    atomicAdd(&c_main[i * 3 + 2], k * (hbz - ka * haz) * ia);
    c_near[(size_t)j * 3 + 0] = k * (hax - kb * hbx) * ib;
    c_near[(size_t)j * 3 + 1] = k * (hay - kb * hby) * ib;
    c_near[(size_t)j * 3 + 2] = k * (haz - kb * hbz) * ib;
  }
  __shared__ float red[4];
  term = wave_sum(term);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = term;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float m = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n_c;
    atomicAdd(&loss_out[3], m);
    atomicAdd(&loss_out[0], weight_c * m);
  }
}
extern "C" int clid_consistency_couple(const float* g_main, const float* g_near, const int64_t* near_index, int32_t n_c, int32_t n_main,
                                       float weight_c, float* c_main, float* c_near, float* loss_out, void* stream) {
  if (!g_main || !g_near || !near_index || n_c <= 0 || n_main <= 0 || !c_main || !c_near || !loss_out) {
    clid_set_error("clid_consistency_couple: bad argument");
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(c_main, 0, sizeof(float) * 3 * (size_t)n_main, s) != hipSuccess) {
    clid_set_error("clid_consistency_couple: %s", hipGetErrorString(hipGetLastError()));
    return CLID_E_HIP;
  }
  hipLaunchKernelGGL(k_consistency_couple, dim3((unsigned)((n_c + 255) / 256)), dim3(256), 0, s, g_main, g_near,
                     reinterpret_cast<const long long*>(near_index), (int)n_c, (int)n_main, weight_c, c_main, c_near, loss_out);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

extern "C" int64_t clid_train_workspace_floats(int32_t bs, int32_t decimation, int32_t eikonal_mode) {
  if (bs <= 0 || decimation <= 0) return -1;
  const long long nfd = eikonal_mode == 1 ? (bs + decimation - 1) / decimation : 0;
  const long long Q = bs + 6 * nfd;
  return (long long)kMaxBwdBlocks * kPartialStride + (long long)rec_buffer_floats((int)Q) + 64;
}

static float eik_weight(const clid_train_args* a, int n_fd) {
  return (a->eikonal_mode == 2 || (a->eikonal_mode == 1 && n_fd > 0)) ? a->weight_e : 0.f;
}

// ---- touched-row workspace --------------------------------------------------------------------------------------------
extern "C" int64_t clid_touch_stride(int32_t M) { return M < 0 ? -1 : (int64_t)touch_stride_of(M); }
extern "C" int64_t clid_touch_workspace_bytes(int32_t M, int32_t chunk_iters) {
  if (M < 0 || chunk_iters < 1 || chunk_iters > kMaxChunkIters) return -1;
  return (int64_t)((touch_bytes(touch_stride_of(M), chunk_iters) + 255) & ~size_t(255));
}
extern "C" int64_t clid_train_search_floats(int32_t bs, int64_t batch_offset, int32_t decimation,
                                            int32_t eikonal_mode, int32_t n_iter) {
  if (bs <= 0 || decimation <= 0 || n_iter < 0) return -1;
  const int first = fd_first(batch_offset, decimation);
  const int n_fd = eikonal_mode == 1 ? fd_count(bs, batch_offset, decimation) : 0;
  return (int64_t)rec_floats_per_iter(make_task_map(bs, n_fd, first, decimation).n_tasks) * n_iter;
}
extern "C" int32_t clid_train_search_tasks(int32_t bs, int64_t batch_offset, int32_t decimation, int32_t eikonal_mode) {
  if (bs <= 0 || decimation <= 0) return -1;
  const int first = fd_first(batch_offset, decimation);
  const int n_fd = eikonal_mode == 1 ? fd_count(bs, batch_offset, decimation) : 0;
  return make_task_map(bs, n_fd, first, decimation).n_tasks;
}
extern "C" int32_t clid_train_chunk_iters(const clid_train_args* a) {
  if (!a || a->bs <= 0 || a->decimation <= 0) return 0;
  int n_fd, first;
  const int Q = n_queries(a, &n_fd, &first);
  const size_t per_iter = (size_t)clid_train_search_floats(a->bs, a->batch_offset, a->decimation, a->eikonal_mode, 1);
  long long chunk = (long long)(rec_buffer_floats(Q) / per_iter);
  if (chunk > kMaxChunkIters) chunk = kMaxChunkIters;
  return chunk < 1 ? 0 : (int32_t)chunk;
}
// the iteration's view of the workspace (bookkeeping off: all NULL)
static TouchIter touch_iter_of(const clid_train_args* a) {
  TouchIter ti{nullptr, nullptr, nullptr, nullptr};
  if (!a || !a->touch_ws || !hoisted(a)) return ti;
  const int chunk = clid_train_chunk_iters(a);
  const TouchWs tw = touch_carve(a->touch_ws, a->touch_stride, chunk);
  const size_t W = (size_t)(a->touch_stride / 32);
  ti.bits = tw.bits + (size_t)a->touch_iter * W;
  ti.cumb = tw.cumb + (size_t)a->touch_iter * W;
  ti.wpre = tw.wpre + (size_t)a->touch_iter * W;
  ti.count = tw.counts + a->touch_iter;
  return ti;
}
static int check_touch(const clid_train_args* a, int M, const char* who) {
  if (!a->touch_ws) return CLID_OK;
  if (a->touch_stride < touch_stride_of(M) || (a->touch_stride & 255) || ((uintptr_t)a->touch_ws & 15) ||
      a->grad_stride != CLID_GRAD_ROW16 || a->touch_iter < 0 || a->touch_iter >= kMaxChunkIters) {
    clid_set_error("%s: touched-row workspace needs touch_stride >= clid_touch_stride(M) = %lld in multiples of 256 (got %lld), "
                   "16-byte alignment, 16-float accumulation rows and 0 <= touch_iter < %d", who, (long long)touch_stride_of(M),
                   (long long)a->touch_stride, kMaxChunkIters);
    return CLID_E_ARG;
  }
  return CLID_OK;
}

extern "C" int clid_train_touch_scan(const clid_train_args* a, int32_t M, int32_t n_it, int32_t it0, int32_t* counts_host,
                                     void* stream) {
  if (!a || !a->touch_ws || M < 0 || n_it < 1 || it0 < 0) {
    clid_set_error("clid_train_touch_scan: bad argument");
    return CLID_E_ARG;
  }
  if (int e = check_touch(a, M, "clid_train_touch_scan")) return e;
  const int chunk = clid_train_chunk_iters(a);
  if (n_it > chunk) {
    clid_set_error("clid_train_touch_scan: %d iterations exceed the chunk of %d", n_it, chunk);
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  const TouchWs tw = touch_carve(a->touch_ws, a->touch_stride, chunk);
  const long long W = a->touch_stride / 32;
  CLID_KLAUNCH(a->prof, 5, k_touch_bits, dim3((unsigned)((W + 63) / 64)), dim3(64), 0, s, tw, (long long)a->touch_stride,
               (int)n_it, (int)it0);
  CLID_KLAUNCH(a->prof, 5, k_touch_scan, dim3((unsigned)n_it), dim3(1024), 0, s, tw, (long long)a->touch_stride);
  CLID_CHECK_LAUNCH();
  if (counts_host)  // (kMaxChunkIters x 4 bytes: fits the pinned landing buffer; synchronises the stream)
    if (int e = clid_read_back(tw.counts, (int32_t)sizeof(int32_t) * n_it, counts_host, stream)) return e;
  return CLID_OK;
}

// partial reduction (+ pack of the touched accumulation rows into the compact exchange buffer) in front of an all-reduce
static int launch_reduce(const clid_map_view* mv, const clid_train_args* a, const float* partial, int nb, int n_fd,
                         hipStream_t s) {
  const TouchIter ti = a->cbuf ? touch_iter_of(a) : TouchIter{nullptr, nullptr, nullptr, nullptr};
  const bool pack = ti.bits != nullptr;
  const long long n_rows = pack ? (long long)mv->M + 1 : 0;
  const unsigned row_blocks = pack ? (unsigned)((n_rows * 2 + 255) / 256) : 0u;
  CLID_KLAUNCH(a->prof, 2, k_reduce_partials, dim3(kColBlocks + row_blocks), dim3(256), 0, s, partial, nb,
               pack ? a->cbuf : a->grad, a->loss_out, a->inv_n_main, a->inv_n_eik, eik_weight(a, n_fd), a->train_decoder, ti,
               a->grad + CLID_GRAD_FEAT_OFFSET16, a->cbuf, n_rows,
               (const float*)((a->eik_mask && a->eik_inv_n) ? a->eik_inv_n + a->touch_iter : nullptr));
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

extern "C" int clid_train_fwd_bwd(const clid_map_view* mv, const clid_train_args* a, void* stream) {
  if (!mv || !a || !mv->tab || !mv->feat || !mv->cert || !a->index || !a->grad || !a->ws || !a->loss_out) {
    clid_set_error("clid_train_fwd_bwd: null argument");
    return CLID_E_ARG;
  }
  if (mv->M >= (1 << kProbeShift)) {
    clid_set_error("clid_train_fwd_bwd: a local map of %d points exceeds the %d the search's packed candidates address", mv->M, 1 << kProbeShift);
    return CLID_E_SHAPE;
  }
  if (mv->P > kMaxProbes) {
    clid_set_error("clid_train_fwd_bwd: neighbourhood of %d cells exceeds the supported %d", mv->P, kMaxProbes);
    return CLID_E_SHAPE;
  }
  if (a->bs <= 0 || a->decimation <= 0) {
    clid_set_error("clid_train_fwd_bwd: bs=%d decimation=%d", a->bs, a->decimation);
    return CLID_E_ARG;
  }
  if (a->decode_each_neighbour) {
    clid_set_error("clid_train_fwd_bwd: decode_each_neighbour (weighted_first: False) runs on the hoisted schedule only");
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  int n_fd, first;
  const int Q = n_queries(a, &n_fd, &first);
  TrainWs ws = carve(a->ws, Q);
  const TaskMap tmap = make_task_map(a->bs, n_fd, first, a->decimation);
  clid_train_args fa = *a;
  fa.pipeline = 0;  // this entry IS the fused schedule (partial_rows / touched-row bookkeeping follow it)
  const int nb = partial_rows(&fa);
  if (a->eikonal_mode == 2) {  // loss.numerical_grad_on: False (utils/mapper.py:57-69, 660-661, 695-696)
    if (int e = clid_launch_train_analytic(mv, a, ws.partial, nullptr, s)) return e;
  } else {
    CLID_KLAUNCH(a->prof, 0, k_train_fused8<0>, dim3(nb), dim3(kFusedBlock), 0, s, *mv, *a, ws.partial, tmap, (float4*)nullptr);
    CLID_CHECK_LAUNCH();
  }
  if (!a->defer_reduce)
    if (int e = launch_reduce(mv, &fa, ws.partial, nb, n_fd, s)) return e;
  return CLID_OK;
}

static AdamK adam_scalars(float lr, float b1, float b2, float eps, float wd, int step) {
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  AdamK k;
  k.one_m_b1 = (float)(1.0 - (double)b1);
  k.b2 = b2;
  k.one_m_b2 = (float)(1.0 - (double)b2);
  k.bc2_sqrt = (float)sqrt(bc2);
  k.eps = eps;
  k.neg_step = (float)(-((double)lr / bc1));
  k.wd = wd;
  return k;
}

extern "C" int clid_adam_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int32_t step, int32_t zero_grad,
                              void* stream) {
  if (!p || !g || !m || !v || n < 0 || step < 1) {
    clid_set_error("clid_adam_step: bad argument");
    return CLID_E_ARG;
  }
  if (n == 0) return CLID_OK;
  long long nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(k_adam, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n,
                     adam_scalars(lr, beta1, beta2, eps, weight_decay, step), zero_grad);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

extern "C" int clid_train_adam(const clid_adam_args* a, const clid_train_args* t, void* stream) {
  if (!a || !a->feat || !a->grad || !a->m || !a->v || a->step < 1) {
    clid_set_error("clid_train_adam: bad argument");
    return CLID_E_ARG;
  }
  if (a->train_decoder && (!a->W1 || !a->b1 || !a->W2 || !a->b2 || !a->m_mlp || !a->v_mlp)) {
    clid_set_error("clid_train_adam: decoder tensors missing");
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  AdamLaunch L;
  L.feat = a->feat; L.grad = a->grad; L.m = a->m; L.v = a->v; L.n_feat = a->n_feat;
  L.W1 = a->W1; L.b1 = a->b1; L.W2 = a->W2; L.b2 = a->b2; L.m_mlp = a->m_mlp; L.v_mlp = a->v_mlp;
  L.partial = nullptr; L.nb = 0; L.loss_out = nullptr; L.inv_n_main = L.inv_n_eik = L.weight_e = 0.f;
  L.ti = TouchIter{nullptr, nullptr, nullptr, nullptr};
  L.cbuf = nullptr;
  L.eik_inv_n = (t && t->eik_mask && t->eik_inv_n) ? t->eik_inv_n + t->touch_iter : nullptr;
  if (t && t->defer_reduce) {  // single-GPU: fold the partial reduction of the forward/backward launch of `t` into this one
    int n_fd, first;
    const int Q = n_queries(t, &n_fd, &first);
    TrainWs ws = carve(t->ws, Q);
    L.partial = ws.partial;
    L.nb = partial_rows(t) + (t->partial_rows_extra > 0 ? t->partial_rows_extra : 0);  // (+ a second batch's rows behind them)
    L.loss_out = t->loss_out;
    L.inv_n_main = t->inv_n_main;
    L.inv_n_eik = t->inv_n_eik;
    L.weight_e = eik_weight(t, n_fd);
  }
  L.zero_partial = 0;
  if (t && !t->defer_reduce && t->dec_copies && !t->cbuf && decode_variant_for(t) && t->eikonal_mode != 2 && !t->decode_each_neighbour) {
    // sharded dense exchange (ABI 8): the all-reduced decoder gradients sit in n_dec_copies copies the decode blocks added to --
    // summed per column like partial rows, and zeroed for the next iteration's adds (the losses went to loss_out directly)
    // The two loss columns hold the RAW sums over all ranks: every rank adds its 1 / dec_ranks share of the normalised total to
    // its loss_out, and the per-call SUM of the ranks' losses (clid_mapping_run_dist / Mapper.mapping) yields the total.
    if (t->dec_ranks < 1) {
      clid_set_error("clid_train_adam: dec_copies needs dec_ranks >= 1 (the ranks the copies were summed over)");
      return CLID_E_ARG;
    }
    int n_fd, first;
    n_queries(t, &n_fd, &first);
    L.partial = t->dec_copies;
    L.nb = t->n_dec_copies;
    L.loss_out = t->loss_out;
    L.inv_n_main = t->inv_n_main / (float)t->dec_ranks;
    L.inv_n_eik = t->inv_n_eik / (float)t->dec_ranks;
    L.weight_e = t->weight_e;  // (a rank without decimated samples of its own still holds its share of the global term)
    L.zero_partial = 1;
  }
  L.train_decoder = a->train_decoder;
  L.gstride = a->grad_stride == CLID_GRAD_ROW16 ? CLID_GRAD_ROW16 : CLID_F;
  L.cert = a->cert;
  L.n_cert = a->cert ? a->n_cert : 0;
  if (L.gstride == CLID_GRAD_ROW16 && (a->n_feat % CLID_F) != 0) {
    clid_set_error("clid_train_adam: n_feat=%lld is not a whole number of rows", (long long)a->n_feat);
    return CLID_E_ARG;
  }
  if (t && t->touch_ws && hoisted(t)) {  // visit only the rows this mapping() call has touched so far
    if (int e = check_touch(t, (int)(a->n_feat / CLID_F) - 1, "clid_train_adam")) return e;
    if (a->weight_decay != 0.f || L.gstride != CLID_GRAD_ROW16) {
      clid_set_error("clid_train_adam: the touched-row sweep needs weight_decay == 0 (every row moves otherwise) and 16-float rows");
      return CLID_E_ARG;
    }
    L.ti = touch_iter_of(t);
    if (!t->defer_reduce) L.cbuf = t->cbuf;  // (sharded: the all-reduced compact buffer; NULL = the caller reduced `grad`)
  }
  L.n_feat_blocks = (int)(((a->n_feat + 3) / 4 + 255) / 256);
  L.k = adam_scalars(a->lr, a->beta1, a->beta2, a->eps, a->weight_decay, a->step);
  CLID_KLAUNCH(t ? t->prof : nullptr, 3, k_adam_all, dim3(L.n_feat_blocks + kColBlocks), dim3(256), 0, s, L.feat, L.grad, L.m, L.v,
               L.partial, L.nb, L);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

// ---- the hoisted-search loop ----------------------------------------------------------------------------
// Within one Mapper.mapping call the map's positions, the voxel table and the drawn sample indices are all
// fixed (training moves features and decoder weights only), so the neighbour searches of ALL iterations are
// independent of the training state.  They are hoisted out of the dependent chain: one large search launch
// per chunk of iterations (a grid that fills the chip and runs at the memory system's gather rate, unlike a
// single iteration's 3.3 k latency-bound waves) parks the winners in HBM, then the chunk's decode/backward +
// Adam launches run back to back.
static bool filter_enabled() {  // CLID_FILTER=0 turns the probe prefilter off (measurement aid; results are unchanged)
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("CLID_FILTER");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  return on != 0;
}

static int check_train_args(const clid_map_view* mv, const clid_train_args* a, const char* who) {
  if (!mv || !a || !mv->tab || !mv->tab_pos || !mv->delta) {
    clid_set_error("%s: null argument", who);
    return CLID_E_ARG;
  }
  if (mv->M >= (1 << kProbeShift)) {
    clid_set_error("%s: a local map of %d points exceeds the %d the search's packed candidates address", who, mv->M, 1 << kProbeShift);
    return CLID_E_SHAPE;
  }
  if (mv->P > kMaxProbes) {
    clid_set_error("%s: neighbourhood of %d cells exceeds the supported %d", who, mv->P, kMaxProbes);
    return CLID_E_SHAPE;
  }
  if (a->bs <= 0 || a->decimation <= 0 || (a->eikonal_mode == 2 && a->decimation != 1)) {
    clid_set_error("%s: bs=%d decimation=%d eikonal_mode=%d", who, a->bs, a->decimation, a->eikonal_mode);
    return CLID_E_ARG;
  }
  if (a->decode_each_neighbour && (a->cbuf || (a->eikonal_mode == 2 && !hoisted(a)))) {
    clid_set_error("%s: decode_each_neighbour (weighted_first: False) needs the hoisted schedule and the plain exchange", who);
    return CLID_E_ARG;
  }
  if (a->main_loss_type) {
    if (a->main_loss_type < 0 || a->main_loss_type > 3 || a->eikonal_mode == 2 || a->decode_each_neighbour || !hoisted(a) ||
        !decode_variant_for(a)) {
      clid_set_error("%s: main_loss_type %d (config.main_loss_type sdf_l1 / sdf_l2 / zhong) needs the hoisted schedule, a tile "
                     "decode kernel, eikonal mode 0 or 1 and weighted_first", who, a->main_loss_type);
      return CLID_E_ARG;
    }
  }
  if (a->g_out || a->c_extra || a->partial_row0 || a->partial_rows_extra) {
    if (a->eikonal_mode != 2 || !hoisted(a) || a->decode_each_neighbour || (a->g_out && a->c_extra) || a->partial_rows_extra < 0) {
      clid_set_error("%s: g_out / c_extra / partial_row0 / partial_rows_extra (config.consistency_loss_on) belong to the analytic "
                     "iteration on the hoisted schedule with weighted_first", who);
      return CLID_E_ARG;
    }
  }
  if (a->proj_correction) {
    if (a->eikonal_mode != 2 || !hoisted(a) || a->decode_each_neighbour || !a->frame_pose || a->n_frame_pose <= 0 ||
        ((uintptr_t)a->frame_pose & 15) != 0 || a->main_loss_type) {
      clid_set_error("%s: proj_correction (config.proj_correction_on) needs the analytic eikonal term on the hoisted schedule, "
                     "weighted_first, the BCE loss and frame_pose [n_frame_pose][12] (16-byte aligned)", who);
      return CLID_E_ARG;
    }
  }
  if (a->pool_pose) {
    if (!a->pool_ts || a->n_pose <= 0 || !hoisted(a) || ((uintptr_t)a->pool_pose & 15) != 0) {
      clid_set_error("%s: pool_pose (Mapper.ba_done_flag) needs pool_ts, n_pose > 0, 16-byte alignment and the hoisted schedule "
                     "(the poses are applied by clid_train_search)", who);
      return CLID_E_ARG;
    }
  }
  if (a->eik_mask) {
    if (a->eik_mask < 0 || a->eik_mask > 2 || !a->eik_inv_n || a->eikonal_mode != 1 || a->decode_each_neighbour ||
        !hoisted(a) || !decode_variant_for(a) || a->cbuf || a->touch_iter < 0 || a->touch_iter >= kMaxChunkIters) {
      clid_set_error("%s: eik_mask (config.ekional_add_to surface / freespace) needs eik_inv_n, the numerical eikonal term, the "
                     "hoisted schedule, a tile decode kernel and one rank", who);
      return CLID_E_ARG;
    }
  }
  return check_touch(a, mv->M, who);
}

// sizes of the chunk's eikonal subsets (config.ekional_add_to != "all"): eik_inv_n[it_rel0 + i], i < n_iter
static int launch_eik_inv_n(const clid_train_args* a, int n_iter, const int64_t* index_base, int64_t index_stride, int it_rel0,
                            hipStream_t s) {
  if (!a->eik_mask) return CLID_OK;
  if (it_rel0 < 0 || it_rel0 + n_iter > kMaxChunkIters) {
    clid_set_error("clid_train_search: eik_inv_n holds %d iterations", kMaxChunkIters);
    return CLID_E_ARG;
  }
  int n_fd, first;
  n_queries(a, &n_fd, &first);
  hipLaunchKernelGGL(k_eik_mask_count, dim3((unsigned)n_iter), dim3(256), 0, s, reinterpret_cast<const long long*>(index_base),
                     (long long)index_stride, a->pool_label, first, a->decimation, n_fd, a->eik_mask, a->eik_mask_range,
                     a->eik_inv_n + it_rel0);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

static int train_search_impl(const clid_map_view* mv, const clid_train_args* a, int32_t n_iter, const int64_t* index_base,
                             int64_t index_stride, float* rec_out, void* stream, int grid_cap);
extern "C" int clid_train_search(const clid_map_view* mv, const clid_train_args* a, int32_t n_iter,
                                 const int64_t* index_base, int64_t index_stride, float* rec_out, void* stream) {
  if (int e = check_train_args(mv, a, "clid_train_search")) return e;
  if (n_iter > 0 && index_base && a->pool_label)
    if (int e = launch_eik_inv_n(a, n_iter, index_base, index_stride, 0, (hipStream_t)stream)) return e;
  return train_search_impl(mv, a, n_iter, index_base, index_stride, rec_out, stream, 0);
}
// grid_cap > 0: at most that many blocks (the side launches of the overlapped schedule leave wave slots to the chain)
static int train_search_impl(const clid_map_view* mv, const clid_train_args* a, int32_t n_iter, const int64_t* index_base,
                             int64_t index_stride, float* rec_out, void* stream, int grid_cap) {
  if (int e = check_train_args(mv, a, "clid_train_search")) return e;
  if (n_iter <= 0 || !index_base || !rec_out || !a->pool_coord) {
    clid_set_error("clid_train_search: bad argument");
    return CLID_E_ARG;
  }
  if (a->touch_ws && n_iter > clid_train_chunk_iters(a)) {
    clid_set_error("clid_train_search: %d iterations exceed the touched-row workspace's chunk of %d", n_iter, clid_train_chunk_iters(a));
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  int n_fd, first;
  n_queries(a, &n_fd, &first);
  const TaskMap tmap = make_task_map(a->bs, n_fd, first, a->decimation);
  if ((long long)tmap.n_tasks * n_iter > 0x7fffffffLL) {
    clid_set_error("clid_train_search: %d tasks x %d iterations overflow", tmap.n_tasks, n_iter);
    return CLID_E_SHAPE;
  }
  clid_train_args t2 = *a;
  t2.index = index_base;
  // probe prefilter (a one-hash Bloom filter over the stored slots; 59 of the 81 probes of a typical query hit nothing):
  // staged in LDS when it fits (<= 32 KB) and the launch is large enough to amortise staging it per block; for large
  // local maps (> 2^17 points: the filter is up to 2 MB, the key table 8+ MB) it is read from global memory, where it
  // stays L2-resident while the bucket loads it saves would each touch a line of the far larger table
  int use_filter = 0;
  if (mv->filter && mv->log2filter >= 10 && filter_enabled()) {
    if (mv->log2filter <= 18) use_filter = 1;
    else use_filter = 2;
  }
  // the window's cell directory (csrc/celldir.hip) when the view carries one: a stencil row is one load + bit tests
  // instead of 2 nc + 1 probes; debug bit 3 keeps the probe kernels (A/B, tests)
  const bool big_map = use_filter == 2;
  if (mv->cdir_hdr && mv->cdir_words && mv->cdir_pos && mv->stencil_rows && mv->stencil_nc >= 1 && mv->stencil_nc <= 2 &&
      mv->P <= kCdHits && !(a->debug_flags & 8))
    use_filter = 3;
  const bool cdir = use_filter == 3;
  if (cdir) use_filter = big_map ? 2 : (mv->filter && mv->log2filter >= 10 && mv->log2filter <= 18 && filter_enabled() ? 1 : 0);
  const size_t dyn_probe = use_filter == 1 ? ((size_t)1 << mv->log2filter) / 8 : 0;
  const size_t dyn_cells = (size_t)(kFusedBlock / 64) * 8 * kCdHits * sizeof(int);
  // iterations of at most kTileLargeFrom tiles: tasks in pairs, the tile's pairs numbered for the decode launch (one tile
  // per wave there); larger ones: per task, the decode kernel numbers in place (train_common.hpp tiles_prenumbered)
  const bool num = clid_tiles_prenumbered(tmap.n_tasks, mv);
  long long sb = num ? ((long long)((tmap.n_tasks + 1) / 2) * n_iter + kFusedBlock / 64 - 1) / (kFusedBlock / 64)
                     : ((long long)tmap.n_tasks * n_iter + kFusedBlock / 64 - 1) / (kFusedBlock / 64);
  const int resident = search_blocks(cdir ? (num ? CLID_CD_WAVES_TILES : CLID_CD_WAVES_TASKS) : CLID_SEARCH_WAVES);
  if (sb > resident) sb = resident;
  if (grid_cap > 0 && sb > grid_cap) sb = grid_cap;
  const bool xm = CLID_XCD_MAP && big_map && sb >= 8;
#define CLID_SEARCH_LAUNCH(K, GRID, DYN)                                                                                   \
  CLID_KLAUNCH(a->prof, 1, K, GRID, dim3(kFusedBlock), DYN, s, *mv, t2, tmap, reinterpret_cast<float4*>(rec_out), n_iter, \
               (long long)index_stride, use_filter)
  if (cdir) {
    // every unit of the directory launch writes its deferred flag; the probing launch behind it shares the flags out
    const int units = num ? (tmap.n_tasks + 1) / 2 : tmap.n_tasks;
    const dim3 g1((unsigned)sb), g2((unsigned)((units + kDeferSlice - 1) / kDeferSlice), (unsigned)n_iter);
    if (num && xm) CLID_SEARCH_LAUNCH((k_search_tiles<true, 1>), g1, dyn_cells);
    else if (num) CLID_SEARCH_LAUNCH((k_search_tiles<false, 1>), g1, dyn_cells);
    else if (xm) CLID_SEARCH_LAUNCH((k_search_tasks<true, 1>), g1, dyn_cells);
    else CLID_SEARCH_LAUNCH((k_search_tasks<false, 1>), g1, dyn_cells);
    if (num) {
      if (!CLID_SEARCH_INLINE_PROBE) CLID_SEARCH_LAUNCH((k_search_tiles<false, 2>), g2, dyn_probe);
    } else CLID_SEARCH_LAUNCH((k_search_tasks<false, 2>), g2, dyn_probe);
  } else {
    const dim3 g1((unsigned)sb);
    if (num && xm) CLID_SEARCH_LAUNCH((k_search_tiles<true, 0>), g1, dyn_probe);
    else if (num) CLID_SEARCH_LAUNCH((k_search_tiles<false, 0>), g1, dyn_probe);
    else if (xm) CLID_SEARCH_LAUNCH((k_search_tasks<true, 0>), g1, dyn_probe);
    else CLID_SEARCH_LAUNCH((k_search_tasks<false, 0>), g1, dyn_probe);
  }
#undef CLID_SEARCH_LAUNCH
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

extern "C" int clid_train_decode(const clid_map_view* mv, const clid_train_args* a, const float* rec,
                                 void* stream) {
  if (int e = check_train_args(mv, a, "clid_train_decode")) return e;
  if (!mv->feat || !mv->cert || !a->grad || !a->ws || !a->loss_out || !rec) {
    clid_set_error("clid_train_decode: null argument");
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  int n_fd, first;
  const int Q = n_queries(a, &n_fd, &first);
  TrainWs ws = carve(a->ws, Q);
  const TaskMap tmap = make_task_map(a->bs, n_fd, first, a->decimation);
  clid_train_args da = *a;
  da.pipeline = 1;  // this entry IS the hoisted schedule
  const int variant = decode_variant_for(&da);
  const int nb = partial_rows(&da);
  if (a->eikonal_mode == 2) {  // loss.numerical_grad_on: False: the analytic iteration from the records (csrc/train_analytic.hip)
    if (a->partial_row0 < 0 || a->partial_row0 + nb > kMaxBwdBlocks) {
      clid_set_error("clid_train_decode: partial_row0 %d + %d rows exceed the workspace's %d", a->partial_row0, nb, kMaxBwdBlocks);
      return CLID_E_ARG;
    }
    if (int e = clid_launch_train_analytic(mv, a, ws.partial + (size_t)a->partial_row0 * kPartialStride, rec, s)) return e;
    if (a->g_out) return CLID_OK;  // (a gradient probe: nothing to reduce)
  } else if (a->decode_each_neighbour) {  // neuralpoints.weighted_first: False (csrc/train_wf0.hip)
    if (int e = clid_launch_train_wf0(mv, a, ws.partial, tmap, const_cast<float*>(rec), s)) return e;
  } else if (variant) {
    if (int e = clid_launch_decode_tile(mv, a, ws.partial, tmap, rec, variant == 2 ? 1 : 0, s)) return e;
  } else {
    CLID_KLAUNCH(a->prof, 0, k_train_fused8<2>, dim3(nb), dim3(kFusedBlock), 0, s, *mv, *a, ws.partial, tmap,
                 reinterpret_cast<float4*>(const_cast<float*>(rec)));
    CLID_CHECK_LAUNCH();
  }
  if (!a->defer_reduce) {
    if (a->cbuf && a->touch_ws && !variant) {
      clid_set_error("clid_train_decode: the compact exchange needs the tile decode kernels (certainty increments in the rows)");
      return CLID_E_ARG;
    }
    // the tile kernels' dense-exchange flush adds the block sums to grad[0 .. 833) / loss_out itself (csrc/train_tile.hip
    // `direct`): nothing left to reduce -- decode -> all-reduce -> Adam, three launches per iteration of the sharded loop
    const bool direct = variant != 0 && a->eikonal_mode != 2 && !a->decode_each_neighbour && a->dec_copies != nullptr;
    if (direct && (a->cbuf || a->n_dec_copies < 1 || ((uintptr_t)a->dec_copies & 15) != 0)) {
      clid_set_error("clid_train_decode: dec_copies belongs to the dense exchange (cbuf NULL), n_dec_copies >= 1, 16-byte aligned");
      return CLID_E_ARG;
    }
    if (!direct)
      if (int e = launch_reduce(mv, &da, ws.partial, nb, n_fd, s)) return e;
  }
  return CLID_OK;
}

// ---- schedule object of the overlapped search schedule (clid_train_args.sched) ------------------------------------------
struct clid_sched {
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr;
  hipEvent_t done[kMaxChunkIters] = {};
  int device = -1;
};
extern "C" int clid_sched_create(const uint32_t* cu_mask, int32_t mask_words, int32_t priority, clid_sched** out) {
  if (!out || mask_words < 0 || (mask_words > 0 && !cu_mask)) {
    clid_set_error("clid_sched_create: bad argument");
    return CLID_E_ARG;
  }
  *out = nullptr;
  clid_sched* s = new clid_sched();
  bool ok = hipGetDevice(&s->device) == hipSuccess;
  if (ok && mask_words > 0) {
    ok = hipExtStreamCreateWithCUMask(&s->side, (uint32_t)mask_words, cu_mask) == hipSuccess;
  } else if (ok) {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    const int prio = priority < 0 ? least : (priority > 0 ? greatest : 0);
    ok = hipStreamCreateWithPriority(&s->side, hipStreamNonBlocking, prio) == hipSuccess;
  }
  ok = ok && hipEventCreateWithFlags(&s->fork, hipEventDisableTiming) == hipSuccess;
  for (int i = 0; ok && i < kMaxChunkIters; ++i) ok = hipEventCreateWithFlags(&s->done[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    clid_set_error("clid_sched_create: %s", hipGetErrorString(hipGetLastError()));
    clid_sched_destroy(s);
    return CLID_E_HIP;
  }
  *out = s;
  return CLID_OK;
}
extern "C" void clid_sched_destroy(clid_sched* s) {
  if (!s) return;
  if (s->side) {
    (void)hipStreamSynchronize(s->side);
    (void)hipStreamDestroy(s->side);
  }
  if (s->fork) (void)hipEventDestroy(s->fork);
  for (int i = 0; i < kMaxChunkIters; ++i)
    if (s->done[i]) (void)hipEventDestroy(s->done[i]);
  delete s;
}

__global__ void __launch_bounds__(64) k_cu_census(int* __restrict__ out, long long hold) {
  unsigned hw = 0, xcc = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const long long t0 = __builtin_readcyclecounter();
  while ((long long)__builtin_readcyclecounter() - t0 < hold) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(((xcc & 0xf) << 16) | (hw & 0xffff));
}
extern "C" int clid_debug_cu_census(clid_sched* sc, int32_t* out_host, int32_t n_blocks, int32_t hold_cycles, void* stream) {
  if (!out_host || n_blocks < 1 || n_blocks > (1 << 20)) {
    clid_set_error("clid_debug_cu_census: bad argument");
    return CLID_E_ARG;
  }
  hipStream_t s = sc ? sc->side : (hipStream_t)stream;
  int* dev = nullptr;
  if (hipMalloc(&dev, sizeof(int) * (size_t)n_blocks) != hipSuccess) return CLID_E_HIP;
  hipLaunchKernelGGL(k_cu_census, dim3((unsigned)n_blocks), dim3(64), 0, s, dev, (long long)hold_cycles);
  const bool ok = hipStreamSynchronize(s) == hipSuccess &&
                  hipMemcpy(out_host, dev, sizeof(int) * (size_t)n_blocks, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(dev);
  if (!ok) {
    clid_set_error("clid_debug_cu_census: %s", hipGetErrorString(hipGetLastError()));
    return CLID_E_HIP;
  }
  return CLID_OK;
}

// The chunk with its searches BESIDE the chain (clid_train_args.sched): iteration it0's search in front of the first decode
// on the launch stream; the searches of the chunk's other iterations on the object's side stream in groups, each followed by
// an event the decode of the group's first iteration waits for.  The side stream starts behind `fork` = everything the launch
// stream has queued so far (the batch draw / ordering of this call, the previous chunk's decodes that still read the record
// buffer).  Every group's event is waited for before the chunk ends: nothing of the call is left on the side stream.
static int chunk_overlapped(const clid_map_view* mv, clid_train_args& ta, clid_adam_args& aa, int it0, int n_it,
                            const int64_t* index_base, int64_t index_stride, float* loss_base, float* rec, size_t per_iter,
                            hipStream_t main) {
  clid_sched* sc = ta.sched;
  if (hipEventRecord(sc->fork, main) != hipSuccess || hipStreamWaitEvent(sc->side, sc->fork, 0) != hipSuccess) {
    clid_set_error("clid_mapping_run: fork of the side stream failed: %s", hipGetErrorString(hipGetLastError()));
    return CLID_E_HIP;
  }
  if (int e = train_search_impl(mv, &ta, 1, index_base + (int64_t)it0 * index_stride, index_stride, rec, main, 0)) return e;
  int group_of[kMaxChunkIters];  // iteration (relative) -> event to wait for in front of its decode, -1 none
  for (int i = 0; i < n_it; ++i) group_of[i] = -1;
  int n_groups = 0;
  for (int i = 1, g = 1; i < n_it; ++n_groups) {
    const int want = ta.side_group > 0 ? ta.side_group : g;
    const int n = (n_it - i) < want ? (n_it - i) : want;
    if (int e = train_search_impl(mv, &ta, n, index_base + (int64_t)(it0 + i) * index_stride, index_stride,
                                  rec + (size_t)i * per_iter, sc->side, ta.side_blocks))
      return e;
    if (hipEventRecord(sc->done[n_groups], sc->side) != hipSuccess) {
      clid_set_error("clid_mapping_run: event record failed: %s", hipGetErrorString(hipGetLastError()));
      return CLID_E_HIP;
    }
    group_of[i] = n_groups;
    i += n;
    g *= 2;
  }
  for (int i = 0; i < n_it; ++i) {
    const int it = it0 + i;
    if (group_of[i] >= 0 && hipStreamWaitEvent(main, sc->done[group_of[i]], 0) != hipSuccess) {
      clid_set_error("clid_mapping_run: event wait failed: %s", hipGetErrorString(hipGetLastError()));
      return CLID_E_HIP;
    }
    ta.index = index_base + (int64_t)it * index_stride;
    ta.loss_out = loss_base + (size_t)it * 4;
    ta.touch_iter = i;
    if (int e = clid_train_decode(mv, &ta, rec + (size_t)i * per_iter, main)) return e;
    aa.step = it + 1;
    if (int e = clid_train_adam(&aa, &ta, main)) return e;
  }
  return CLID_OK;
}

static int mapping_run_hoisted(const clid_map_view* mv, clid_train_args& ta, clid_adam_args& aa, int iters,
                               const int64_t* index_base, int64_t index_stride, float* loss_base, void* stream) {
  int n_fd, first;
  const int Q = n_queries(&ta, &n_fd, &first);
  TrainWs ws = carve(ta.ws, Q);
  const size_t per_iter = (size_t)clid_train_search_floats(ta.bs, ta.batch_offset, ta.decimation, ta.eikonal_mode, 1);
  const int chunk = clid_train_chunk_iters(&ta);
  if (chunk < 1) {
    clid_set_error("clid_mapping_run: the task records exceed the workspace bound");
    return CLID_E_SHAPE;
  }
  for (int it0 = 0; it0 < iters; it0 += chunk) {
    const int n_it = (iters - it0) < chunk ? (iters - it0) : chunk;
    if (int e = launch_eik_inv_n(&ta, n_it, index_base + (int64_t)it0 * index_stride, index_stride, 0, (hipStream_t)stream)) return e;
    if (ta.sched && !ta.touch_ws && n_it >= 2) {
      int dev = -1;
      if (hipGetDevice(&dev) != hipSuccess || dev != ta.sched->device) {
        clid_set_error("clid_mapping_run: the schedule object belongs to device %d, the current device is %d", ta.sched->device, dev);
        return CLID_E_ARG;
      }
      if (int e = chunk_overlapped(mv, ta, aa, it0, n_it, index_base, index_stride, loss_base, ws.rec, per_iter, (hipStream_t)stream))
        return e;
      continue;
    }
    if (int e = train_search_impl(mv, &ta, n_it, index_base + (int64_t)it0 * index_stride, index_stride, ws.rec, stream, 0))
      return e;
    if (ta.touch_ws)
      if (int e = clid_train_touch_scan(&ta, mv->M, n_it, it0, nullptr, stream)) return e;
    for (int it = it0; it < it0 + n_it; ++it) {
      ta.index = index_base + (int64_t)it * index_stride;
      ta.loss_out = loss_base + (size_t)it * 4;
      ta.touch_iter = it - it0;
      if (int e = clid_train_decode(mv, &ta, ws.rec + (size_t)(it - it0) * per_iter, stream)) return e;
      aa.step = it + 1;
      if (int e = clid_train_adam(&aa, &ta, stream)) return e;
    }
  }
  return CLID_OK;
}

// The whole single-GPU loop of Mapper.mapping (utils/mapper.py:642-860) enqueued by ONE host call:
// t->pipeline 1 = the neighbour searches hoisted into one launch per chunk of iterations, then per iteration the
// decode/backward kernel and the reduce+Adam kernel; 0 = per iteration the fused search+decode kernel and the reduce+Adam
// kernel (the results are identical up to the order of the atomic accumulations).
extern "C" int clid_mapping_run(const clid_map_view* mv, const clid_train_args* t, const clid_adam_args* a,
                                int32_t iters, const int64_t* index_base, int64_t index_stride,
                                float* loss_base, void* stream) {
  if (!mv || !t || !a || iters < 0 || !index_base || !loss_base) {
    clid_set_error("clid_mapping_run: bad argument");
    return CLID_E_ARG;
  }
  clid_train_args ta = *t;
  clid_adam_args aa = *a;
  ta.defer_reduce = 1;
  ta.cbuf = nullptr;
  if (hoisted(&ta) && iters > 0)
    return mapping_run_hoisted(mv, ta, aa, iters, index_base, index_stride, loss_base, stream);
  ta.pipeline = 0;
  for (int it = 0; it < iters; ++it) {
    ta.index = index_base + (int64_t)it * index_stride;
    ta.loss_out = loss_base + (size_t)it * 4;
    if (int e = clid_train_fwd_bwd(mv, &ta, stream)) return e;
    aa.step = it + 1;
    if (int e = clid_train_adam(&aa, &ta, stream)) return e;
  }
  return CLID_OK;
}

// The sharded loop of one rank, stream-resident: hoisted searches of this rank's slice of every
// batch, then per iteration decode/backward -> partial reduction -> RCCL all-reduce ON THE LAUNCH STREAM -> the identical
// Adam step on every replica.  `index_base` points at this rank's slice of iteration 0 (row stride index_stride);
// t->batch_offset / inv_n_* carry the global lattice phase and normalisers.  After the loop the per-iteration losses (SUM)
// and the update stamps (MAX) are merged once.
//   Exchange.  Dense (t->touch_ws / t->cbuf NULL): the whole fused buffer `grad` [grad_floats] = 848 + 16 (M + 1) floats per
//   iteration.  Compact: the chunk's searches flag every row every iteration will touch; the flags are MAX-reduced over
//   the ranks once per chunk (M bytes per iteration), scanned into per-iteration row lists whose lengths come back to the
//   host (ONE synchronisation per chunk), and each iteration all-reduces [848 | 9 x rows it touches] floats.
struct clid_comm;
extern "C" int clid_comm_allreduce(clid_comm* comm, void* buf, int64_t count, int32_t dtype, int32_t op_max, void* stream);

extern "C" int clid_mapping_run_dist(const clid_map_view* mv, const clid_train_args* t, const clid_adam_args* a,
                                     int32_t iters, const int64_t* index_base, int64_t index_stride, float* loss_base,
                                     clid_comm* comm, int64_t grad_floats, int64_t* exchanged_floats_host, void* stream) {
  if (!mv || !t || !a || iters < 0 || !index_base || !loss_base || !comm || grad_floats <= 0) {
    clid_set_error("clid_mapping_run_dist: bad argument");
    return CLID_E_ARG;
  }
  clid_train_args ta = *t;
  clid_adam_args aa = *a;
  ta.defer_reduce = 0;  // the decoder gradients must sit in `grad` / `cbuf` before the all-reduce
  const bool hoist = hoisted(&ta);
  if (!hoist) ta.pipeline = 0;
  const bool compact = hoist && ta.touch_ws && ta.cbuf;
  if (!compact) ta.cbuf = nullptr;
  if (!hoist) ta.touch_ws = nullptr;
  int n_fd, first;
  const int Q = n_queries(&ta, &n_fd, &first);
  TrainWs ws = carve(ta.ws, Q);
  size_t per_iter = 0;
  int chunk = 1;
  if (hoist) {
    per_iter = (size_t)clid_train_search_floats(ta.bs, ta.batch_offset, ta.decimation, ta.eikonal_mode, 1);
    chunk = clid_train_chunk_iters(&ta);
    if (chunk < 1) {
      clid_set_error("clid_mapping_run_dist: the task records exceed the workspace bound");
      return CLID_E_SHAPE;
    }
  }
  int32_t counts[kMaxChunkIters];
  long long moved = 0;
  // peer-mapped exchange (csrc/p2p.hip) for the compact buffer when the object fits this call -- a decision every rank
  // makes alike (same world, same M, same capacity)
  clid_p2p* px = compact ? ta.p2p : nullptr;
  if (px && (clid_p2p_world(px) != clid_comm_size(comm) ||
             clid_p2p_capacity(px) < (int64_t)sizeof(float) * (CLID_GRAD_FEAT_OFFSET16 + 9LL * (mv->M + 1) + 4)))
    px = nullptr;
  for (int it = 0; it < iters; ++it) {
    ta.index = index_base + (int64_t)it * index_stride;
    ta.loss_out = loss_base + (size_t)it * 4;
    ta.touch_iter = it % chunk;
    if (hoist) {
      if (it % chunk == 0) {
        const int n_it = (iters - it) < chunk ? (iters - it) : chunk;
        if (int e = clid_train_search(mv, &ta, n_it, ta.index, index_stride, ws.rec, stream)) return e;
        if (ta.touch_ws && ta.touch_all && compact) {
          // small local map: every row is on every iteration's list (nothing to agree on, nothing to read back)
          if (mv->M > 0 && hipMemset2DAsync(ta.touch_ws, (size_t)ta.touch_stride, 1, (size_t)mv->M, (size_t)n_it,
                                            (hipStream_t)stream) != hipSuccess) {
            clid_set_error("clid_mapping_run_dist: flag fill failed");
            return CLID_E_HIP;
          }
          if (int e = clid_train_touch_scan(&ta, mv->M, n_it, it, nullptr, stream)) return e;
          for (int i = 0; i < n_it; ++i) counts[i] = mv->M;
        } else if (ta.touch_ws) {
          // the union over the ranks of the rows each iteration touches: every rank then packs the same list
          const int64_t flag_bytes = (int64_t)n_it * ta.touch_stride;
          if (px && flag_bytes + 16 <= clid_p2p_capacity(px)) {
            if (int e = clid_p2p_allreduce_or(px, ta.touch_ws, flag_bytes, stream)) return e;
          } else if (int e = clid_comm_allreduce(comm, ta.touch_ws, flag_bytes, 2, 1, stream)) {
            return e;
          }
          moved += (flag_bytes + 3) / 4;
          if (int e = clid_train_touch_scan(&ta, mv->M, n_it, it, compact ? counts : nullptr, stream)) return e;
        }
      }
      // (after the chunk's flag exchange, which takes one turn of the two exchange buffers itself)
      if (px) ta.cbuf = static_cast<float*>(clid_p2p_buffer(px));
      if (int e = clid_train_decode(mv, &ta, ws.rec + (size_t)(it % chunk) * per_iter, stream)) return e;
    } else {
      if (int e = clid_train_fwd_bwd(mv, &ta, stream)) return e;
    }
    if (compact) {
      const int64_t n = CLID_GRAD_FEAT_OFFSET16 + 9LL * counts[it % chunk];
      if (px) {
        if (int e = clid_p2p_allreduce(px, n, stream)) return e;
      } else if (int e = clid_comm_allreduce(comm, ta.cbuf, n, 0, 0, stream)) {
        return e;
      }
      moved += n;
    } else {
      if (int e = clid_comm_allreduce(comm, ta.grad, grad_floats, 0, 0, stream)) return e;
      moved += grad_floats;
    }
    aa.step = it + 1;
    if (int e = clid_train_adam(&aa, &ta, stream)) return e;
  }
  if (iters > 0) {
    if (int e = clid_comm_allreduce(comm, loss_base, (int64_t)iters * 4, 0, 0, stream)) return e;
    if (mv->ts_update && mv->M > 0)
      if (int e = clid_comm_allreduce(comm, mv->ts_update, mv->M, 1, 1, stream)) return e;
  }
  if (exchanged_floats_host) *exchanged_floats_host = moved;
  // a flag wait that gave up on ANY rank invalidates the call on EVERY rank: agreed through the communicator, so all ranks
  // return the same code and the host repeats the call over RCCL from its saved state
  if (px && iters > 0)
    if (int e = clid_p2p_agree(px, comm, stream)) return e;
  return CLID_OK;
}

// Host-side enumeration of the task -> query mapping the fused kernel uses (same code, compiled for the
// host): counts how often each local batch position is trained as a batch sample (must be exactly 1)
// and how many shifted copies it gets (6 on the decimation lattice, else 0).  CPU-only; used by tests.
extern "C" int clid_debug_task_cover(int32_t bs, int64_t batch_offset, int32_t decimation, int32_t eikonal_mode,
                                     int32_t* main_count_host, int32_t* fd_count_host, int32_t* n_tasks_host) {
  if (bs <= 0 || decimation <= 0 || !main_count_host || !fd_count_host) return CLID_E_ARG;
  const int first = fd_first(batch_offset, decimation);
  const int n_fd = eikonal_mode == 1 ? fd_count(bs, batch_offset, decimation) : 0;
  const TaskMap tm = make_task_map(bs, n_fd, first, decimation);
  for (int i = 0; i < bs; ++i) main_count_host[i] = fd_count_host[i] = 0;
  for (int task = 0; task < tm.n_tasks; ++task)
    for (int round = 0; round < 2; ++round)
      for (int grp = 0; grp < 4; ++grp) {
        const QDesc q = task_query(tm, task, round, grp);
        if (q.p < 0) continue;
        if (q.p >= bs) return CLID_E_ARG;
        if (q.axis < 0) main_count_host[q.p]++;
        else fd_count_host[q.p]++;
      }
  if (n_tasks_host) *n_tasks_host = tm.n_tasks;
  return CLID_OK;
}
