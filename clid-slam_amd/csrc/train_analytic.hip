// Analytic-eikonal mapping iteration (`loss.numerical_grad_on: False`): utils/mapper.py:57-69, 660-661,
// 695-696 -> g = get_gradient(coord, sdf_pred) (utils/tools.py:298-311, create_graph=True), eikonal loss
// on EVERY batch sample (gradient_decimation = 1, utils/config.py:645-646), backward THROUGH g.
//
// Closed forms (verified against the reference's double backward by G6):
//   alpha_k = 2 r_k omega_k,  abar = sum_j w_j alpha_j,  dw_k = w_k (abar - alpha_k)        (d w_k / d x)
//   u = s (W2 .* a) W1  (== the decoder's d sdf / d f),   g = sum_k (u . v_k) dw_k + (sum_k w_k) u[8:11]
//   c = dL/dg;  s_k = c . dw_k;  t = J c = sum_k v_k s_k + [0_8 ; (sum w) c]
//   dW1[h,:] += s W2[h] a[h] t,  dW2[h] += s a[h] (W1[h,:] . t),  d feat_k += (w_k delta + s_k) u[0:8]
// One round = 4 queries per wave, forward and backward back to back in registers (no shifted copies, so
// no cross-query dependency).  Lane layout of the gather as in train.hip: lane16 = 2k + half.
#include <type_traits>

#include "train_common.hpp"

namespace clid {


#define CLID_BFLY(x) x += dpp_mov<0x128>(x); x += dpp_mov<0x124>(x); x += dpp_mov<0x122>(x);
#define CLID_SWAP(x) dpp_mov<0xB1>(x)  // quad_perm [1,0,3,2]: the other lane of the pair

// HOISTED: neighbours and IDW weights come from the records of the chunk's search launch (clid_train_search: plain tasks of 8
// samples, task = position / 8) -- positions only, nothing the training writes -- instead of a search inside the iteration's
// dependent chain; omega_k is recomputed from x - p_k with the search's own operation order (bit-identical).
// This lane's share of the decoder lives in registers (52 floats) and the kernel is compiled for 2 waves per SIMD -- what a launch
// of <= 512 blocks puts there anyway: 16.2 -> 14.6 us against the weights in LDS at 3 waves (with 3 waves it spills: 23 us)
#ifndef CLID_ANALYTIC_WAVES
#define CLID_ANALYTIC_WAVES 2
#endif
#ifndef CLID_ANALYTIC_REGW
#define CLID_ANALYTIC_REGW 1
#endif
// PC: config.proj_correction_on (utils/mapper.py:712-714): label' = label |cos(g, x - origin of the sample's frame)| with g in the
// graph, so the BCE term reaches the parameters through g as well: dL/dg gains dL/dlabel' label d|cos|/dg (its own instantiation)
// CX: config.consistency_loss_on -- 1 = gradient probe (g of every sample to ta.g_out, nothing else), 2 = ta.c_extra added to dL/dg
template <bool HOISTED, bool PC = false, int CX = 0>
__global__ void __launch_bounds__(CLID_BLOCK, CLID_ANALYTIC_WAVES)
k_train_analytic(clid_map_view mv, clid_train_args ta, float* __restrict__ partial, int n_rounds, const float4* __restrict__ rec) {
  __shared__ MlpLds mlp;
  __shared__ typename std::conditional<HOISTED, DeltaLds, SearchLds>::type dl;  // (in-kernel search: + the window's cell directory)
  __shared__ float red[(CLID_BLOCK / 64) * kRedFloats];
  __shared__ PairLds pairs[CLID_BLOCK / 64];  // merged scatter on 16-float accumulation rows (train_common.hpp)
  if (HOISTED) stage_mlp(mlp, ta.W1, ta.b1, ta.W2, ta.b2);
  else stage_mlp_and_delta(mlp, dl, mv, ta.W1, ta.b1, ta.W2, ta.b2);
  const int lane = threadIdx.x & 63, lane16 = lane & 15, gbase = lane & 48, grp = lane >> 4;
  const int wave = threadIdx.x >> 6, waves_per_block = CLID_BLOCK / 64;
  const int my_k = lane16 >> 1;
  const bool odd = lane16 & 1;
  MlpAcc acc;
  acc.zero();
  float bce_acc = 0.f, eik_acc = 0.f;
#if CLID_ANALYTIC_REGW
  // this lane's share of the decoder (hidden units lane16 + 16 u) in registers: the forward pass, u = s (W2 .* a) W1 and
  // W1 t each walk it once per round -- three LDS reads per weight and round otherwise
  float w1r[CLID_HPL][CLID_D], b1r[CLID_HPL], w2r[CLID_HPL];
#pragma unroll
  for (int uu = 0; uu < CLID_HPL; ++uu) {
    const int h = lane16 + CLID_G * uu;
#pragma unroll
    for (int c = 0; c < CLID_D; ++c) w1r[uu][c] = mlp.w[h * CLID_D + c];
    b1r[uu] = mlp.w[CLID_H * CLID_D + h];
    w2r[uu] = mlp.w[CLID_H * CLID_D + CLID_H + h];
  }
  const float b2r = mlp.w[CLID_MLP_PARAMS - 1];
#endif
  const int gstride = ta.grad_stride == CLID_GRAD_ROW16 ? CLID_GRAD_ROW16 : CLID_F;
  const bool merged = gstride == CLID_GRAD_ROW16;
  float* g_theta = ta.grad + CLID_GRAD_OFFSET(gstride);
  const float inv_sigma = fdiv(1.0f, ta.sigma);
  const float4* pos4 = reinterpret_cast<const float4*>(mv.pos4);
  const float sc = ta.sdf_scale;

  // HOISTED: the lane's slice of a round's record (10 registers; the kernel has 36 to spare at two waves per SIMD), requested one
  // round ahead -- in front of the current round's gathers, so their wait covers it and a round does not open with a wait that
  // also drains the previous round's gradient atomics (vmcnt is one in-order counter for loads, stores and atomics)
#ifndef CLID_ANALYTIC_PF
#define CLID_ANALYTIC_PF 1
#endif
  struct RoundRec {
    float4 qi, qd;
    float2 wn;
  };
  auto load_round = [&](int rd_) -> RoundRec {
    const int p = rd_ * 4 + grp;
    const bool lv = p < ta.bs;
    const float4* r = rec + (size_t)((lv ? p : 0) >> 3) * 48;  // qinfo[8] | qdesc[8] | win[8][8] float2
    const int sl = lv ? (p & 7) : 0;
    RoundRec o;
    o.qi = r[sl];
    o.qd = r[8 + sl];
    o.wn = reinterpret_cast<const float2*>(r + 16)[sl * 8 + (my_k < CLID_K ? my_k : 0)];
    return o;
  };
  RoundRec nxt;
  if constexpr (HOISTED && CLID_ANALYTIC_PF) {
    const int rd0 = blockIdx.x * waves_per_block + wave;
    if (rd0 < n_rounds) nxt = load_round(rd0);
  }
  for (int rd = blockIdx.x * waves_per_block + wave; rd < n_rounds; rd += gridDim.x * waves_per_block) {
    const int p_raw = rd * 4 + grp;
    bool live = p_raw < ta.bs;
    long long s = 0;
    float px, py, pz, rec_label = 0.f, rec_wt = 1.f;
    int rec_ts = 0;
    int my_j = -1;
    float my_w = 0.f, my_om = 0.f;
    if constexpr (HOISTED && CLID_ANALYTIC_PF) {
      const RoundRec cur = nxt;
      const int rdn = rd + gridDim.x * waves_per_block;
      if (rdn < n_rounds) nxt = load_round(rdn);
      const float4 qi = cur.qi, qd = cur.qd;
      live = live && __float_as_int(qd.x) >= 0;
      px = qi.x; py = qi.y; pz = qi.z;
      rec_ts = __float_as_int(qi.w);
      rec_label = qd.z; rec_wt = qd.w;
      if (my_k < CLID_K) {
        my_w = cur.wn.x;
        my_j = __float_as_int(cur.wn.y);
      }
    } else if constexpr (HOISTED) {
      const float4* r = rec + (size_t)(p_raw >> 3) * 48;  // qinfo[8] | qdesc[8] | win[8][8] float2
      const int slot = p_raw & 7;
      const float4 qi = r[live ? slot : 0], qd = r[8 + (live ? slot : 0)];
      live = live && __float_as_int(qd.x) >= 0;
      px = qi.x; py = qi.y; pz = qi.z;
      rec_ts = __float_as_int(qi.w);
      rec_label = qd.z; rec_wt = qd.w;
      const float2 wn = reinterpret_cast<const float2*>(r + 16)[(live ? slot : 0) * 8 + (my_k < CLID_K ? my_k : 0)];
      if (my_k < CLID_K) {
        my_w = wn.x;
        my_j = __float_as_int(wn.y);
      }
    } else {
      s = ta.index[live ? p_raw : 0];
      px = ta.pool_coord[s * 3 + 0]; py = ta.pool_coord[s * 3 + 1]; pz = ta.pool_coord[s * 3 + 2];
      TopK t;
      search_topk(mv, dl, px, py, pz, lane16, gbase, t);
      float w[CLID_K], omega[CLID_K];
      idw_weights(t, w, omega);
#pragma unroll
      for (int k = 0; k < CLID_K; ++k) {
        my_j = (my_k == k) ? t.j[k] : my_j;
        my_w = (my_k == k) ? w[k] : my_w;
        my_om = (my_k == k) ? omega[k] : my_om;
      }
    }
    if (!live) my_j = -1;
    const bool valid = my_j >= 0;
    if (!valid) { my_w = 0.f; my_om = 0.f; }
    const int jc = valid ? my_j : 0;
    float4 v = reinterpret_cast<const float4*>(mv.feat)[(size_t)jc * 2 + (odd ? 1 : 0)];
    const float4 pj = pos4[jc];
    // (np.py:719-728) amax is idempotent: only a newer stamp needs the atomic -- the old one is read with the gather
    const int ts_old = (HOISTED && valid && !odd && mv.ts_update) ? mv.ts_update[jc] : 0x7fffffff;
    if (!valid) v = make_float4(0.f, 0.f, 0.f, 0.f);
    float rstd = 1.f;
    if (mv.layer_norm) {  // np.py:632-633; an all-zero (invalid) row stays zero
      float s1 = (v.x + v.y) + (v.z + v.w);
      s1 += CLID_SWAP(s1);
      const float mu = s1 * (1.0f / CLID_F);
      v.x -= mu; v.y -= mu; v.z -= mu; v.w -= mu;
      float s2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      s2 += CLID_SWAP(s2);
      rstd = 1.0f / sqrtf(s2 * (1.0f / CLID_F) + 1e-5f);
      v.x *= rstd; v.y *= rstd; v.z *= rstd; v.w *= rstd;
    }
    const float rx = valid ? fsub(px, pj.x) : 0.f, ry = valid ? fsub(py, pj.y) : 0.f, rz = valid ? fsub(pz, pj.z) : 0.f;
    if constexpr (HOISTED)  // np.py:688-693 on dist2 = |p_k - x|^2 in the search's operation order
      my_om = valid ? fdiv(1.0f, fadd(fadd(fadd(fmul(rx, rx), fmul(ry, ry)), fmul(rz, rz)), 1e-15f)) : 0.f;

    // ---- blended decoder input f (replicated)
    float f[CLID_D];
    {
      float a0 = v.x * my_w, a1 = v.y * my_w, a2 = v.z * my_w, a3 = v.w * my_w;
      float r0 = rx * my_w, r1 = ry * my_w, r2 = rz * my_w;  // carried by both lanes of the pair
      CLID_BFLY(a0) CLID_BFLY(a1) CLID_BFLY(a2) CLID_BFLY(a3) CLID_BFLY(r0) CLID_BFLY(r1) CLID_BFLY(r2)
      const float b0 = CLID_SWAP(a0), b1 = CLID_SWAP(a1), b2 = CLID_SWAP(a2), b3 = CLID_SWAP(a3);
      f[0] = odd ? b0 : a0; f[1] = odd ? b1 : a1; f[2] = odd ? b2 : a2; f[3] = odd ? b3 : a3;
      f[4] = odd ? a0 : b0; f[5] = odd ? a1 : b1; f[6] = odd ? a2 : b2; f[7] = odd ? a3 : b3;
      f[8] = r0; f[9] = r1; f[10] = r2;
    }
    float pre[CLID_HPL];
#if CLID_ANALYTIC_REGW
    float sdf;
    {
      float part = 0.f;
#pragma unroll
      for (int uu = 0; uu < CLID_HPL; ++uu) {
        float a = b1r[uu];
#pragma unroll
        for (int c = 0; c < CLID_D; ++c) a = fmaf(w1r[uu][c], f[c], a);
        pre[uu] = a;
        part = fmaf(w2r[uu], fmaxf(a, 0.f), part);
      }
      sdf = sc * (group_sum(part) + b2r);
    }
    const int l16 = lane16;
#else
    const float sdf = mlp_forward(mlp, f, lane16, sc, pre);
    const int l16 = lane16 + opaque_zero();
#endif
    // ---- u = s (W2 .* a) W1   (replicated)
    float u[CLID_D];
    float e[CLID_HPL];
#pragma unroll
    for (int uu = 0; uu < CLID_HPL; ++uu)
#if CLID_ANALYTIC_REGW
      e[uu] = pre[uu] > 0.f ? sc * w2r[uu] : 0.f;
#else
      e[uu] = pre[uu] > 0.f ? sc * mlp.w[CLID_H * CLID_D + CLID_H + l16 + CLID_G * uu] : 0.f;
#endif
#pragma unroll
    for (int c = 0; c < CLID_D; ++c) {
      float part = 0.f;
#pragma unroll
#if CLID_ANALYTIC_REGW
      for (int uu = 0; uu < CLID_HPL; ++uu) part = fmaf(w1r[uu][c], e[uu], part);
#else
      for (int uu = 0; uu < CLID_HPL; ++uu) part = fmaf(mlp.w[(l16 + CLID_G * uu) * CLID_D + c], e[uu], part);
#endif
      u[c] = group_sum(part);
    }
    // ---- d w_k / d x and g = d sdf / d x
    const float alx = 2.f * rx * my_om, aly = 2.f * ry * my_om, alz = 2.f * rz * my_om;
    float abx = my_w * alx, aby = my_w * aly, abz = my_w * alz, wsum = my_w;
    CLID_BFLY(abx) CLID_BFLY(aby) CLID_BFLY(abz) CLID_BFLY(wsum)
    const float dwx = my_w * (abx - alx), dwy = my_w * (aby - aly), dwz = my_w * (abz - alz);
    const float us0 = odd ? u[4] : u[0], us1 = odd ? u[5] : u[1], us2 = odd ? u[6] : u[2], us3 = odd ? u[7] : u[3];
    float dot = us0 * v.x + us1 * v.y + us2 * v.z + us3 * v.w;
    if (!odd) dot += u[8] * rx + u[9] * ry + u[10] * rz;
    dot += CLID_SWAP(dot);
    float gx = dot * dwx, gy = dot * dwy, gz = dot * dwz;
    CLID_BFLY(gx) CLID_BFLY(gy) CLID_BFLY(gz)
    gx += wsum * u[8]; gy += wsum * u[9]; gz += wsum * u[10];
    if constexpr (CX == 1) {  // gradient probe: g is all the caller wants of this pass
      if (live && lane16 == 0) {
        ta.g_out[(size_t)p_raw * 3 + 0] = gx; ta.g_out[(size_t)p_raw * 3 + 1] = gy; ta.g_out[(size_t)p_raw * 3 + 2] = gz;
      }
      continue;
    }

    // ---- training_mode side effects (np.py:708-733)
    if (valid && !odd && lane16 < 2 * CLID_K) {
      if (!merged) atomicAdd(&mv.cert[my_j], my_w);
      if constexpr (HOISTED) {
        if (ts_old < rec_ts) atomicMax(&mv.ts_update[my_j], rec_ts);
      } else {
        if (mv.ts_update) atomicMax(&mv.ts_update[my_j], ta.pool_ts[s]);
      }
    }

    // ---- losses and their derivatives
    float delta = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
    if (live) {
      float label = HOISTED ? rec_label : ta.pool_label[s];
      const float wt = HOISTED ? rec_wt : (ta.loss_weight_on ? fabsf(ta.pool_weight[s]) : 1.0f);  // mapper.py:747-749
      float dcx = 0.f, dcy = 0.f, dcz = 0.f, label0 = label;  // (PC) d|cos| / dg, the unscaled label
      if constexpr (PC) {
        // F.cosine_similarity(g, coord - origins) (mapper.py:713): both vectors divided by max(norm, 1e-8), then the dot product
        int fr = rec_ts;
        fr = fr < 0 ? 0 : (fr >= ta.n_frame_pose ? ta.n_frame_pose - 1 : fr);
        const float4* T = reinterpret_cast<const float4*>(ta.frame_pose) + (size_t)fr * 3;
        const float dx = px - T[0].w, dy = py - T[1].w, dz = pz - T[2].w;
        const float ng = sqrtf(gx * gx + gy * gy + gz * gz), nd = sqrtf(dx * dx + dy * dy + dz * dz);
        const float ig = 1.0f / fmaxf(ng, 1e-8f), id = 1.0f / fmaxf(nd, 1e-8f);
        const float hx = gx * ig, hy = gy * ig, hz = gz * ig, ex = dx * id, ey = dy * id, ez_ = dz * id;
        const float dot = hx * ex + hy * ey + hz * ez_;
        const float sgn = dot > 0.f ? 1.0f : (dot < 0.f ? -1.0f : 0.f);
        // d(dot)/dg = (e - dot h) / |g| (|g| above the clamp; below it h = g / eps is linear in g: e / eps)
        const float k = ng > 1e-8f ? 1.0f : 0.f;
        dcx = sgn * (ex - k * dot * hx) * ig; dcy = sgn * (ey - k * dot * hy) * ig; dcz = sgn * (ez_ - k * dot * hz) * ig;
        label *= fabsf(dot);
      }
      const float z = sdf * inv_sigma;
      const float tgt = __frcp_rn(1.0f + __expf(-label * inv_sigma));         // loss.py:60
      const float ez = __expf(-fabsf(z));
      const float sg = (z >= 0.f ? 1.0f : ez) * __frcp_rn(1.0f + ez);
      const float li = fmaxf(z, 0.f) - z * tgt + __logf(1.0f + ez);           // BCEWithLogits
      if (lane16 == 0) bce_acc += wt * li;
      delta = wt * (sg - tgt) * inv_sigma * ta.inv_n_main;
      if (ta.weight_e > 0.f) {
        const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
        if (lane16 == 0) eik_acc += (nrm - 1.f) * (nrm - 1.f);
        // d/dg of weight_e * mean((|g|-1)^2); 0 at |g| == 0 (torch norm subgradient)
        const float coef = nrm > 0.f ? ta.weight_e * 2.f * (nrm - 1.f) * ta.inv_n_eik / nrm : 0.f;
        cx = coef * gx; cy = coef * gy; cz = coef * gz;
      }
      if constexpr (CX == 2) {  // the consistency term's dL/dg of this sample (clid_consistency_couple)
        cx += ta.c_extra[(size_t)p_raw * 3 + 0]; cy += ta.c_extra[(size_t)p_raw * 3 + 1]; cz += ta.c_extra[(size_t)p_raw * 3 + 2];
      }
      if constexpr (PC) {
        // the BCE term through the scaled label: d/dt of BCEWithLogits = -z, t = sigmoid(label' / sigma), label' = label |cos|
        const float dcos = wt * (-z) * ta.inv_n_main * tgt * (1.0f - tgt) * inv_sigma * label0;
        cx = fmaf(dcos, dcx, cx); cy = fmaf(dcos, dcy, cy); cz = fmaf(dcos, dcz, cz);
      }
    }
    const float sk = cx * dwx + cy * dwy + cz * dwz;  // c . dw_k

    // ---- decoder gradients
    if (ta.train_decoder) {
      float tv[CLID_D];
      {
        float t0 = v.x * sk, t1 = v.y * sk, t2 = v.z * sk, t3 = v.w * sk;
        float q0 = rx * sk, q1 = ry * sk, q2 = rz * sk;
        CLID_BFLY(t0) CLID_BFLY(t1) CLID_BFLY(t2) CLID_BFLY(t3) CLID_BFLY(q0) CLID_BFLY(q1) CLID_BFLY(q2)
        const float b0 = CLID_SWAP(t0), b1 = CLID_SWAP(t1), b2 = CLID_SWAP(t2), b3 = CLID_SWAP(t3);
        tv[0] = odd ? b0 : t0; tv[1] = odd ? b1 : t1; tv[2] = odd ? b2 : t2; tv[3] = odd ? b3 : t3;
        tv[4] = odd ? t0 : b0; tv[5] = odd ? t1 : b1; tv[6] = odd ? t2 : b2; tv[7] = odd ? t3 : b3;
        tv[8] = q0 + wsum * cx; tv[9] = q1 + wsum * cy; tv[10] = q2 + wsum * cz;
      }
      const float dz = sc * delta;
      float fb = (lane16 == CLID_D) ? 1.0f : 0.f, tb = 0.f;
#pragma unroll
      for (int c = 0; c < CLID_D; ++c) {
        fb = (lane16 == c) ? f[c] : fb;
        tb = (lane16 == c) ? tv[c] : tb;
      }
#pragma unroll
      for (int uu = 0; uu < CLID_HPL; ++uu) {
        const int h = l16 + CLID_G * uu;
        const bool on = pre[uu] > 0.f;
#if CLID_ANALYTIC_REGW
        const float dh = on ? dz * w2r[uu] : 0.f;
        float w1t = 0.f;
#pragma unroll
        for (int c = 0; c < CLID_D; ++c) w1t = fmaf(w1r[uu][c], tv[c], w1t);
#else
        const float dh = on ? dz * mlp.w[CLID_H * CLID_D + CLID_H + h] : 0.f;
        float w1t = 0.f;
#pragma unroll
        for (int c = 0; c < CLID_D; ++c) w1t = fmaf(mlp.w[h * CLID_D + c], tv[c], w1t);
#endif
        acc.dW2[uu] += on ? (dz * pre[uu] + sc * w1t) : 0.f;
        acc.dW1[uu] = __builtin_amdgcn_mfma_f32_16x16x4f32(dh, fb, acc.dW1[uu], 0, 0, 0);
        acc.dW1[uu] = __builtin_amdgcn_mfma_f32_16x16x4f32(e[uu], tb, acc.dW1[uu], 0, 0, 0);
      }
      if (lane16 == 0) acc.db2 += dz;
    }

    // ---- feature gradients: upstream (w_k delta + s_k) u[0:8] on the (normalised) row of neighbour k
    {
      const float ck = my_w * delta + sk;
      float d0 = ck * us0, d1 = ck * us1, d2 = ck * us2, d3 = ck * us3;
      if (mv.layer_norm) {  // dx = rstd (dy - mean(dy) - xhat mean(dy xhat)) over the 8 features (pair of lanes)
        float m1 = (d0 + d1) + (d2 + d3);
        float m2 = (d0 * v.x + d1 * v.y) + (d2 * v.z + d3 * v.w);
        m1 += CLID_SWAP(m1);
        m2 += CLID_SWAP(m2);
        m1 *= (1.0f / CLID_F);
        m2 *= (1.0f / CLID_F);
        d0 = rstd * (d0 - m1 - v.x * m2); d1 = rstd * (d1 - m1 - v.y * m2);
        d2 = rstd * (d2 - m1 - v.z * m2); d3 = rstd * (d3 - m1 - v.w * m2);
      }
      if (merged) {
        scatter_pairs_rows16(pairs[wave], lane, grp, my_k, odd, valid ? my_j : -1, d0, d1, d2, d3, valid ? my_w : 0.f, g_theta);
      } else if (valid && lane16 < 2 * CLID_K && ck != 0.f) {
        float* dst = g_theta + (size_t)my_j * gstride + (odd ? 4 : 0);
        atomicAdd(dst + 0, d0); atomicAdd(dst + 1, d1); atomicAdd(dst + 2, d2); atomicAdd(dst + 3, d3);
      }
    }
  }
  if constexpr (CX == 1) return;  // (a probe leaves no partial row)
  flush_mlp_acc(acc, bce_acc, eik_acc, red, partial + (size_t)blockIdx.x * kPartialStride, ta.train_decoder != 0);
}


// ---- `neuralpoints.weighted_first: False` with the analytic eikonal term (utils/mapper.py:676-680, 695-696) ---------------
// Every neighbour's own input in_k = [feat_k | x - p_k] is decoded and the K SDFs are blended: sdf = sum_k w_k s_k, so
//   g = d sdf / d x = sum_k (dw_k s_k + w_k u_k[8:11]),      u_k = s (W2 .* a_k) W1  (the decoder's d s_k / d in_k)
// and with delta = dL/dsdf, c = dL/dg the backward THROUGH g has first-order form per neighbour (a_k is piecewise constant):
//   dL/ds_k = w_k delta + c . dw_k =: ck_k,        dL/du_k = [0_8 ; w_k c] =: t_k
//   dW1[h,:] += ck_k s W2[h] a_k[h] [in_k | 1] + s W2[h] a_k[h] t_k,   dW2[h] += ck_k s relu(pre_k[h]) + s a_k[h] (W1[h,:] . t_k)
//   db2 += ck_k s,     d feat_k = ck_k u_k[0:8]  (then the row's layer-norm backward)
// Hoisted schedule only (records of plain tasks).  Layout as k_train_analytic: 4 queries per wave round, lane16 = 2 k + half;
// the decoder is evaluated once per neighbour with in_k broadcast over the query's 16 lanes (pass 1: s_k, u_k -> g, losses;
// pass 2: the same evaluation again for the gradients -- recomputing 44 FMAs per lane beats keeping 6 x 4 pre-activations).
__global__ void __launch_bounds__(CLID_BLOCK, 1)  // (208 B of scratch per lane when held to 256 registers)
k_train_analytic_wf0(clid_map_view mv, clid_train_args ta, float* __restrict__ partial, int n_rounds, const float4* __restrict__ rec) {
  __shared__ MlpLds mlp;
  __shared__ float red[(CLID_BLOCK / 64) * kRedFloats];
  __shared__ PairLds pairs[CLID_BLOCK / 64];
  stage_mlp(mlp, ta.W1, ta.b1, ta.W2, ta.b2);
  const int lane = threadIdx.x & 63, lane16 = lane & 15, gbase = lane & 48, grp = lane >> 4;
  const int wave = threadIdx.x >> 6, waves_per_block = CLID_BLOCK / 64;
  const int my_k = lane16 >> 1;
  const bool odd = lane16 & 1;
  MlpAcc acc;
  acc.zero();
  float bce_acc = 0.f, eik_acc = 0.f;
  float w1r[CLID_HPL][CLID_D], b1r[CLID_HPL], w2r[CLID_HPL];
#pragma unroll
  for (int uu = 0; uu < CLID_HPL; ++uu) {
    const int h = lane16 + CLID_G * uu;
#pragma unroll
    for (int c = 0; c < CLID_D; ++c) w1r[uu][c] = mlp.w[h * CLID_D + c];
    b1r[uu] = mlp.w[CLID_H * CLID_D + h];
    w2r[uu] = mlp.w[CLID_H * CLID_D + CLID_H + h];
  }
  const float b2r = mlp.w[CLID_MLP_PARAMS - 1];
  const int gstride = ta.grad_stride == CLID_GRAD_ROW16 ? CLID_GRAD_ROW16 : CLID_F;
  const bool merged = gstride == CLID_GRAD_ROW16;
  float* g_theta = ta.grad + CLID_GRAD_OFFSET(gstride);
  const float inv_sigma = fdiv(1.0f, ta.sigma);
  const float4* pos4 = reinterpret_cast<const float4*>(mv.pos4);
  const float sc = ta.sdf_scale;

  for (int rd = blockIdx.x * waves_per_block + wave; rd < n_rounds; rd += gridDim.x * waves_per_block) {
    const int p_raw = rd * 4 + grp;
    bool live = p_raw < ta.bs;
    const float4* r = rec + (size_t)(p_raw >> 3) * 48;  // qinfo[8] | qdesc[8] | win[8][8] float2
    const int slot = p_raw & 7;
    const float4 qi = r[live ? slot : 0], qd = r[8 + (live ? slot : 0)];
    live = live && __float_as_int(qd.x) >= 0;
    const float px = qi.x, py = qi.y, pz = qi.z;
    const int rec_ts = __float_as_int(qi.w);
    const float rec_label = qd.z, rec_wt = qd.w;
    const float2 wn = reinterpret_cast<const float2*>(r + 16)[(live ? slot : 0) * 8 + (my_k < CLID_K ? my_k : 0)];
    float my_w = 0.f;
    int my_j = -1;
    if (my_k < CLID_K) {
      my_w = wn.x;
      my_j = __float_as_int(wn.y);
    }
    if (!live) my_j = -1;
    const bool valid = my_j >= 0;
    if (!valid) my_w = 0.f;
    const int jc = valid ? my_j : 0;
    float4 v = reinterpret_cast<const float4*>(mv.feat)[(size_t)jc * 2 + (odd ? 1 : 0)];
    const float4 pj = pos4[jc];
    const int ts_old = (valid && !odd && mv.ts_update) ? mv.ts_update[jc] : 0x7fffffff;
    if (!valid) v = make_float4(0.f, 0.f, 0.f, 0.f);
    float rstd = 1.f;
    if (mv.layer_norm) {  // np.py:632-633; an all-zero (invalid) row stays zero
      float s1 = (v.x + v.y) + (v.z + v.w);
      s1 += CLID_SWAP(s1);
      const float mu = s1 * (1.0f / CLID_F);
      v.x -= mu; v.y -= mu; v.z -= mu; v.w -= mu;
      float s2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      s2 += CLID_SWAP(s2);
      rstd = 1.0f / sqrtf(s2 * (1.0f / CLID_F) + 1e-5f);
      v.x *= rstd; v.y *= rstd; v.z *= rstd; v.w *= rstd;
    }
    const float rx = valid ? fsub(px, pj.x) : 0.f, ry = valid ? fsub(py, pj.y) : 0.f, rz = valid ? fsub(pz, pj.z) : 0.f;
    const float my_om = valid ? fdiv(1.0f, fadd(fadd(fadd(fmul(rx, rx), fmul(ry, ry)), fmul(rz, rz)), 1e-15f)) : 0.f;
    // d w_k / d x
    const float alx = 2.f * rx * my_om, aly = 2.f * ry * my_om, alz = 2.f * rz * my_om;
    float abx = my_w * alx, aby = my_w * aly, abz = my_w * alz;
    CLID_BFLY(abx) CLID_BFLY(aby) CLID_BFLY(abz)
    const float dwx = my_w * (abx - alx), dwy = my_w * (aby - aly), dwz = my_w * (abz - alz);

    // the decoder on neighbour k's input, broadcast from its lane pair: pre-activations of this lane's hidden units
    auto load_in = [&](int k, float (&fk)[CLID_D]) {
      const int l0 = gbase + 2 * k;
      fk[0] = __shfl(v.x, l0, 64); fk[1] = __shfl(v.y, l0, 64); fk[2] = __shfl(v.z, l0, 64); fk[3] = __shfl(v.w, l0, 64);
      fk[4] = __shfl(v.x, l0 + 1, 64); fk[5] = __shfl(v.y, l0 + 1, 64); fk[6] = __shfl(v.z, l0 + 1, 64); fk[7] = __shfl(v.w, l0 + 1, 64);
      fk[8] = __shfl(rx, l0, 64); fk[9] = __shfl(ry, l0, 64); fk[10] = __shfl(rz, l0, 64);
    };
    auto eval = [&](const float (&fk)[CLID_D], float (&pre)[CLID_HPL]) -> float {
      float part = 0.f;
#pragma unroll
      for (int uu = 0; uu < CLID_HPL; ++uu) {
        float a = b1r[uu];
#pragma unroll
        for (int c = 0; c < CLID_D; ++c) a = fmaf(w1r[uu][c], fk[c], a);
        pre[uu] = a;
        part = fmaf(w2r[uu], fmaxf(a, 0.f), part);
      }
      return sc * (group_sum(part) + b2r);
    };

    // ---- pass 1: s_k and u_k of every neighbour; this lane keeps those of ITS neighbour
    float s_mine = 0.f, us0 = 0.f, us1 = 0.f, us2 = 0.f, us3 = 0.f, u8 = 0.f, u9 = 0.f, u10 = 0.f;
#pragma unroll
    for (int k = 0; k < CLID_K; ++k) {
      float fk[CLID_D], pre[CLID_HPL], e[CLID_HPL], u[CLID_D];
      load_in(k, fk);
      const float sk_val = eval(fk, pre);
#pragma unroll
      for (int uu = 0; uu < CLID_HPL; ++uu) e[uu] = pre[uu] > 0.f ? sc * w2r[uu] : 0.f;
#pragma unroll
      for (int c = 0; c < CLID_D; ++c) {
        float part = 0.f;
#pragma unroll
        for (int uu = 0; uu < CLID_HPL; ++uu) part = fmaf(w1r[uu][c], e[uu], part);
        u[c] = group_sum(part);
      }
      if (my_k == k) {
        s_mine = sk_val;
        us0 = odd ? u[4] : u[0]; us1 = odd ? u[5] : u[1]; us2 = odd ? u[6] : u[2]; us3 = odd ? u[7] : u[3];
        u8 = u[8]; u9 = u[9]; u10 = u[10];
      }
    }
    // blended SDF and g (the sums run over one parity class = the K neighbours; both classes carry the same value)
    float sdf = my_w * s_mine;
    float gx = s_mine * dwx + my_w * u8, gy = s_mine * dwy + my_w * u9, gz = s_mine * dwz + my_w * u10;
    CLID_BFLY(sdf) CLID_BFLY(gx) CLID_BFLY(gy) CLID_BFLY(gz)

    // ---- training_mode side effects (np.py:708-733)
    if (valid && !odd && lane16 < 2 * CLID_K) {
      if (!merged) atomicAdd(&mv.cert[my_j], my_w);
      if (ts_old < rec_ts) atomicMax(&mv.ts_update[my_j], rec_ts);
    }

    // ---- losses and their derivatives
    float delta = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
    if (live) {
      const float z = sdf * inv_sigma;
      const float tgt = __frcp_rn(1.0f + __expf(-rec_label * inv_sigma));     // loss.py:60
      const float ez = __expf(-fabsf(z));
      const float sg = (z >= 0.f ? 1.0f : ez) * __frcp_rn(1.0f + ez);
      const float li = fmaxf(z, 0.f) - z * tgt + __logf(1.0f + ez);           // BCEWithLogits
      if (lane16 == 0) bce_acc += rec_wt * li;
      delta = rec_wt * (sg - tgt) * inv_sigma * ta.inv_n_main;
      if (ta.weight_e > 0.f) {
        const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
        if (lane16 == 0) eik_acc += (nrm - 1.f) * (nrm - 1.f);
        const float coef = nrm > 0.f ? ta.weight_e * 2.f * (nrm - 1.f) * ta.inv_n_eik / nrm : 0.f;
        cx = coef * gx; cy = coef * gy; cz = coef * gz;
      }
    }
    const float ck = my_w * delta + (cx * dwx + cy * dwy + cz * dwz);  // dL/ds_k of this lane's neighbour

    // ---- pass 2: decoder gradients, neighbour by neighbour
    if (ta.train_decoder) {
#pragma unroll
      for (int k = 0; k < CLID_K; ++k) {
        float fk[CLID_D], pre[CLID_HPL];
        load_in(k, fk);
        (void)eval(fk, pre);
        const float ck_k = __shfl(ck, gbase + 2 * k, 64), w_k = __shfl(my_w, gbase + 2 * k, 64);
        const float dz = sc * ck_k;
        const float tq0 = w_k * cx, tq1 = w_k * cy, tq2 = w_k * cz;  // dL/du_k[8:11]
        float fb = (lane16 == CLID_D) ? 1.0f : 0.f, tb = 0.f;
#pragma unroll
        for (int c = 0; c < CLID_D; ++c) fb = (lane16 == c) ? fk[c] : fb;
        tb = lane16 == 8 ? tq0 : (lane16 == 9 ? tq1 : (lane16 == 10 ? tq2 : 0.f));
#pragma unroll
        for (int uu = 0; uu < CLID_HPL; ++uu) {
          const bool on = pre[uu] > 0.f;
          const float e = on ? sc * w2r[uu] : 0.f;
          const float dh = on ? dz * w2r[uu] : 0.f;
          const float w1t = fmaf(w1r[uu][8], tq0, fmaf(w1r[uu][9], tq1, w1r[uu][10] * tq2));
          acc.dW2[uu] += on ? (dz * pre[uu] + sc * w1t) : 0.f;
          acc.dW1[uu] = __builtin_amdgcn_mfma_f32_16x16x4f32(dh, fb, acc.dW1[uu], 0, 0, 0);
          acc.dW1[uu] = __builtin_amdgcn_mfma_f32_16x16x4f32(e, tb, acc.dW1[uu], 0, 0, 0);
        }
        if (lane16 == 0) acc.db2 += dz;
      }
    }

    // ---- feature gradients: d feat_k = ck_k u_k[0:8] on the (normalised) row of neighbour k
    {
      float d0 = ck * us0, d1 = ck * us1, d2 = ck * us2, d3 = ck * us3;
      if (mv.layer_norm) {
        float m1 = (d0 + d1) + (d2 + d3);
        float m2 = (d0 * v.x + d1 * v.y) + (d2 * v.z + d3 * v.w);
        m1 += CLID_SWAP(m1);
        m2 += CLID_SWAP(m2);
        m1 *= (1.0f / CLID_F);
        m2 *= (1.0f / CLID_F);
        d0 = rstd * (d0 - m1 - v.x * m2); d1 = rstd * (d1 - m1 - v.y * m2);
        d2 = rstd * (d2 - m1 - v.z * m2); d3 = rstd * (d3 - m1 - v.w * m2);
      }
      if (merged) {
        scatter_pairs_rows16(pairs[wave], lane, grp, my_k, odd, valid ? my_j : -1, d0, d1, d2, d3, valid ? my_w : 0.f, g_theta);
      } else if (valid && lane16 < 2 * CLID_K && ck != 0.f) {
        float* dst = g_theta + (size_t)my_j * gstride + (odd ? 4 : 0);
        atomicAdd(dst + 0, d0); atomicAdd(dst + 1, d1); atomicAdd(dst + 2, d2); atomicAdd(dst + 3, d3);
      }
    }
  }
  flush_mlp_acc(acc, bce_acc, eik_acc, red, partial + (size_t)blockIdx.x * kPartialStride, ta.train_decoder != 0);
}

#undef CLID_BFLY
#undef CLID_SWAP

}  // namespace clid

#ifndef CLID_ANALYTIC_MAX_BLOCKS
#define CLID_ANALYTIC_MAX_BLOCKS 512  // two rounds per wave, two blocks per CU: 20.3 -> 17.0 us, and k_adam_all sums 512 partial rows (7.8 -> 5.2 us)
#endif
int clid_train_analytic_blocks(int bs) {
  const int rounds = (bs + 3) / 4;
  int nb = (rounds + CLID_BLOCK / 64 - 1) / (CLID_BLOCK / 64);
  return nb > CLID_ANALYTIC_MAX_BLOCKS ? CLID_ANALYTIC_MAX_BLOCKS : (nb < 1 ? 1 : nb);
}

int clid_launch_train_analytic(const clid_map_view* mv, const clid_train_args* a, float* partial, const float* rec, hipStream_t s) {
  if (a->decimation != 1) {
    clid_set_error("clid_train_fwd_bwd: analytic eikonal mode expects gradient_decimation == 1 (utils/config.py:645-646), got %d",
                   a->decimation);
    return CLID_E_ARG;
  }
  const int rounds = (a->bs + 3) / 4;
  if (a->decode_each_neighbour) {  // neuralpoints.weighted_first: False
    if (!rec) {
      clid_set_error("clid_train_fwd_bwd: decode_each_neighbour with the analytic eikonal term runs on the hoisted schedule only");
      return CLID_E_ARG;
    }
    CLID_KLAUNCH(a->prof, 0, clid::k_train_analytic_wf0, dim3(clid_train_analytic_blocks(a->bs)), dim3(CLID_BLOCK), 0, s, *mv, *a,
                 partial, rounds, reinterpret_cast<const float4*>(rec));
    CLID_CHECK_LAUNCH();
    return CLID_OK;
  }
  if (rec && (a->g_out || a->c_extra)) {  // config.consistency_loss_on: the probe / the coupled backward
    const dim3 grid(clid_train_analytic_blocks(a->bs)), block(CLID_BLOCK);
    const float4* r4 = reinterpret_cast<const float4*>(rec);
    if (a->g_out) CLID_KLAUNCH(a->prof, 0, (clid::k_train_analytic<true, false, 1>), grid, block, 0, s, *mv, *a, partial, rounds, r4);
    else if (a->proj_correction) CLID_KLAUNCH(a->prof, 0, (clid::k_train_analytic<true, true, 2>), grid, block, 0, s, *mv, *a, partial, rounds, r4);
    else CLID_KLAUNCH(a->prof, 0, (clid::k_train_analytic<true, false, 2>), grid, block, 0, s, *mv, *a, partial, rounds, r4);
  } else if (rec && a->proj_correction)
    CLID_KLAUNCH(a->prof, 0, (clid::k_train_analytic<true, true>), dim3(clid_train_analytic_blocks(a->bs)), dim3(CLID_BLOCK), 0, s, *mv, *a,
                 partial, rounds, reinterpret_cast<const float4*>(rec));
  else if (rec)
    CLID_KLAUNCH(a->prof, 0, clid::k_train_analytic<true>, dim3(clid_train_analytic_blocks(a->bs)), dim3(CLID_BLOCK), 0, s, *mv, *a,
                 partial, rounds, reinterpret_cast<const float4*>(rec));
  else
    CLID_KLAUNCH(a->prof, 0, clid::k_train_analytic<false>, dim3(clid_train_analytic_blocks(a->bs)), dim3(CLID_BLOCK), 0, s, *mv, *a,
                 partial, rounds, static_cast<const float4*>(nullptr));
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}
