// Mapping iteration with `weighted_first: False` (utils/mapper.py:679-680, model/neural_points.py:745-754 skipped): every
// neighbour's own decoder input [LN(feat_k) | x - p_k] is decoded and the K SDFs are blended with the IDW weights,
//   sdf(x) = sum_k w_k s (W2 relu(W1 v_k + b1) + b2),
// for the batch samples AND the six shifted copies of every `decimation`-th sample (Mapper.sdf / get_numerical_gradient,
// utils/mapper.py:968-1034), BCE + numerical eikonal loss (utils/loss.py:44-62, utils/mapper.py:746-798), backward.  No shipped
// config uses it; it was the one mode of Mapper.mapping that still ran the reference's op sequence on autograd kernels (and
// could not be sharded).  Hoisted schedule only: neighbours and weights come from the search records.
//
// Six decoder evaluations per query point instead of one, so the finite-difference coupling is resolved BETWEEN launches
// instead of inside a wave (three launches per iteration, 16 lanes per query as csrc/train_analytic.hip):
//   pass 0  shifted copies, forward only: SDF -> the copy's record slot
//   pass 1  batch samples: forward, BCE; a decimated sample also reads its six copies' SDFs (same task record), adds the
//           eikonal term and leaves the copies' upstream gradients in their slots; backward of the sample
//   pass 2  shifted copies again: forward, backward with that upstream gradient
// The two scratch values of a shifted copy live in the label / loss-weight fields of ITS OWN record slot (qdesc.z / .w, which
// mean nothing for a copy): no side buffer, no capacity bound, and a record can be decoded again (pass 0 rewrites them).
// Per neighbour k the backward is the single-query decoder backward (train_common.hpp mlp_backward) with dz = s w_k dL/dsdf;
// the rows' gradients go through the layer-norm backward on the two lanes that loaded the row and out as ONE request per
// (query, neighbour) pair (scatter_pairs_rows16, train_common.hpp).
#include "train_common.hpp"

namespace clid {

constexpr int kWf0Block = 256, kWf0Groups = kWf0Block / CLID_G;  // 16 query points per block and round
#ifndef CLID_WF0_BLOCKS
#define CLID_WF0_BLOCKS 512
#endif
constexpr int kWf0MaxBlocks = CLID_WF0_BLOCKS;  // per pass; passes 1 and 2 leave one partial row per block
static_assert(2 * kWf0MaxBlocks <= kMaxBwdBlocks, "partial rows");

struct Wf0Lds {
  float v[kWf0Groups][CLID_K][12];  // per query group and neighbour: the decoder input (11) | 1 (the bias column of dW1)
};

template <int PASS>
__global__ void __launch_bounds__(kWf0Block, 2)  // (2 waves per SIMD: what <= 512 blocks put there)
k_train_wf0(clid_map_view mv, clid_train_args ta, float* __restrict__ partial, int row_base, TaskMap tmap, float4* __restrict__ rec) {
  __shared__ MlpLds mlp;
  __shared__ Wf0Lds wl;
  __shared__ float red[(kWf0Block / 64) * kRedFloats];
  __shared__ PairLds pairs[kWf0Block / 64];  // merged scatter on 16-float accumulation rows (train_common.hpp)
  stage_mlp(mlp, ta.W1, ta.b1, ta.W2, ta.b2);
  const int lane = threadIdx.x & 63, lane16 = lane & 15, gib = threadIdx.x >> 4;
  const int my_k = lane16 >> 1;
  const bool odd = lane16 & 1;
#ifndef CLID_WF0_REGW
#define CLID_WF0_REGW 1
#endif
#if CLID_WF0_REGW
  // this lane's share of the decoder (hidden units lane16 + 16 u) in registers: six forward and six backward walks per query point
  const int l16 = lane16;
  float w1r[CLID_HPL][CLID_D], b1r[CLID_HPL], w2r[CLID_HPL];
#pragma unroll
  for (int uu = 0; uu < CLID_HPL; ++uu) {
    const int h = lane16 + CLID_G * uu;
#pragma unroll
    for (int c = 0; c < CLID_D; ++c) w1r[uu][c] = mlp.w[h * CLID_D + c];
    b1r[uu] = mlp.w[CLID_H * CLID_D + h];
    w2r[uu] = mlp.w[CLID_H * CLID_D + CLID_H + h];
  }
#define WF0_W1(uu, h, c) w1r[uu][c]
#define WF0_B1(uu, h) b1r[uu]
#define WF0_W2(uu, h) w2r[uu]
#else
  const int l16 = lane16 + opaque_zero();  // (keeps the decoder weights in LDS, common.hpp)
#define WF0_W1(uu, h, c) mlp.w[(h) * CLID_D + (c)]
#define WF0_B1(uu, h) mlp.w[CLID_H * CLID_D + (h)]
#define WF0_W2(uu, h) mlp.w[CLID_H * CLID_D + CLID_H + (h)]
#endif
  MlpAcc acc;
  acc.zero();
  float bce_acc = 0.f, eik_acc = 0.f;
  const int gstride = ta.grad_stride == CLID_GRAD_ROW16 ? CLID_GRAD_ROW16 : CLID_F;
  float* g_theta = ta.grad + CLID_GRAD_OFFSET(gstride);
  const bool merged = gstride == CLID_GRAD_ROW16;
  const float inv_sigma = fdiv(1.0f, ta.sigma);
  const float inv_two_eps = fdiv(1.0f, 2.0f * ta.fd_eps);
  const float4* pos4 = reinterpret_cast<const float4*>(mv.pos4);
  const float sc = ta.sdf_scale;
  const int n_fd = tmap.n_fd;
  const int n_slots = (PASS == 1 ? tmap.n_tasks : n_fd) * 8;  // the shifted copies live in the bundle tasks 0 .. n_fd-1 (train_common.hpp TaskMap)
  const bool train = ta.train_decoder != 0;

  for (int u0 = blockIdx.x * kWf0Groups; u0 < n_slots; u0 += gridDim.x * kWf0Groups) {
    const int u = u0 + gib;
    const bool inr = u < n_slots;
    const int task = (inr ? u : 0) >> 3, slot = (inr ? u : 0) & 7;
    float4* r = rec + (size_t)task * 48;  // qinfo[8] | qdesc[8] | win[8][8] float2
    const float4 qi = r[slot], qd = r[8 + slot];
    const int p = __float_as_int(qd.x), code = __float_as_int(qd.y);
    const bool act = inr && p >= 0 && (PASS == 1 ? code < 0 : code >= 0);
    // bundle task j < n_fd (train_common.hpp task_query): slots 0..5 = the copies x+ x- y+ y- z+ z- (code = slot ^ 1),
    // slot 6 = the decimated sample itself, slot 7 = the sample in front of it
    const bool on_lattice = task < n_fd && slot == 6;
    float* scratch = reinterpret_cast<float*>(r + 8);  // qdesc[s] = scratch[4 s ..]: .z = SDF of copy s, .w = its upstream gradient
    const float2* win = reinterpret_cast<const float2*>(r + 16) + slot * 8;
    float w6[CLID_K];
    float my_w = 0.f;
    int my_j = -1;
#pragma unroll
    for (int k = 0; k < CLID_K; ++k) {
      const float2 wn = win[k];
      const bool has = act && __float_as_int(wn.y) >= 0;
      w6[k] = has ? wn.x : 0.f;  // IDW weight from the search record (np.py:688-706)
      if (my_k == k && has) {
        my_w = wn.x;
        my_j = __float_as_int(wn.y);
      }
    }
    const bool valid = my_j >= 0;
    const int jc = valid ? my_j : 0;
    float4 v = reinterpret_cast<const float4*>(mv.feat)[(size_t)jc * 2 + (odd ? 1 : 0)];
    const float4 pj = pos4[jc];
    if (!valid) v = make_float4(0.f, 0.f, 0.f, 0.f);
    float rstd = 1.f;
    if (mv.layer_norm) {  // np.py:632-633; an all-zero (invalid) row stays zero
      float s1 = (v.x + v.y) + (v.z + v.w);
      s1 += dpp_mov<0xB1>(s1);
      const float mu = s1 * (1.0f / CLID_F);
      v.x -= mu; v.y -= mu; v.z -= mu; v.w -= mu;
      float s2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      s2 += dpp_mov<0xB1>(s2);
      rstd = 1.0f / sqrtf(s2 * (1.0f / CLID_F) + 1e-5f);
      v.x *= rstd; v.y *= rstd; v.z *= rstd; v.w *= rstd;
    }
    if (my_k < CLID_K) {
      float* dst = &wl.v[gib][my_k][0];
      *reinterpret_cast<float4*>(dst + (odd ? 4 : 0)) = v;
      if (!odd) {
        dst[8] = valid ? fsub(qi.x, pj.x) : 0.f;
        dst[9] = valid ? fsub(qi.y, pj.y) : 0.f;
        dst[10] = valid ? fsub(qi.z, pj.z) : 0.f;
        dst[11] = 1.0f;
      }
    }
    wave_lds_fence();

    // ================= forward: one decoder evaluation per neighbour, blended SDF
    float pre[CLID_K][CLID_HPL];
    float sdf = 0.f;
#pragma unroll
    for (int k = 0; k < CLID_K; ++k) {
      float vk[CLID_D];
#pragma unroll
      for (int c = 0; c < CLID_D; ++c) vk[c] = wl.v[gib][k][c];
      float part = 0.f;
#pragma unroll
      for (int uu = 0; uu < CLID_HPL; ++uu) {
        const int h = l16 + CLID_G * uu;
        float a = WF0_B1(uu, h);
#pragma unroll
        for (int c = 0; c < CLID_D; ++c) a = fmaf(WF0_W1(uu, h, c), vk[c], a);
        pre[k][uu] = a;
        part = fmaf(WF0_W2(uu, h), fmaxf(a, 0.f), part);
      }
      const float sdf_k = sc * (group_sum(part) + mlp.w[CLID_MLP_PARAMS - 1]);
      sdf = fadd(sdf, fmul(sdf_k, w6[k]));  // utils/mapper.py:679-680 (a neighbour slot without a point has weight 0)
    }

    if (PASS == 0) {
      if (act && lane16 == 0) scratch[4 * slot + 2] = sdf;
      wave_lds_fence();
      continue;
    }

    // ================= upstream gradient of this query point's SDF
    float dsdf = 0.f;
    if (PASS == 1) {
      if (act) {
        const float label = qd.z, wt = qd.w;  // (|weight| or 1: resolved by the search launch, mapper.py:747-749)
        const float z = sdf * inv_sigma;
        const float tgt = __frcp_rn(1.0f + __expf(-label * inv_sigma));  // loss.py:60
        const float ez = __expf(-fabsf(z));
        const float sg = (z >= 0.f ? 1.0f : ez) * __frcp_rn(1.0f + ez);
        const float li = fmaxf(z, 0.f) - z * tgt + __logf(1.0f + ez);    // BCEWithLogits
        if (lane16 == 0) bce_acc += wt * li;
        dsdf = wt * (sg - tgt) * inv_sigma * ta.inv_n_main;
        if (on_lattice && ta.weight_e > 0.f) {  // numerical eikonal term of this sample (mapper.py:1011-1013, 795-797)
          const float gx = (scratch[4 * 0 + 2] - scratch[4 * 1 + 2]) * inv_two_eps;
          const float gy = (scratch[4 * 2 + 2] - scratch[4 * 3 + 2]) * inv_two_eps;
          const float gz = (scratch[4 * 4 + 2] - scratch[4 * 5 + 2]) * inv_two_eps;
          const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
          if (lane16 == 0) eik_acc += (nrm - 1.f) * (nrm - 1.f);
          const float ecoef = nrm > 0.f ? ta.weight_e * 2.f * (nrm - 1.f) * ta.inv_n_eik * inv_two_eps / nrm : 0.f;
          if (lane16 < 6) {  // slot s: axis s >> 1, the + copy on the even slot
            const float ga = lane16 < 2 ? gx : (lane16 < 4 ? gy : gz);
            scratch[4 * lane16 + 3] = ((lane16 & 1) ? -1.f : 1.f) * ecoef * ga;
          }
        }
      }
    } else {
      dsdf = act ? scratch[4 * slot + 3] : 0.f;
    }

    // ================= training_mode side effects (np.py:708-733): batch samples and shifted copies alike
    if (valid && !odd) {
      if (!merged) atomicAdd(&mv.cert[my_j], my_w);
      const int ts = __float_as_int(qi.w);  // amax is idempotent: only a newer stamp needs the atomic
      if (PASS == 1 && mv.ts_update && ta.pool_ts && mv.ts_update[my_j] < ts) atomicMax(&mv.ts_update[my_j], ts);
    }

    // ================= backward, neighbour by neighbour
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;  // this lane's half of d L / d (normalised) row of ITS neighbour
#pragma unroll
    for (int k = 0; k < CLID_K; ++k) {
      const float dz = sc * dsdf * w6[k];  // (0 for an inactive query point or an empty neighbour slot)
      float dh[CLID_HPL];
#pragma unroll
      for (int uu = 0; uu < CLID_HPL; ++uu) {
        const int h = l16 + CLID_G * uu;
        const bool on = pre[k][uu] > 0.f;
        dh[uu] = on ? dz * WF0_W2(uu, h) : 0.f;
        if (train) acc.dW2[uu] += on ? dz * pre[k][uu] : 0.f;
      }
      if (train) {
        const float fb = lane16 < 12 ? wl.v[gib][k][lane16] : 0.f;  // input c on lane c, 1 on lane 11 (db1)
#pragma unroll
        for (int uu = 0; uu < CLID_HPL; ++uu)
          acc.dW1[uu] = __builtin_amdgcn_mfma_f32_16x16x4f32(dh[uu], fb, acc.dW1[uu], 0, 0, 0);
        if (lane16 == 0) acc.db2 += dz;
      }
      float dv[CLID_F];
#pragma unroll
      for (int c = 0; c < CLID_F; ++c) {
        float part = 0.f;
#pragma unroll
        for (int uu = 0; uu < CLID_HPL; ++uu) part = fmaf(WF0_W1(uu, l16 + CLID_G * uu, c), dh[uu], part);
        dv[c] = group_sum(part);
      }
      if (my_k == k) {
        d0 = odd ? dv[4] : dv[0]; d1 = odd ? dv[5] : dv[1]; d2 = odd ? dv[6] : dv[2]; d3 = odd ? dv[7] : dv[3];
      }
    }
    if (mv.layer_norm) {  // dx = rstd (dy - mean(dy) - xhat mean(dy xhat)) over the 8 features (pair of lanes)
      float m1 = (d0 + d1) + (d2 + d3);
      float m2 = (d0 * v.x + d1 * v.y) + (d2 * v.z + d3 * v.w);
      m1 += dpp_mov<0xB1>(m1);
      m2 += dpp_mov<0xB1>(m2);
      m1 *= (1.0f / CLID_F);
      m2 *= (1.0f / CLID_F);
      d0 = rstd * (d0 - m1 - v.x * m2); d1 = rstd * (d1 - m1 - v.y * m2);
      d2 = rstd * (d2 - m1 - v.z * m2); d3 = rstd * (d3 - m1 - v.w * m2);
    }
    if (merged) {
      scatter_pairs_rows16(pairs[threadIdx.x >> 6], lane, lane >> 4, my_k, odd, valid ? my_j : -1, d0, d1, d2, d3,
                           valid ? my_w : 0.f, g_theta);
    } else if (valid && dsdf != 0.f) {
      float* dst = g_theta + (size_t)my_j * gstride + (odd ? 4 : 0);
      atomicAdd(dst + 0, d0); atomicAdd(dst + 1, d1); atomicAdd(dst + 2, d2); atomicAdd(dst + 3, d3);
    }
    wave_lds_fence();
  }
  if (PASS != 0)
    flush_mlp_acc(acc, bce_acc, eik_acc, red, partial + (size_t)(row_base + blockIdx.x) * kPartialStride, train);
}

}  // namespace clid

using namespace clid;

static int wf0_blocks(int n_tasks) {
  const int nb = (n_tasks * 8 + kWf0Groups - 1) / kWf0Groups;
  return nb > kWf0MaxBlocks ? kWf0MaxBlocks : (nb < 1 ? 1 : nb);
}
// partial rows an iteration leaves for clid_train_adam: one set per gradient pass
int clid_train_wf0_rows(int n_tasks, int n_fd) { return wf0_blocks(n_tasks) + (n_fd > 0 ? wf0_blocks(n_fd) : 0); }

int clid_launch_train_wf0(const clid_map_view* mv, const clid_train_args* a, float* partial, const TaskMap& tmap, float* rec,
                          hipStream_t s) {
  const int nb = wf0_blocks(tmap.n_tasks), nb_fd = wf0_blocks(tmap.n_fd);
  float4* r4 = reinterpret_cast<float4*>(rec);
  if (tmap.n_fd > 0) CLID_KLAUNCH(a->prof, 0, k_train_wf0<0>, dim3(nb_fd), dim3(kWf0Block), 0, s, *mv, *a, partial, 0, tmap, r4);
  CLID_KLAUNCH(a->prof, 0, k_train_wf0<1>, dim3(nb), dim3(kWf0Block), 0, s, *mv, *a, partial, 0, tmap, r4);
  if (tmap.n_fd > 0) CLID_KLAUNCH(a->prof, 0, k_train_wf0<2>, dim3(nb_fd), dim3(kWf0Block), 0, s, *mv, *a, partial, nb, tmap, r4);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}
