// Device-side building blocks shared by the gfx950 kernels.  Wave = 64 lanes; a QUERY GROUP is one
// DPP row (16 lanes): 4 queries per wave.  Everything after the top-K selection is replicated across
// the 16 lanes of a group except the 64 hidden units of the decoder (4 per lane).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/clid_native.h"

#define CLID_G 16                 // lanes per query
#define CLID_HPL (CLID_H / CLID_G)  // hidden units per lane = 4
#define CLID_BLOCK 256
#define CLID_QPB (CLID_BLOCK / CLID_G)  // queries per block per round = 16

namespace clid {

// ---- optional in-kernel phase timing (tools/phase_timing.py builds with -DCLID_TIMING) -----------------
#ifdef CLID_TIMING
static __device__ long long clid_stamps[256 * 32];
#define CLID_STAMP(k)                                                                   \
  do {                                                                                  \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                          \
    if (threadIdx.x == 0 && blockIdx.x < 256 && (k) >= 0 && (k) < 32)                    \
      clid_stamps[blockIdx.x * 32 + (k)] = __builtin_amdgcn_s_memtime();                 \
  } while (0)
#else
#define CLID_STAMP(k) do { } while (0)
#endif

// ---- exact fp32 helpers (no FMA contraction where the reference's op order matters) -------------
__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float fdiv(float a, float b) { return __fdiv_rn(a, b); }

// ---- DPP row (16-lane) primitives ---------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// row_ror:n == 0x120 + n : every lane of the row ends with the full reduction
__device__ __forceinline__ float group_sum(float v) {
  v += dpp_mov<0x128>(v);
  v += dpp_mov<0x124>(v);
  v += dpp_mov<0x122>(v);
  v += dpp_mov<0x121>(v);
  return v;
}
__device__ __forceinline__ float group_min(float v) {
  v = fminf(v, dpp_mov<0x128>(v));
  v = fminf(v, dpp_mov<0x124>(v));
  v = fminf(v, dpp_mov<0x122>(v));
  v = fminf(v, dpp_mov<0x121>(v));
  return v;
}
__device__ __forceinline__ int group_sum_i(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xF, 0xF, false);
  return v;
}
// sum over the 4 groups of a wave (lanes l, l^16, l^32, l^48)
__device__ __forceinline__ float cross_group_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  return cross_group_sum(group_sum(v));
}

// ---- voxel hashing (model/neural_points.py:984-999) ---------------------------------------------
// cell = floor(x / res) with a TRUE fp32 divide; h = sum cell_c * prime_c in int64;
// slot = h mod B taken non-negative (fmod + negative index wrap).
// The int64 modulo is done in fp64: |h| < 2^52 for |cell| < 2^24, so h, q*B and h - q*B are exact and
// q = floor(h / B) is off by at most one, fixed by the two conditional corrections (exact result, ~15
// instructions instead of a ~150-instruction software 64-bit division).
__device__ __forceinline__ int base_slot(float x, float y, float z, float res, int B) {
  const double cx = (double)floorf(fdiv(x, res));
  const double cy = (double)floorf(fdiv(y, res));
  const double cz = (double)floorf(fdiv(z, res));
  const double h = fma(cx, 73856093.0, fma(cy, 19349669.0, cz * 83492791.0));  // exact: all terms < 2^51
  const double Bd = (double)B;
  const double q = floor(h / Bd);
  double r = fma(-q, Bd, h);
  if (r < 0.0) r += Bd;
  if (r >= Bd) r -= Bd;
  return (int)r;
}

// ---- compact probe table: 4-key buckets (one 16-byte load per probe) ------------------------------------
// keys[bucket] = 4 slot numbers, -1 empty, filled in order; tab_pos[bucket][m] = {x, y, z, id bits} of key m,
// so a hit costs ONE more 16-byte load that returns the position and the id together.  At <= 0.5 keys per
// bucket the chance that a bucket is full (>= 4 keys) without a match is ~2e-4 per probe, so the walk to
// the next bucket is a rare slow path instead of an extra dependent latency for every wave.
// bucket of a slot: xor-fold instead of a multiplicative hash (v_mul_lo_u32 is a quarter-rate instruction and
// this runs 81 times per query); slot = (sum cell*prime) mod B is already well mixed
__device__ __forceinline__ unsigned tab_home(int slot, int log2nb) {
  const unsigned u = (unsigned)slot;
  return (u ^ (u >> 11)) & ((1u << log2nb) - 1u);
}
// bit of a slot in the probe prefilter (a one-hash Bloom filter over the keys of the table): the low bits of the slot number
// itself -- slot = (sum cell * prime) mod B is already well mixed, and the 81 probes of a query pay for every operation here
// (a 4-term xor-fold for the filter and a 3-term one for the bucket cost 4.6 % of the search kernel)
__host__ __device__ __forceinline__ unsigned filter_bit(int slot, int log2bits) {
  return (unsigned)slot & ((1u << log2bits) - 1u);
}
__device__ __forceinline__ int bucket_match(const int4 b, int slot) {
  return (b.x == slot) ? 0 : ((b.y == slot) ? 1 : ((b.z == slot) ? 2 : ((b.w == slot) ? 3 : -1)));
}
// cell index (bucket*4 + m) of `slot`, or -1
__device__ __forceinline__ int tab_find(const int4* __restrict__ tab, int log2nb, int slot, unsigned home, int4 b) {
  const unsigned mask = (1u << log2nb) - 1u;
  for (;;) {
    const int m = bucket_match(b, slot);
    if (m >= 0) return (int)(home * 4u) + m;
    if (b.w < 0) return -1;  // not full: the key would have been here
    home = (home + 1) & mask;
    b = tab[home];
  }
}

// ---- per-lane sorted candidate list + group top-K ------------------------------------------------
struct Cand {
  float d[CLID_K];
  int j[CLID_K];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int s = 0; s < CLID_K; ++s) {
      d[s] = __builtin_inff();
      j[s] = -1;
    }
  }
  __device__ __forceinline__ void insert(float nd, int nj) {
#pragma unroll
    for (int s = 0; s < CLID_K; ++s) {
      const bool lt = nd < d[s];
      const float td = lt ? d[s] : nd;
      const int tj = lt ? j[s] : nj;
      d[s] = lt ? nd : d[s];
      j[s] = lt ? nj : j[s];
      nd = td;
      nj = tj;
    }
  }
  __device__ __forceinline__ void pop() {
#pragma unroll
    for (int s = 0; s < CLID_K - 1; ++s) {
      d[s] = d[s + 1];
      j[s] = j[s + 1];
    }
    d[CLID_K - 1] = __builtin_inff();
    j[CLID_K - 1] = -1;
  }
};

struct TopK {  // replicated across the group
  float d2[CLID_K];
  int j[CLID_K];  // -1 invalid
  int nn;         // valid probes over all P (np.py:600-602)
};

// Per-offset slot deltas staged in LDS, padded with zeros to a multiple of kProbeChunk
constexpr int kProbesPerLane = 6;
constexpr int kProbeChunk = CLID_G * kProbesPerLane;  // 96 >= 81
constexpr int kMaxProbes = 4 * kProbeChunk;           // 384 >= 343 (num_nei_cells = 3)
struct DeltaLds {
  int d[kMaxProbes];
};
// call from all threads of the block BEFORE a __syncthreads()
__device__ __forceinline__ void stage_delta(DeltaLds& s, const clid_map_view& mv) {
  const int padded = (mv.P + kProbeChunk - 1) / kProbeChunk * kProbeChunk;
  for (int i = threadIdx.x; i < padded; i += blockDim.x) s.d[i] = i < mv.P ? mv.delta[i] : 0;
}

// Search the P probe cells of one query (x,y,z) with the 16 lanes of a group and select the K
// nearest valid neighbours, ascending (np.py:971-1030 + 595-612).  All probe loads of a chunk are
// issued before any is consumed (the dependent chain is bucket -> position, not 6x that).
__device__ __forceinline__ void search_topk(const clid_map_view& mv, const DeltaLds& dl, float x, float y,
                                            float z, int lane16, int gbase, TopK& out, int tm = -100) {
  const int4* __restrict__ tab = reinterpret_cast<const int4*>(mv.tab);
  const int B = mv.buffer_size;
  const int r0 = base_slot(x, y, z, mv.resolution, B);
  Cand c;
  c.init();
  int nvalid = 0;
  const float4* __restrict__ tpos = reinterpret_cast<const float4*>(mv.tab_pos);
  // lane l owns the contiguous probes [R l, R (l + 1)), R = ceil(P / 16), in ascending order: with the strict < of the insert
  // and the lowest lane winning ties in the selection below, equal distances are resolved by probe index -- a stable sort
  // of the reference's dist2 row (np.py:607-609, whose torch.sort leaves the order of equal distances undefined)
  const int R = (mv.P + CLID_G - 1) / CLID_G;
  const int obase = R * lane16;
  for (int o0 = 0; o0 < R; o0 += kProbesPerLane) {
    int slot[kProbesPerLane];
    unsigned home[kProbesPerLane];
    int4 bk[kProbesPerLane];
#pragma unroll
    for (int t = 0; t < kProbesPerLane; ++t) {
      const int o = obase + o0 + t;
      const bool in = (o0 + t < R) && (o < mv.P);
      int sl = r0 + dl.d[in ? o : 0];
      if (sl >= B) sl -= B;
      slot[t] = in ? sl : -2;  // -2 never matches a key
      home[t] = tab_home(sl, mv.log2cap);
      bk[t] = tab[slot[t] != -2 ? home[t] : 0];
    }
    int cell[kProbesPerLane];
    bool walk = false;
#pragma unroll
    for (int t = 0; t < kProbesPerLane; ++t) {
      const int m = bucket_match(bk[t], slot[t]);
      cell[t] = m >= 0 ? (int)(home[t] * 4u) + m : -1;
      walk |= (m < 0) && (bk[t].w >= 0) && (slot[t] != -2);
    }
    CLID_STAMP(tm + 1);
    if (__any(walk)) {  // rare: a full bucket without a match
#pragma unroll
      for (int t = 0; t < kProbesPerLane; ++t)
        if (cell[t] < 0 && slot[t] != -2 && bk[t].w >= 0) cell[t] = tab_find(tab, mv.log2cap, slot[t], home[t], bk[t]);
    }
    float4 pp[kProbesPerLane];
#pragma unroll
    for (int t = 0; t < kProbesPerLane; ++t) pp[t] = tpos[cell[t] >= 0 ? cell[t] : 0];
    CLID_STAMP(tm + 2);
#pragma unroll
    for (int t = 0; t < kProbesPerLane; ++t) {
      const float ax = fsub(pp[t].x, x), ay = fsub(pp[t].y, y), az = fsub(pp[t].z, z);
      const float d2 = fadd(fadd(fmul(ax, ax), fmul(ay, ay)), fmul(az, az));
      if (cell[t] >= 0 && !(d2 > mv.max_valid_dist2)) {  // np.py:1016-1020
        c.insert(d2, __float_as_int(pp[t].w));
        ++nvalid;
      }
    }
  }
  out.nn = group_sum_i(nvalid);
  CLID_STAMP(tm + 3);
#pragma unroll
  for (int k = 0; k < CLID_K; ++k) {
    const float head = c.d[0];
    const float m = group_min(head);
    const bool mine = (head == m) && (c.j[0] >= 0);
    const unsigned long long b = __ballot(mine);
    const unsigned gb = (unsigned)(b >> gbase) & 0xFFFFu;
    int owner = gb ? (__ffs(gb) - 1) : 0;
    const int wj = __shfl(c.j[0], gbase + owner, 64);
    out.d2[k] = gb ? m : 9e3f;  // np.py:606
    out.j[k] = gb ? wj : -1;
    if (gb && lane16 == owner) c.pop();
  }
}

// IDW weights (np.py:688-706): w_k = valid_k/(d2_k+eps) normalised; all zero when no neighbour.
__device__ __forceinline__ void idw_weights(const TopK& t, float (&w)[CLID_K], float (&omega)[CLID_K]) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < CLID_K; ++k) {
    omega[k] = (t.j[k] >= 0) ? fdiv(1.0f, fadd(t.d2[k], 1e-15f)) : 0.f;
    s = fadd(s, omega[k]);
  }
  const float inv_s = fdiv(1.0f, s);  // one division instead of six (<= 1 ulp from omega/s)
#pragma unroll
  for (int k = 0; k < CLID_K; ++k) w[k] = (t.j[k] >= 0) ? fmul(omega[k], inv_s) : 0.f;
}

__device__ __forceinline__ void load_feat(const float* __restrict__ feat, int j, float (&v)[CLID_F]) {
  const float4* p = reinterpret_cast<const float4*>(feat + (size_t)j * CLID_F);
  const float4 a = p[0], b = p[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// F.layer_norm over the F features, no affine, eps 1e-5, biased variance (np.py:632-633)
__device__ __forceinline__ void layer_norm8(float (&v)[CLID_F], float& rstd) {
  float mu = 0.f;
#pragma unroll
  for (int c = 0; c < CLID_F; ++c) mu += v[c];
  mu *= (1.0f / CLID_F);
  float var = 0.f;
#pragma unroll
  for (int c = 0; c < CLID_F; ++c) {
    const float d = v[c] - mu;
    var += d * d;
  }
  var *= (1.0f / CLID_F);
  rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
  for (int c = 0; c < CLID_F; ++c) v[c] = (v[c] - mu) * rstd;
}
// dx = rstd * (dy - mean(dy) - xhat * mean(dy*xhat))
__device__ __forceinline__ void layer_norm8_bwd(const float (&xhat)[CLID_F], float rstd, float (&dy)[CLID_F]) {
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int c = 0; c < CLID_F; ++c) {
    m1 += dy[c];
    m2 += dy[c] * xhat[c];
  }
  m1 *= (1.0f / CLID_F);
  m2 *= (1.0f / CLID_F);
#pragma unroll
  for (int c = 0; c < CLID_F; ++c) dy[c] = rstd * (dy[c] - m1 - xhat[c] * m2);
}

// ---- decoder weights staged in LDS ---------------------------------------------------------------
// layout: W1 [H][D] | b1 [H] | W2 [H] | b2 [1]  (== the 833-float parameter/gradient vector)
struct MlpLds {
  float w[CLID_MLP_PARAMS + 3];
};
__device__ __forceinline__ void stage_mlp(MlpLds& s, const float* W1, const float* b1, const float* W2,
                                          const float* b2) {
  for (int i = threadIdx.x; i < CLID_H * CLID_D; i += blockDim.x) s.w[i] = W1[i];
  for (int i = threadIdx.x; i < CLID_H; i += blockDim.x) {
    s.w[CLID_H * CLID_D + i] = b1[i];
    s.w[CLID_H * CLID_D + CLID_H + i] = W2[i];
  }
  if (threadIdx.x == 0) s.w[CLID_MLP_PARAMS - 1] = b2[0];
  __syncthreads();
}
// weights + probe deltas with ONE barrier
__device__ __forceinline__ void stage_mlp_and_delta(MlpLds& s, DeltaLds& dl, const clid_map_view& mv,
                                                    const float* W1, const float* b1, const float* W2,
                                                    const float* b2) {
  stage_delta(dl, mv);
  stage_mlp(s, W1, b1, W2, b2);  // ends with __syncthreads()
}
// An index the optimiser cannot see through: keeps the decoder weights IN LDS.  Without it LICM hoists
// all 116 weight reads out of the persistent task loop into VGPRs (201 VGPRs, 2 waves/SIMD).
__device__ __forceinline__ int opaque_zero() {
  int z = 0;
  asm volatile("" : "+v"(z));
  return z;
}
// lane16 owns hidden units h = lane16 + 16*u, u = 0..3
__device__ __forceinline__ float mlp_forward(const MlpLds& s, const float (&f)[CLID_D], int lane16,
                                             float scale, float (&pre)[CLID_HPL]) {
  float part = 0.f;
  lane16 += opaque_zero();
#pragma unroll
  for (int u = 0; u < CLID_HPL; ++u) {
    const int h = lane16 + CLID_G * u;
    float a = s.w[CLID_H * CLID_D + h];
#pragma unroll
    for (int c = 0; c < CLID_D; ++c) a = fmaf(s.w[h * CLID_D + c], f[c], a);
    pre[u] = a;
    part = fmaf(s.w[CLID_H * CLID_D + CLID_H + h], fmaxf(a, 0.f), part);
  }
  const float tot = group_sum(part);
  return scale * (tot + s.w[CLID_MLP_PARAMS - 1]);
}

}  // namespace clid

// ---- host-side error plumbing (api.hip) -----------------------------------------------------------
extern "C" void clid_set_error(const char* fmt, ...);
#define CLID_CHECK_LAUNCH()                                              \
  do {                                                                   \
    hipError_t e__ = hipGetLastError();                                  \
    if (e__ != hipSuccess) {                                             \
      clid_set_error("%s:%d HIP error: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return CLID_E_HIP;                                                 \
    }                                                                    \
  } while (0)
