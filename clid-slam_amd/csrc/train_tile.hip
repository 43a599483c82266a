// Decode / backward of one mapping iteration on the matrix cores: one wave per TILE of 16 query points.
//
// Replaces, for the numerical-eikonal / no-eikonal modes, the same reference code as k_train_fused8<2>
// (csrc/train.hip): label/weight gathers of Mapper.get_batch (utils/mapper.py:501-507), the blend of
// NeuralPoints.query_feature (model/neural_points.py:618-754) from the hoisted search records, Decoder.sdf
// (model/decoder.py:58-82), the finite-difference eikonal term (mapper.py:697-704, 985-1034), BCE + eikonal
// loss (utils/loss.py:44-62, mapper.py:746-798) and the backward of all of it.
//
// Geometry.  lane = (q = lane & 15, g = lane >> 4): 16 query slots x 4 lanes.  A tile is two consecutive wave
// tasks of the search kernel (8 query slots each, records of kRecFloat4 float4), so the six shifted copies of a
// decimated sample and the sample itself sit in lanes q0 .. q0+6 of one DPP row.
//   gather   lane g = 0 / 1 loads feature columns 0-3 / 4-7 of the query's 6 neighbours (one 16-byte load each),
//            g = 2 their positions, g = 3 the update stamps; every lane blends ITS 4 columns of the decoder input
//            f = [sum_k w_k feat_k | sum_k w_k (x - p_k) | 1] in registers -- no cross-lane traffic at all.
//   layer 1  pre[h][q] = sum_c W1e[h][c] f[c][q] as D = A.B on v_mfma_f32_16x16x4_f32: A[i = h][k] = W1e (the
//            bias is column 11, f[11] = 1), B[k = g][j = q] = the lane's own blended column 4g + s for K-step s.
//            The accumulator D_u[r] of lane (q, g) is hidden unit 16u + 4g + r of query q.
//   layer 2  16 in-lane FMAs + two permlane swaps (lane ^ 16, lane ^ 32).
//   backward d f[q][c] = sum_h dh[q][h] W1[h][c] with the operands SWAPPED (A = dh straight from the accumulator
//            registers, B = W1 rows): the result has the feature column on lane & 15 and the query on
//            (lane >> 4, register).
//   scatter  The fp32 atomic units at the memory side retire ~17 G REQUESTS/s whatever their width up to 64 bytes
//            (tools/ubench_atomic.hip) and this launch is bound by them, so the tile first merges its 96 (query,
//            neighbour) pairs per map row: the distinct rows are numbered through a small LDS hash (integer CAS),
//            the pair weights form a matrix Wm[row][query], and G[row][c] = sum_q Wm[row][q] d f[q][c] is one more
//            MFMA with B = the d f registers as they are (column 8 of B is 1, so column 8 of G is the row's
//            certainty increment, np.py:714).  9 consecutive lanes then add one row = ONE request per distinct row
//            of the tile (the 7 queries around a decimated sample share nearly all their neighbours).  With layer
//            norm the row's backward operator is linear and depends on the row only, so it is applied once to the
//            merged gradient.
//   dW1      = sum_q dh_q (x) f_q contracts over the queries, which sit on lane & 15: dh and f take one trip
//            through LDS to put them on the K axis (skipped when the decoder is frozen).
// PREC = 1 runs the three contractions on v_mfma_f32_16x16x32_bf16 (bf16 operands, fp32 accumulation,
// BASELINE.json configs[2]); everything outside the MFMAs is unchanged fp32.
#include <type_traits>

#include "train_common.hpp"

namespace clid {

// waves (= tiles in flight) per block: 4 while one round of blocks covers the batch (the reference's 16 384 samples: one
// partial row per 4 tiles keeps k_adam_all's column sums short), 2 beyond that (finer tail; 45.1 -> 41.1 us at 65 536)
#ifndef CLID_TILE_WAVES_SMALL
#define CLID_TILE_WAVES_SMALL 4
#endif
constexpr int kTileWavesSmall = CLID_TILE_WAVES_SMALL, kTileWavesLarge = 2;
#ifndef CLID_TILE_EARLY_REC
#define CLID_TILE_EARLY_REC 1  // the one-tile-per-wave fp32 kernels request their record before the weight staging (0: A/B)
#endif
#ifndef CLID_TILE_PF
#define CLID_TILE_PF 1  // the multi-tile launches fetch the NEXT tile's search record straight into LDS while the current tile runs (0: A/B)
#endif
#ifndef CLID_TILE_BLK
#define CLID_TILE_BLK 1  // block-level dW1 flush of the one-tile-per-wave launches (0: the per-wave form, A/B)
#endif
__host__ inline int tile_waves_for(int n_tiles) { return n_tiles > kTileLargeFrom ? kTileWavesLarge : kTileWavesSmall; }
constexpr int kRecF4 = 48;       // float4 per search record (== kRecFloat4 of train.hip)
constexpr int kDhStride = 84;    // floats per query row of the dh transposition buffer (conflict-free b128 stores)
constexpr int kFStride = 20;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kHash = 128;      // LDS hash slots for the <= 96 distinct map rows of a tile
constexpr int kMaxRows = 96;
struct alignas(16) TileLds {
  union {                      // the numbering's hash is dead before dh is staged (both per wave; a fence in between)
    float dh[16 * kDhStride];  // [q][h]
    struct {
      int hkey[kHash];         // (in-kernel numbering only) hash slot -> map row id, -1 empty
      int hrow[kHash];         // (in-kernel numbering only) hash slot -> row number inside the tile
    };
  };
  float f[16 * kFStride];    // [q][c], c = 0..15 (11 = the bias input 1, 12..15 = 0)
  float wm[kMaxRows * 16];   // [row][q]: weight of query q on the tile's distinct map row `row`
  int rowid[kMaxRows];       // row number -> map row id
  int count;                 // (in-kernel numbering only) distinct rows of the tile
  int pad_[3];
};
// The <= 168-register instantiation (WPS == 3) keeps the decoder's MFMA operands in LDS instead of 48 registers per lane, which
// costs 5.4 KB per block; its tile buffers give that back: the 16 x 16 staging rows of f live in the padding of dh's rows
// (columns 64..79 of the 84-float rows), so a tile wave needs 11 920 bytes and a 4-wave block 47 680 + 5 392 = 53 072 (three per CU).
struct alignas(16) TileLdsC {
  union {
    float dh[16 * kDhStride];  // [q][0..63] = dh, [q][64..79] = f (c = 0..15)
    struct {
      int hkey[kHash];
      int hrow[kHash];
    };
  };
  float wm[kMaxRows * 16];
  int rowid[kMaxRows];
  int count;
  int pad_[3];
};
constexpr int kWStride = 20;                       // floats per hidden unit of the LDS weight table: W1[h][0..10] | b1[h] | 0 0 0 0 | pad
constexpr int kWOffW2 = CLID_H * kWStride;         // then W2[64], then b2
constexpr int kWFloats = kWOffW2 + CLID_H + 4;
template <class TL> struct tile_f;                 // where the f staging rows live
template <> struct tile_f<TileLds> {
  static __device__ __forceinline__ float* at(TileLds& tl, int row) { return &tl.f[row * kFStride]; }
};
template <> struct tile_f<TileLdsC> {
  static __device__ __forceinline__ float* at(TileLdsC& tl, int row) { return &tl.dh[row * kDhStride + 64]; }
};
// 13 200 bytes per wave: three 4-wave blocks (52 800 B) or six 2-wave blocks (26 400 B) fit a CU's 160 KB of LDS, i.e. LDS
// admits 3 waves per SIMD (with the hash in its own 1 KB the 4-wave block was 56 896 B: two per CU)
static_assert(sizeof(TileLds) * 12 <= 160 * 1024, "LDS plan: 12 tile waves per CU");

// layer-norm variants: what F.layer_norm's backward needs of every distinct row, saved by the forward pass
struct alignas(16) TileLnLds {
  float xh[kMaxRows * CLID_F];  // [row][c]: normalised features
  float rs[kMaxRows];           // [row]: 1 / sqrt(var + eps)
  int rown[16 * 8];             // [q][k]: row number of neighbour k of query q (-1: none)
};

__device__ __forceinline__ float xsum16(float v) {  // v[lane] + v[lane ^ 16]
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xsum32(float v) {  // v[lane] + v[lane ^ 32]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// two floats -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32); `lo` lands in bits 0..15
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{lo, hi}), bf16x2));
}
__device__ __forceinline__ float bf16_round(float x) { return __uint_as_float(pack_bf16(x, 0.f) << 16); }
__device__ __forceinline__ bf16x8 bf16_frag(unsigned a, unsigned b, unsigned c, unsigned d) {
  return __builtin_bit_cast(bf16x8, (u32x4{a, b, c, d}));
}

__device__ __forceinline__ void tile_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// WPS = waves per SIMD the instantiation is compiled for (launch bounds): 0 = the default -- layer norm pins 2 (VGPRs + AGPRs
// <= 256; it lands on 1 otherwise), the others ask for 1 and get 2 --, 3 = the large-launch instantiation at <= 168 registers
// (65 536 samples and up: many tiles per SIMD, the launch is bound by resident waves, not by one tile's chain)
#ifndef CLID_TILE_WAVES
#define CLID_TILE_WAVES (WPS ? WPS : ((LN || EM) ? 2 : 1))
#endif
// PRE: the tiles' row numbers come from the search launch's number blocks (small launches on maps within one L2: see
// clid_tiles_prenumbered); otherwise the kernel numbers in place
// EM: the reference's non-default loop branches -- config.ekional_add_to "surface" / "freespace" and config.main_loss_type
// "sdf_l1" / "sdf_l2" / "zhong" (their own instantiations: the default ones are register-tight, three more live values put the
// headline kernel at 226 + 32 > 256 registers = one wave per SIMD)
template <int PREC, bool LN, int TW, bool PRE, int WPS = 0, bool EM = false>
__global__ void __launch_bounds__(TW * 64, CLID_TILE_WAVES)
k_decode_tile(const float4* __restrict__ rec, const int* __restrict__ tnum, int n_tiles, float* __restrict__ partial,
              float* __restrict__ sdf_dbg, TaskMap tmap, clid_map_view mv, clid_train_args ta) {
  // (argument order: what the wave's first loads need -- the record, the number block -- leads the kernel-argument segment, ahead
  // of the two by-value structs: with -mllvm -amdgpu-kernarg-preload-count these arrive in SGPRs with the wave)
  constexpr bool LW = WPS == 3;  // decoder operands read from LDS per use (the <= 168-register instantiation)
  // BLK (launches of one tile per wave: the pre-numbered instantiation): dW1 / db1 are contracted once per BLOCK behind the
  // block's barrier -- wave w takes hidden units 16 w .. 16 w + 15 over the 64 queries whose dh / f rows the four waves staged
  // in LDS anyway -- and leave straight from the accumulator registers; only dW2 / db2 / the two loss sums (67 numbers per wave)
  // still cross the waves through LDS.  The per-wave form wrote 835 floats per wave to LDS and summed them behind a second barrier.
  constexpr bool BLK = CLID_TILE_BLK && PRE && TW == 4 && !LW && PREC == 0;
  using TL = typename std::conditional<LW, TileLdsC, TileLds>::type;
  __shared__ TL tls[TW];
  __shared__ TileLnLds lns[LN ? TW : 1];
  __shared__ float wq[LW ? kWFloats : 4];
  static_assert(sizeof(TL) % 16 == 0 && sizeof(TL) * TW >= TW * kRedFloats * sizeof(float), "LDS plan");
  static_assert(!LW || (sizeof(TileLdsC) * TW + kWFloats * 4) * (12 / TW) <= 160 * 1024, "LDS plan: 3 waves per SIMD");
  // PF (launches of several tiles per wave): the two task records of the wave's NEXT tile (96 float4) arrive by LDS-DMA
  // (global_load_lds: no registers -- the kernel has none to spare) while the current tile runs.  The request goes out in front
  // of the current tile's feature gathers, so the wait at their first use covers it: at the top of the next iteration the record
  // is simply there, and that iteration no longer opens with a wait that also drains the previous tile's gradient atomics.
  constexpr bool PF = CLID_TILE_PF && !PRE && !LW;
  __shared__ float4 recbuf[PF ? TW : 1][PF ? 2 * kRecF4 : 1];
  float* red = reinterpret_cast<float*>(tls);  // the block flush reuses the tile buffers (after a barrier)
  const int lane = threadIdx.x & 63, q = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
  TL& tl = tls[wave];
  TileLnLds& ln = lns[LN ? wave : 0];
  const bool train = ta.train_decoder != 0;
  const float sc = ta.sdf_scale;
  const float inv_sigma = fdiv(1.0f, ta.sigma);
  const float inv_two_eps = fdiv(1.0f, 2.0f * ta.fd_eps);
  // config.ekional_add_to (utils/mapper.py:779-789): "surface" / "freespace" restrict the eikonal mean to the decimated samples
  // with |label| below / not below the surface range; the subset's size was counted by the search launch
  const float inv_n_eik = (EM && ta.eik_mask) ? ta.eik_inv_n[ta.touch_iter] : ta.inv_n_eik;  // (uniform: a scalar load)
  float* __restrict__ rows = ta.grad + CLID_GRAD_FEAT_OFFSET16;
  const float4* __restrict__ feat4 = reinterpret_cast<const float4*>(mv.feat);
  const float4* __restrict__ pos4 = reinterpret_cast<const float4*>(mv.pos4);

  // the lane's slice of a tile's search record (+ its number block): 31 registers
  struct TileRec {
    float4 qi, qq, w01, w23, w45, wf;
    int4 rid4;
    int n_rows, rnum0, rnum1;
  };
  auto load_rec = [&](int tile) -> TileRec {
    TileRec rc;
    const int task = 2 * tile + (q >> 3), slot = q & 7;
    const float4* r = rec + (size_t)(task < tmap.n_tasks ? task : 0) * kRecF4;
    rc.qi = r[slot];
    rc.qq = r[8 + slot];
    rc.w01 = r[16 + slot * 4];
    rc.w23 = r[16 + slot * 4 + 1];
    rc.w45 = r[16 + slot * 4 + 2];
    rc.wf = r[16 + slot * 4 + 3];  // (fx, fy | fz, -): blended offset, decoder inputs 8..10
    rc.rid4 = make_int4(0, 0, 0, 0);
    rc.n_rows = 0;
    rc.rnum0 = rc.rnum1 = 255;
    if constexpr (PRE) {
      // PRE (launches of one tile per wave): the tile's pairs were numbered per distinct map row by the search launch
      // (k_search_tiles, train.hip) -- the number block sits behind the iteration's task records
      const int* __restrict__ tn = tnum + (size_t)tile * kTileNumWords;
      if (lane < kMaxRows / 4) rc.rid4 = *reinterpret_cast<const int4*>(tn + 4 * lane);
      rc.n_rows = tn[kTileNumCount];
      const unsigned char* __restrict__ rbytes = reinterpret_cast<const unsigned char*>(tn + kTileNumBytes);
      rc.rnum0 = rbytes[q * CLID_K + g];
      if (g < 2) rc.rnum1 = rbytes[q * CLID_K + g + 4];
    }
    return rc;
  };
  auto prefetch_rec = [&](int tile) {  // (PF) lane l brings float4 l and, l < 32, float4 64 + l of the tile's 96
    const size_t f0 = (size_t)2 * tile * kRecF4, last = (size_t)tmap.n_tasks * kRecF4 - 1;  // (an odd task count: the tile's second half is clamped)
    const size_t a = f0 + lane < last ? f0 + lane : last, b = f0 + 64 + lane < last ? f0 + 64 + lane : last;
    __builtin_amdgcn_global_load_lds(rec + a, (__attribute__((address_space(3))) void*)&recbuf[PF ? wave : 0][0], 16, 0, 0);
    if (lane < 2 * kRecF4 - 64)
      __builtin_amdgcn_global_load_lds(rec + b, (__attribute__((address_space(3))) void*)&recbuf[PF ? wave : 0][PF ? 64 : 0], 16, 0, 0);
  };
  auto read_rec = [&]() -> TileRec {  // (PF) the lane's slice of the record in LDS
    TileRec rc;
    const float4* r = &recbuf[PF ? wave : 0][PF ? (q >> 3) * kRecF4 : 0];
    const int slot = q & 7;
    rc.qi = r[slot];
    rc.qq = r[8 + slot];
    rc.w01 = r[16 + slot * 4];
    rc.w23 = r[16 + slot * 4 + 1];
    rc.w45 = r[16 + slot * 4 + 2];
    rc.wf = r[16 + slot * 4 + 3];
    rc.rid4 = make_int4(0, 0, 0, 0);
    rc.n_rows = 0;
    rc.rnum0 = rc.rnum1 = 255;
    return rc;
  };
  // EARLY (one tile per wave, fp32): the wave's record is requested BEFORE the decoder weights are staged -- its ~1 us of latency
  // runs under the staging's two barriers (decode 11.93 -> 11.65 us, three alternating runs).  (Round 4 measured this on the
  // 224 + 32-register kernel: 31 more live registers put it at one wave per SIMD; the block-level dW1 flush freed them.)
  // Measured and NOT kept on top of it (profiles/r05_decode_chain_experiments.jsonl): the block flush with all 32 operands requested
  // from LDS in front of four independent product chains, and the merge's 16-row product chains issued in pairs -- together
  // 11.65 -> 12.0 us: the compiler's own interleaving of LDS reads and products was the better schedule.
  constexpr bool EARLY = CLID_TILE_EARLY_REC && BLK;
  TileRec early;
  if constexpr (EARLY) {
    const int t0 = blockIdx.x * TW + wave;
    early = load_rec(t0 < n_tiles ? t0 : 0);
  }
  if constexpr (PF) {
    const int t0 = blockIdx.x * TW + wave;
    if (t0 < n_tiles) prefetch_rec(t0);  // (lands under the weight staging's barriers)
  }

  // ---- constant MFMA operands: the decoder weights (3.3 KB) are staged through LDS with one coalesced load per
  // thread (48 strided global loads per lane cost ~1.5 us of address-unit time at the start of every wave)
  //   A1[u][s] = W1e[16u + q][4g + s]     (W1e = [W1 | b1 | 0 0 0 0])            layer 1, A[i = lane & 15][k = lane >> 4]
  //   W2r[u][r] = W2[16u + 4g + r]                                                 layer 2 / dh, accumulator layout
  //   A2[u][r] = W1[16u + 4g + r][q], q < 8                                        d f, B[k = lane >> 4][j = lane & 15]
  float A1[4][4], W2r[4][4], A2[4][4];
  float b2;
  if constexpr (LW) {
    // the operand table stays in LDS for the whole launch: W1e rows padded to 20 floats (conflict-free 16-byte reads of
    // A1 = W1e[16u + q][4g ..], column reads of A2 = W1[16u + 4g + r][q]), bf16 variants pre-rounded like the register path
    for (int i = threadIdx.x; i < CLID_H * 16; i += (TW * 64)) {
      const int h = i >> 4, c = i & 15;
      float a = c < CLID_D ? ta.W1[h * CLID_D + c] : (c == CLID_D ? ta.b1[h] : 0.f);
      if (PREC == 1 && c != CLID_D) a = bf16_round(a);
      wq[h * kWStride + c] = a;
    }
    if (threadIdx.x < CLID_H) wq[kWOffW2 + threadIdx.x] = ta.W2[threadIdx.x];
    if (threadIdx.x == 0) wq[kWOffW2 + CLID_H] = ta.b2[0];
    __syncthreads();
    b2 = wq[kWOffW2 + CLID_H];
  } else {
  {
    float* wl = reinterpret_cast<float*>(tls);  // [W1 704 | b1 64 | W2 64 | b2 1]; overwritten by the first tile's fences later
    for (int i = threadIdx.x; i < CLID_H * CLID_D; i += (TW * 64)) wl[i] = ta.W1[i];
    if (threadIdx.x < CLID_H) {
      wl[CLID_H * CLID_D + threadIdx.x] = ta.b1[threadIdx.x];
      wl[CLID_H * CLID_D + CLID_H + threadIdx.x] = ta.W2[threadIdx.x];
    }
    if (threadIdx.x == 0) wl[CLID_MLP_PARAMS - 1] = ta.b2[0];
  }
  __syncthreads();
  {
    const float* wl = reinterpret_cast<const float*>(tls);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int h = 16 * u + q, c = 4 * g + s;
        float a = 0.f;
        if (c < CLID_D) a = wl[h * CLID_D + c];
        else if (c == CLID_D) a = wl[CLID_H * CLID_D + h];
        const int h2 = 16 * u + 4 * g + s;
        const float w2 = wl[CLID_H * CLID_D + CLID_H + h2];
        float a2 = q < CLID_F ? wl[h2 * CLID_D + q] : 0.f;
        if (PREC == 1) {  // bf16 operands; the bias column stays exact (it is added as hi + lo, see below)
          if (c != CLID_D) a = bf16_round(a);
          a2 = bf16_round(a2);
        }
        A1[u][s] = a;
        W2r[u][s] = w2;
        A2[u][s] = a2;
      }
    b2 = wl[CLID_MLP_PARAMS - 1];
  }
  __syncthreads();  // the tile buffers may be written from here on
  }
  // operand access: registers, or (LW) the LDS table
  auto ldA1 = [&](int u) -> float4 {
    if constexpr (LW) return *reinterpret_cast<const float4*>(&wq[(16 * u + q) * kWStride + 4 * g]);
    else return make_float4(A1[u][0], A1[u][1], A1[u][2], A1[u][3]);
  };
  auto ldW2 = [&](int u) -> float4 {
    if constexpr (LW) return *reinterpret_cast<const float4*>(&wq[kWOffW2 + 16 * u + 4 * g]);
    else return make_float4(W2r[u][0], W2r[u][1], W2r[u][2], W2r[u][3]);
  };
  auto ldA2 = [&](int u, int rr) -> float {
    if constexpr (LW) return q < CLID_F ? wq[(16 * u + 4 * g + rr) * kWStride + q] : 0.f;
    else return A2[u][rr];
  };
  CLID_STAMP(0);
  if constexpr (!LW) asm volatile("" ::"v"(A1[3][3]), "v"(W2r[3][3]), "v"(A2[3][3]));
  CLID_STAMP(1);

  f32x4 dW1a[4];
  float dW2a[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    dW1a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) dW2a[u][r] = 0.f;
  }
  float db2a = 0.f, bce_acc = 0.f, eik_acc = 0.f;

  for (int tile = blockIdx.x * TW + wave; tile < n_tiles; tile += gridDim.x * TW) {
    // ================= record of this lane's query slot
    const int task = 2 * tile + (q >> 3), slot = q & 7;
    const bool tlive = task < tmap.n_tasks;
    TileRec rc;
    if constexpr (EARLY) rc = early;
    else if constexpr (PF) {
      if (tile == (int)(blockIdx.x * TW + wave))  // the wave's first tile: its request is the only memory operation in flight;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every later record landed under its predecessor's gather wait
      rc = read_rec();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the record is in registers: its LDS slot may be overwritten
      const int nxt = tile + gridDim.x * TW;
      if (nxt < n_tiles) prefetch_rec(nxt);
    } else rc = load_rec(tile);
    const float4 qi = rc.qi, qq = rc.qq, w01 = rc.w01, w23 = rc.w23, w45 = rc.w45, wf = rc.wf;
    const int sidx = tlive ? __float_as_int(qi.w) : -1;  // time stamp of the sample, -1 = padding slot
    const bool bundle = tlive && task < tmap.n_fd;
    const int4 rid4 = rc.rid4;
    int n_rows = rc.n_rows;
    const int rnum[2] = {rc.rnum0, rc.rnum1};
    // IDW weights and neighbour ids come from the search record (np.py:688-706)
    float w[CLID_K] = {w01.x, w01.z, w23.x, w23.z, w45.x, w45.z};
    int j[CLID_K] = {__float_as_int(w01.y), __float_as_int(w01.w), __float_as_int(w23.y),
                     __float_as_int(w23.w), __float_as_int(w45.y), __float_as_int(w45.w)};
#pragma unroll
    for (int k = 0; k < CLID_K; ++k) {
      if (sidx < 0) j[k] = -1;
      if (j[k] < 0) w[k] = 0.f;
    }
    CLID_STAMP(2);
    // ================= gather of this lane's 4 decoder-input columns: the loads are issued first ...
    float4 v[CLID_K];
    int ts_old[CLID_K];
#pragma unroll
    for (int k = 0; k < CLID_K; ++k) {
      const int jc = j[k] >= 0 ? j[k] : 0;
      v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      ts_old[k] = 0x7fffffff;
      if (g < 2) v[k] = feat4[(size_t)jc * 2 + g];
      else if (g == 3 && mv.ts_update && j[k] >= 0) ts_old[k] = mv.ts_update[jc];
    }
    // ================= ... and in their shadow Wm[row][query] is filled.  The pairs' row numbers depend on the records only,
    // never on the training state: small launches (one tile per wave: the launch is one tile's dependent chain long) read
    // them from the number block the search launch wrote; large launches number in place through the LDS hash -- there the
    // other waves of the SIMD hide it, while numbering every tile in the search costs that launch more than this one gains
    // (65 536 samples: search +6 us, decode -1 us per iteration).
    if constexpr (PRE) {
#pragma unroll
      for (int i = 0; i < kMaxRows * 16 / (64 * 4); ++i)
        *reinterpret_cast<float4*>(&tl.wm[(i * 64 + lane) * 4]) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane < kMaxRows / 4) *reinterpret_cast<int4*>(&tl.rowid[4 * lane]) = rid4;
      tile_lds_fence();
      // lane (q, g) places neighbours k = g and (g < 2) k = g + 4 of its query.  A query's own list may name a row twice
      // (two colliding cells returning the same point): the first occurrence carries the sum of the weights
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int k = g + 4 * t;
        int jk = -1;
#pragma unroll
        for (int kk = 0; kk < CLID_K; ++kk) jk = (kk == k) ? j[kk] : jk;
        bool fst = true;
        float tot = 0.f;
#pragma unroll
        for (int kk = 0; kk < CLID_K; ++kk) {  // ascending, so the sum has the order of a sequential fold
          const bool same = j[kk] == jk;
          if (same && kk < k) fst = false;
          if (same) tot += w[kk];
        }
        const int row = (k < CLID_K && jk >= 0) ? rnum[t] : -1;
        if (row >= 0 && fst) tl.wm[row * 16 + q] = tot;
        if (LN && k < CLID_K) ln.rown[q * 8 + k] = row;
      }
      tile_lds_fence();
    } else {
      tl.hkey[lane] = -1;
      tl.hkey[lane + 64] = -1;
      if (lane == 0) tl.count = 0;
#pragma unroll
      for (int i = 0; i < kMaxRows * 16 / (64 * 4); ++i)
        *reinterpret_cast<float4*>(&tl.wm[(i * 64 + lane) * 4]) = make_float4(0.f, 0.f, 0.f, 0.f);
      tile_lds_fence();
      // lane (q, g) numbers neighbours k = g and (g < 2) k = g + 4 of its query through the LDS hash (integer ds_cmpst)
      int hs[2] = {-1, -1};
      float wsum[2] = {0.f, 0.f};
      bool first[2] = {false, false};
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int k = g + 4 * t;
        int jk = -1;
#pragma unroll
        for (int kk = 0; kk < CLID_K; ++kk) jk = (kk == k) ? j[kk] : jk;
        bool fst = true;
        float tot = 0.f;
#pragma unroll
        for (int kk = 0; kk < CLID_K; ++kk) {  // ascending, so the sum has the order of a sequential fold
          const bool same = j[kk] == jk;
          if (same && kk < k) fst = false;
          if (same) tot += w[kk];
        }
        if (k < CLID_K && jk >= 0) {
          unsigned h = ((unsigned)jk * 2654435761u) >> 25;  // 7 bits
          for (;;) {
            const int old = atomicCAS(&tl.hkey[h], -1, jk);
            if (old == -1) {  // first pair of this row in the tile: take the next row number
              const int d = atomicAdd(&tl.count, 1);
              tl.hrow[h] = d;
              tl.rowid[d] = jk;
              break;
            }
            if (old == jk) break;
            h = (h + 1) & (kHash - 1);
          }
          hs[t] = (int)h;
          first[t] = fst;
          wsum[t] = tot;
        }
      }
      tile_lds_fence();
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int k = g + 4 * t;
        const int row = hs[t] >= 0 ? tl.hrow[hs[t]] : -1;
        if (first[t]) tl.wm[row * 16 + q] = wsum[t];
        if (LN && k < CLID_K) ln.rown[q * 8 + k] = row;
      }
      tile_lds_fence();
    }
    CLID_STAMP(8);
    if (LN) {  // F.layer_norm over the 8 features of every neighbour row (np.py:632-633); lanes g = 0,1 hold the halves
#pragma unroll
      for (int k = 0; k < CLID_K; ++k) {
        const float mu = xsum16((v[k].x + v[k].y) + (v[k].z + v[k].w)) * (1.0f / CLID_F);
        const float4 c = make_float4(v[k].x - mu, v[k].y - mu, v[k].z - mu, v[k].w - mu);
        const float var = xsum16((c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w)) * (1.0f / CLID_F);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        v[k] = make_float4(c.x * rstd, c.y * rstd, c.z * rstd, c.w * rstd);
        // the backward of this row (applied once to the merged gradient below) reads what autograd would have saved;
        // pairs that share the row store identical values
        const int row = ln.rown[q * 8 + k];
        if (g < 2 && row >= 0) {
          *reinterpret_cast<float4*>(&ln.xh[row * CLID_F + 4 * g]) = v[k];
          if (g == 0) ln.rs[row] = rstd;
        }
      }
      tile_lds_fence();
    }
    CLID_STAMP(3);
    float pc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CLID_K; ++k) {
      pc[0] = fmaf(v[k].x, w[k], pc[0]);
      pc[1] = fmaf(v[k].y, w[k], pc[1]);
      pc[2] = fmaf(v[k].z, w[k], pc[2]);
      pc[3] = fmaf(v[k].w, w[k], pc[3]);
    }
    if (g == 2) {  // x - neighbour position, blended by the search (np.py:653-674), and the bias input
      pc[0] = wf.x; pc[1] = wf.y; pc[2] = wf.z; pc[3] = 1.0f;
    } else if (g == 3) {
      pc[0] = pc[1] = pc[2] = pc[3] = 0.f;
    }
    CLID_STAMP(4);
    // ================= layer 1 on the matrix cores
    f32x4 D[4];
    if (PREC == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        D[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float4 a = ldA1(u);
        D[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, pc[0], D[u], 0, 0, 0);
        D[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, pc[1], D[u], 0, 0, 0);
        D[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, pc[2], D[u], 0, 0, 0);
        D[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, pc[3], D[u], 0, 0, 0);
      }
    } else {
      // K = 32: lane (q, g) owns K slots 8g .. 8g+7 = its 4 columns (bf16) | 4 spare slots; the spare slots of the
      // g = 2 lanes carry the bias a second time: b1 = hi + lo with the input 1.0 in slots 3 and 4
      const bf16x8 bq = bf16_frag(pack_bf16(pc[0], pc[1]), pack_bf16(pc[2], pc[3]), g == 2 ? pack_bf16(1.0f, 0.f) : 0u, 0u);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 a = ldA1(u);
        const float blo = a.w - bf16_round(a.w);  // g == 2: the bias' low part
        const bf16x8 aq = bf16_frag(pack_bf16(a.x, a.y), pack_bf16(a.z, a.w),
                                    g == 2 ? pack_bf16(blo, 0.f) : 0u, 0u);
        D[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq, bq, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      }
    }
    // ================= layer 2 (decoder.py:76-82)
    float part = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 w2 = ldW2(u);
      part = fmaf(w2.x, fmaxf(D[u][0], 0.f), part);
      part = fmaf(w2.y, fmaxf(D[u][1], 0.f), part);
      part = fmaf(w2.z, fmaxf(D[u][2], 0.f), part);
      part = fmaf(w2.w, fmaxf(D[u][3], 0.f), part);
    }
    const float sdf = sc * (xsum32(xsum16(part)) + b2);
    if (sdf_dbg && g == 0 && tlive) sdf_dbg[(size_t)task * 8 + slot] = sdf;  // tests: SDF per record slot
    CLID_STAMP(5);
    // ================= losses
    const int p = tlive ? __float_as_int(qq.x) : -1;
    const int code = __float_as_int(qq.y);  // -1 = a batch sample, else 2*axis + (sign > 0)
    float delta = 0.f;
    {
      const int b8 = lane & ~7;
      const float s0 = __shfl(sdf, b8 + 0, 64), s1 = __shfl(sdf, b8 + 1, 64), s2 = __shfl(sdf, b8 + 2, 64);
      const float s3 = __shfl(sdf, b8 + 3, 64), s4 = __shfl(sdf, b8 + 4, 64), s5 = __shfl(sdf, b8 + 5, 64);
      float gx = 0.f, gy = 0.f, gz = 0.f, ecoef = 0.f;
      if (bundle) {
        gx = (s0 - s1) * inv_two_eps;  // mapper.py:1011-1013
        gy = (s2 - s3) * inv_two_eps;
        gz = (s4 - s5) * inv_two_eps;
        const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
        bool inmask = true;
        if constexpr (EM) {  // the decimated sample (slot 6 of the bundle) decides for its six copies
          const float lab6 = __shfl(qq.z, b8 + 6, 64);
          if (ta.eik_mask) inmask = (fabsf(lab6) < ta.eik_mask_range) == (ta.eik_mask == 1);
        }
        if (slot == 6 && g == 0 && inmask) eik_acc += (nrm - 1.f) * (nrm - 1.f);
        ecoef = (nrm > 0.f && inmask) ? ta.weight_e * 2.f * (nrm - 1.f) * inv_n_eik * inv_two_eps / nrm : 0.f;
      }
      if (p >= 0) {
        if (code < 0) {
          const float label = qq.z, wt = qq.w;
          if (!EM || ta.main_loss_type == 0) {
            const float z = sdf * inv_sigma;
            const float tgt = __frcp_rn(1.0f + __expf(-label * inv_sigma));  // loss.py:60
            const float ez = __expf(-fabsf(z));
            const float sg = (z >= 0.f ? 1.0f : ez) * __frcp_rn(1.0f + ez);
            const float li = fmaxf(z, 0.f) - z * tgt + __logf(1.0f + ez);    // BCEWithLogits
            if (g == 0) bce_acc += wt * li;
            delta = wt * (sg - tgt) * inv_sigma * ta.inv_n_main;
          } else if (ta.main_loss_type == 3) {
            // sdf_zhong_loss (loss.py:66-84, trunc_dist None): |pred - label / 2| - |label / 2| where positive, else 0
            const float mid = label * 0.5f, sh = sdf - mid;
            const bool on = fabsf(sh) > fabsf(mid);
            if (g == 0 && on) bce_acc += wt * (fabsf(sh) - fabsf(mid));
            delta = on ? wt * (sh > 0.f ? 1.0f : (sh < 0.f ? -1.0f : 0.f)) * ta.inv_n_main : 0.f;
          } else {
            // sdf_diff_loss (loss.py:9-17, scale 1): weight * diff^2 ("sdf_l2") or weight * |diff| ("sdf_l1"), summed / count
            const float diff = sdf - label;
            const bool l2 = ta.main_loss_type == 2;
            if (g == 0) bce_acc += wt * (l2 ? diff * diff : fabsf(diff));
            delta = wt * (l2 ? 2.0f * diff : (diff > 0.f ? 1.0f : (diff < 0.f ? -1.0f : 0.f))) * ta.inv_n_main;
          }
        } else {
          const int axis = code >> 1;
          const float ga = axis == 0 ? gx : (axis == 1 ? gy : gz);
          delta = ((code & 1) ? 1.0f : -1.0f) * ecoef * ga;
        }
      }
    }
    CLID_STAMP(6);
    // ================= backward through the decoder
    const float dz = sc * delta;
    float dh[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 w2v = ldW2(u);
      const float w2[4] = {w2v.x, w2v.y, w2v.z, w2v.w};
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const bool on = D[u][rr] > 0.f;
        dh[u][rr] = on ? dz * w2[rr] : 0.f;
        if (train) dW2a[u][rr] += on ? dz * D[u][rr] : 0.f;
      }
    }
    if (train && g == 0) db2a += dz;
    // d f, operands swapped: Df[rr] of lane (c = lane & 15, G = lane >> 4) = d f[c] of query slot 4G + rr
    f32x4 Df = {0.f, 0.f, 0.f, 0.f};
    if (PREC == 0) {
      f32x4 Df2 = {0.f, 0.f, 0.f, 0.f};  // two independent chains (40-cycle dependent latency vs 32-cycle issue)
#pragma unroll
      for (int u = 0; u < 4; u += 2)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          Df = __builtin_amdgcn_mfma_f32_16x16x4f32(dh[u][rr], ldA2(u, rr), Df, 0, 0, 0);
          Df2 = __builtin_amdgcn_mfma_f32_16x16x4f32(dh[u + 1][rr], ldA2(u + 1, rr), Df2, 0, 0, 0);
        }
      Df += Df2;
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {  // K slot 8g + e <-> hidden 16(2t + e/4) + 4g + e%4
        const bf16x8 aq = bf16_frag(pack_bf16(dh[2 * t][0], dh[2 * t][1]), pack_bf16(dh[2 * t][2], dh[2 * t][3]),
                                    pack_bf16(dh[2 * t + 1][0], dh[2 * t + 1][1]), pack_bf16(dh[2 * t + 1][2], dh[2 * t + 1][3]));
        const bf16x8 bq = bf16_frag(pack_bf16(ldA2(2 * t, 0), ldA2(2 * t, 1)), pack_bf16(ldA2(2 * t, 2), ldA2(2 * t, 3)),
                                    pack_bf16(ldA2(2 * t + 1, 0), ldA2(2 * t + 1, 1)), pack_bf16(ldA2(2 * t + 1, 2), ldA2(2 * t + 1, 3)));
        Df = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq, bq, Df, 0, 0, 0);
      }
    }
    CLID_STAMP(7);
    // ================= scatter: the tile's pairs were numbered per map row above; G = Wm x d f, one request per row
    if (train) {  // stage dh and f for the dW1 contraction
#pragma unroll
      for (int u = 0; u < 4; ++u)
        *reinterpret_cast<float4*>(&tl.dh[q * kDhStride + 16 * u + 4 * g]) = make_float4(dh[u][0], dh[u][1], dh[u][2], dh[u][3]);
      *reinterpret_cast<float4*>(tile_f<TL>::at(tl, q) + 4 * g) = make_float4(pc[0], pc[1], pc[2], pc[3]);
      tile_lds_fence();
    }
    {
      const bool do_cert = !(ta.debug_flags & 1), do_grad = !(ta.debug_flags & 2);
      const bool act = q < CLID_F ? do_grad : (q == CLID_F && do_cert);
      if constexpr (!PRE) n_rows = tl.count;
      // B[k = G][j = c] of K-step rr = d f[c] of query 4G + rr (the registers as they are); column 8 = 1
      float Bx[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) Bx[rr] = q < CLID_F ? Df[rr] : (q == CLID_F ? 1.0f : 0.f);
      for (int t0 = 0; t0 < n_rows; t0 += 16) {
        const float4 a = *reinterpret_cast<const float4*>(&tl.wm[(t0 + q) * 16 + 4 * g]);  // Wm[row t0 + q][4G .. 4G+3]
        f32x4 G = {0.f, 0.f, 0.f, 0.f};
        G = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, Bx[0], G, 0, 0, 0);
        G = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, Bx[1], G, 0, 0, 0);
        G = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, Bx[2], G, 0, 0, 0);
        G = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, Bx[3], G, 0, 0, 0);
        const int4 ids = *reinterpret_cast<const int4*>(&tl.rowid[t0 + 4 * g]);
        const int id4[4] = {ids.x, ids.y, ids.z, ids.w};
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {  // G[rr] of lane (c, G) = merged gradient column c of row t0 + 4G + rr
          const bool live_row = t0 + 4 * g + rr < n_rows;
          float val = G[rr];
          if (LN) {  // layer-norm backward of the row, once (linear in the incoming gradient; np.py:632-633)
            const int rl = live_row ? t0 + 4 * g + rr : 0;
            const float xh = q < CLID_F ? ln.xh[rl * CLID_F + q] : 0.f;
            const float rstd = ln.rs[rl];
            float gsum = q < CLID_F ? val : 0.f;
            float gx = gsum * xh;
            gsum += dpp_mov<0xB1>(gsum); gsum += dpp_mov<0x4E>(gsum); gsum += dpp_mov<0x141>(gsum);
            gx += dpp_mov<0xB1>(gx); gx += dpp_mov<0x4E>(gx); gx += dpp_mov<0x141>(gx);
            if (q < CLID_F) val = rstd * (val - gsum * (1.0f / CLID_F) - xh * gx * (1.0f / CLID_F));
          }
          if (act && live_row) atomicAdd(&rows[(size_t)id4[rr] * CLID_GRAD_ROW16 + q], val);
        }
      }
    }
    CLID_STAMP(9);
    // time stamps (np.py:719-728): amax is idempotent, only a newer stamp needs the atomic
    if (g == 3 && code < 0 && p >= 0 && mv.ts_update && !(ta.debug_flags & 1)) {
#pragma unroll
      for (int k = 0; k < CLID_K; ++k)
        if (j[k] >= 0 && ts_old[k] < sidx) atomicMax(&mv.ts_update[j[k]], sidx);
    }
    CLID_STAMP(10);
    // ================= dW1 (+ db1 through the bias column) on the matrix cores
    if (train && !BLK) {
      if (PREC == 0) {
        float bt[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) bt[s] = tile_f<TL>::at(tl, 4 * s + g)[q];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            dW1a[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(tl.dh[(4 * s + g) * kDhStride + 16 * u + q], bt[s], dW1a[u], 0, 0, 0);
      } else {  // K slot 8g + e <-> query 8g + e (g < 2), zero above
        float fq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) fq[e] = g < 2 ? tile_f<TL>::at(tl, 8 * g + e)[q] : 0.f;
        const bf16x8 bq = bf16_frag(pack_bf16(fq[0], fq[1]), pack_bf16(fq[2], fq[3]), pack_bf16(fq[4], fq[5]), pack_bf16(fq[6], fq[7]));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float dq[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) dq[e] = g < 2 ? tl.dh[(8 * g + e) * kDhStride + 16 * u + q] : 0.f;
          const bf16x8 aq = bf16_frag(pack_bf16(dq[0], dq[1]), pack_bf16(dq[2], dq[3]), pack_bf16(dq[4], dq[5]), pack_bf16(dq[6], dq[7]));
          dW1a[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq, bq, dW1a[u], 0, 0, 0);
        }
      }
    }
    tile_lds_fence();
    CLID_STAMP(11);
  }

  CLID_STAMP(12);
  float* out = partial + (size_t)blockIdx.x * kPartialStride;
  // Sharded runs with the dense exchange (clid_train_args.dec_copies, ABI 8): the block's sums -- 833 decoder gradients and the
  // two raw loss sums -- are ADDED to one of the copies inside the buffer the all-reduce reads, instead of stored to a partial row
  // that a reduction launch between decode and all-reduce would add up: one dependent launch less per iteration.  Same-line
  // atomics serialise (~170 ns each): hence the copies, and hence no adds to loss_out from here (ONE line for every block).
  const bool direct = ta.dec_copies != nullptr && !ta.defer_reduce;
  float* copy = direct ? ta.dec_copies + (size_t)(blockIdx.x % (unsigned)ta.n_dec_copies) * kPartialStride : nullptr;
  auto flush = [&](int i, float v) {
    if (!direct) out[i] = v;
    else atomicAdd(copy + i, v);
  };
  if constexpr (BLK) {
    __shared__ float aux[TW][72];  // per wave: dW2 [64] | db2 | bce sum | eikonal sum
    const float bce_w = wave_sum(bce_acc), eik_w = wave_sum(eik_acc);
    if (train) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float s2 = group_sum(dW2a[u][rr]);
          if (q == 0) aux[wave][16 * u + 4 * g + rr] = s2;
        }
      const float db2_w = wave_sum(db2a);
      if (lane == 0) aux[wave][CLID_H] = db2_w;
    }
    if (lane == 0) {
      aux[wave][CLID_H + 1] = bce_w;
      aux[wave][CLID_H + 2] = eik_w;
    }
    __syncthreads();  // every wave's tile is done: its dh / f rows are complete
    if (train) {
      // dW1[16 wave + i][c] = sum over the block's (up to) 64 queries of dh[query][16 wave + i] f[query][c]: A[i = q][k] = dh of
      // query k, B[k][j = q] = f of query k; lane (q, g) ends with rows 16 wave + 4 g + r of column q (column 11 = db1)
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sw = 0; sw < TW; ++sw) {
        if (blockIdx.x * TW + sw >= n_tiles) break;  // (uniform: the block's trailing waves had no tile, their buffers are stale)
        TL& ts = tls[sw];
#pragma unroll
        for (int sq = 0; sq < 4; sq += 2) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ts.dh[(4 * sq + g) * kDhStride + 16 * wave + q], tile_f<TL>::at(ts, 4 * sq + g)[q], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ts.dh[(4 * sq + 4 + g) * kDhStride + 16 * wave + q], tile_f<TL>::at(ts, 4 * sq + 4 + g)[q], acc1, 0, 0, 0);
        }
      }
      acc0 += acc1;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int h = 16 * wave + 4 * g + rr;
        if (q < CLID_D) flush(h * CLID_D + q, acc0[rr]);
        else if (q == CLID_D) flush(CLID_H * CLID_D + h, acc0[rr]);
      }
    }
    if (threadIdx.x < CLID_H + 3 && (train || threadIdx.x > CLID_H)) {
      float sacc = 0.f;
#pragma unroll
      for (int wv = 0; wv < TW; ++wv) sacc += aux[wv][threadIdx.x];
      flush(CLID_H * CLID_D + CLID_H + threadIdx.x, sacc);  // W2 [64] | b2 | bce | eik: consecutive in the partial row
    }
  } else {
  __syncthreads();
  // ---- block flush: one partial row [833 decoder gradients | bce | eik] per block
  float* mine = red + wave * kRedFloats;
  const float bce_w = wave_sum(bce_acc), eik_w = wave_sum(eik_acc);
  if (train) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int h = 16 * u + 4 * g + rr;
        if (q < CLID_D) mine[h * CLID_D + q] = dW1a[u][rr];
        else if (q == CLID_D) mine[CLID_H * CLID_D + h] = dW1a[u][rr];
        const float s2 = group_sum(dW2a[u][rr]);
        if (q == 0) mine[CLID_H * CLID_D + CLID_H + h] = s2;
      }
    const float db2_w = wave_sum(db2a);
    if (lane == 0) mine[CLID_MLP_PARAMS - 1] = db2_w;
  }
  if (lane == 0) {
    mine[CLID_MLP_PARAMS] = bce_w;
    mine[CLID_MLP_PARAMS + 1] = eik_w;
  }
  __syncthreads();
  for (int i = train ? threadIdx.x : CLID_MLP_PARAMS + threadIdx.x; i < CLID_MLP_PARAMS + 2; i += (TW * 64)) {
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < TW; ++wv) s += red[wv * kRedFloats + i];
    flush(i, s);
  }
  }
  CLID_STAMP(13);
}

}  // namespace clid

using namespace clid;

#ifdef CLID_TIMING
extern "C" int clid_debug_read_stamps_tile(long long* out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(clid::clid_stamps), sizeof(long long) * 256 * 32) == hipSuccess ? 0 : -3;
}
#endif
// do the tiles of a launch over n_tasks tasks read their row numbers from the search launch's number blocks?
bool clid_tiles_prenumbered(int n_tasks, const clid_map_view* mv) {
  // (i) at most kTileLargeFrom tiles: the decode launch then runs one tile per wave and is one tile's dependent chain long
  // (14.5 -> 12.6 us at 16 384 samples for +1.0 us per iteration in the search launch); (ii) with the PROBING search, a local map
  // whose probe table still sits in one L2 (the launcher's LDS-prefilter regime, M <= 2^17): at M = 243 k that search paid 4.5 us
  // per iteration for the numbering and the decode gained 4.4.  The cell-directory search pays 1.9 us there and the decode gains
  // 4.2 (19.5 -> 15.3 us at M = 231 k, layer norm, frozen decoder): with a directory in the view every small launch pre-numbers.
  return tiles_prenumbered(n_tasks) && (mv->cdir_hdr != nullptr || !(mv->filter && mv->log2filter > 18));
}

int clid_decode_tile_blocks(int n_tasks) {
  const int tiles = (n_tasks + 1) / 2;
  const int tw = tile_waves_for(tiles);
  int nb = (tiles + tw - 1) / tw;
  return nb > kMaxBwdBlocks ? kMaxBwdBlocks : (nb < 1 ? 1 : nb);
}

int clid_launch_decode_tile(const clid_map_view* mv, const clid_train_args* a, float* partial, const TaskMap& tmap,
                            const float* rec, int prec, hipStream_t s) {
  if (a->grad_stride != CLID_GRAD_ROW16) {
    clid_set_error("clid_train_decode: the tile kernels need grad_stride == %d (got %d)", CLID_GRAD_ROW16, a->grad_stride);
    return CLID_E_ARG;
  }
  if (((uintptr_t)a->grad & 63) != 0) {
    clid_set_error("clid_train_decode: grad must be 64-byte aligned for the 16-float accumulation rows");
    return CLID_E_ARG;
  }
  const int n_tiles = (tmap.n_tasks + 1) / 2;
  const int nb = clid_decode_tile_blocks(tmap.n_tasks);
  const float4* r4 = reinterpret_cast<const float4*>(rec);
  const int* tn = reinterpret_cast<const int*>(rec + (size_t)tmap.n_tasks * kRecFloatsPerTask);  // (rec_floats_per_iter layout)
  const bool pre = clid_tiles_prenumbered(tmap.n_tasks, mv);
  int nb3 = (n_tiles + 3) / 4;  // (the 4-wave blocks of the 3-waves-per-SIMD instantiation: never more partial rows than `nb`)
  if (nb3 > nb) nb3 = nb;
#define CLID_TILE_LAUNCH_K(K, TWV)                                                                                    \
  CLID_KLAUNCH(a->prof, 0, K, dim3(nb), dim3((TWV) * 64), 0, s, r4, tn, n_tiles, partial, a->sdf_dbg, tmap, *mv, *a)
#define CLID_TILE_LAUNCH(P, L)                                                                                        \
  do {                                                                                                                \
    const bool small = tile_waves_for(n_tiles) == kTileWavesSmall;                                                    \
    if (a->eik_mask || a->main_loss_type) { /* config.ekional_add_to surface / freespace, main_loss_type != bce */     \
      if (small && pre) CLID_TILE_LAUNCH_K((k_decode_tile<P, L, kTileWavesSmall, true, 0, true>), kTileWavesSmall);   \
      else if (small) CLID_TILE_LAUNCH_K((k_decode_tile<P, L, kTileWavesSmall, false, 0, true>), kTileWavesSmall);    \
      else CLID_TILE_LAUNCH_K((k_decode_tile<P, L, kTileWavesLarge, false, 0, true>), kTileWavesLarge);               \
    } else if (small && pre)                                                                                          \
      CLID_TILE_LAUNCH_K((k_decode_tile<P, L, kTileWavesSmall, true>), kTileWavesSmall);                              \
    else if (small)                                                                                                   \
      CLID_TILE_LAUNCH_K((k_decode_tile<P, L, kTileWavesSmall, false>), kTileWavesSmall);                             \
    else if (!L && (a->debug_flags & 16)) /* debug bit 4: the 3-waves-per-SIMD instantiation (operands from LDS; 4-wave blocks) */ \
      CLID_KLAUNCH(a->prof, 0, (k_decode_tile<P, L, 4, false, L ? 0 : 3>), dim3(nb3), dim3(256), 0, s, r4, tn, n_tiles, partial, \
                   a->sdf_dbg, tmap, *mv, *a);                                                                         \
    else                                                                                                              \
      CLID_TILE_LAUNCH_K((k_decode_tile<P, L, kTileWavesLarge, false>), kTileWavesLarge);                             \
  } while (0)
  if (prec == 1) {
    if (mv->layer_norm) CLID_TILE_LAUNCH(1, true);
    else CLID_TILE_LAUNCH(1, false);
  } else {
    if (mv->layer_norm) CLID_TILE_LAUNCH(0, true);
    else CLID_TILE_LAUNCH(0, false);
  }
#undef CLID_TILE_LAUNCH_K
#undef CLID_TILE_LAUNCH
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}
