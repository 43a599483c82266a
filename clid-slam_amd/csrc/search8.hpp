// The 8-lane neighbour searches (81-cell stencil of NeuralPoints.radius_neighborhood_search, model/neural_points.py:971-1030):
// probing the compact table (search8) and walking the window's cell directory (search_cells).  Shared by the training search
// launches (csrc/train.hip) and the dense SDF query on the matrix cores (csrc/query_tile.hip).
#pragma once
#include "train_common.hpp"

namespace clid {

__device__ __forceinline__ float group8_min(float v) {
  v = fminf(v, dpp_mov<0xB1>(v));   // quad_perm [1,0,3,2]
  v = fminf(v, dpp_mov<0x4E>(v));   // quad_perm [2,3,0,1]
  v = fminf(v, dpp_mov<0x141>(v));  // row_half_mirror: the other quad of the 8-lane group
  return v;
}
__device__ __forceinline__ int group8_sum_i(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);
  return v;
}

__device__ __forceinline__ float group8_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  return v;
}

#ifndef CLID_PROBE_ROWS
#define CLID_PROBE_ROWS 6
#endif
constexpr int kProbeRows8 = CLID_PROBE_ROWS;  // probes per lane per chunk; chunk = 8 * rows slots

// Per-lane sorted candidate list of DEPTH entries.  The 81 probes of a query are spread over 8 lanes, so a lane
// almost never owns more than 3 of the 6 winners: the throughput (search-only) kernel runs with DEPTH = 3 -- half
// the compare/select work of the insert, which is 40 % of the search's instructions -- and remembers the best
// distance it ever pushed out; if that could have been a winner the wave repeats the search at full depth.
template <int DEPTH>
struct CandN {
  float d[DEPTH];
  int j[DEPTH];
  float dropped;
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      d[s] = __builtin_inff();
      j[s] = -1;
    }
    dropped = __builtin_inff();
  }
  __device__ __forceinline__ void insert(float nd, int nj) {
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      const bool lt = nd < d[s];
      const float td = lt ? d[s] : nd;
      const int tj = lt ? j[s] : nj;
      d[s] = lt ? nd : d[s];
      j[s] = lt ? nj : j[s];
      nd = td;
      nj = tj;
    }
    if (DEPTH < CLID_K) dropped = fminf(dropped, nd);
  }
  __device__ __forceinline__ void pop() {
#pragma unroll
    for (int s = 0; s < DEPTH - 1; ++s) {
      d[s] = d[s + 1];
      j[s] = j[s + 1];
    }
    d[DEPTH - 1] = __builtin_inff();
    j[DEPTH - 1] = -1;
  }
};

// The K winners of the group's sorted per-lane lists -> win[0..K), ascending distance.  Candidates carry their probe index
// above their id (kProbeShift): when several lanes hold the same distance the lowest index wins -- a stable sort of the
// reference's dist2 row (np.py:607-609).  Returns the distance of the K-th winner (inf when fewer were found).
// PSH (here and below): the id's width in a candidate -- kProbeShift for the training searches (local window), kProbeShiftWide for
// inference over global tables (common.hpp)
template <int DEPTH, int PSH = kProbeShift>
__device__ __forceinline__ float select_packed(CandN<DEPTH>& c, int lane8, int gshift, float2* __restrict__ win) {
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < CLID_K; ++k) {
    const float head = c.d[0];
    m = group8_min(head);
    const bool mine = (head == m) && (c.j[0] >= 0);
    const unsigned long long b = __ballot(mine);
    const unsigned gb = (unsigned)(b >> gshift) & 0xFFu;
    bool take = gb && lane8 == (int)(__ffs(gb) - 1);
    if (__any((gb & (gb - 1u)) != 0u)) {  // rare: several lanes hold the same distance
      const int key = mine ? (c.j[0] >> PSH) : 0x7fffffff;
      int kmin = min(key, __builtin_amdgcn_update_dpp(0x7fffffff, key, 0xB1, 0xF, 0xF, false));
      kmin = min(kmin, __builtin_amdgcn_update_dpp(0x7fffffff, kmin, 0x4E, 0xF, 0xF, false));
      kmin = min(kmin, __builtin_amdgcn_update_dpp(0x7fffffff, kmin, 0x141, 0xF, 0xF, false));
      take = mine && key == kmin;
    }
    if (take) {
      win[k] = make_float2(m, __int_as_float(c.j[0] & ((1 << PSH) - 1)));
      c.pop();
    }
  }
  return m;
}

// 81-cell search of one query by the 8 lanes of a group; winners -> win[0..K).  Returns true when a DEPTH < K
// list may have lost a winner (the caller then repeats with DEPTH = K).
// `filt` (optional, LDS): bit per stored slot; a probe whose bit is clear cannot match and is not loaded
// COUNT: *nvalid += this lane's probes that hit a point within max_valid_dist2 (np.py:600-602; the dense SDF query's mask)
template <bool FILTER, int DEPTH, bool COUNT = false, int PSH = kProbeShift>
__device__ __forceinline__ bool search8(const clid_map_view& mv, const DeltaLds& dl, float x, float y, float z,
                                        int lane8, int gshift, float2* __restrict__ win,
                                        const unsigned* __restrict__ filt = nullptr, int* nvalid = nullptr) {
  const int4* __restrict__ tab = reinterpret_cast<const int4*>(mv.tab);
  const float4* __restrict__ tpos = reinterpret_cast<const float4*>(mv.tab_pos);
  const int B = mv.buffer_size;
  const int r0 = base_slot(x, y, z, mv.resolution, B);
  CandN<DEPTH> c;
  c.init();
  if (lane8 < CLID_K) win[lane8] = make_float2(9e3f, __int_as_float(-1));  // np.py:606
  // Lane l takes the probes o = l, l + 8, l + 16, ... in ascending order (interleaved: the winners of a query, which cluster in
  // the middle of the probe order, spread evenly over the lanes -- with contiguous ranges per lane a DEPTH-3 list overflowed in
  // most waves: search 8.3 -> 15.1 us).  A candidate carries its probe index next to its id (kProbeShift), so that equal
  // distances can be resolved by probe index -- the order of a STABLE sort of the reference's dist2 row (np.py:607-609; torch's
  // own sort there is unstable: its choice among equidistant candidates at the K-th place is implementation-defined).  Inside
  // a lane the strict < of the insert keeps equals in probe order; between lanes the selection below compares the indices.
  for (int o0 = 0; o0 < mv.P; o0 += 8 * kProbeRows8) {
    int slot[kProbeRows8];
    unsigned home[kProbeRows8];
    int4 bk[kProbeRows8];
#pragma unroll
    for (int t = 0; t < kProbeRows8; ++t) {  // straight-line (predicated) so all loads of the chunk batch
      const int o = o0 + 8 * t + lane8;
      int sl = r0 + dl.d[o];
      sl = (int)min((unsigned)sl, (unsigned)(sl - B));  // sl < 2 B: one conditional subtraction as add / sub / min
      bool in = o < mv.P;
      slot[t] = in ? sl : -2;
      home[t] = tab_home(sl, mv.log2cap);
      if constexpr (FILTER) {
        const unsigned b = filter_bit(sl, mv.log2filter);
        in = in && ((filt[b >> 5] >> (b & 31)) & 1u);
        bk[t] = make_int4(-1, -1, -1, -1);  // "empty bucket": no match, no walk
        if (in) bk[t] = tab[home[t]];
      } else {
        bk[t] = tab[in ? home[t] : 0];
      }
    }
    int cell[kProbeRows8];
    bool walk = false;
#pragma unroll
    for (int t = 0; t < kProbeRows8; ++t) {
      const int m = bucket_match(bk[t], slot[t]);
      cell[t] = m >= 0 ? (int)(home[t] * 4u) + m : -1;
      walk |= (m < 0) && (bk[t].w >= 0) && (slot[t] != -2);
    }
    if (__any(walk)) {  // rare: a full bucket without a match
#pragma unroll
      for (int t = 0; t < kProbeRows8; ++t)
        if (cell[t] < 0 && slot[t] != -2 && bk[t].w >= 0) cell[t] = tab_find(tab, mv.log2cap, slot[t], home[t], bk[t]);
    }
    float4 pp[kProbeRows8];
#pragma unroll
    for (int t = 0; t < kProbeRows8; ++t) pp[t] = tpos[cell[t] >= 0 ? cell[t] : 0];
#pragma unroll
    for (int t = 0; t < kProbeRows8; ++t) {
      const float ax = fsub(pp[t].x, x), ay = fsub(pp[t].y, y), az = fsub(pp[t].z, z);
      const float d2 = fadd(fadd(fmul(ax, ax), fmul(ay, ay)), fmul(az, az));
      if (cell[t] >= 0 && !(d2 > mv.max_valid_dist2)) {  // np.py:1016-1020
        c.insert(d2, __float_as_int(pp[t].w) | ((o0 + 8 * t + lane8) << PSH));
        if constexpr (COUNT) ++*nvalid;
      }
    }
  }
  CLID_STAMP(2);
  const float m = select_packed<DEPTH, PSH>(c, lane8, gshift, win);
  // m = distance of the 6th winner (inf when fewer were found): anything pushed out at or below it is suspect
  return DEPTH < CLID_K && c.dropped <= m && c.dropped < __builtin_inff();
}

// ---- the same search over the window's CELL DIRECTORY (csrc/celldir.hip) ------------------------------------------------
// search8 pays per PROBE: slot arithmetic, a prefilter bit, a 4-key bucket compare -- 81 times per query although 3 of 4
// probed cells are empty.  The directory answers a whole stencil row of 2 nc + 1 z-adjacent cells with ONE 8-byte load:
// occupancy bits of 32 cells | rank of the word's first hit (the straddling bits of the next word packed beside it); the
// hits of a row are consecutive rows of `cdir_pos`.  Row-major (dx, dy) with dz fastest IS the probe order of np.py:931-969,
// so hit number g of a query is its g-th valid probe.  Lane l of the query's 8 lanes takes rows [4 l, 4 l + 4), the lanes
// agree on the hits' numbering with one 8-lane scan, expand their rows' rank ranges into an LDS list, and then every lane
// takes an equal share of the hits (l, l + 8, ...): one position load + distance + insert per hit instead of per probe.
#ifndef CLID_CD_WAVES_TASKS
#define CLID_CD_WAVES_TASKS 8  // waves per SIMD the directory-search instantiations are compiled for (64 VGPRs without the
#endif                         // probing code: 8 waves hide the words -> list -> position chain better than 6)
#ifndef CLID_CD_WAVES_TILES
#define CLID_CD_WAVES_TILES 6
#endif
// inclusive sum over the lanes 0 .. lane8 of an 8-lane group (row_shr never reaches across the group: guarded by lane8)
__device__ __forceinline__ int group8_scan_i(int v, int lane8) {
  int t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
  v += lane8 >= 1 ? t : 0;
  t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
  v += lane8 >= 2 ? t : 0;
  t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
  v += lane8 >= 4 ? t : 0;
  return v;
}
// The hits of one query, listed in LDS by rank in probe order: lane l takes the hits l, l + 8, ... (interleaved like search8's
// probes: the winners spread over the lanes), kCdBatch position loads in flight per lane and trip; candidates carry their hit
// number above the id, ties resolve as in search8.  `trips` is the wave's maximum (uniform).  Returns search8's "repeat at full depth".
constexpr int kCdBatch = 3;
template <int DEPTH, bool COUNT = false, int PSH = kProbeShift>
__device__ __forceinline__ bool consume_hits(const clid_map_view& mv, const int* __restrict__ list, int H, int trips, float x, float y,
                                             float z, int lane8, int gshift, float2* __restrict__ win, int* nvalid = nullptr) {
  const float4* __restrict__ cpos = reinterpret_cast<const float4*>(mv.cdir_pos);
  CandN<DEPTH> c;
  c.init();
  for (int i0 = 0; i0 < trips; ++i0) {
    float4 pp[kCdBatch];
    int g[kCdBatch];
#pragma unroll
    for (int t = 0; t < kCdBatch; ++t) {
      g[t] = lane8 + 8 * (kCdBatch * i0 + t);
      const bool ok = g[t] < H;
      pp[t] = cpos[ok ? list[ok ? g[t] : 0] : 0];
    }
#pragma unroll
    for (int t = 0; t < kCdBatch; ++t) {
      const float ax = fsub(pp[t].x, x), ay = fsub(pp[t].y, y), az = fsub(pp[t].z, z);
      const float d2 = fadd(fadd(fmul(ax, ax), fmul(ay, ay)), fmul(az, az));
      if (g[t] < H && !(d2 > mv.max_valid_dist2)) {  // np.py:1016-1020
        c.insert(d2, __float_as_int(pp[t].w) | (g[t] << PSH));
        if constexpr (COUNT) ++*nvalid;
      }
    }
  }
  const float m = select_packed<DEPTH, PSH>(c, lane8, gshift, win);
  return DEPTH < CLID_K && c.dropped <= m && c.dropped < __builtin_inff();
}
// One query of the task: (rx, ry) = cell - origin, rz0 = cell_z - origin_z - nc (all in range: the caller checked).  `list`:
// this query slot's kCdHits ints in LDS.
template <bool COUNT = false, int PSH = kProbeShift>
__device__ __forceinline__ void search_cells(const clid_map_view& mv, const CellLds& cl, int* __restrict__ list, float x, float y,
                                             float z, int rx, int ry, int rz0, int lane8, int gshift, float2* __restrict__ win,
                                             bool full_depth, int* nvalid = nullptr) {
  const uint2* __restrict__ words = reinterpret_cast<const uint2*>(mv.cdir_words);
  const int sh = rz0 & 31;
  const int qbase = (rx * cl.ny + ry) * cl.nzw + (rz0 >> 5);
  const unsigned low = (1u << sh) - 1u;
  int4 rw[4];
  uint2 e[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    rw[t] = cl.row[4 * lane8 + t];
    e[t] = make_uint2(0u, 0u);
    if (rw[t].y) e[t] = words[qbase + rw[t].x];
  }
  CLID_STAMP(14);  // cell coordinates, stencil rows, directory words requested
  int rf[4], cnt[4], n_l = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const unsigned a = __builtin_amdgcn_alignbit(e[t].y >> 24, e[t].x, (unsigned)sh);  // cells rz0 .. of the row's column
    cnt[t] = __popc(a & (unsigned)rw[t].y);
    rf[t] = (int)(e[t].y & 0xFFFFFFu) + __popc(e[t].x & low) + __popc(a & (unsigned)rw[t].z);  // rank of the row's first hit
    n_l += cnt[t];
  }
  const int incl = group8_scan_i(n_l, lane8);
  const int H = group8_sum_i(n_l);
  int p = incl - n_l;
#pragma unroll
  for (int t = 0; t < 4; ++t) {  // the stencil bits of a row are one run (a ball), so its hits have consecutive ranks
#pragma unroll
    for (int k = 0; k < 5; ++k)
      if (k < cnt[t]) list[p + k] = rf[t] + k;
    p += cnt[t];
  }
  wave_lds_fence();
  CLID_STAMP(16);  // directory words arrived, hit ranks expanded into the LDS list
  int hmax = H;  // the wave's largest hit count -> uniform trip count
  hmax = max(hmax, __shfl_xor(hmax, 8, 64));
  hmax = max(hmax, __shfl_xor(hmax, 16, 64));
  hmax = max(hmax, __shfl_xor(hmax, 32, 64));
  const int trips = (__builtin_amdgcn_readfirstlane(hmax) + 8 * kCdBatch - 1) / (8 * kCdBatch);
  if (lane8 < CLID_K) win[lane8] = make_float2(9e3f, __int_as_float(-1));  // np.py:606
  bool redo = full_depth;
  if (!full_depth) redo = consume_hits<3, COUNT, PSH>(mv, list, H, trips, x, y, z, lane8, gshift, win, nvalid);
  if (__any(redo)) {  // a 3-deep list may have pushed a winner out (or debug bit 2): once more at full depth, from the same list
    if (lane8 < CLID_K) win[lane8] = make_float2(9e3f, __int_as_float(-1));
    if (full_depth) consume_hits<CLID_K, COUNT, PSH>(mv, list, H, trips, x, y, z, lane8, gshift, win, nvalid);
    else consume_hits<CLID_K, false, PSH>(mv, list, H, trips, x, y, z, lane8, gshift, win);  // (counted by the first pass)
  }
}

}  // namespace clid
