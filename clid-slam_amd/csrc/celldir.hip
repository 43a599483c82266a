// Cell directory of the local window (include/clid_native.h clid_cdir_build): the occupancy of the window's voxels as
// the reference's probe chain sees it -- buffer_pt_index[hash(cell) mod B] -> travel-distance filter -> global2local
// (model/neural_points.py:984-1009, 595-598) -- as one bit per cell over the bounding box of the window's points, a rank
// per 32 cells and the (x, y, z, id) rows of the hits in cell order.  Built from the compact probe table (csrc/table.hip),
// which already IS that chain keyed by slot number, so collisions of the big table are reproduced bit for bit:
//   k_cdir_box    cell bounding box of the window's points (biased non-negative extremes over a zeroed header)
//   k_cdir_bits   every block derives origin = min - margin, dims and word count from the extremes (block 0 stores them); 8 lanes
//                 per 32 z-adjacent cells: the cells' slot numbers by one exact hash + modular additions, prefilter bit, bucket
//                 compare on the 12 % that pass; occupancy word + hit count per 32 words
//   k_cdir_scan   exclusive scan of those counts (one block), hit total, validity against the row capacity
//   k_cdir_rows   rank of every word's first hit, the hits' rows copied from the table in cell order, the next word's low
//                 bits packed next to the rank (a stencil row that straddles two words is still one 8-byte load)
// No launch depends on a host read-back: the arrays have capacities, the header carries the sizes and a validity word.
#include <limits.h>
#include <string.h>

#include "common.hpp"

namespace clid {

constexpr int kCdirBlock = 256;

struct CdirHdr {       // CLID_CDIR_HDR_INTS = 16 ints
  int ox, oy, oz;      // cell coordinates of the box's corner
  int nx, ny, nz;      // cells per axis
  int nzw;             // 32-cell words per (x, y) column
  int words;           // nx * ny * nzw
  int valid;           // 1 = the directory describes the window; 0 = it did not fit (searches probe the table instead)
  int n_hits;
  int mn[3], mx[3];    // scratch of the bounding-box reduction
};
static_assert(sizeof(CdirHdr) == CLID_CDIR_HDR_INTS * 4, "header layout");

__device__ __forceinline__ int cell_of(float v, float res) { return (int)floorf(fdiv(v, res)); }

// Bounding box over a header zeroed by a memset: the six extremes are kept as non-negative numbers (kBias + c and kBias - c,
// so that 0 is the neutral element of the max).  |cell| < kBias = 2^22 also keeps every product below in int32 and inside the
// exact range of the fp64 hash (common.hpp base_slot); a point beyond it raises mx[0] to INT_MAX: no directory.
// (No "last block finishes the job" here or below: a device-scope release fence per block is an L2 write-back per block on
// this multi-XCD part -- a ticket version of k_cdir_bits took 217 us instead of 25.)
constexpr int kBias = 1 << 22;
__global__ void __launch_bounds__(kCdirBlock) k_cdir_box(const float4* __restrict__ pos4, int n, float res, CdirHdr* hdr) {
  __shared__ int smx[6][kCdirBlock / 64];
  int v[6] = {0, 0, 0, 0, 0, 0};  // [0..3): max(kBias - c), [3..6): max(kBias + c)
  bool bad = false;
  for (int j = blockIdx.x * kCdirBlock + threadIdx.x; j < n; j += gridDim.x * kCdirBlock) {
    const float4 p = pos4[j];
    const int c[3] = {cell_of(p.x, res), cell_of(p.y, res), cell_of(p.z, res)};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      bad = bad || c[a] <= -kBias || c[a] >= kBias;
      const int cc = min(max(c[a], 1 - kBias), kBias - 1);
      v[a] = max(v[a], kBias - cc);
      v[3 + a] = max(v[3 + a], kBias + cc);
    }
  }
  if (bad) v[3] = INT_MAX;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    for (int o = 32; o > 0; o >>= 1) v[a] = max(v[a], __shfl_xor(v[a], o, 64));
    if ((threadIdx.x & 63) == 0) smx[a][threadIdx.x >> 6] = v[a];
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    int m = smx[threadIdx.x][0];
    for (int w = 1; w < kCdirBlock / 64; ++w) m = max(m, smx[threadIdx.x][w]);
    atomicMax(threadIdx.x < 3 ? &hdr->mn[threadIdx.x] : &hdr->mx[threadIdx.x - 3], m);
  }
}
// origin, dims, word count, validity from the reduced extremes (every block of k_cdir_bits derives them; its block 0 stores them)
__device__ __forceinline__ CdirHdr box_of(const CdirHdr* hdr, int n, long long words_cap) {
  CdirHdr h;
  bool ok = n > 0 && hdr->mx[0] != INT_MAX && hdr->mx[0] > 0;
  for (int a = 0; a < 3; ++a) {
    h.mn[a] = kBias - hdr->mn[a];
    h.mx[a] = hdr->mx[a] - kBias;
  }
  long long words = 0;
  if (ok) {
    // margins: a query point farther than (margin - nc) cells outside the points' box defers its task to the probing launch
    // (one dependent chain of ~12 us even for a single task), so x / y get the wide margin -- the sample pool reaches a little
    // beyond the window's points there.  Along z a cell costs a 32nd of a word per column: the column gets the fewest words
    // that leave >= kMinMarginZ (= the largest nc) cells on either side, and the cells those words have to spare are the margin
    constexpr int kMinMarginZ = 2;
    h.ox = h.mn[0] - CLID_CDIR_MARGIN_XY;
    h.oy = h.mn[1] - CLID_CDIR_MARGIN_XY;
    h.nx = h.mx[0] - h.mn[0] + 1 + 2 * CLID_CDIR_MARGIN_XY;
    h.ny = h.mx[1] - h.mn[1] + 1 + 2 * CLID_CDIR_MARGIN_XY;
    const int ez = h.mx[2] - h.mn[2] + 1;
    h.nzw = (ez + 2 * kMinMarginZ + 31) >> 5;
    h.nz = 32 * h.nzw;
    h.oz = h.mn[2] - (h.nz - ez) / 2;
    words = (long long)h.nx * h.ny * h.nzw;
    ok = words <= words_cap && words < (1LL << 30);
  }
  if (!ok) {
    h.ox = h.oy = h.oz = 0;
    h.nx = h.ny = h.nz = h.nzw = 0;
    words = 0;
  }
  h.words = (int)words;
  h.valid = ok ? 1 : 0;
  h.n_hits = 0;
  return h;
}

// slot of cell (cx, cy, cz): the exact hash of common.hpp base_slot from integer cell coordinates
__device__ __forceinline__ int slot_of_cell(int cx, int cy, int cz, int B) {
  const double h = fma((double)cx, 73856093.0, fma((double)cy, 19349669.0, (double)cz * 83492791.0));
  const double Bd = (double)B;
  const double q = floor(h / Bd);
  double r = fma(-q, Bd, h);
  if (r < 0.0) r += Bd;
  if (r >= Bd) r -= Bd;
  return (int)r;
}

__device__ __forceinline__ void word_cell(const CdirHdr& h, int w, int& cx, int& cy, int& cz0, int& ncell) {
  const int col = w / h.nzw, izw = w - col * h.nzw;
  const int ix = col / h.ny, iy = col - ix * h.ny;
  cx = h.ox + ix;
  cy = h.oy + iy;
  cz0 = h.oz + 32 * izw;
  ncell = min(32, h.nz - 32 * izw);
}

// Both word kernels: 8 lanes per word (4 z-adjacent cells each: their loads are in flight together -- a thread walking its 32
// cells one lookup after the other was one dependent chain of 32 round trips, 24 us on a 70 k-word box), 32 words per
// 256-thread "logical block" (the unit of the rank scan), a persistent grid striding over the logical blocks (the box is
// sized on the device; the launch is not).
constexpr int kWordsPerBlock = kCdirBlock / 8;

__global__ void __launch_bounds__(kCdirBlock) k_cdir_bits(CdirHdr* __restrict__ hdr, int n, long long words_cap, const int4* __restrict__ tab,
                                                          int log2cap, const unsigned* __restrict__ filter, int log2filter, int B, int p3mod,
                                                          uint2* __restrict__ words, int* __restrict__ block_count) {
  __shared__ int wsum[kCdirBlock / 64];
  const CdirHdr h = box_of(hdr, n, words_cap);
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // (the fields the other blocks read -- mn, mx -- keep their values)
    hdr->ox = h.ox; hdr->oy = h.oy; hdr->oz = h.oz;
    hdr->nx = h.nx; hdr->ny = h.ny; hdr->nz = h.nz;
    hdr->nzw = h.nzw; hdr->words = h.words; hdr->valid = h.valid; hdr->n_hits = 0;
  }
  const int used = h.valid ? (h.words + kWordsPerBlock - 1) / kWordsPerBlock : 0;
  const int chunk = threadIdx.x & 7;
  for (int lb = blockIdx.x; lb < used; lb += gridDim.x) {
    const int w = lb * kWordsPerBlock + (threadIdx.x >> 3);
    unsigned bits = 0;
    if (w < h.words) {
      int cx, cy, cz0, ncell;
      word_cell(h, w, cx, cy, cz0, ncell);
      int slot[4];
      slot[0] = slot_of_cell(cx, cy, cz0 + 4 * chunk, B);  // hash(c + ez) = hash(c) + prime_z (mod B)
#pragma unroll
      for (int b = 1; b < 4; ++b) {
        slot[b] = slot[b - 1] + p3mod;
        if (slot[b] >= B) slot[b] -= B;
      }
      bool pass[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {  // prefilter bits of the 4 cells: all loads in flight
        pass[b] = 4 * chunk + b < ncell;
        if (filter && pass[b]) {
          const unsigned f = filter_bit(slot[b], log2filter);
          pass[b] = (filter[f >> 5] >> (f & 31)) & 1u;
        }
      }
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (pass[b]) {
          const unsigned home = tab_home(slot[b], log2cap);
          if (tab_find(tab, log2cap, slot[b], home, tab[home]) >= 0) bits |= 1u << (4 * chunk + b);
        }
    }
    bits |= __shfl_xor(bits, 1, 64);
    bits |= __shfl_xor(bits, 2, 64);
    bits |= __shfl_xor(bits, 4, 64);
    if (chunk == 0 && w < h.words) words[w] = make_uint2(bits, 0u);
    int c = chunk == 0 ? __popc(bits) : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int i = 0; i < kCdirBlock / 64; ++i) t += wsum[i];
      block_count[lb] = t;
    }
    __syncthreads();
  }
}

// exclusive scan of the logical blocks' hit counts in place (one block); total -> hdr->n_hits, validity against hits_cap
__global__ void __launch_bounds__(1024) k_cdir_scan(CdirHdr* hdr, int* __restrict__ block_count, long long hits_cap) {
  __shared__ int part[1024];
  const int used = hdr->valid ? (hdr->words + kWordsPerBlock - 1) / kWordsPerBlock : 0;
  const int per = (used + 1023) / 1024;
  const int b0 = threadIdx.x * per, b1 = min(used, b0 + per);
  int s = 0;
  for (int b = b0; b < b1; ++b) s += block_count[b];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
    const int v = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = part[threadIdx.x] - s;
  for (int b = b0; b < b1; ++b) {
    const int c = block_count[b];
    block_count[b] = run;
    run += c;
  }
  if (threadIdx.x == 1023) {
    const int total = part[1023];
    hdr->n_hits = total;
    if (total > hits_cap || total >= (1 << 24)) hdr->valid = 0;  // (the rank shares its word with 8 bits of the next word)
  }
}

__global__ void __launch_bounds__(kCdirBlock) k_cdir_rows(const CdirHdr* __restrict__ hdr, const int4* __restrict__ tab,
                                                          const float4* __restrict__ tab_pos, int log2cap, int B, int p3mod,
                                                          uint2* __restrict__ words, const int* __restrict__ block_prefix,
                                                          float4* __restrict__ pos_out) {
  __shared__ int wpop[kWordsPerBlock];
  const CdirHdr h = *hdr;
  if (!h.valid) return;  // (uniform)
  const int used = (h.words + kWordsPerBlock - 1) / kWordsPerBlock;
  const int chunk = threadIdx.x & 7, wi = threadIdx.x >> 3;
  for (int lb = blockIdx.x; lb < used; lb += gridDim.x) {
    const int w = lb * kWordsPerBlock + wi;
    const bool live = w < h.words;
    const unsigned bits = live ? words[w].x : 0u;
    if (chunk == 0) wpop[wi] = __popc(bits);
    __syncthreads();
    int rank0 = block_prefix[lb];
    for (int i = 0; i < wi; ++i) rank0 += wpop[i];
    __syncthreads();
    if (!live) continue;
    if (chunk == 0) {
      // the next word of the SAME column (a stencil row never leaves its column: the box has a margin of >= nc cells)
      const bool has_next = (w + 1 < h.words) && ((w + 1) % h.nzw != 0);
      const unsigned next_low = has_next ? (words[w + 1].x & 0xFFu) : 0u;
      words[w].y = (unsigned)rank0 | (next_low << 24);
    }
    const unsigned mine = (bits >> (4 * chunk)) & 15u;
    if (!mine) continue;
    int cx, cy, cz0, ncell;
    word_cell(h, w, cx, cy, cz0, ncell);
    int slot = slot_of_cell(cx, cy, cz0 + 4 * chunk, B);
    int r = rank0 + __popc(bits & ((1u << (4 * chunk)) - 1u));
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if ((mine >> b) & 1u) {  // (the bit says the slot is in the table: no prefilter test)
        const unsigned home = tab_home(slot, log2cap);
        const int cell = tab_find(tab, log2cap, slot, home, tab[home]);
        pos_out[r++] = tab_pos[cell >= 0 ? cell : 0];
      }
      slot += p3mod;
      if (slot >= B) slot -= B;
    }
  }
}

}  // namespace clid

extern "C" int clid_cdir_build(const float* pos4, int32_t n, const int32_t* tab, const float* tab_pos, int32_t log2cap,
                               const uint32_t* filter, int32_t log2filter, int64_t buffer_size, float resolution, int32_t* hdr_out,
                               uint32_t* words_out, int64_t words_cap, float* pos_out, int64_t hits_cap, int32_t* scratch, void* stream) {
  using namespace clid;
  if (n < 0 || !tab || !tab_pos || !hdr_out || !words_out || !pos_out || !scratch || words_cap < kCdirBlock || hits_cap < 1 ||
      words_cap >= (1LL << 30) || buffer_size <= 0 || buffer_size >= (1LL << 30) || !(resolution > 0.f) || (n > 0 && !pos4)) {
    clid_set_error("clid_cdir_build: bad argument (n=%d words_cap=%lld hits_cap=%lld)", n, (long long)words_cap, (long long)hits_cap);
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  CdirHdr* hdr = reinterpret_cast<CdirHdr*>(hdr_out);
  if (hipMemsetAsync(hdr, 0, sizeof(CdirHdr), s) != hipSuccess) {
    clid_set_error("clid_cdir_build: header reset failed");
    return CLID_E_HIP;
  }
  const int bb = (n + kCdirBlock - 1) / kCdirBlock;
  hipLaunchKernelGGL(k_cdir_box, dim3(bb < 1 ? 1 : (bb < 128 ? bb : 128)), dim3(kCdirBlock), 0, s, reinterpret_cast<const float4*>(pos4), n,
                     resolution, hdr);
  const long long lb_cap = (words_cap + kWordsPerBlock - 1) / kWordsPerBlock;
  const int grid = (int)(lb_cap < 2048 ? lb_cap : 2048);
  const int B = (int)buffer_size;
  const int p3mod = (int)(83492791LL % buffer_size);
  const int4* t4 = reinterpret_cast<const int4*>(tab);
  uint2* w2 = reinterpret_cast<uint2*>(words_out);
  hipLaunchKernelGGL(k_cdir_bits, dim3(grid), dim3(kCdirBlock), 0, s, hdr, n, (long long)words_cap, t4, log2cap, filter, log2filter, B,
                     p3mod, w2, scratch);
  hipLaunchKernelGGL(k_cdir_scan, dim3(1), dim3(1024), 0, s, hdr, scratch, (long long)hits_cap);
  hipLaunchKernelGGL(k_cdir_rows, dim3(grid), dim3(kCdirBlock), 0, s, hdr, t4, reinterpret_cast<const float4*>(tab_pos), log2cap, B,
                     p3mod, w2, scratch, reinterpret_cast<float4*>(pos_out));
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}
