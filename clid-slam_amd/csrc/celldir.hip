// Cell directory of the local window (include/clid_native.h clid_cdir_build): the occupancy of the window's voxels as
// the reference's probe chain sees it -- buffer_pt_index[hash(cell) mod B] -> travel-distance filter -> global2local
// (model/neural_points.py:984-1009, 595-598) -- as one bit per cell over the bounding box of the window's points, a rank
// per 32 cells and the (x, y, z, id) rows of the hits in cell order.  Built from the compact probe table (csrc/table.hip),
// which already IS that chain keyed by slot number, so collisions of the big table are reproduced bit for bit:
//   k_cdir_bbox   cell bounding box of the window's points (block-reduced, 6 atomics per block)
//   k_cdir_setup  origin = min - margin, dims, word count, validity (one thread)
//   k_cdir_bits   one thread per 32 z-adjacent cells: the cells' slot numbers by ONE exact hash + 31 modular additions,
//                 prefilter bit, bucket compare on the 12 % that pass; occupancy word + per-block hit counts
//   k_cdir_scan   exclusive scan of the block counts (one block)
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

__global__ void k_cdir_init(CdirHdr* h, CdirHdr v) { *h = v; }  // (the initial header travels as a kernel argument)

__global__ void __launch_bounds__(kCdirBlock) k_cdir_bbox(const float4* __restrict__ pos4, int n, float res, CdirHdr* hdr) {
  __shared__ int smn[3][kCdirBlock / 64], smx[3][kCdirBlock / 64];
  int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
  for (int j = blockIdx.x * kCdirBlock + threadIdx.x; j < n; j += gridDim.x * kCdirBlock) {
    const float4 p = pos4[j];
    const int c[3] = {cell_of(p.x, res), cell_of(p.y, res), cell_of(p.z, res)};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      mn[a] = min(mn[a], c[a]);
      mx[a] = max(mx[a], c[a]);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    for (int o = 32; o > 0; o >>= 1) {
      mn[a] = min(mn[a], __shfl_xor(mn[a], o, 64));
      mx[a] = max(mx[a], __shfl_xor(mx[a], o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      smn[a][threadIdx.x >> 6] = mn[a];
      smx[a][threadIdx.x >> 6] = mx[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    int lo = smn[a][0], hi = smx[a][0];
    for (int w = 1; w < kCdirBlock / 64; ++w) {
      lo = min(lo, smn[a][w]);
      hi = max(hi, smx[a][w]);
    }
    atomicMin(&hdr->mn[a], lo);
    atomicMax(&hdr->mx[a], hi);
  }
}

__global__ void k_cdir_setup(CdirHdr* hdr, int n, long long words_cap) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  CdirHdr h = *hdr;
  bool ok = n > 0;
  long long words = 0;
  if (ok) {
    // |cell| < 2^22 keeps every product below in int32 / the exact range of the fp64 hash (common.hpp base_slot)
    for (int a = 0; a < 3; ++a) ok = ok && h.mn[a] > -(1 << 22) && h.mx[a] < (1 << 22);
  }
  if (ok) {
    h.ox = h.mn[0] - CLID_CDIR_MARGIN;
    h.oy = h.mn[1] - CLID_CDIR_MARGIN;
    h.oz = h.mn[2] - CLID_CDIR_MARGIN;
    h.nx = h.mx[0] - h.mn[0] + 1 + 2 * CLID_CDIR_MARGIN;
    h.ny = h.mx[1] - h.mn[1] + 1 + 2 * CLID_CDIR_MARGIN;
    h.nz = h.mx[2] - h.mn[2] + 1 + 2 * CLID_CDIR_MARGIN;
    h.nzw = (h.nz + 31) >> 5;
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
  *hdr = h;
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

// table cell (bucket * 4 + key) holding `slot`, or -1: prefilter bit first (never a false negative)
__device__ __forceinline__ int cdir_lookup(const int4* __restrict__ tab, int log2cap, const unsigned* __restrict__ filter,
                                           int log2filter, int slot) {
  if (filter) {
    const unsigned b = filter_bit(slot, log2filter);
    if (!((filter[b >> 5] >> (b & 31)) & 1u)) return -1;
  }
  const unsigned home = tab_home(slot, log2cap);
  return tab_find(tab, log2cap, slot, home, tab[home]);
}

__device__ __forceinline__ void word_cell(const CdirHdr& h, int w, int& cx, int& cy, int& cz0, int& ncell) {
  const int col = w / h.nzw, izw = w - col * h.nzw;
  const int ix = col / h.ny, iy = col - ix * h.ny;
  cx = h.ox + ix;
  cy = h.oy + iy;
  cz0 = h.oz + 32 * izw;
  ncell = min(32, h.nz - 32 * izw);
}

__global__ void __launch_bounds__(kCdirBlock) k_cdir_bits(const CdirHdr* __restrict__ hdr, const int4* __restrict__ tab, int log2cap,
                                                          const unsigned* __restrict__ filter, int log2filter, int B, int p3mod,
                                                          uint2* __restrict__ words, int* __restrict__ block_count) {
  __shared__ int wsum[kCdirBlock / 64];
  const CdirHdr h = *hdr;
  const int w = blockIdx.x * kCdirBlock + threadIdx.x;
  unsigned bits = 0;
  if (h.valid && w < h.words) {
    int cx, cy, cz0, ncell;
    word_cell(h, w, cx, cy, cz0, ncell);
    int slot = slot_of_cell(cx, cy, cz0, B);
    for (int b = 0; b < ncell; ++b) {  // hash(c + ez) = hash(c) + prime_z (mod B)
      if (cdir_lookup(tab, log2cap, filter, log2filter, slot) >= 0) bits |= 1u << b;
      slot += p3mod;
      if (slot >= B) slot -= B;
    }
    words[w] = make_uint2(bits, 0u);
  }
  int c = __popc(bits);
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int i = 0; i < kCdirBlock / 64; ++i) t += wsum[i];
    block_count[blockIdx.x] = t;
  }
}

// exclusive scan of the block counts in place (one block of 1024 threads); total -> hdr->n_hits, validity against hits_cap
__global__ void __launch_bounds__(1024) k_cdir_scan(CdirHdr* hdr, int* __restrict__ block_count, int n_blocks_cap, long long hits_cap) {
  __shared__ int part[1024];
  const int used = hdr->valid ? (hdr->words + kCdirBlock - 1) / kCdirBlock : 0;
  const int per = (used + 1023) / 1024;
  const int b0 = threadIdx.x * per, b1 = min(used, b0 + per);
  int s = 0;
  for (int b = b0; b < b1; ++b) s += block_count[b];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
    const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
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
  (void)n_blocks_cap;
}

__global__ void __launch_bounds__(kCdirBlock) k_cdir_rows(const CdirHdr* __restrict__ hdr, const int4* __restrict__ tab,
                                                          const float4* __restrict__ tab_pos, int log2cap,
                                                          const unsigned* __restrict__ filter, int log2filter, int B, int p3mod,
                                                          uint2* __restrict__ words, const int* __restrict__ block_prefix,
                                                          float4* __restrict__ pos_out) {
  __shared__ int wsum[kCdirBlock / 64];
  const CdirHdr h = *hdr;
  if (!h.valid) return;  // (uniform)
  const int w = blockIdx.x * kCdirBlock + threadIdx.x;
  const bool live = w < h.words;
  const unsigned bits = live ? words[w].x : 0u;
  const int c = __popc(bits);
  // exclusive scan of the popcounts inside the block: wave scan + the waves' totals
  int incl = c;
  const int lane = threadIdx.x & 63;
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  int base = block_prefix[blockIdx.x];
  for (int i = 0; i < (int)(threadIdx.x >> 6); ++i) base += wsum[i];
  const int rank0 = base + incl - c;
  if (!live) return;
  // the next word of the SAME column (a stencil row never leaves its column: the box has a margin of >= nc cells)
  const bool has_next = (w + 1 < h.words) && ((w + 1) % h.nzw != 0);
  const unsigned next_low = has_next ? (words[w + 1].x & 0xFFu) : 0u;
  words[w].y = (unsigned)rank0 | (next_low << 24);
  if (!bits) return;
  int cx, cy, cz0, ncell;
  word_cell(h, w, cx, cy, cz0, ncell);
  int slot = slot_of_cell(cx, cy, cz0, B);
  int r = rank0;
  for (int b = 0; b < ncell; ++b) {
    if ((bits >> b) & 1u) {
      const int cell = cdir_lookup(tab, log2cap, filter, log2filter, slot);
      pos_out[r++] = tab_pos[cell >= 0 ? cell : 0];
    }
    slot += p3mod;
    if (slot >= B) slot -= B;
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
  CdirHdr init;
  memset(&init, 0, sizeof(init));
  for (int a = 0; a < 3; ++a) {
    init.mn[a] = INT_MAX;
    init.mx[a] = INT_MIN;
  }
  CdirHdr* hdr = reinterpret_cast<CdirHdr*>(hdr_out);
  hipLaunchKernelGGL(k_cdir_init, dim3(1), dim3(1), 0, s, hdr, init);
  if (n > 0) {
    const int bb = (n + kCdirBlock - 1) / kCdirBlock;
    hipLaunchKernelGGL(k_cdir_bbox, dim3(bb < 256 ? bb : 256), dim3(kCdirBlock), 0, s, reinterpret_cast<const float4*>(pos4), n,
                       resolution, hdr);
  }
  hipLaunchKernelGGL(k_cdir_setup, dim3(1), dim3(1), 0, s, hdr, n, (long long)words_cap);
  const int nb = (int)((words_cap + kCdirBlock - 1) / kCdirBlock);
  const int B = (int)buffer_size;
  const int p3mod = (int)(83492791LL % buffer_size);
  const int4* t4 = reinterpret_cast<const int4*>(tab);
  uint2* w2 = reinterpret_cast<uint2*>(words_out);
  hipLaunchKernelGGL(k_cdir_bits, dim3(nb), dim3(kCdirBlock), 0, s, hdr, t4, log2cap, filter, log2filter, B, p3mod, w2, scratch);
  hipLaunchKernelGGL(k_cdir_scan, dim3(1), dim3(1024), 0, s, hdr, scratch, nb, (long long)hits_cap);
  hipLaunchKernelGGL(k_cdir_rows, dim3(nb), dim3(kCdirBlock), 0, s, hdr, t4, reinterpret_cast<const float4*>(tab_pos), log2cap,
                     filter, log2filter, B, p3mod, w2, scratch, reinterpret_cast<float4*>(pos_out));
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}
