// Map maintenance on the device ("next" row N4 of SURVEY.md section 8f): the per-frame steps around the hot path
// that the reference runs as long chains of small torch ops.
//
//   clid_voxel_down_sample   voxel_down_sample_torch (utils/tools.py:639-682): one point per voxel, the one closest
//                            to the voxel centre (distance quantised to 1000 levels, lowest index among equals),
//                            voxels in ascending order of the reference's linear voxel id.  The reference does it
//                            with unique + scatter_reduce(amin) (non-deterministic on the GPU by its own comment);
//                            here: bounding box reduction -> hash-table insert with a 64-bit atomicMin of
//                            (quantised distance, index) per voxel -> compaction -> radix sort of the surviving
//                            voxels only.  Deterministic.
//
// HBM-bound integer work: no LDS tiling, no MFMA; atomics are the scattered kind (one per point).
#include <hipcub/hipcub.hpp>

#include "common.hpp"

namespace clid {

// ---- exclusive prefix sums (hand-written since round 5; hipcub::DeviceScan / BlockScan before) ----------------------------
// Every compaction of this file is flags -> exclusive scan -> scatter over 1e4 .. 1e6 elements (the pool: 4e4 block counts).
// A block of 1024 threads scans a tile of 4096 elements: 4 consecutive elements per thread, an inclusive scan of the thread
// sums across the wave by lane shifts, the 16 wave totals through LDS.  Up to kScanOneMax elements (one tile) that is the whole
// scan -- one launch; beyond that two launches over tiles of 1024, tile totals first, then every block adds up the totals in
// front of its tile (<= a few hundred, cache-resident) and scans it: no look-back chain, no spin, nothing order-dependent.
constexpr int kScanThreads = 1024, kScanTile = 4 * kScanThreads;  // the one-block form (<= kScanOneMax elements)
constexpr int kScanThreadsM = 256, kScanTileM = 4 * kScanThreadsM;  // the two-launch form: small tiles, so that 1e5 elements
                                                                    // already spread over ~100 CUs (both launches are latency-bound)
constexpr long long kScanOneMax = kScanTile;  // one tile: a longer walk of one block over several tiles measured slower than the
                                               // two-launch form (13.8 us for 10 tiles against 4.6 + 5.0 us, tracer attached)
template <class T>
__device__ __forceinline__ T wave_incl_scan(T v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const T o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}
// exclusive prefix of one value per thread over a block of NT threads (NT / 64 wave totals in `ws`); *total = the block's sum
template <class T, int NT>
__device__ __forceinline__ T block_excl_scan(T v, T* ws, T* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const T incl = wave_incl_scan(v, lane);
  if (lane == 63) ws[w] = incl;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) {
    const T x = ws[i];
    base += i < w ? x : (T)0;
    tot += x;
  }
  __syncthreads();  // (ws may be written again)
  *total = tot;
  return base + incl - v;
}
// one tile of 4 NT elements: base + 4 t .. base + 4 t + 3 of thread t; out = carry + exclusive prefix; returns carry + the tile's sum
template <class T, int NT, bool WRITE>
__device__ __forceinline__ T scan_tile(const T* __restrict__ in, T* __restrict__ out, long long base, long long n, T carry, T* ws) {
  const long long i0 = base + 4LL * threadIdx.x;
  T v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = i0 + k < n ? in[i0 + k] : (T)0;
  T tot;
  T ex = block_excl_scan<T, NT>((v[0] + v[1]) + (v[2] + v[3]), ws, &tot) + carry;
  if (WRITE) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i0 + k < n) out[i0 + k] = ex;
      ex += v[k];
    }
  }
  return carry + tot;
}
// (all of a thread's elements are requested before the first tile is scanned: tile by tile behind the barriers of the block scan
// every tile paid its own memory latency -- 26.8 us for the pool's 39 k block counts, 10 tiles)
template <class T>
__global__ void __launch_bounds__(kScanThreads) k_scan_one(const T* __restrict__ in, T* __restrict__ out, long long n) {
  __shared__ T ws[kScanThreads / 64];
  constexpr int kTiles = (int)(kScanOneMax / kScanTile);
  T v[kTiles][4];
#pragma unroll
  for (int t = 0; t < kTiles; ++t) {
    const long long i0 = (long long)t * kScanTile + 4LL * threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[t][k] = i0 + k < n ? in[i0 + k] : (T)0;
  }
  T carry = 0;
#pragma unroll
  for (int t = 0; t < kTiles; ++t) {
    if ((long long)t * kScanTile >= n) break;  // (uniform)
    const long long i0 = (long long)t * kScanTile + 4LL * threadIdx.x;
    T tot;
    T ex = block_excl_scan<T, kScanThreads>((v[t][0] + v[t][1]) + (v[t][2] + v[t][3]), ws, &tot) + carry;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i0 + k < n) out[i0 + k] = ex;
      ex += v[t][k];
    }
    carry += tot;
  }
}
template <class T>
__global__ void __launch_bounds__(kScanThreadsM) k_scan_totals(const T* __restrict__ in, T* __restrict__ totals, long long n) {
  __shared__ T ws[kScanThreadsM / 64];
  const T tot = scan_tile<T, kScanThreadsM, false>(in, nullptr, (long long)blockIdx.x * kScanTileM, n, (T)0, ws);
  if (threadIdx.x == 0) totals[blockIdx.x] = tot;
}
template <class T>
__global__ void __launch_bounds__(kScanThreadsM) k_scan_apply(const T* __restrict__ in, T* __restrict__ out,
                                                              const T* __restrict__ totals, long long n) {
  __shared__ T ws[kScanThreadsM / 64];
  T mine = 0;  // the totals of the tiles in front of this one, summed in a fixed order per thread, then over the block
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += kScanThreadsM) mine += totals[i];
  T carry;
  (void)block_excl_scan<T, kScanThreadsM>(mine, ws, &carry);
  (void)scan_tile<T, kScanThreadsM, true>(in, out, (long long)blockIdx.x * kScanTileM, n, carry, ws);
}
// bytes of `scratch` scan_exclusive needs for n elements (the tile totals of the two-launch form)
#ifndef CLID_SCAN_LIB
#define CLID_SCAN_LIB 0  // 1: hipcub::DeviceScan (the scans of rounds 2-4; A/B)
#endif
static size_t scan_scratch_bytes(long long n) {
  size_t need = ((size_t)((n + kScanTileM - 1) / kScanTileM) + 1) * 8 + 256;
#if CLID_SCAN_LIB
  size_t tmp = 0;
  hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (int)n);
  need = need > tmp + 256 ? need : tmp + 256;
#endif
  return need;
}
// out[i] = in[0] + .. + in[i - 1]; in != out; integer T (exact, order-free)
template <class T>
static void scan_exclusive(const T* in, T* out, long long n, void* scratch, hipStream_t s) {
  if (n <= 0) return;
#if CLID_SCAN_LIB
  size_t bytes = scan_scratch_bytes(n);
  (void)hipcub::DeviceScan::ExclusiveSum(scratch, bytes, in, out, (int)n, s);
  return;
#endif
  if (n <= kScanOneMax && sizeof(T) == 4) {  // (64-bit elements would spill in the one-block form: they take the two-launch form)
    hipLaunchKernelGGL(k_scan_one<T>, dim3(1), dim3(kScanThreads), 0, s, in, out, n);
    return;
  }
  const unsigned nb = (unsigned)((n + kScanTileM - 1) / kScanTileM);
  T* totals = static_cast<T*>(scratch);
  hipLaunchKernelGGL(k_scan_totals<T>, dim3(nb), dim3(kScanThreadsM), 0, s, in, totals, n);
  hipLaunchKernelGGL(k_scan_apply<T>, dim3(nb), dim3(kScanThreadsM), 0, s, in, out, (const T*)totals, n);
}

// the seven reduced values of the bounding-box pass, ONE per 128-byte line (atomics on one line retire one after the other
// whichever word they hit): [lo x, lo y, lo z, hi x, hi y, hi z, dmax] at box[k * kBoxStride]
constexpr int kBoxStride = 32;
struct VoxStats {
  int lo[3];         // min coordinate per axis, order-preserving int encoding of the float
  int hi[3];         // max coordinate per axis
  unsigned dmax;     // max distance to the voxel centre (non-negative float bits order like unsigned ints)
  unsigned count;    // number of occupied voxels (filled by the compaction)
  long long stride;  // v = max over axes of (cell - offset), the reference's linearisation stride
  unsigned overflow; // the bucketed ordering gave up (a bucket beyond its capacity): the caller orders with the library
  unsigned spilled;  // ... or, without a host in the loop (clid_voxel_down_sample_async), entries parked in the spill list
};

__device__ __forceinline__ int ordered(float f) {
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__host__ __device__ __forceinline__ float unordered(int i) {
  const int b = i >= 0 ? i : i ^ 0x7fffffff;
#ifdef __HIP_DEVICE_COMPILE__
  return __int_as_float(b);
#else
  float f;
  memcpy(&f, &b, 4);
  return f;
#endif
}

__device__ __forceinline__ float centre_dist(float x, float y, float z, float v, float* cx, float* cy, float* cz) {
  *cx = floorf(fdiv(x, v)); *cy = floorf(fdiv(y, v)); *cz = floorf(fdiv(z, v));  // tools.py:654
  const float dx = fsub(x, fmul(fadd(*cx, 0.5f), v)), dy = fsub(y, fmul(fadd(*cy, 0.5f), v)),
              dz = fsub(z, fmul(fadd(*cz, 0.5f), v));                               // :655-656
  return sqrtf(fadd(fadd(fmul(dx, dx), fmul(dy, dy)), fmul(dz, dz)));
}

// `value` != NULL: the per-point score of voxel_down_sample_min_value_torch (utils/tools.py:685-724; non-negative)
// takes the place of the distance to the voxel centre.
// n_dev != NULL: the number of points is read on the device (at most n: the grid's bound) -- a caller that has not read the
// count of the pass that produced the points (Mapper.process_frame: the sampler's compaction)
__device__ __forceinline__ int vox_n(int n, const long long* __restrict__ n_dev) {
  if (!n_dev) return n;
  const long long m = *n_dev;
  return m < (long long)n ? (int)m : n;
}
__global__ void __launch_bounds__(256) k_vox_stats(const float* __restrict__ pts, int n_bound, float v, int* __restrict__ box,
                                                   const float* __restrict__ value, const long long* __restrict__ n_dev) {
  const int n = vox_n(n_bound, n_dev);
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  unsigned dm = 0u;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    float cx, cy, cz;
    float d = centre_dist(x, y, z, v, &cx, &cy, &cz);
    if (value) d = value[i];
    const int ox = ordered(x), oy = ordered(y), oz = ordered(z);
    lo[0] = min(lo[0], ox); lo[1] = min(lo[1], oy); lo[2] = min(lo[2], oz);
    hi[0] = max(hi[0], ox); hi[1] = max(hi[1], oy); hi[2] = max(hi[2], oz);
    dm = max(dm, (unsigned)__float_as_int(d));
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      lo[a] = min(lo[a], __shfl_xor(lo[a], o, 64));
      hi[a] = max(hi[a], __shfl_xor(hi[a], o, 64));
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) dm = max(dm, (unsigned)__shfl_xor((int)dm, o, 64));
  // one set of (contended, same-address) atomics per BLOCK: combine the block's waves through LDS first
  __shared__ int part[4][7];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      part[wave][a] = lo[a];
      part[wave][3 + a] = hi[a];
    }
    part[wave][6] = (int)dm;
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    const int c = threadIdx.x;
    int v0 = part[0][c];
    for (int w = 1; w < 4; ++w) {
      const int o = part[w][c];
      v0 = c < 3 ? min(v0, o) : (c < 6 ? max(v0, o) : (int)max((unsigned)v0, (unsigned)o));
    }
    if (c < 6) {
      if (c < 3) atomicMin(&box[c * kBoxStride], v0);
      else atomicMax(&box[c * kBoxStride], v0);
    } else {
      atomicMax(reinterpret_cast<unsigned*>(&box[6 * kBoxStride]), (unsigned)v0);
    }
  }
}

__global__ void k_vox_init(VoxStats* st, int* box) {
  if (threadIdx.x == 0) {
    for (int a = 0; a < 3; ++a) {
      box[a * kBoxStride] = 0x7fffffff;
      box[(3 + a) * kBoxStride] = (int)0x80000000;
    }
    box[6 * kBoxStride] = 0;
    st->count = 0u;
    st->stride = 0;
    st->overflow = 0u;
    st->spilled = 0u;
  }
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33;
  return k;
}

// With a device-side count below the bound the table was sized (and cleared) for, only its first 2^l slots are used, l by
// the host's rule on the live count: the near-surface rows of a frame are a third of the sampler rows the host knows about
// (an 8 MB region of a 32 MB table: the atomics and the partition pass stay closer to the caches).
__device__ __forceinline__ int vox_eff_log2cap(int log2cap, int n_bound, const long long* __restrict__ n_dev) {
  if (!n_dev) return log2cap;
  const int n = vox_n(n_bound, n_dev);
  int l = 10;
  while ((1LL << l) < 2LL * n) ++l;
  return l < log2cap ? l : log2cap;
}

__global__ void __launch_bounds__(256) k_vox_insert(const float* __restrict__ pts, int n_bound, float v, VoxStats* st,
                                                    long long* keys, unsigned long long* vals, int log2cap,
                                                    const float* __restrict__ value, const int* __restrict__ box,
                                                    const long long* __restrict__ n_dev) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < vox_n(n_bound, n_dev);
  // tools.py:653, :661-663: offset = floor(min / v); stride = max(cell - offset) over ALL axes (not max + 1: voxels
  // whose coordinate equals the stride alias another voxel, reproduced on purpose -- it decides which points exist)
  long long off[3], stride = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    off[a] = (long long)floorf(fdiv(unordered(box[a * kBoxStride]), v));
    const long long top = (long long)floorf(fdiv(unordered(box[(3 + a) * kBoxStride]), v)) - off[a];
    stride = top > stride ? top : stride;
  }
  if (i == 0) st->stride = stride;
  long long flat = -1;
  unsigned long long pack = ~0ULL;
  if (live) {
    const float x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    float cx, cy, cz;
    float d = centre_dist(x, y, z, v, &cx, &cy, &cz);
    if (value) d = value[i];
    const float dmax = __int_as_float(box[6 * kBoxStride]);
    const long long q = dmax > 0.f ? (long long)fmul(fdiv(d, dmax), 999.0f) : 0;  // :657-659
    const long long gx = (long long)cx - off[0], gy = (long long)cy - off[1], gz = (long long)cz - off[2];
    flat = gx + gy * stride + gz * stride * stride;
    pack = ((unsigned long long)q << 32) | (unsigned)i;  // min == (smallest q, then smallest index)
  }
  // Neighbouring inputs fall into the same voxel (the samples of one ray around its end point; ~10 inputs per voxel on
  // the map growth's pass) and the table's atomics are what the launch waits for: the lanes of a wave that share a voxel
  // first agree on their minimum, ONE of them goes to the table.  No memory operation inside the loop.
  const int lane = threadIdx.x & 63;
  bool todo = live, rep = false;
  // (worth it only where neighbours do share: a wave with fewer than 16 lanes equal to their successor goes as it is)
  if (__popcll(__ballot(live && flat == __shfl_down(flat, 1, 64) && lane < 63)) < 16) {
    rep = live;
    todo = false;
  }
  for (int round = 0;; ++round) {
    const unsigned long long m = __ballot(todo);
    if (!m) break;
    if (round == 16) {  // a wave of (mostly) distinct voxels: the rest goes as it is, the table's atomicMin sorts it out
      rep = rep || todo;
      break;
    }
    const int leader = __ffsll((long long)m) - 1;
    const long long lk = __shfl(flat, leader, 64);
    const bool same = todo && flat == lk;
    unsigned long long p = same ? pack : ~0ULL;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const unsigned long long q2 = __shfl_xor(p, o, 64);
      p = q2 < p ? q2 : p;
    }
    if (lane == leader) {
      rep = true;
      pack = p;
    }
    todo = todo && !same;
  }
  if (!rep) return;
  const unsigned long long mask = (1ULL << vox_eff_log2cap(log2cap, n_bound, n_dev)) - 1ULL;
  unsigned long long h = mix64((unsigned long long)flat) & mask;
  for (;;) {
    const long long prev = (long long)atomicCAS(reinterpret_cast<unsigned long long*>(&keys[h]), ~0ULL, (unsigned long long)flat);
    if (prev == -1LL || prev == flat) {
      atomicMin(&vals[h], pack);
      return;
    }
    h = (h + 1) & mask;
  }
}

// occupied slots -> dense (flat, index) lists.  The output position comes from ONE counter: same-address atomics
// retire at ~5 ns each, so the bump is aggregated per 1024-thread block (ballot + popcount per wave, the 16 wave
// totals combined in LDS): 1 k atomics for a 1 M-slot table instead of 16 k (per wave) or 60 k (per element).
__global__ void __launch_bounds__(1024) k_vox_compact(const long long* __restrict__ keys,
                                                      const unsigned long long* __restrict__ vals, int log2cap,
                                                      VoxStats* st, long long* flat_out, long long* idx_out) {
  __shared__ unsigned wave_cnt[16];
  __shared__ unsigned block_base;
  const long long h = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long k = h < (1LL << log2cap) ? keys[h] : -1LL;
  const bool live = k != -1LL;
  const unsigned long long m = __ballot(live);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wave] = (unsigned)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned tot = 0;
    for (int w = 0; w < 16; ++w) {
      const unsigned c = wave_cnt[w];
      wave_cnt[w] = tot;  // exclusive prefix
      tot += c;
    }
    block_base = tot ? atomicAdd(&st->count, tot) : 0u;
  }
  __syncthreads();
  if (!live) return;
  const unsigned pos = block_base + wave_cnt[wave] + (unsigned)__popcll(m & ((1ULL << lane) - 1ULL));
  flat_out[pos] = k;
  idx_out[pos] = (long long)(vals[h] & 0xffffffffULL);
}


// ---- ordering of the occupied voxels by their linear id in three small launches (work proportional to m) ------------------
// The library sort of m ~ 1e5 (id, index) pairs is a block sort and seven dependent merge launches (70 - 90 us, twice per
// frame) behind a host round trip (it needs m on the host).  Here: (1) ONE block derives 511 splitters from the ids of
// 1024 sampled input points; (2) the pass over the hash table that used to compact the occupied slots appends every
// (id, index) to the list of its splitter bucket instead (one atomic per voxel on 512 counters); (3) one block per
// bucket orders its list in registers / LDS (bitonic network over packed 64-bit values, lane shuffles for partners inside
// a wave) and writes the indices behind the smaller buckets.  Buckets have a fixed capacity (8 x the mean at 1.3e5
// voxels); a bucket beyond it raises VoxStats.overflow and the host orders with the library as before (first frame,
// whole-map rebuilds).  Everything runs on the device-side count: the one round trip comes after the last launch.
constexpr int kVbBuckets = 512, kVbCap = 4096, kVbThreads = 1024, kVbSamples = 1024;
constexpr int kVbPerBucket = kVbSamples / kVbBuckets;  // samples per bucket
constexpr int kVbPosBits = 12;  // position inside a bucket list
constexpr int kVbCntStride = 32;  // one bucket counter per 128-byte line: atomics on ONE line retire one after the other
                                  // (256 adjacent counters = 8 lines: the partition launch took 93 us for 1e5 voxels)
constexpr unsigned long long kVbPad = ~0ULL;

// bitonic network over the first P (power of two, <= E * 1024) of E * 1024 values, element r * 1024 + tid in register r
template <int E>
__device__ __forceinline__ void vb_bitonic(unsigned long long (&v)[E], unsigned long long* sval, int P) {
  const int tid = threadIdx.x;
  for (int kk = 2; kk <= P; kk <<= 1) {
    for (int j = kk >> 1; j > 0; j >>= 1) {
      if (j >= 64) {
#pragma unroll
        for (int r = 0; r < E; ++r)
          if (r * kVbThreads + tid < P) sval[r * kVbThreads + tid] = v[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < E; ++r) {
          const int i = r * kVbThreads + tid;
          if (i < P) {
            const unsigned long long o = sval[i ^ j];
            const bool keep_min = ((i & j) == 0) == ((i & kk) == 0);
            v[r] = (keep_min == (o < v[r])) ? o : v[r];
          }
        }
        __syncthreads();
      } else if ((tid & ~63) < P) {  // (waves entirely beyond P idle: the network's cost follows P)
#pragma unroll
        for (int r = 0; r < E; ++r) {
          const int i = r * kVbThreads + tid;
          const unsigned long long o = __shfl_xor(v[r], j, 64);
          const bool keep_min = ((i & j) == 0) == ((i & kk) == 0);
          if (i < P) v[r] = (keep_min == (o < v[r])) ? o : v[r];
        }
      }
    }
  }
}

__device__ __forceinline__ long long vox_flat(const float* __restrict__ pts, int i, float v, const int* __restrict__ box) {
  long long off[3], stride = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    off[a] = (long long)floorf(fdiv(unordered(box[a * kBoxStride]), v));
    const long long top = (long long)floorf(fdiv(unordered(box[(3 + a) * kBoxStride]), v)) - off[a];
    stride = top > stride ? top : stride;
  }
  const long long gx = (long long)floorf(fdiv(pts[i * 3 + 0], v)) - off[0], gy = (long long)floorf(fdiv(pts[i * 3 + 1], v)) - off[1],
                  gz = (long long)floorf(fdiv(pts[i * 3 + 2], v)) - off[2];
  return gx + gy * stride + gz * stride * stride;
}

// splitters[k] = sample of rank kVbPerBucket (k + 1), k = 0 .. kVbBuckets - 2, among the ids of 1024 evenly spaced input points.
// Ranks by counting on (id, sample number) composites, 256 samples per block of 256 threads: one 1024-thread block (a bitonic
// network, 11.8 us) needs a whole CU's wave slots at once and starved next to the pool's compaction on the side stream
// (138 us, profiles/r04_frame_trace.txt); small blocks fit wherever one of that launch's blocks leaves.
constexpr int kVsThreads = 256, kVsShift = 10;
static_assert(kVbSamples == (1 << kVsShift) && kVbSamples % kVsThreads == 0, "sample number in the composite's low bits");
__global__ void __launch_bounds__(kVsThreads) k_vox_splitters(const float* __restrict__ pts, int n_bound, float v, const int* __restrict__ box,
                                                              long long* __restrict__ split, unsigned* __restrict__ bucket_cnt,
                                                              const long long* __restrict__ n_dev) {
  __shared__ __attribute__((aligned(16))) unsigned long long sval[kVbSamples];
  const int tid = threadIdx.x;
  const int n = vox_n(n_bound, n_dev);
  if (blockIdx.x == 0)
    for (int i = tid; i < kVbBuckets; i += kVsThreads) bucket_cnt[i * kVbCntStride] = 0u;
  for (int sidx = tid; sidx < kVbSamples; sidx += kVsThreads) {
    const unsigned long long id = n > 0 ? (unsigned long long)vox_flat(pts, (int)((long long)sidx * n / kVbSamples), v, box) : 0ULL;
    sval[sidx] = (id << kVsShift) | (unsigned)sidx;  // (ids beyond 51 bits take the library path / raise: mapops vox_finish)
  }
  __syncthreads();
  const unsigned long long c = sval[blockIdx.x * kVsThreads + tid];
  const ulonglong2* __restrict__ s2 = reinterpret_cast<const ulonglong2*>(sval);
  int rank = 0;
#pragma unroll 8
  for (int j = 0; j < kVbSamples / 2; ++j) {
    const ulonglong2 o = s2[j];
    rank += (o.x < c ? 1 : 0) + (o.y < c ? 1 : 0);
  }
  if (rank % kVbPerBucket == 0 && rank > 0) split[rank / kVbPerBucket - 1] = (long long)(c >> kVsShift);
}

// occupied slots of the hash table -> bucket lists
__global__ void __launch_bounds__(256) k_vox_partition(const long long* __restrict__ keys, const unsigned long long* __restrict__ vals,
                                                       int log2cap, VoxStats* st, const long long* __restrict__ split,
                                                       unsigned* __restrict__ bucket_cnt, long long* __restrict__ bkeys,
                                                       long long* __restrict__ bidx, long long* __restrict__ spill_keys,
                                                       long long* __restrict__ spill_idx, int n_bound,
                                                       const long long* __restrict__ n_dev) {
  __shared__ long long sp[kVbBuckets];
  if ((long long)blockIdx.x * blockDim.x >= (1LL << vox_eff_log2cap(log2cap, n_bound, n_dev))) return;  // (block-uniform)
  for (int i = threadIdx.x; i < kVbBuckets - 1; i += blockDim.x) sp[i] = split[i];
  __syncthreads();
  const long long h = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long k = h < (1LL << log2cap) ? keys[h] : -1LL;
  if (k == -1LL) return;
  int b = 0;  // number of splitters <= k
#pragma unroll
  for (int step = kVbBuckets / 2; step >= 1; step >>= 1)
    if (b + step <= kVbBuckets - 1 && sp[b + step - 1] <= k) b += step;
  const unsigned pos = atomicAdd(&bucket_cnt[b * kVbCntStride], 1u);
  if (pos >= (unsigned)kVbCap) {
    if (spill_keys) {  // no host in the loop: parked for k_vox_bucket_big
      const unsigned q = atomicAdd(&st->spilled, 1u);
      spill_keys[q] = k;
      spill_idx[q] = (long long)(vals[h] & 0xffffffffULL);
    } else {
      st->overflow = 1u;
    }
    return;
  }
  bkeys[(size_t)b * kVbCap + pos] = k;
  bidx[(size_t)b * kVbCap + pos] = (long long)(vals[h] & 0xffffffffULL);
}

template <int E>
__device__ __forceinline__ void vb_emit(int c, long long below, const long long* __restrict__ bk, const long long* __restrict__ bi,
                                        unsigned long long* sval, long long* __restrict__ out) {
  const int tid = threadIdx.x;
  int P = 64;
  while (P < c) P <<= 1;
  unsigned long long v[E];
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int i = r * kVbThreads + tid;
    v[r] = i < c ? (((unsigned long long)bk[i] << kVbPosBits) | (unsigned)i) : kVbPad;
  }
  vb_bitonic<E>(v, sval, P);
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int i = r * kVbThreads + tid;
    if (i < c) out[below + i] = bi[v[r] & ((1ULL << kVbPosBits) - 1ULL)];
  }
}

__global__ void __launch_bounds__(kVbThreads) k_vox_bucket_sort(const unsigned* __restrict__ bucket_cnt, const long long* __restrict__ bkeys,
                                                                const long long* __restrict__ bidx, VoxStats* st,
                                                                long long* __restrict__ out) {
  __shared__ unsigned long long sval[kVbCap];
  __shared__ unsigned wsum[kVbThreads / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  // the number of voxels below this bucket -- and, from the last block, the total (the counters count every arrival, also
  // the ones an overflowing bucket could not store)
  unsigned mine = tid < (b == kVbBuckets - 1 ? kVbBuckets : b) ? bucket_cnt[tid * kVbCntStride] : 0u;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if ((tid & 63) == 0) wsum[tid >> 6] = mine;
  __syncthreads();
  long long below = 0;
  for (int w = 0; w < kVbThreads / 64; ++w) below += wsum[w];
  const int c = (int)bucket_cnt[b * kVbCntStride];
  if (b == kVbBuckets - 1) {
    if (tid == 0) st->count = (unsigned)below;
    below -= c;
  }
  if (st->overflow || c == 0 || c > kVbCap) return;  // (block-uniform; c > capacity only without a host in the loop: k_vox_bucket_big)
  const long long* bk = bkeys + (size_t)b * kVbCap;
  const long long* bi = bidx + (size_t)b * kVbCap;
  if (c <= kVbThreads) vb_emit<1>(c, below, bk, bi, sval, out);
  else if (c <= 2 * kVbThreads) vb_emit<2>(c, below, bk, bi, sval, out);
  else vb_emit<4>(c, below, bk, bi, sval, out);
}

// Without a host in the loop (clid_voxel_down_sample_async) a bucket beyond its capacity cannot fall back to the library's
// sort, which needs the count on the host.  Its entries -- kVbCap stored, the rest in the spill list shared by all buckets --
// are ranked by counting instead (ids are unique): quadratic, but correct for any input, and it never runs on LiDAR frames
// (a bucket holds 1/512 of the voxels on average; beyond capacity = 35 x that at 6e4 voxels).  Always launched: block 0 also
// publishes [number of voxels | ids too wide for the packed sort values] where the consumers' kernels read them.
__global__ void __launch_bounds__(kVbThreads) k_vox_bucket_big(const unsigned* __restrict__ bucket_cnt, const long long* __restrict__ bkeys,
                                                               const long long* __restrict__ bidx, const long long* __restrict__ spill_keys,
                                                               const long long* __restrict__ spill_idx, const long long* __restrict__ split,
                                                               const VoxStats* __restrict__ st, long long* __restrict__ out,
                                                               long long* __restrict__ count_out) {
  __shared__ long long skey[kVbCap];
  __shared__ unsigned wsum[kVbThreads / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b == 0 && tid == 0) {
    const unsigned long long sd = (unsigned long long)st->stride;
    const unsigned long long top = sd * (1ULL + sd + sd * sd);
    int bits = 1;
    while (bits < 63 && (top >> bits)) ++bits;
    // ids too wide for the packed sort values: the index list is unordered.  The consumers size their work by count_out[0] on the
    // device (clid_map_insert / clid_cloud_update: n_dev), so a failed pass publishes ZERO voxels -- nothing is inserted from an
    // unordered list -- and the flag; the host sees it at its next read-back and repeats the step through the library sort
    const bool bad = !(bits <= 64 - kVbPosBits - 1);
    count_out[0] = bad ? 0LL : (long long)st->count;
    count_out[1] = bad ? 1 : 0;
  }
  const int c = (int)bucket_cnt[b * kVbCntStride];
  if (c <= kVbCap) return;  // (block-uniform)
  unsigned mine = tid < b ? bucket_cnt[tid * kVbCntStride] : 0u;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if ((tid & 63) == 0) wsum[tid >> 6] = mine;
  for (int i = tid; i < kVbCap; i += kVbThreads) skey[i] = bkeys[(size_t)b * kVbCap + i];
  __syncthreads();
  long long below = 0;
  for (int w = 0; w < kVbThreads / 64; ++w) below += wsum[w];
  const bool has_lo = b > 0, has_hi = b < kVbBuckets - 1;
  const long long lo = has_lo ? split[b - 1] : 0, hi = has_hi ? split[b] : 0;  // bucket b: split[b-1] <= id < split[b]
  const int S = (int)st->spilled;
  for (int e = tid; e < kVbCap + S; e += kVbThreads) {
    long long k, src;
    if (e < kVbCap) {
      k = skey[e];
      src = bidx[(size_t)b * kVbCap + e];
    } else {
      k = spill_keys[e - kVbCap];
      if ((has_lo && k < lo) || (has_hi && k >= hi)) continue;  // another bucket's
      src = spill_idx[e - kVbCap];
    }
    long long rank = 0;
    for (int i = 0; i < kVbCap; ++i) rank += skey[i] < k ? 1 : 0;
    for (int i = 0; i < S; ++i) {
      const long long o = spill_keys[i];
      rank += (o < k && !(has_lo && o < lo)) ? 1 : 0;  // (o < k < hi already)
    }
    out[below + rank] = src;
  }
}

struct Pose12 {
  float T[12];
};
// transform_torch (utils/tools.py:590-609): y = R x + t in fp32
__global__ void __launch_bounds__(256) k_transform(const float* __restrict__ pts, int n, Pose12 p, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
  out[i * 3 + 0] = fmaf(z, p.T[2], fmaf(y, p.T[1], fmaf(x, p.T[0], p.T[3])));
  out[i * 3 + 1] = fmaf(z, p.T[6], fmaf(y, p.T[5], fmaf(x, p.T[4], p.T[7])));
  out[i * 3 + 2] = fmaf(z, p.T[10], fmaf(y, p.T[9], fmaf(x, p.T[8], p.T[11])));
}

// NeuralPoints.assign_local_to_global (model/neural_points.py:538-549): the three masked assignments in one launch.
// ids [n] = global index of local point i (ascending, == nonzero(local_mask[:-1])); the padding row n of the local
// feature table goes to global row `pad_row` (the reference's mask includes the last, padding, element).
__global__ void __launch_bounds__(256)
k_local_to_global(const long long* __restrict__ ids, int n, long long pad_row, const float4* __restrict__ lfeat,
                  const float* __restrict__ lcert, const int* __restrict__ lts, float4* __restrict__ gfeat,
                  float* __restrict__ gcert, int* __restrict__ gts) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = t >> 1, half = t & 1;
  if (row > n) return;
  const long long dst = row < n ? ids[row] : pad_row;
  gfeat[dst * 2 + half] = lfeat[(size_t)row * 2 + half];
  if (half == 0 && row < n) {
    gcert[dst] = lcert[row];
    gts[dst] = lts[row];
  }
}

// ---- training-pool maintenance (utils/mapper.py:297-392) ---------------------------------------------------------
// The reference concatenates the frame's samples onto five pool arrays, masks the pool by distance to the sensor
// (fp64 under type promotion with the float64 pose), drops random picks above `pool_capacity` and compacts every array
// with a boolean mask: ~25 torch ops and 3 host round trips over 1e7 samples per frame.  Here: flags -> exclusive scan
// -> (capacity drop -> scan) -> one scatter that moves all five arrays, sizes kept in device memory until the caller
// reads the two counts it needs.  The order of the samples is preserved (stable compaction) like the boolean mask does.
struct PoolSrc {
  const float* coord; const float* gcoord; const float* label; const float* weight; const int* time;
  long long n;
  const long long* ndev;  // device, may be NULL: only the first min(*ndev, n) samples exist (n is then the arrays' bound)
};
__device__ __forceinline__ const float* pool_gcoord(const PoolSrc& a, const PoolSrc& b, long long i) {
  return i < a.n ? a.gcoord + i * 3 : b.gcoord + (i - a.n) * 3;
}
// Compaction by per-block counts: the flag pass leaves one byte per sample and one count per 256-sample block; a scan over
// the 40 k block counts (not over 1e7 samples: two 52-us device scans per frame) gives every block its offset and the
// consumers rebuild the position inside the block from ballots.
__device__ __forceinline__ int block_prefix256(bool f, int* total) {  // exclusive prefix of f over the block, 256 threads
  __shared__ int wsum[4];
  const unsigned long long bal = __ballot(f);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wsum[wave] = (int)__popcll(bal);
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    base += w < wave ? wsum[w] : 0;
    tot += wsum[w];
  }
  __syncthreads();
  if (total) *total = tot;
  return base + (int)__popcll(bal & ((1ULL << lane) - 1ULL));
}
__global__ void __launch_bounds__(256)
k_pool_flags(PoolSrc a, PoolSrc b, double ox, double oy, double oz, double r2, unsigned char* __restrict__ flag,
             int* __restrict__ block_cnt) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool f = false;
  long long nb = b.n;
  if (b.ndev) nb = *b.ndev < nb ? *b.ndev : nb;
  if (i >= a.n + nb && i < a.n + b.n) flag[i] = 0;  // rows of the bound beyond the frame's samples: the later passes skip them
  if (i < a.n + nb) {
    const float* g = pool_gcoord(a, b, i);
    const double dx = (double)g[0] - ox, dy = (double)g[1] - oy, dz = (double)g[2] - oz;  // mapper.py:346-349, float64
    f = ((dx * dx + dy * dy) + dz * dz) < r2;
    flag[i] = f ? 1 : 0;
  }
  int tot;
  block_prefix256(f, &tot);
  if (threadIdx.x == 0) block_cnt[blockIdx.x] = tot;
}
// per-block counts of the flags after the capacity drop cleared some: one THREAD per 256-flag block (sixteen 16-byte
// loads; a 256-thread block with two barriers per 256 bytes took 17 us for 1e7 flags, this takes 10)
__global__ void __launch_bounds__(256)
k_pool_count(const unsigned char* __restrict__ flag, long long n, int* __restrict__ block_cnt) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long i0 = b * 256;
  if (i0 >= n) return;
  int tot = 0;
  if (i0 + 256 <= n) {
    const uint4* f = reinterpret_cast<const uint4*>(flag + i0);
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const uint4 v = f[w];  // (flag bytes are 0 or 1)
      tot += __popc(v.x & 0x01010101u) + __popc(v.y & 0x01010101u) + __popc(v.z & 0x01010101u) + __popc(v.w & 0x01010101u);
    }
  } else {
    for (long long i = i0; i < n; ++i) tot += flag[i] ? 1 : 0;
  }
  block_cnt[b] = tot;
}
// kept_list[rank] = index for the samples that passed the window test; counts[2] = kept (before the capacity drop)
__global__ void __launch_bounds__(256)
k_pool_list(const unsigned char* __restrict__ flag, const int* __restrict__ block_off, const int* __restrict__ block_cnt,
            long long n, int* __restrict__ kept_list, long long* __restrict__ counts) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool f = i < n && flag[i];
  const int r = block_prefix256(f, nullptr);
  if (f) kept_list[block_off[blockIdx.x] + r] = (int)i;
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) counts[2] = (long long)block_off[blockIdx.x] + block_cnt[blockIdx.x];
}
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}
// mapper.py:352-361: `kept - capacity` uniform picks WITH replacement among the kept samples are dropped
__global__ void __launch_bounds__(256)
k_pool_drop(unsigned char* __restrict__ flag, const int* __restrict__ kept_list, const long long* __restrict__ counts,
            long long capacity, unsigned long long seed) {
  const long long kept = counts[2];
  const long long excess = kept - capacity;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < excess; t += (long long)gridDim.x * blockDim.x) {
    const unsigned long long r = splitmix64(seed + (unsigned long long)t);
    flag[kept_list[(long long)(r % (unsigned long long)kept)]] = 0;
  }
}
struct PoolDst {
  float* coord; float* gcoord; float* label; float* weight; int* time;
};
// One block per 256 consecutive samples.  (A bounded, grid-stride launch -- 2 / 4 / 6 blocks per CU, to leave wave slots to the
// map growth's small launches on the main stream -- was measured: process_frame 1.10-1.17 / 1.01-1.08 / 0.95-0.96 ms against
// 0.94-0.95: this launch is bandwidth-bound, wants every slot, and the pool chain it ends is the frame's critical path.)
__global__ void __launch_bounds__(256)
k_pool_scatter(PoolSrc a, PoolSrc b, const unsigned char* __restrict__ flag, const int* __restrict__ block_off,
               const int* __restrict__ block_cnt, PoolDst d, long long* __restrict__ counts) {
  const long long n = a.n + b.n, n_vblocks = gridDim.x;
  {
    const long long vb = blockIdx.x;
    const long long i = vb * blockDim.x + threadIdx.x;
    const bool f = i < n && flag[i];
    const long long j = (long long)block_off[vb] + block_prefix256(f, nullptr);
    if (vb == n_vblocks - 1 && threadIdx.x == 0) {
      const long long kept = (long long)block_off[vb] + block_cnt[vb];
      counts[0] = kept;  // samples kept in total
      // ... of which from this frame (the tail): kept minus what survived of the old pool (positions < a.n)
      long long kept_old = 0;
      if (a.n > 0) {
        const long long ba = a.n / 256;
        kept_old = ba < n_vblocks ? (long long)block_off[ba] : kept;
        for (long long q = ba * 256; q < a.n && ba < n_vblocks; ++q) kept_old += flag[q];
      }
      counts[1] = kept - kept_old;
    }
    // coordinates: 3 floats per sample.  A block whose 256 samples come from ONE source at a 16-byte aligned offset stages the
    // two 3 KB tiles through LDS -- contiguous 16-byte loads in, the kept samples' floats compacted, contiguous stores out --
    // instead of six stride-12 loads and six stride-12 stores per thread
    __shared__ __attribute__((aligned(16))) float tile[2][768];
    __shared__ float packed[2][768];
    const long long i0 = vb * blockDim.x;
    const bool one_src = i0 + 256 <= a.n || i0 >= a.n;
    const PoolSrc& sb = i0 < a.n ? a : b;
    const long long k0 = i0 < a.n ? i0 : i0 - a.n;
    const bool full = i0 + 256 <= n;
    const bool fast = one_src && full && (((uintptr_t)(sb.coord + 3 * k0) | (uintptr_t)(sb.gcoord + 3 * k0)) & 15) == 0;  // (block-uniform)
    if (fast) {
      if (threadIdx.x < 192) {
        reinterpret_cast<float4*>(tile[0])[threadIdx.x] = reinterpret_cast<const float4*>(sb.coord + 3 * k0)[threadIdx.x];
        reinterpret_cast<float4*>(tile[1])[threadIdx.x] = reinterpret_cast<const float4*>(sb.gcoord + 3 * k0)[threadIdx.x];
      }
      __syncthreads();
      const int jl = (int)(j - block_off[vb]);  // position among the block's kept samples
      if (f) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          packed[0][3 * jl + c] = tile[0][3 * threadIdx.x + c];
          packed[1][3 * jl + c] = tile[1][3 * threadIdx.x + c];
        }
      }
      __syncthreads();
      const int n3 = 3 * block_cnt[vb];
      float* __restrict__ oc = d.coord + 3 * (long long)block_off[vb];
      float* __restrict__ og = d.gcoord + 3 * (long long)block_off[vb];
      for (int x = threadIdx.x; x < n3; x += 256) {
        oc[x] = packed[0][x];
        og[x] = packed[1][x];
      }
      if (!f) return;
      d.label[j] = sb.label[k0 + threadIdx.x];
      d.weight[j] = sb.weight[k0 + threadIdx.x];
      d.time[j] = sb.time[k0 + threadIdx.x];
      return;
    }
    if (!f) return;
    const bool old = i < a.n;
    const PoolSrc& s = old ? a : b;
    const long long k = old ? i : i - a.n;
    d.coord[j * 3 + 0] = s.coord[k * 3 + 0]; d.coord[j * 3 + 1] = s.coord[k * 3 + 1]; d.coord[j * 3 + 2] = s.coord[k * 3 + 2];
    d.gcoord[j * 3 + 0] = s.gcoord[k * 3 + 0]; d.gcoord[j * 3 + 1] = s.gcoord[k * 3 + 1]; d.gcoord[j * 3 + 2] = s.gcoord[k * 3 + 2];
    d.label[j] = s.label[k];
    d.weight[j] = s.weight[k];
    d.time[j] = s.time[k];
  }
}

// ---- local-window selection (model/neural_points.py:439-536) -------------------------------------------------------
// reset_local_map: travel-distance (or frame-count) window AND distance to the sensor select the trainable local map;
// the reference then builds global2local / local_mask and gathers six local arrays with ~40 torch ops and two host round
// trips.  Here: flags (+ count of the time window) -> scan -> one gather pass; the caller reads ONE count.
constexpr unsigned kFlagBlocks = 512;  // blocks of the flag kernels that end in a same-address atomic
struct WindowArgs {
  const float* points; const int* ts_create; const int* ts_update; const float* travel; long long n;
  const long long* n_extra;  // device, may be NULL: the map holds n + *n_extra points (the count of an insert still in flight)
  int cur_ts, use_mid_ts, temporal, use_travel, diff_ts_local, reboot_ts, reboot_map;
  float diff_travel;
  double sx, sy, sz, r2;
  int pos_f64;  // the sensor position is float64 (pose dtype): the distance test then runs in float64 by type promotion
};
__device__ __forceinline__ long long window_n(const WindowArgs& a) { return a.n + (a.n_extra ? *a.n_extra : 0); }
__global__ void __launch_bounds__(256) k_window_flags(WindowArgs a, unsigned char* __restrict__ bits, long long* __restrict__ counts) {
  const long long n = window_n(a);
  int mine = 0;
  // grid-stride over at most kFlagBlocks blocks: the launch ends in ONE same-address atomic per block, and those retire
  // one after the other (2.7 k blocks at 690 k points: 27 us of a 5 us kernel; per wave it was 56 us)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    bool t_ok = true;
    bool d_ok = false;
    if (a.temporal) {
      int ts = a.ts_create[i];
      if (a.use_mid_ts) ts = (int)(((float)a.ts_create[i] + (float)a.ts_update[i]) / 2.0f);  // ((a + b) / 2).int(), :449
      if (a.use_travel) t_ok = fabsf(fsub(a.travel[a.cur_ts], a.travel[ts])) < a.diff_travel;  // :452-455
      else t_ok = abs(a.cur_ts - ts) < a.diff_ts_local;
      if (a.reboot_map) t_ok = t_ok && ts >= a.reboot_ts;
    }
    if (a.pos_f64) {
      const double dx = (double)a.points[i * 3 + 0] - a.sx, dy = (double)a.points[i * 3 + 1] - a.sy, dz = (double)a.points[i * 3 + 2] - a.sz;
      d_ok = ((dx * dx + dy * dy) + dz * dz) < a.r2;  // :474-478
    } else {
      const float dx = fsub(a.points[i * 3 + 0], (float)a.sx), dy = fsub(a.points[i * 3 + 1], (float)a.sy),
                  dz = fsub(a.points[i * 3 + 2], (float)a.sz);
      d_ok = fadd(fadd(fmul(dx, dx), fmul(dy, dy)), fmul(dz, dz)) < (float)a.r2;
    }
    bits[i] = (unsigned char)((t_ok ? 1 : 0) | (d_ok ? 2 : 0));
    mine += t_ok ? 1 : 0;
  }
  __shared__ int wave_cnt[4];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    if (tot) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[0]), (unsigned long long)tot);
  }
}
// fewer than 100 points inside the time window -> the window is dropped (:462-466)
__global__ void __launch_bounds__(256) k_window_combine(WindowArgs a, const unsigned char* __restrict__ bits, long long n_upper,
                                                        const long long* __restrict__ counts, int* __restrict__ flag) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_upper) return;
  if (i >= window_n(a)) {  // rows of the capacity beyond the map: nothing there (the scan runs over the upper bound)
    flag[i] = 0;
    return;
  }
  const bool use_time = a.temporal && counts[0] >= 100;
  flag[i] = ((bits[i] & 2) && (!use_time || (bits[i] & 1))) ? 1 : 0;
}
struct WindowOut {
  long long cap;  // rows the local arrays can hold (local_ids, points, ..., features cap + 1): rows beyond it are not written
  long long* local_ids; long long* g2l; unsigned char* local_mask;
  float* l_points; float* l_orient; float* l_cert; int* l_ts; float* l_feat;
  const float* g_orient; const float* g_cert; const float* g_feat;
};
__global__ void __launch_bounds__(256) k_window_gather(WindowArgs a, const int* __restrict__ flag, const int* __restrict__ pos,
                                                       WindowOut o, long long* __restrict__ counts) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = window_n(a);
  if (i > n) return;
  if (i == n) {  // the padding element: always part of the mask, never of the map (:518-530)
    const long long m = n > 0 ? pos[n - 1] + flag[n - 1] : 0;
    counts[1] = m;
    o.g2l[i] = -1;
    o.local_mask[i] = 1;
    if (m <= o.cap)
      for (int c = 0; c < CLID_F; ++c) o.l_feat[m * CLID_F + c] = o.g_feat[n * CLID_F + c];
    return;
  }
  const bool in = flag[i] != 0;
  o.local_mask[i] = in ? 1 : 0;
  o.g2l[i] = in ? (long long)pos[i] : -1;
  if (!in) return;
  const long long j = pos[i];
  if (j >= o.cap) return;  // (the caller sees m > capacity in counts[1] and repeats the call with room for m rows)
  o.local_ids[j] = i;
  o.l_points[j * 3 + 0] = a.points[i * 3 + 0]; o.l_points[j * 3 + 1] = a.points[i * 3 + 1]; o.l_points[j * 3 + 2] = a.points[i * 3 + 2];
  reinterpret_cast<float4*>(o.l_orient)[j] = reinterpret_cast<const float4*>(o.g_orient)[i];
  o.l_cert[j] = o.g_cert[i];
  o.l_ts[j] = a.ts_update[i];
  reinterpret_cast<float4*>(o.l_feat)[j * 2] = reinterpret_cast<const float4*>(o.g_feat)[i * 2];
  reinterpret_cast<float4*>(o.l_feat)[j * 2 + 1] = reinterpret_cast<const float4*>(o.g_feat)[i * 2 + 1];
}

// ---- neural-point insertion (model/neural_points.py:324-437) --------------------------------------------------------
// NeuralPoints.update after the voxel down-sampling: per sample, the slot of its voxel in the reference's own table, the
// point held there, the take test (empty slot | held point farther than sqrt(3) voxels | held point stale by travelled
// distance), ranks of the taken samples, the table update with the reference's sequential semantics (among several samples
// naming one slot the LAST one wins, whether it was taken or not) and the append of positions / orientations / stamps /
// certainties -- ~45 torch ops incl. an argsort in the reference-style chain, four launches + a scan here.
struct InsertArgs {
  const float* samples; int n;            // voxel-down-sampled points [n][3] ...
  const long long* s_idx;                 // ... or, with a list: sample i = row s_idx[i] of `samples`,
  const long long* n_dev;                 //     i < *n_dev (n is then the upper bound the grids are sized for)
  long long* table; int buffer_size;      // buffer_pt_index
  float* points; float* orient; int* ts_create; int* ts_update; float* cert;  // global arrays with room for n more rows
  float4* feat;                           // optional: feature rows of the added points and the padding row behind them <- 0
  const float* travel;                    // travel_dist or NULL
  long long base;                         // points in the map before the insert
  int test_on;                            // 0: empty map / reboot frame -> every sample is taken (:370-371)
  int temporal;
  int cur_ts;
  float res, far2, diff_travel;
};
constexpr long long kClaimBase = 1LL << 40;  // above any point index: marks a slot as claimed by sample (value - base)
__global__ void __launch_bounds__(256) k_insert_probe(InsertArgs a, int* __restrict__ phys, long long* __restrict__ held,
                                                      int* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  if (a.n_dev && i >= *a.n_dev) {  // beyond the device-side count: takes part in the scan as "not taken"
    flag[i] = 0;
    return;
  }
  const long long r = a.s_idx ? a.s_idx[i] : (long long)i;
  const float x = a.samples[r * 3 + 0], y = a.samples[r * 3 + 1], z = a.samples[r * 3 + 2];
  const int slot = base_slot(x, y, z, a.res, a.buffer_size);  // == fmod(sum cell*prime, B) taken non-negative (:355-361)
  const long long h = a.table[slot];
  bool take = true;
  if (a.test_on) {
    take = h == -1;
    if (!take) {
      const float dx = fsub(a.points[h * 3 + 0], x), dy = fsub(a.points[h * 3 + 1], y), dz = fsub(a.points[h * 3 + 2], z);
      take = fadd(fadd(fmul(dx, dx), fmul(dy, dy)), fmul(dz, dz)) > a.far2;                       // :373-377
      if (!take && a.temporal) take = fsub(a.travel[a.cur_ts], a.travel[a.ts_update[h]]) > a.diff_travel;  // :379-385
    }
  }
  phys[i] = slot;
  held[i] = h;
  flag[i] = take ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_insert_claim(InsertArgs a, const int* __restrict__ phys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n || (a.n_dev && i >= *a.n_dev)) return;
  atomicMax(reinterpret_cast<long long*>(&a.table[phys[i]]), kClaimBase + i);
}
__global__ void __launch_bounds__(256) k_insert_commit(InsertArgs a, const int* __restrict__ phys, const long long* __restrict__ held,
                                                       const int* __restrict__ flag, const int* __restrict__ pos,
                                                       long long* __restrict__ counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  if (i == a.n - 1) {
    counts[0] = pos[i] + flag[i];
    if (a.feat) {  // the padding row moves behind the last added point (zero-initialised features: geo_feature_std == 0)
      const long long pad = a.base + pos[i] + flag[i];
      a.feat[pad * 2] = make_float4(0.f, 0.f, 0.f, 0.f);
      a.feat[pad * 2 + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (a.n_dev && i >= *a.n_dev) return;
  const bool take = flag[i] != 0;
  const long long dst = a.base + pos[i];
  if (a.table[phys[i]] == kClaimBase + i) a.table[phys[i]] = take ? dst : held[i];  // the last sample naming this slot decides it
  if (!take) return;
  const long long r = a.s_idx ? a.s_idx[i] : (long long)i;
  a.points[dst * 3 + 0] = a.samples[r * 3 + 0]; a.points[dst * 3 + 1] = a.samples[r * 3 + 1]; a.points[dst * 3 + 2] = a.samples[r * 3 + 2];
  reinterpret_cast<float4*>(a.orient)[dst] = make_float4(1.f, 0.f, 0.f, 0.f);
  a.ts_create[dst] = a.cur_ts;
  a.ts_update[dst] = a.cur_ts;
  a.cert[dst] = 0.f;
  if (a.feat) {
    a.feat[dst * 2] = make_float4(0.f, 0.f, 0.f, 0.f);
    a.feat[dst * 2 + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// ---- raw-point map maintenance (model/local_point_cloud_map.py:43-72) -------------------------------------------------
// LocalPointCloudMap.update_map: scan points whose voxel slot is still empty are appended, the map is cropped to
// `map_size` around the sensor and the slot table is rebuilt from scratch (every frame).  Here: probe -> flags over
// [existing | samples] -> scan -> compaction into the other half of a ping-pong buffer -> table fill + amax scatter.
__device__ __forceinline__ int cloud_slot(float x, float y, float z, float res, int B) {  // :38-41, ITS OWN middle prime
  const double cx = (double)floorf(fdiv(x, res)), cy = (double)floorf(fdiv(y, res)), cz = (double)floorf(fdiv(z, res));
  const double h = fma(cx, 73856093.0, fma(cy, 19349663.0, cz * 83492791.0));
  const double Bd = (double)B;
  const double q = floor(h / Bd);
  double r = fma(-q, Bd, h);
  if (r < 0.0) r += Bd;
  if (r >= Bd) r -= Bd;
  return (int)r;
}
struct CloudArgs {
  const float* old_pts; long long n_a;   // map before the update
  const float* samples; long long n_s;   // voxel-down-sampled scan points (world frame) ...
  const long long* s_idx;                // ... or, with a list: sample i = row s_idx[i] of `samples`,
  const long long* n_s_dev;              //     i < *n_s_dev (n_s is then the upper bound the grids are sized for)
  const long long* table_old; long long* table_new; int buffer_size;
  float res;
  double sx, sy, sz, map_size;
  int pos_f64;
};
__device__ __forceinline__ bool cloud_near(const CloudArgs& a, const float* p) {
  if (a.pos_f64) {
    const double dx = (double)p[0] - a.sx, dy = (double)p[1] - a.sy, dz = (double)p[2] - a.sz;
    return sqrt((dx * dx + dy * dy) + dz * dz) < a.map_size;  // torch.norm(...) < map_size in float64 (:66)
  }
  const float dx = fsub(p[0], (float)a.sx), dy = fsub(p[1], (float)a.sy), dz = fsub(p[2], (float)a.sz);
  return sqrtf(fadd(fadd(fmul(dx, dx), fmul(dy, dy)), fmul(dz, dz))) < (float)a.map_size;
}
__global__ void __launch_bounds__(256) k_cloud_flags(CloudArgs a, int* __restrict__ flag, long long* __restrict__ counts) {
  const long long n = a.n_a + a.n_s;
  int mine = 0;  // (grid-stride, one same-address atomic per block: see k_window_flags)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    bool fresh = false, keep = false;
    if (i < a.n_a) {
      keep = cloud_near(a, a.old_pts + i * 3);
    } else if (!a.n_s_dev || i - a.n_a < *a.n_s_dev) {
      const long long r = i - a.n_a;
      const float* p = a.samples + (a.s_idx ? a.s_idx[r] : r) * 3;
      fresh = a.table_old[cloud_slot(p[0], p[1], p[2], a.res, a.buffer_size)] == -1;  // :49-56: slot still empty
      keep = fresh && cloud_near(a, p);
    }
    flag[i] = keep ? 1 : 0;
    mine += fresh ? 1 : 0;
  }
  __shared__ int wave_cnt[4];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    if (tot) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[1]), (unsigned long long)tot);
  }
}
__global__ void __launch_bounds__(256) k_cloud_scatter(CloudArgs a, const int* __restrict__ flag, const int* __restrict__ pos,
                                                       float* __restrict__ out, long long* __restrict__ counts) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = a.n_a + a.n_s;
  if (i >= n) return;
  if (i == n - 1) counts[0] = pos[i] + flag[i];
  if (!flag[i]) return;
  const float* p = i < a.n_a ? a.old_pts + i * 3 : a.samples + (a.s_idx ? a.s_idx[i - a.n_a] : i - a.n_a) * 3;
  const long long j = pos[i];
  out[j * 3 + 0] = p[0]; out[j * 3 + 1] = p[1]; out[j * 3 + 2] = p[2];
  // the rebuilt table: several points may share a slot, the largest index stays (:69-71, amax == last writer)
  atomicMax(&a.table_new[cloud_slot(p[0], p[1], p[2], a.res, a.buffer_size)], j);
}


// ---- NeuralPoints.recreate_hash / prune_map (model/neural_points.py:771-812, 840-929) ---------------------------------
// Table fill of recreate_hash (:886-892 keeping, :918-925 merging): buffer_pt_index[slot(points[src])] = src for the
// list entries p = 0..m-1 (src = idx[p], or p without a list).  Several entries may name one slot; the reference's
// indexed assignment keeps the LAST one on the CPU.  Two launches: every entry bids (p + 1, src) with an atomicMax,
// then the winner replaces its bid by src (a loser sees either the winning bid or a bare index, whose upper half is 0).
__global__ void __launch_bounds__(256) k_rehash_claim(const float* __restrict__ pts, const long long* __restrict__ idx, int m,
                                                      float res, long long* __restrict__ table, int B) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m) return;
  const long long src = idx ? idx[p] : (long long)p;
  const int slot = base_slot(pts[src * 3 + 0], pts[src * 3 + 1], pts[src * 3 + 2], res, B);
  atomicMax(&table[slot], ((long long)(p + 1) << 32) | src);
}
__global__ void __launch_bounds__(256) k_rehash_commit(const float* __restrict__ pts, const long long* __restrict__ idx, int m,
                                                       float res, long long* __restrict__ table, int B) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m) return;
  const long long src = idx ? idx[p] : (long long)p;
  const int slot = base_slot(pts[src * 3 + 0], pts[src * 3 + 1], pts[src * 3 + 2], res, B);
  if ((table[slot] >> 32) == (long long)(p + 1)) table[slot] = src;
}

// rows idx[0..m) of the global arrays into fresh arrays (prune_map :795-808, the merging branch of recreate_hash
// :898-913); feature row m (the padding row) comes from row `pad_src`.  One thread per (row, 16-byte half of the feature row).
struct MapRows {
  const float* points; const float4* orient; const int* ts_create; const int* ts_update; const float* cert; const float4* feat;
};
struct MapRowsOut {
  float* points; float4* orient; int* ts_create; int* ts_update; float* cert; float4* feat;
};
__global__ void __launch_bounds__(256) k_map_gather(const long long* __restrict__ idx, int m, long long pad_src, MapRows a,
                                                    MapRowsOut o) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int p = t >> 1, half = t & 1;
  if (p > m) return;
  const long long src = p < m ? idx[p] : pad_src;
  o.feat[(long long)p * 2 + half] = a.feat[src * 2 + half];
  if (p == m) return;
  if (half == 0) {
    o.points[(long long)p * 3 + 0] = a.points[src * 3 + 0];
    o.points[(long long)p * 3 + 1] = a.points[src * 3 + 1];
    o.points[(long long)p * 3 + 2] = a.points[src * 3 + 2];
    o.cert[p] = a.cert[src];
  } else {
    o.orient[p] = a.orient[src];
    o.ts_create[p] = a.ts_create[src];
    o.ts_update[p] = a.ts_update[src];
  }
}

// prune_map :779-789: a point goes when it is uncertain and (unless `global`) has left the travel-distance window
__global__ void __launch_bounds__(256) k_prune_flags(const int* __restrict__ ts_update, const float* __restrict__ cert, int n,
                                                     const float* __restrict__ travel, int cur_ts, float thre, float diff_travel,
                                                     int global, int* __restrict__ keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool prune = cert[i] < thre;
  if (prune && !global) prune = fabsf(fsub(travel[cur_ts], travel[ts_update[i]])) > diff_travel;
  keep[i] = prune ? 0 : 1;
}
static int vox_log2cap(int n) {
  int l = 10;
  while ((1LL << l) < 2LL * n) ++l;
  return l;
}

}  // namespace clid

using namespace clid;

static size_t align256(size_t b) { return (b + 255) & ~size_t(255); }

struct VoxLayout {
  size_t stats, box, keys, vals, flat_a, idx_a, flat_b, cub, split, bcnt, bkeys, bidx, total, cub_bytes;
};

static VoxLayout vox_layout(int n) {
  VoxLayout L;
  const size_t cap = (size_t)1 << vox_log2cap(n);
  size_t o = 0;
  L.stats = o; o += align256(sizeof(VoxStats));
  L.box = o; o += align256(7 * kBoxStride * sizeof(int));
  L.keys = o; o += align256(cap * 8);
  L.vals = o; o += align256(cap * 8);
  L.flat_a = o; o += align256((size_t)n * 8);
  L.idx_a = o; o += align256((size_t)n * 8);
  L.flat_b = o; o += align256((size_t)n * 8);
  size_t tmp = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, (const long long*)nullptr, (long long*)nullptr, (const long long*)nullptr,
                                     (long long*)nullptr, n);
  L.cub_bytes = tmp;
  L.cub = o; o += align256(tmp);
  L.split = o; o += align256(kVbBuckets * 8);
  L.bcnt = o; o += align256((size_t)kVbBuckets * kVbCntStride * 4);
  L.bkeys = o; o += align256((size_t)kVbBuckets * kVbCap * 8);
  L.bidx = o; o += align256((size_t)kVbBuckets * kVbCap * 8);
  L.total = o;
  return L;
}

extern "C" int64_t clid_voxel_workspace_bytes(int32_t n) {
  if (n <= 0) return 256;
  return (int64_t)vox_layout(n).total;
}

static int vox_launch(const float* points, int32_t n, float voxel_size, const float* value, void* workspace,
                      int64_t* idx_out, void* stream, const int64_t* n_dev_in = nullptr, int64_t* count_out_dev = nullptr) {
  const long long* n_dev = reinterpret_cast<const long long*>(n_dev_in);
  if (n < 0 || !(voxel_size > 0.f) || (n > 0 && (!points || !workspace || !idx_out))) {
    clid_set_error("clid_voxel_down_sample: bad argument");
    return CLID_E_ARG;
  }
  if (n == 0) return CLID_OK;
  hipStream_t s = (hipStream_t)stream;
  const VoxLayout L = vox_layout(n);
  char* ws = static_cast<char*>(workspace);
  VoxStats* st = reinterpret_cast<VoxStats*>(ws + L.stats);
  long long* keys = reinterpret_cast<long long*>(ws + L.keys);
  unsigned long long* vals = reinterpret_cast<unsigned long long*>(ws + L.vals);
  long long* flat_a = reinterpret_cast<long long*>(ws + L.flat_a);
  long long* idx_a = reinterpret_cast<long long*>(ws + L.idx_a);
  long long* flat_b = reinterpret_cast<long long*>(ws + L.flat_b);
  const int log2cap = vox_log2cap(n);
  if (hipMemsetAsync(keys, 0xFF, ((size_t)1 << log2cap) * 16, s) != hipSuccess) {  // keys and vals are adjacent
    clid_set_error("clid_voxel_down_sample: workspace init failed");
    return CLID_E_HIP;
  }
  int* box = reinterpret_cast<int*>(ws + L.box);
  hipLaunchKernelGGL(k_vox_init, dim3(1), dim3(64), 0, s, st, box);
  int sb = (n + 255) / 256;
  if (sb > 256) sb = 256;  // every block ends in 7 atomics, one per cache line (on ONE line 512 blocks spent 15 of 18 us there)
  hipLaunchKernelGGL(k_vox_stats, dim3(sb), dim3(256), 0, s, points, n, voxel_size, box, value, n_dev);
  hipLaunchKernelGGL(k_vox_insert, dim3((n + 255) / 256), dim3(256), 0, s, points, n, voxel_size, st, keys, vals, log2cap,
                     value, box, n_dev);
  long long* split = reinterpret_cast<long long*>(ws + L.split);
  unsigned* bcnt = reinterpret_cast<unsigned*>(ws + L.bcnt);
  long long* bkeys = reinterpret_cast<long long*>(ws + L.bkeys);
  long long* bidx = reinterpret_cast<long long*>(ws + L.bidx);
  const unsigned table_blocks1k = (unsigned)((((size_t)1 << log2cap) + 1023) / 1024);
  const bool bucketed = n <= (1 << 21);  // beyond: the buckets would overflow anyway
  if (bucketed) {
    hipLaunchKernelGGL(k_vox_splitters, dim3(kVbSamples / kVsThreads), dim3(kVsThreads), 0, s, points, n, voxel_size, box, split, bcnt, n_dev);
    // count_out_dev: nobody will come back for the count (or for an overflowing bucket): spill list + k_vox_bucket_big
    hipLaunchKernelGGL(k_vox_partition, dim3(table_blocks1k * 4), dim3(256), 0, s, keys, vals, log2cap, st, split, bcnt, bkeys, bidx,
                       count_out_dev ? flat_a : nullptr, count_out_dev ? idx_a : nullptr, n, n_dev);
    hipLaunchKernelGGL(k_vox_bucket_sort, dim3(kVbBuckets), dim3(kVbThreads), 0, s, bcnt, bkeys, bidx, st,
                       reinterpret_cast<long long*>(idx_out));
    if (count_out_dev)
      hipLaunchKernelGGL(k_vox_bucket_big, dim3(kVbBuckets), dim3(kVbThreads), 0, s, bcnt, bkeys, bidx, flat_a, idx_a, split, st,
                         reinterpret_cast<long long*>(idx_out), reinterpret_cast<long long*>(count_out_dev));
  } else {
    hipLaunchKernelGGL(k_vox_compact, dim3(table_blocks1k), dim3(1024), 0, s, keys, vals, log2cap, st, flat_a, idx_a);
  }
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

// second half: the one round trip (sizes the caller's tensors) and, where the bucketed ordering gave up, the library sort
static int vox_finish(int32_t n, void* workspace, int64_t* idx_out, void* stream) {
  if (n < 0 || (n > 0 && (!workspace || !idx_out))) {
    clid_set_error("clid_voxel_down_sample: bad argument");
    return CLID_E_ARG;
  }
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const VoxLayout L = vox_layout(n);
  char* ws = static_cast<char*>(workspace);
  VoxStats* st = reinterpret_cast<VoxStats*>(ws + L.stats);
  long long* keys = reinterpret_cast<long long*>(ws + L.keys);
  unsigned long long* vals = reinterpret_cast<unsigned long long*>(ws + L.vals);
  long long* flat_a = reinterpret_cast<long long*>(ws + L.flat_a);
  long long* idx_a = reinterpret_cast<long long*>(ws + L.idx_a);
  long long* flat_b = reinterpret_cast<long long*>(ws + L.flat_b);
  const int log2cap = vox_log2cap(n);
  const unsigned table_blocks1k = (unsigned)((((size_t)1 << log2cap) + 1023) / 1024);
  const bool bucketed = n <= (1 << 21);
  // the output size is data dependent: ONE host round trip (the caller sizes its tensors with it), through the library's
  // pinned landing buffer (a pageable destination makes hipMemcpyAsync stage and block for ~150 us)
  VoxStats got;
  if (int rc = clid_read_back(st, (int32_t)sizeof(VoxStats), &got, stream)) return rc;  // (polls an event: no interrupt wake-up)
  const int m = (int)got.count;
  int bits = 1;
  {
    const unsigned long long top = (unsigned long long)got.stride * (1ULL + (unsigned long long)got.stride +
                                                                     (unsigned long long)got.stride * got.stride);
    while (bits < 63 && (top >> bits)) ++bits;
  }
  if (bucketed) {
    if (!got.overflow && bits <= 64 - kVbPosBits - 1) return m;  // ordered on the device already
    // a bucket beyond its capacity (or ids too wide for the packed sort values): the library orders the compacted slots
    if (hipMemsetAsync(&st->count, 0, sizeof(unsigned), s) != hipSuccess) return CLID_E_HIP;
    hipLaunchKernelGGL(k_vox_compact, dim3(table_blocks1k), dim3(1024), 0, s, keys, vals, log2cap, st, flat_a, idx_a);
    CLID_CHECK_LAUNCH();
  }
  size_t tmp = L.cub_bytes;
  if (hipcub::DeviceRadixSort::SortPairs(ws + L.cub, tmp, flat_a, flat_b, idx_a, reinterpret_cast<long long*>(idx_out), m,
                                         0, bits, s) != hipSuccess) {
    clid_set_error("clid_voxel_down_sample: sort failed");
    return CLID_E_HIP;
  }
  return m;
}


static int vox_down_sample(const float* points, int32_t n, float voxel_size, const float* value, void* workspace,
                           int64_t* idx_out, void* stream) {
  if (int e = vox_launch(points, n, voxel_size, value, workspace, idx_out, stream)) return e;
  return vox_finish(n, workspace, idx_out, stream);
}

// the two halves on their own: a caller that has other work to enqueue puts it between them (it then runs on the device /
// is prepared on the host while the first half executes); nothing else may use `workspace` in between
extern "C" int clid_voxel_down_sample_launch(const float* points, int32_t n, float voxel_size, const float* value,
                                             const int64_t* n_dev, void* workspace, int64_t* idx_out, void* stream) {
  return vox_launch(points, n, voxel_size, value, workspace, idx_out, stream, n_dev);
}
extern "C" int clid_voxel_down_sample_finish(int32_t n, void* workspace, int64_t* idx_out, void* stream) {
  return vox_finish(n, workspace, idx_out, stream);
}
// The whole pass without a host round trip: count_out (device int64[2]) receives [number of voxels m | 1 if the voxel ids
// were too wide for the device-side ordering (then idx_out is not ordered: the caller treats the frame as failed)] and
// idx_out[0..m) the indices; consumers take both on the device (clid_cloud_update / clid_map_insert with sample_idx).
extern "C" int clid_voxel_down_sample_async(const float* points, int32_t n, float voxel_size, const int64_t* n_dev, void* workspace,
                                            int64_t* idx_out, int64_t* count_out, void* stream) {
  if (n <= 0 || n > (1 << 21) || !count_out) {
    clid_set_error("clid_voxel_down_sample_async: 1 .. 2^21 points and a count block");
    return CLID_E_ARG;
  }
  return vox_launch(points, n, voxel_size, nullptr, workspace, idx_out, stream, n_dev, count_out);
}

extern "C" int clid_voxel_down_sample(const float* points, int32_t n, float voxel_size, void* workspace,
                                      int64_t* idx_out, void* stream) {
  return vox_down_sample(points, n, voxel_size, nullptr, workspace, idx_out, stream);
}

extern "C" int clid_voxel_down_sample_min_value(const float* points, int32_t n, float voxel_size, const float* value,
                                                void* workspace, int64_t* idx_out, void* stream) {
  if (n > 0 && !value) {
    clid_set_error("clid_voxel_down_sample_min_value: bad argument");
    return CLID_E_ARG;
  }
  return vox_down_sample(points, n, voxel_size, value, workspace, idx_out, stream);
}

extern "C" int clid_transform_points(const float* points, int32_t n, const float* pose12_host, float* out, void* stream) {
  if (n < 0 || !pose12_host || (n > 0 && (!points || !out))) {
    clid_set_error("clid_transform_points: bad argument");
    return CLID_E_ARG;
  }
  if (n == 0) return CLID_OK;
  Pose12 p;
  for (int i = 0; i < 12; ++i) p.T[i] = pose12_host[i];
  hipLaunchKernelGGL(k_transform, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, points, n, p, out);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

extern "C" int clid_local_to_global(const int64_t* ids, int32_t n, int64_t pad_row, const float* local_feat,
                                    const float* local_cert, const int32_t* local_ts, float* global_feat,
                                    float* global_cert, int32_t* global_ts, void* stream) {
  // an empty local map (n == 0) still copies the padding row, which needs the feature arrays only: the id / certainty /
  // stamp tensors of an empty map have no storage (data_ptr() == 0)
  if (n < 0 || !local_feat || !global_feat || (n > 0 && (!ids || !local_cert || !local_ts || !global_cert || !global_ts))) {
    clid_set_error("clid_local_to_global: bad argument");
    return CLID_E_ARG;
  }
  const int threads = 2 * (n + 1);
  hipLaunchKernelGGL(k_local_to_global, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(ids), n, (long long)pad_row,
                     reinterpret_cast<const float4*>(local_feat), local_cert, local_ts,
                     reinterpret_cast<float4*>(global_feat), global_cert, global_ts);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

// ---- per-call preparation of Mapper.mapping: workspace reset + batch index draw in ONE launch ----------------------
// Batch composition of utils/mapper.py:473-500 for every iteration of the call: bs - bs_new uniform draws from the pool
// followed by bs_new uniform draws from the newest frame's samples (new_idx).  Counter-based generator: element e of
// call `counter` is mix64(seed, counter, e) -> multiply-high into the range (bias < 2^-40), so the draw is a pure function
// of (seed, counter, position): identical on every rank of a multi-GPU run and reproducible on the host (tests).
__host__ __device__ inline unsigned long long clid_mix64(unsigned long long seed, unsigned long long counter,
                                                           unsigned long long e) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (counter + 1) + 0xD1B54A32D192ED03ull * (e + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;  // splitmix64 finaliser, twice
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// 8 bits per axis of the sample's voxel coordinate, interleaved (x lowest): the order the sorted variant presents a batch in
constexpr unsigned kSortClassBit = 1u << 24;  // above the 24-bit code in a sort key
__device__ __forceinline__ unsigned morton24(int cx, int cy, int cz) {
  auto spread = [](unsigned v) {
    v &= 0xFFu;
    v = (v | (v << 8)) & 0x00F00Fu;
    v = (v | (v << 4)) & 0x0C30C3u;
    v = (v | (v << 2)) & 0x249249u;
    return v;
  };
  return spread((unsigned)cx) | (spread((unsigned)cy) << 1) | (spread((unsigned)cz) << 2);
}

__global__ void __launch_bounds__(256)
k_mapping_prep(float4* __restrict__ zero4, long long n_zero4, long long* __restrict__ index_out, long long n_index, int bs,
               int bs_new, unsigned long long pool_count, const long long* __restrict__ new_idx, unsigned long long n_new,
               unsigned long long seed, unsigned long long counter, const float* __restrict__ pool_coord, float resolution,
               unsigned* __restrict__ key_out, int col0, int ncols, int decim) {
  const long long stride = (long long)gridDim.x * 256;
  const long long t0 = (long long)blockIdx.x * 256 + threadIdx.x;
  for (long long i = t0; i < n_zero4; i += stride) zero4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int n_hist = bs - bs_new;
  // n_index counts the elements of the column window [col0, col0 + ncols) of every iteration; the draw is a function of the
  // element's position e in the FULL [iters][bs] array, so a rank that draws only its shard draws the same values
  for (long long u = t0; u < n_index; u += stride) {
    const long long it = u / ncols;
    const int col = col0 + (int)(u - it * ncols);
    const long long e = it * bs + col;
    const unsigned long long r = clid_mix64(seed, counter, (unsigned long long)e);
    long long v;
    if (col < n_hist) v = (long long)__umul64hi(r, pool_count);
    else v = new_idx[__umul64hi(r, n_new)];
    index_out[e] = v;
    if (key_out) {  // Morton code of the sample's voxel (8 bits per axis: wraps every 256 voxels): the sort key of k_batch_sort.
      // Bit 24 = the position's class (see Lattice below): 0 on the eikonal lattice col % decim == 0, 1 elsewhere -- the order is
      // by (class, code, position), so the class costs the ordering launch nothing
      struct F3 {
        float x, y, z;
      };
      const F3 c = reinterpret_cast<const F3*>(pool_coord)[v];
      key_out[e] = morton24((int)floorf(fdiv(c.x, resolution)), (int)floorf(fdiv(c.y, resolution)),
                            (int)floorf(fdiv(c.z, resolution))) | ((decim > 1 && col % decim != 0) ? kSortClassBit : 0u);
    }
  }
}

// Spatially ordered variant: one 1024-thread block per (iteration, 16 384-sample segment of the batch) redraws the
// segment's samples (the draw is a pure function of the position), keys them by the Morton code of their voxel (8 bits per
// axis: wraps every 256 voxels) and sorts (key, position) in LDS with a stable block radix sort, so the order is a function
// of the draws alone: identical on every rank.  The zero fill is shared by all blocks.
constexpr int kSortSeg = 16384, kSortThreads = 1024, kSortBins = 256, kSortWaves = kSortThreads / 64;
// (the radix passes below order the TAIL of a batch that is no multiple of a segment; full segments: k_batch_sort_bucket)
struct BinScan {  // LDS of block_excl_scan<unsigned, kSortThreads> (the top of this file): the block's wave totals
  struct TempStorage {
    unsigned w[kSortThreads / 64];
  };
};

// lanes of the wave that hold the same 8-bit digit as this lane (an OR-mask row in LDS per wave does the same with two LDS
// operations, but a sorting block is LDS-throughput-bound on its one CU)
__device__ __forceinline__ unsigned long long match8(unsigned d) {
  unsigned long long m = ~0ULL;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const unsigned long long s = __ballot((d >> b) & 1u);
    m &= ((d >> b) & 1u) ? s : ~s;
  }
  return m;
}

// One stable counting-sort pass over a sequence of ITEMS * 1024 elements held ITEMS per thread in WAVE-BLOCKED order:
// element (wave, r, lane) is sequence position wave * 64 * ITEMS + r * 64 + lane.  A wave walks its elements in order, so
// the rank of an element among the equal digits of ITS wave is (running count of the digit in the wave's column of `tab`)
// + (equal digits in lower lanes, from ballots); one exclusive scan over tab[digit][wave] then turns the columns into
// global offsets.  No atomic decides an order: the permutation is a function of the digits alone.
template <int ITEMS>
__device__ __forceinline__ void counting_pass(const unsigned (&digit)[ITEMS], unsigned (&dest)[ITEMS], unsigned* tab,
                                              typename BinScan::TempStorage& scan_tmp) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < kSortBins * kSortWaves; i += kSortThreads) tab[i] = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const unsigned d = digit[r];
    const unsigned long long peers = match8(d);
    const unsigned rk = (unsigned)__popcll(peers & ((1ULL << lane) - 1ULL));
    const int leader = __ffsll((long long)peers) - 1;
    unsigned old = 0;
    if (rk == 0) {  // (lane == leader) only this wave touches its column, in program order
      old = tab[d * kSortWaves + wave];
      tab[d * kSortWaves + wave] = old + (unsigned)__popcll(peers);
    }
    dest[r] = (unsigned)__shfl((int)old, leader, 64) + rk;  // rank among the wave's equal digits so far
  }
  __syncthreads();
  {  // exclusive scan over [digit][wave]: 4 consecutive entries per thread
    constexpr int kPer = kSortBins * kSortWaves / kSortThreads;
    const int b0 = threadIdx.x * kPer;
    unsigned c[kPer], sum = 0;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      c[i] = tab[b0 + i];
      sum += c[i];
    }
    unsigned off;
    unsigned all_;
    off = block_excl_scan<unsigned, kSortThreads>(sum, scan_tmp.w, &all_);
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      tab[b0 + i] = off;
      off += c[i];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) dest[r] += tab[digit[r] * kSortWaves + wave];
  __syncthreads();
}

// Three stable 8-bit passes over (key24, pos14) pairs held ITEMS per thread (wave-blocked): on return element r of the
// thread is the one with draw position pos[r] and its place in the order by (key, sequence position) is dest[r].
template <int ITEMS>
__device__ __forceinline__ void sort_pairs(unsigned (&key)[ITEMS], unsigned (&pos)[ITEMS], unsigned (&dest)[ITEMS], unsigned* tab,
                                           unsigned* seq, typename BinScan::TempStorage& scan_tmp) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i0 = wave * (ITEMS * 64) + lane;  // sequence position of item r: i0 + 64 r
  unsigned digit[ITEMS], s[ITEMS];
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) digit[r] = key[r] & 0xFFu;
  counting_pass<ITEMS>(digit, dest, tab, scan_tmp);
  CLID_STAMP(7);
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) seq[dest[r]] = ((key[r] >> 8) << 14) | pos[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    s[r] = seq[i0 + 64 * r];
    digit[r] = (s[r] >> 14) & 0xFFu;
  }
  counting_pass<ITEMS>(digit, dest, tab, scan_tmp);  // (its barriers also cover the reads of seq above)
  CLID_STAMP(8);
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) seq[dest[r]] = ((s[r] >> 22) << 14) | (s[r] & 0x3FFFu);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    s[r] = seq[i0 + 64 * r];
    digit[r] = (s[r] >> 14) & 0xFFu;
    pos[r] = s[r] & 0x3FFFu;
  }
  counting_pass<ITEMS>(digit, dest, tab, scan_tmp);
}

// ---- class-preserving placement ------------------------------------------------------------------------------------------
// The eikonal term of an iteration runs on coord[::decim] of its batch (utils/mapper.py:700-704): the draws at the batch
// positions col with col % decim == 0 -- a uniform random tenth.  Ordering a batch in space must not change WHICH draws those
// are, so the order is applied within the two classes: the lattice positions of a segment receive the segment's lattice draws
// in (Morton code, position) order, the other positions the other draws in that order.  The eikonal subset -- and with it every
// loss term -- is then the reference's on the unordered batch, sample for sample; only the order of the work changes.
// (decim == 1: every position is a lattice position and the rule is the plain order.)
struct Lattice {
  int decim;  // config.gradient_decimation when the numerical eikonal term is on, else 1
  int off;    // position inside the segment of its first lattice column
};
__device__ __forceinline__ Lattice lattice_of(long long seg_col0, int decim) {
  return Lattice{decim, (int)((decim - (int)(seg_col0 % decim)) % decim)};
}
__device__ __forceinline__ bool lattice_has(const Lattice& L, unsigned p) { return L.decim == 1 || ((int)p - L.off + L.decim) % L.decim == 0; }
// position of the r-th lattice (cls) / the r-th other position of the segment
__device__ __forceinline__ unsigned lattice_pos(const Lattice& L, bool cls, unsigned r) {
  if (cls) return (unsigned)L.off + r * (unsigned)L.decim;
  if (r < (unsigned)L.off) return r;
  const unsigned r2 = r - (unsigned)L.off, d1 = (unsigned)L.decim - 1u;
  return (unsigned)L.off + (r2 / d1) * (unsigned)L.decim + 1u + r2 % d1;
}
// rD[r] = number of lattice-class elements of the block placed before element r (places `dest`, a permutation of 0 .. m-1; only
// dest < m counts).  words / wpre: NW = ceil(m_cap / 32) unsigned each, in LDS (the radix table is free after sort_pairs).
template <int ITEMS>
__device__ __forceinline__ void class_ranks(const unsigned (&dest)[ITEMS], const bool (&cls)[ITEMS], unsigned m, unsigned (&rD)[ITEMS],
                                            unsigned* words, unsigned* wpre, int NW, typename BinScan::TempStorage& scan_tmp) {
  for (int i = threadIdx.x; i < NW; i += kSortThreads) words[i] = 0u;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ITEMS; ++r)
    if (dest[r] < m && cls[r]) atomicOr(&words[dest[r] >> 5], 1u << (dest[r] & 31u));
  __syncthreads();
  {
    const unsigned c = (int)threadIdx.x < NW ? (unsigned)__popc(words[threadIdx.x]) : 0u;
    unsigned off;
    unsigned all_;
    off = block_excl_scan<unsigned, kSortThreads>(c, scan_tmp.w, &all_);
    if ((int)threadIdx.x < NW) wpre[threadIdx.x] = off;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const unsigned d = dest[r] < m ? dest[r] : 0u;
    rD[r] = wpre[d >> 5] + (unsigned)__popc(words[d >> 5] & ((1u << (d & 31u)) - 1u));
  }
}

// Spatially ordered batches.  A segment = 16 384 consecutive samples of one iteration's batch; its draws are ordered by
// (Morton code of the sample's voxel, draw position) -- a total order, so the result is a function of the draws alone,
// identical on every rank.  draws / keys come from k_mapping_prep (computed by the whole chip: the 16 k random 12-byte
// gathers of a segment take 25 us through one CU's L1).
//
// k_batch_sort_tail: one block orders a whole (short) segment: the tail of a batch that is no multiple of 16 384.
__global__ void __launch_bounds__(kSortThreads)
k_batch_sort_tail(const long long* __restrict__ draws, const unsigned* __restrict__ keys, long long* __restrict__ index_out, int bs,
                  int base, int decim) {
  constexpr int ITEMS = kSortSeg / kSortThreads;
  __shared__ unsigned tab[kSortBins * kSortWaves];  // [digit][wave]
  __shared__ unsigned seq[kSortSeg];                // the sequence between passes: (remaining code << 14) | draw position
  __shared__ typename BinScan::TempStorage scan_tmp;
  const int n = bs - base;
  const long long e0 = (long long)blockIdx.x * bs + base;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned key[ITEMS], pos[ITEMS], dest[ITEMS];
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int p = wave * (ITEMS * 64) + r * 64 + lane;
    pos[r] = (unsigned)p;
    key[r] = p < n ? (keys[e0 + p] & 0xFFFFFFu) : 0xFFFFFFu;  // padding: behind every real element (largest code, larger position)
  }
  sort_pairs<ITEMS>(key, pos, dest, tab, seq, scan_tmp);
  if (decim > 1) {  // (uniform) the order within the lattice class and within the others (see Lattice)
    const Lattice L = lattice_of(base, decim);
    bool cls[ITEMS];
    unsigned rD[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) cls[r] = lattice_has(L, pos[r]);
    static_assert(kSortSeg / 32 * 2 <= kSortBins * kSortWaves, "class_ranks' words + prefix fit the radix table");
    class_ranks<ITEMS>(dest, cls, (unsigned)n, rD, tab, tab + kSortSeg / 32, kSortSeg / 32, scan_tmp);
#pragma unroll
    for (int r = 0; r < ITEMS; ++r)
      if ((int)dest[r] < n) index_out[e0 + lattice_pos(L, cls[r], cls[r] ? rD[r] : dest[r] - rD[r])] = draws[e0 + pos[r]];
    return;
  }
#pragma unroll
  for (int r = 0; r < ITEMS; ++r)
    if ((int)dest[r] < n) index_out[e0 + dest[r]] = draws[e0 + pos[r]];
}

// k_batch_sort_bucket: a FULL segment is shared by 12 .. 16 blocks.  Every block derives the same splitters from the same 512
// sample keys (positions 0, 32, 64, ...), reads the segment's keys (64 KB, 16 consecutive ones per thread), keeps the elements
// of ITS key range in position order, orders them by (key, position) and writes them behind the smaller ranges.  No
// communication between the blocks.
//   Round 6 (34.4 -> 21.9 us at the bench shape, same box: profiles/r06_sort_ab.jsonl; phases: tools/sort_timing.py):
//   - the position's class rides in the key (bit 24, written by k_mapping_prep) and a splitter is forced onto the class boundary:
//     no range holds both classes, the place of an element is its rank in (class, code, position) order mapped through the
//     lattice -- no second ranking pass, no per-class counters in the scan (the scan + keep phases: 30 k -> 11 k cycles);
//   - the scan is thread-blocked (four 16-byte loads, two compares per key) with ONE block scan of packed (mine | below) counts
//     instead of two ballots + population counts per key and wave;
//   - at most 2048 elements per block, ordered by a bitonic network on composites held two per thread: partner distance 1 is
//     the thread's own pair, 2 .. 64 are lane exchanges (DPP moves up to 8 lanes, permlane swaps for 16 / 32: no LDS crossbar,
//     of which a 1024-thread block has one), 128 and up go through LDS (double-buffered: one barrier per stage) -- 66
//     compare-exchange stages, 10 barriers, against three counting passes of 8 ballots + a dependent LDS update per element
//     and a 4096-entry table scan each (26 k -> 12 k cycles).  Composites are 32 bits ((key - lo) << 11 | slot) whenever the
//     kept keys lie within 2^21 of lo, 64 bits (key << 14 | position) otherwise; the network is sized to the element count;
//   - the splitters come from the same network (512 sample VALUES, 45 stages on four waves).
constexpr int kSortBucketsMax = 16, kSortBucketsMin = 12, kSortCap = 2048, kSortSamples = 512;
// blocks per segment: 16; 12 .. 15 when that keeps every block of the call alone on its compute unit (two 1024-thread blocks on a
// CU take twice as long, and the launch lasts as long as its slowest block); decided by the CALL's shape (the chip's 256 CUs as a
// constant), so every rank of a data-parallel run orders a shared segment the same way
__host__ inline int sort_buckets_for(long long call_segments) {
  if (call_segments * kSortBucketsMax <= 256 || call_segments * kSortBucketsMin > 256) return kSortBucketsMax;
  return (int)(256 / call_segments);
}
static_assert(kSortCap == 2 * kSortThreads, "two composites per thread");
// value of lane (lane ^ J): DPP moves up to 8 lanes (no LDS crossbar: a 1024-thread block has ONE), permlane swaps for 16 / 32
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int J>
__device__ __forceinline__ unsigned lane_xor(unsigned v) {
  if constexpr (J == 1) return dpp_u32<0xB1>(v);                        // quad_perm [1,0,3,2]
  else if constexpr (J == 2) return dpp_u32<0x4E>(v);                   // quad_perm [2,3,0,1]
  else if constexpr (J == 4) return dpp_u32<0x1B>(dpp_u32<0x141>(v));   // row_half_mirror (^7), then quad_perm [3,2,1,0] (^3)
  else if constexpr (J == 8) return dpp_u32<0x141>(dpp_u32<0x140>(v));  // row_mirror (^15), then row_half_mirror (^7)
  else if constexpr (J == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // one of the pair is the lane's own value
    return r[0] == v ? r[1] : r[0];
  } else {
    static_assert(J == 32, "lane distance");
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return r[0] == v ? r[1] : r[0];
  }
}
template <int J>
__device__ __forceinline__ unsigned long long lane_xor(unsigned long long v) {
  return ((unsigned long long)lane_xor<J>((unsigned)(v >> 32)) << 32) | lane_xor<J>((unsigned)v);
}
template <class T>
__device__ __forceinline__ T tmin(T a, T b) { return a < b ? a : b; }
template <class T>
__device__ __forceinline__ T tmax(T a, T b) { return a < b ? b : a; }
template <class T>
struct Pair2 {
  T x, y;
};
// ascending bitonic network over N <= 2048 composites: thread t < N / 2 holds elements 2 t and 2 t + 1 (the other threads of the
// block only keep the barriers).  Partner distance 1: the thread's own pair; 2 .. 64: a lane exchange (lane ^ 1 .. 32); 128 and up:
// through LDS, two buffers in turn (one barrier per stage; the first such stage writes buffer 1: buffer 0 still holds the
// unordered elements other threads may be reading).
// keep the smaller (keep_min) or the larger of x and the partner's p.  32 bits: min / max (the compiler folds the DPP move into
// them); 64 bits: ONE compare and two selects (min and max separately cost two 64-bit compares and six selects)
__device__ __forceinline__ void keep_one(unsigned& x, unsigned p, bool keep_min) { x = keep_min ? min(x, p) : max(x, p); }
__device__ __forceinline__ void keep_one(unsigned long long& x, unsigned long long p, bool keep_min) {
  x = ((p < x) == keep_min) ? p : x;  // (p == x: either)
}
template <class T, int N, int K, int J>
__device__ __forceinline__ void bitonic_stage(T& x0, T& x1, T* __restrict__ buf, int& flip, bool active) {
  const unsigned t = threadIdx.x;
  const bool up = K == N || (t & (unsigned)(K >> 1)) == 0u;  // direction of the element's K-block (index 2 t)
  if constexpr (J == 1) {
    const bool swap = (x1 < x0) == up;
    const T a = swap ? x1 : x0, b = swap ? x0 : x1;
    x0 = a;
    x1 = b;
  } else {
    T p0 = x0, p1 = x1;
    if constexpr (J <= 64) {
      p0 = lane_xor<J / 2>(x0);
      p1 = lane_xor<J / 2>(x1);
    } else {
      Pair2<T>* __restrict__ b2 = reinterpret_cast<Pair2<T>*>(buf + (size_t)flip * kSortCap);
      if (active) b2[t] = Pair2<T>{x0, x1};
      __syncthreads();
      if (active) {
        const Pair2<T> p = b2[t ^ (unsigned)(J >> 1)];
        p0 = p.x;
        p1 = p.y;
      }
      flip ^= 1;
    }
    const bool keep_min = ((t & (unsigned)(J >> 1)) == 0u) == up;
    keep_one(x0, p0, keep_min);
    keep_one(x1, p1, keep_min);
  }
}
template <class T, int N, int K, int J>
__device__ __forceinline__ void bitonic_phase(T& x0, T& x1, T* __restrict__ buf, int& flip, bool active) {
  if (J > 64 || active) bitonic_stage<T, N, K, J>(x0, x1, buf, flip, active);  // (wave-uniform: N / 2 is a multiple of 64)
  if constexpr (J > 1) bitonic_phase<T, N, K, J / 2>(x0, x1, buf, flip, active);
}
template <class T, int N, int K = 2>
__device__ __forceinline__ void bitonic_sort(T& x0, T& x1, T* __restrict__ buf, int& flip, bool active) {
  bitonic_phase<T, N, K, K / 2>(x0, x1, buf, flip, active);
  if constexpr (K < N) bitonic_sort<T, N, 2 * K>(x0, x1, buf, flip, active);
}
// the network of the smallest size that holds m elements (block-uniform): a lattice-class range of a few hundred elements
// occupies four of the block's sixteen waves for 45 stages, not all of them for 66
template <class T>
__device__ __forceinline__ void bitonic_sort_m(T& x0, T& x1, T* __restrict__ buf, unsigned m) {
  int flip = 1;
  if (m <= 512u) bitonic_sort<T, 512>(x0, x1, buf, flip, threadIdx.x < 256);
  else if (m <= 1024u) bitonic_sort<T, 1024>(x0, x1, buf, flip, threadIdx.x < 512);
  else bitonic_sort<T, kSortCap>(x0, x1, buf, flip, true);
}
// place of the element of rank R in (class, code, position) order of a full segment: the lattice positions take the ranks below n_lat
__device__ __forceinline__ unsigned lattice_place(const Lattice& L, unsigned n_lat, unsigned R) {
  if (L.decim == 1) return R;
  return R < n_lat ? lattice_pos(L, true, R) : lattice_pos(L, false, R - n_lat);
}

__global__ void __launch_bounds__(kSortThreads)
k_batch_sort_bucket(const long long* __restrict__ draws, const unsigned* __restrict__ keys, long long* __restrict__ index_out,
                    int bs, int full_segs, int seg0, int decim, int buckets, int force_wide) {
  constexpr int SCAN = kSortSeg / kSortThreads;
  __shared__ __attribute__((aligned(16))) unsigned long long buf[2 * kSortCap];
  __shared__ unsigned sorted[kSortSamples];
  __shared__ unsigned short selpos[kSortCap];
  __shared__ unsigned scan_ws[kSortWaves];
  const int bk = blockIdx.x % buckets;
  const int sg = blockIdx.x / buckets;
  const int it = sg / full_segs, seg = seg0 + (sg - it * full_segs);
  const long long e0 = (long long)it * bs + (long long)seg * kSortSeg;
  CLID_STAMP(0);
  // the segment's keys, 16 consecutive ones per thread, requested first: they arrive under the splitter ranking
  unsigned key[SCAN];
  {
    const unsigned* __restrict__ kp = keys + e0 + SCAN * threadIdx.x;
    if ((e0 & 3) == 0) {  // (uniform) 16-byte loads when the segment starts on a 16-byte boundary of the key array
#pragma unroll
      for (int q = 0; q < SCAN / 4; ++q) {
        const uint4 v = reinterpret_cast<const uint4*>(kp)[q];
        key[4 * q] = v.x;
        key[4 * q + 1] = v.y;
        key[4 * q + 2] = v.z;
        key[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int r = 0; r < SCAN; ++r) key[r] = kp[r];
    }
  }
  // ---- splitters: the 512 keys at positions 0, 32, 64, ... ordered by (key, sample number) on the same network (four waves, 45
  // stages); 512 samples keep a range of twice the mean size from ever (< 1e-4 per range at 12 blocks per segment) exceeding
  // a block's capacity, 256 did not (1.5 % / 0.1 % at 12 / 16 blocks)
  constexpr int kSampStep = kSortSeg / kSortSamples;
  const bool samp_thread = threadIdx.x < kSortSamples / 2;
  unsigned q0 = 0xFFFFFFFFu, q1 = 0xFFFFFFFFu;  // (the splitters are VALUES at given ranks: equal sample keys need no order)
  if (samp_thread) {
    q0 = keys[e0 + kSampStep * (2 * threadIdx.x)];
    q1 = keys[e0 + kSampStep * (2 * threadIdx.x + 1)];
  }
  CLID_STAMP(1);
  {
    int flip = 1;
    bitonic_sort<unsigned, kSortSamples>(q0, q1, reinterpret_cast<unsigned*>(buf), flip, samp_thread);
  }
  if (samp_thread) {
    sorted[2 * threadIdx.x] = q0;
    sorted[2 * threadIdx.x + 1] = q1;
  }
  // samples of the lattice class (they rank first); the barriers also publish `sorted` and end the network's use of buf
  const unsigned s0 = (unsigned)__syncthreads_count(samp_thread && q0 < kSortClassBit) +
                      (unsigned)__syncthreads_count(samp_thread && q1 < kSortClassBit);
  CLID_STAMP(2);
  // this block's key range [lo, hi).  The classes share the blocks by their sizes (one position in `decim` is a lattice position):
  // nb0 blocks split the lattice class at its own samples' quantiles, the others the rest; the class boundary is a splitter.
  constexpr unsigned kLast = kSortSamples - 1;
  const int nb0 = decim == 1 ? buckets : min(max((buckets + decim / 2) / decim, 1), buckets - 1);
  unsigned lo, hi;
  if (bk < nb0) {
    lo = bk > 0 ? sorted[min((unsigned)bk * s0 / (unsigned)nb0, kLast)] : 0u;
    hi = bk + 1 < nb0 ? sorted[min((unsigned)(bk + 1) * s0 / (unsigned)nb0, kLast)] : (decim == 1 ? 0xFFFFFFFFu : kSortClassBit);
    if (s0 == 0u) lo = 0u, hi = bk + 1 < nb0 ? 0u : kSortClassBit;  // (no sample of the class: its last block takes all of it)
  } else {
    const unsigned j = (unsigned)(bk - nb0), nb1 = (unsigned)(buckets - nb0), s1 = (unsigned)kSortSamples - s0;
    lo = j > 0 ? sorted[min(s0 + j * s1 / nb1, kLast)] : kSortClassBit;
    hi = j + 1 < nb1 ? sorted[min(s0 + (j + 1) * s1 / nb1, kLast)] : 0xFFFFFFFFu;
    if (s1 == 0u) lo = j + 1 < nb1 ? 0xFFFFFFFFu : kSortClassBit, hi = 0xFFFFFFFFu;
  }
  // ---- scan: count, one block scan of the packed counts, keep this block's range in position order
  unsigned cnt = 0;   // mine | below << 16
  bool far = false;   // a kept key 2^21 or more above lo: the narrow composites below do not hold it
#pragma unroll
  for (int r = 0; r < SCAN; ++r) {
    const bool mine = key[r] >= lo && key[r] < hi;
    cnt += (key[r] < lo ? 0x10000u : 0u) + (mine ? 1u : 0u);
    far |= mine && key[r] - lo >= (1u << 21);
  }
  buf[2 * threadIdx.x] = ~0ULL;  // padding behind every real element (all ones at either width)
  buf[2 * threadIdx.x + 1] = ~0ULL;
  unsigned tot;
  const unsigned ex = block_excl_scan<unsigned, kSortThreads>(cnt, scan_ws, &tot);
  CLID_STAMP(3);
  const unsigned m = tot & 0xFFFFu, lower = tot >> 16;
  const Lattice L = lattice_of((long long)seg * kSortSeg, decim);
  const unsigned n_lat = decim == 1 ? 0u : (unsigned)((kSortSeg - L.off + decim - 1) / decim);
  if (m > (unsigned)kSortCap) {  // (block-uniform) a key range more than twice the mean: keeps the draw order
    unsigned slot = ex & 0xFFFFu;
#pragma unroll
    for (int r = 0; r < SCAN; ++r)
      if (key[r] >= lo && key[r] < hi) {
        index_out[e0 + lattice_place(L, n_lat, lower + slot)] = draws[e0 + (unsigned)(SCAN * threadIdx.x + r)];
        ++slot;
      }
    return;
  }
  // Composites.  Narrow (the rule): (key - lo) << 11 | slot, 32 bits -- the slots are in position order, so the order by
  // composite is the order by (key, position); the position comes back through selpos.  Wide (a kept key 2^21 or more above lo:
  // the one range that holds the whole lattice class, or two far clusters in one range): key << 14 | position, 64 bits.
  // (the barrier also orders the padding stores above in front of the composites)
  const bool narrow = !__syncthreads_or(far || force_wide);
  unsigned* __restrict__ buf32 = reinterpret_cast<unsigned*>(buf);
  {
    unsigned slot = ex & 0xFFFFu;
#pragma unroll
    for (int r = 0; r < SCAN; ++r)
      if (key[r] >= lo && key[r] < hi) {
        const unsigned p = (unsigned)(SCAN * threadIdx.x + r);
        if (narrow) {
          buf32[slot] = ((key[r] - lo) << 11) | slot;
          selpos[slot] = (unsigned short)p;
        } else {
          buf[slot] = ((unsigned long long)key[r] << 14) | p;
        }
        ++slot;
      }
  }
  __syncthreads();
  CLID_STAMP(4);
  unsigned p0, p1;  // positions (in the segment) of the elements of rank 2 t and 2 t + 1 of this block
  if (narrow) {
    unsigned x0 = buf32[2 * threadIdx.x], x1 = buf32[2 * threadIdx.x + 1];
    bitonic_sort_m<unsigned>(x0, x1, buf32, m);
    p0 = selpos[x0 & 0x7FFu];
    p1 = selpos[x1 & 0x7FFu];
  } else {
    unsigned long long x0 = buf[2 * threadIdx.x], x1 = buf[2 * threadIdx.x + 1];
    bitonic_sort_m<unsigned long long>(x0, x1, buf, m);
    p0 = (unsigned)(x0 & 0x3FFFu);
    p1 = (unsigned)(x1 & 0x3FFFu);
  }
  CLID_STAMP(5);
  if (2 * threadIdx.x < m) index_out[e0 + lattice_place(L, n_lat, lower + 2 * threadIdx.x)] = draws[e0 + p0];
  if (2 * threadIdx.x + 1 < m) index_out[e0 + lattice_place(L, n_lat, lower + 2 * threadIdx.x + 1)] = draws[e0 + p1];
  CLID_STAMP(6);
}
#ifdef CLID_TIMING
extern "C" int clid_debug_read_stamps_mapops(long long* out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(clid::clid_stamps), sizeof(long long) * 256 * 32) == hipSuccess ? 0 : -3;
}
#endif

static int sort_force_wide() {  // CLID_SORT_WIDE=1: the 64-bit composites everywhere (tests: the path of far-apart key clusters)
  const char* e = getenv("CLID_SORT_WIDE");
  return (e && e[0] == '1') ? 1 : 0;
}
extern "C" int64_t clid_mapping_prep_workspace_bytes(int32_t iters, int32_t bs) {
  const long long n = (long long)iters * bs;
  if (n <= 0) return 256;
  return (int64_t)(align256((size_t)n * 8) + align256((size_t)n * 4) + 256);  // unsorted draws | keys
}

extern "C" int clid_mapping_prep(float* zero_base, int64_t zero_floats, int64_t* index_out, int32_t iters, int32_t bs,
                                 int32_t bs_new, int64_t pool_count, const int64_t* new_idx, int64_t n_new, uint64_t seed,
                                 uint64_t counter, const float* pool_coord, float resolution, void* sort_workspace,
                                 int32_t col0, int32_t ncols, int32_t decimation, void* stream) {
  if (decimation < 1) {
    clid_set_error("clid_mapping_prep: decimation %d", decimation);
    return CLID_E_ARG;
  }
  if (zero_floats < 0 || (zero_floats && (!zero_base || (zero_floats & 3) || ((uintptr_t)zero_base & 15))) || iters < 0 ||
      bs < 0 || bs_new < 0 || bs_new > bs || (iters && bs && !index_out) || (index_out && iters && bs && pool_count <= 0) ||
      (bs_new > 0 && (!new_idx || n_new <= 0)) || (sort_workspace && (!pool_coord || !(resolution > 0.f))) ||
      (ncols > 0 && (col0 < 0 || col0 + (long long)ncols > bs))) {
    clid_set_error("clid_mapping_prep: bad argument");
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  const bool want_sort = sort_workspace != nullptr;
  // the column window of every iteration that gets written: all of them, or the caller's shard -- widened to whole
  // 16 384-sample segments when the batches are ordered (a segment is ordered as a unit)
  int c0 = 0, nc = bs;
  if (ncols > 0 && index_out) {
    c0 = col0;
    int c1 = col0 + ncols;
    if (want_sort) {
      c0 = c0 / kSortSeg * kSortSeg;
      c1 = (c1 + kSortSeg - 1) / kSortSeg * kSortSeg;
      if (c1 > bs) c1 = bs;
    }
    nc = c1 - c0;
  }
  const long long n_index = index_out ? (long long)iters * nc : 0;
  const long long work = (zero_floats / 4 > n_index ? zero_floats / 4 : n_index);
  if (work == 0) return CLID_OK;
  const int seg0 = c0 / kSortSeg;                          // first segment of the window
  const int seg_full_end = (c0 + nc) / kSortSeg;           // one past its last FULL segment
  const int full_segs = seg_full_end > seg0 ? seg_full_end - seg0 : 0;
  const int tail_base = seg_full_end * kSortSeg;           // a shorter last segment exists iff the window reaches bs
  const int tail = (c0 + nc) - tail_base;                  // > 0 only then (c0 + nc == bs, bs no multiple of the segment)
  const int buckets = sort_buckets_for((long long)iters * ((bs + kSortSeg - 1) / kSortSeg));
  const long long sort_blocks = (long long)iters * full_segs * buckets;
  const bool sorted = want_sort && n_index > 0 && sort_blocks < (1LL << 31);
  char* ws = static_cast<char*>(sort_workspace);
  const long long n_all = (long long)iters * bs;
  long long* draws = sorted ? reinterpret_cast<long long*>(ws) : reinterpret_cast<long long*>(index_out);
  unsigned* keys = sorted ? reinterpret_cast<unsigned*>(ws + align256((size_t)n_all * 8)) : nullptr;
  long long blocks = (work + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_mapping_prep, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<float4*>(zero_base),
                     (long long)(zero_floats / 4), draws, n_index, bs, bs_new, (unsigned long long)pool_count,
                     reinterpret_cast<const long long*>(new_idx), (unsigned long long)n_new, (unsigned long long)seed,
                     (unsigned long long)counter, pool_coord, resolution, keys, c0, nc > 0 ? nc : 1, (int)decimation);
  if (sorted && full_segs > 0)
  {
    hipLaunchKernelGGL(k_batch_sort_bucket, dim3((unsigned)sort_blocks), dim3(kSortThreads), 0, s, draws, keys,
                       reinterpret_cast<long long*>(index_out), bs, full_segs, seg0, (int)decimation, buckets, sort_force_wide());
  }
  if (sorted && tail > 0)
    hipLaunchKernelGGL(k_batch_sort_tail, dim3((unsigned)iters), dim3(kSortThreads), 0, s, draws, keys,
                       reinterpret_cast<long long*>(index_out), bs, tail_base, (int)decimation);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

// host restatement of one draw (tests): the value k_mapping_prep writes at flat position e for a uniform column
extern "C" int64_t clid_debug_prep_draw(uint64_t seed, uint64_t counter, uint64_t e, uint64_t range) {
  const unsigned long long r = clid_mix64(seed, counter, e);
  return (int64_t)(unsigned long long)(((unsigned __int128)r * range) >> 64);
}

// test entry of the hand-written exclusive scans every compaction of this file runs on: elem_bytes 4 (int32) or 8 (uint64);
// scratch: clid_debug_scan_scratch_bytes(n) bytes
extern "C" int64_t clid_debug_scan_scratch_bytes(int64_t n) { return (int64_t)scan_scratch_bytes(n); }
extern "C" int clid_debug_scan(const void* in, void* out, int64_t n, int32_t elem_bytes, void* scratch, void* stream) {
  if (n < 0 || (n > 0 && (!in || !out || !scratch || in == out)) || (elem_bytes != 4 && elem_bytes != 8)) {
    clid_set_error("clid_debug_scan: bad argument");
    return CLID_E_ARG;
  }
  if (elem_bytes == 4) scan_exclusive(static_cast<const int*>(in), static_cast<int*>(out), (long long)n, scratch, (hipStream_t)stream);
  else scan_exclusive(static_cast<const unsigned long long*>(in), static_cast<unsigned long long*>(out), (long long)n, scratch, (hipStream_t)stream);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

// ---- pool maintenance --------------------------------------------------------------------------------------------
static size_t pool_scan_bytes(long long n) { return scan_scratch_bytes(n); }
extern "C" int64_t clid_pool_workspace_bytes(int64_t n_total) {
  if (n_total <= 0) return 256;
  const size_t nblk = ((size_t)n_total + 255) / 256;  // flags (1 B) | block counts | block offsets | kept list | scan scratch
  return (int64_t)(align256((size_t)n_total) + 2 * align256(nblk * 4) + align256((size_t)n_total * 4) +
                   align256(pool_scan_bytes((long long)nblk)) + 256);
}

extern "C" int clid_pool_filter_after(const float* coord_a, const float* gcoord_a, const float* label_a, const float* weight_a,
                                      const int32_t* time_a, int64_t n_a, const float* coord_b, const float* gcoord_b,
                                      const float* label_b, const float* weight_b, const int32_t* time_b, int64_t n_b,
                                      const double* origin_host, double radius2, int64_t capacity, uint64_t seed, float* coord_out,
                                      float* gcoord_out, float* label_out, float* weight_out, int32_t* time_out,
                                      int64_t* counts_out, void* workspace, const int64_t* n_b_dev, void* stream,
                                      void* scatter_after_event);
extern "C" int clid_pool_filter(const float* coord_a, const float* gcoord_a, const float* label_a, const float* weight_a,
                                const int32_t* time_a, int64_t n_a, const float* coord_b, const float* gcoord_b,
                                const float* label_b, const float* weight_b, const int32_t* time_b, int64_t n_b,
                                const double* origin_host, double radius2, int64_t capacity, uint64_t seed, float* coord_out,
                                float* gcoord_out, float* label_out, float* weight_out, int32_t* time_out,
                                int64_t* counts_out, void* workspace, const int64_t* n_b_dev, void* stream) {
  return clid_pool_filter_after(coord_a, gcoord_a, label_a, weight_a, time_a, n_a, coord_b, gcoord_b, label_b, weight_b, time_b, n_b,
                                origin_host, radius2, capacity, seed, coord_out, gcoord_out, label_out, weight_out, time_out,
                                counts_out, workspace, n_b_dev, stream, nullptr);
}
// scatter_after_event != NULL (a recorded hipEvent_t): the five-array compaction -- the frame's one bandwidth-bound launch, which
// takes every wave slot of the chip for its duration -- is held back until that event; the flag / list / drop passes in front of it
// are not.  (process_frame records it behind the map growth's voxel pass, whose small dependent launches starve next to it.)
extern "C" int clid_pool_filter_after(const float* coord_a, const float* gcoord_a, const float* label_a, const float* weight_a,
                                      const int32_t* time_a, int64_t n_a, const float* coord_b, const float* gcoord_b,
                                      const float* label_b, const float* weight_b, const int32_t* time_b, int64_t n_b,
                                      const double* origin_host, double radius2, int64_t capacity, uint64_t seed, float* coord_out,
                                      float* gcoord_out, float* label_out, float* weight_out, int32_t* time_out,
                                      int64_t* counts_out, void* workspace, const int64_t* n_b_dev, void* stream,
                                      void* scatter_after_event) {
  const long long n = n_a + n_b;
  if (n_a < 0 || n_b < 0 || n >= (1LL << 31) || !origin_host || !counts_out || !workspace || capacity < 0 ||
      (n_a > 0 && (!coord_a || !gcoord_a || !label_a || !weight_a || !time_a)) ||
      (n_b > 0 && (!coord_b || !gcoord_b || !label_b || !weight_b || !time_b)) ||
      (n > 0 && (!coord_out || !gcoord_out || !label_out || !weight_out || !time_out))) {
    clid_set_error("clid_pool_filter: bad argument");
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  long long* counts = reinterpret_cast<long long*>(counts_out);
  if (n == 0) return hipMemsetAsync(counts, 0, 3 * sizeof(long long), s) == hipSuccess ? CLID_OK : CLID_E_HIP;
  char* ws = static_cast<char*>(workspace);
  const long long nblk = (n + 255) / 256;
  unsigned char* flag = reinterpret_cast<unsigned char*>(ws);
  int* block_cnt = reinterpret_cast<int*>(ws + align256((size_t)n));
  int* block_off = reinterpret_cast<int*>(ws + align256((size_t)n) + align256((size_t)nblk * 4));
  int* kept_list = reinterpret_cast<int*>(ws + align256((size_t)n) + 2 * align256((size_t)nblk * 4));
  void* cub = ws + align256((size_t)n) + 2 * align256((size_t)nblk * 4) + align256((size_t)n * 4);
  const PoolSrc a{coord_a, gcoord_a, label_a, weight_a, time_a, n_a, nullptr},
      b{coord_b, gcoord_b, label_b, weight_b, time_b, n_b, reinterpret_cast<const long long*>(n_b_dev)};
  const unsigned blocks = (unsigned)nblk;
  hipLaunchKernelGGL(k_pool_flags, dim3(blocks), dim3(256), 0, s, a, b, origin_host[0], origin_host[1], origin_host[2], radius2,
                     flag, block_cnt);
  scan_exclusive(block_cnt, block_off, (long long)nblk, cub, s);
  if (n > capacity) {  // only then can more than `capacity` samples survive the window test
    hipLaunchKernelGGL(k_pool_list, dim3(blocks), dim3(256), 0, s, flag, block_off, block_cnt, n, kept_list, counts);
    hipLaunchKernelGGL(k_pool_drop, dim3(1024), dim3(256), 0, s, flag, kept_list, counts, (long long)capacity,
                       (unsigned long long)seed);
    hipLaunchKernelGGL(k_pool_count, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, s, flag, n, block_cnt);
    scan_exclusive(block_cnt, block_off, (long long)nblk, cub, s);
  }
  const PoolDst d{coord_out, gcoord_out, label_out, weight_out, time_out};
  if (scatter_after_event && hipStreamWaitEvent(s, (hipEvent_t)scatter_after_event, 0) != hipSuccess) {
    clid_set_error("clid_pool_filter_after: event wait failed: %s", hipGetErrorString(hipGetLastError()));
    return CLID_E_HIP;
  }
  hipLaunchKernelGGL(k_pool_scatter, dim3(blocks), dim3(256), 0, s, a, b, flag, block_off, block_cnt, d, counts);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

// ---- sampler output -> this frame's pool rows + the points that grow the map (utils/mapper.py:240-283, :297-310) -------
// The sampler kernel leaves [rays x samples] rows with a keep mask; the reference then runs boolean-mask indexing on three
// arrays, a second mask |sdf| < surface_range * ratio, another indexed copy and two rigid transforms.  One flag pass, ONE
// scan over packed (kept, near) 64-bit counters and one scatter: compacted coord / label / weight, the frame stamp, the
// world-frame coordinates (pool) and the world-frame near-surface subset (NeuralPoints.update), stable order.
static size_t scan64_bytes(long long n) { return scan_scratch_bytes(n); }

__global__ void __launch_bounds__(256)
k_compact_flags(const unsigned char* __restrict__ keep, const float* __restrict__ label, long long n, float near_range,
                unsigned long long* __restrict__ flag) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const bool k = keep[i] != 0;
  const bool nr = k && fabsf(label[i]) < near_range;
  flag[i] = (k ? 1ULL : 0ULL) | (nr ? (1ULL << 32) : 0ULL);
}

__global__ void __launch_bounds__(256)
k_compact_scatter(const float* __restrict__ coord, const float* __restrict__ label, const float* __restrict__ weight,
                  const unsigned long long* __restrict__ flag, const unsigned long long* __restrict__ pos, long long n,
                  Pose12 p, int stamp, float* __restrict__ coord_out, float* __restrict__ gcoord_out,
                  float* __restrict__ label_out, float* __restrict__ weight_out, int* __restrict__ stamp_out,
                  float* __restrict__ update_out, long long* __restrict__ counts) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned long long f = flag[i], q = pos[i];
  if (i == n - 1) {
    counts[0] = (long long)((q + f) & 0xffffffffULL);
    counts[1] = (long long)((q + f) >> 32);
  }
  if (!(f & 1ULL)) return;
  const long long o = (long long)(q & 0xffffffffULL);
  const float x = coord[i * 3 + 0], y = coord[i * 3 + 1], z = coord[i * 3 + 2];
  const float gx = fmaf(z, p.T[2], fmaf(y, p.T[1], fmaf(x, p.T[0], p.T[3])));  // transform_torch (utils/tools.py:590-609)
  const float gy = fmaf(z, p.T[6], fmaf(y, p.T[5], fmaf(x, p.T[4], p.T[7])));
  const float gz = fmaf(z, p.T[10], fmaf(y, p.T[9], fmaf(x, p.T[8], p.T[11])));
  coord_out[o * 3 + 0] = x; coord_out[o * 3 + 1] = y; coord_out[o * 3 + 2] = z;
  gcoord_out[o * 3 + 0] = gx; gcoord_out[o * 3 + 1] = gy; gcoord_out[o * 3 + 2] = gz;
  label_out[o] = label[i];
  weight_out[o] = weight[i];
  stamp_out[o] = stamp;
  if (f >> 32) {
    const long long u = (long long)(q >> 32);
    update_out[u * 3 + 0] = gx; update_out[u * 3 + 1] = gy; update_out[u * 3 + 2] = gz;
  }
}

extern "C" int64_t clid_sample_compact_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  return (int64_t)(2 * align256((size_t)n * 8) + align256(scan64_bytes(n)) + 256);
}

extern "C" int clid_sample_compact(const float* coord, const float* label, const float* weight, const uint8_t* keep, int64_t n,
                                   const float* pose12_host, float near_range, int32_t stamp, float* coord_out,
                                   float* gcoord_out, float* label_out, float* weight_out, int32_t* stamp_out,
                                   float* update_out, int64_t* counts_out, void* workspace, void* stream) {
  if (n < 0 || n >= (1LL << 31) || !pose12_host || !counts_out || !workspace ||
      (n > 0 && (!coord || !label || !weight || !keep || !coord_out || !gcoord_out || !label_out || !weight_out || !stamp_out ||
                 !update_out))) {
    clid_set_error("clid_sample_compact: bad argument");
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  long long* counts = reinterpret_cast<long long*>(counts_out);
  if (n == 0) return hipMemsetAsync(counts, 0, 2 * sizeof(long long), s) == hipSuccess ? CLID_OK : CLID_E_HIP;
  char* ws = static_cast<char*>(workspace);
  unsigned long long* flag = reinterpret_cast<unsigned long long*>(ws);
  unsigned long long* pos = reinterpret_cast<unsigned long long*>(ws + align256((size_t)n * 8));
  void* cub = ws + 2 * align256((size_t)n * 8);
  Pose12 p;
  for (int i = 0; i < 12; ++i) p.T[i] = pose12_host[i];
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_compact_flags, dim3(blocks), dim3(256), 0, s, keep, label, (long long)n, near_range, flag);
  scan_exclusive(flag, pos, (long long)n, cub, s);
  hipLaunchKernelGGL(k_compact_scatter, dim3(blocks), dim3(256), 0, s, coord, label, weight, flag, pos, (long long)n, p,
                     (int)stamp, coord_out, gcoord_out, label_out, weight_out, stamp_out, update_out, counts);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

// ---- newly observed samples (utils/mapper.py:400-423): 1-stencil certainty probe of the GLOBAL map + the two tests +
// the ascending index list, one flag pass and one scan
__global__ void __launch_bounds__(256)
k_new_sample_flags(const long long* __restrict__ table, int buffer_size, const float* __restrict__ points,
                   const float* __restrict__ cert, const int* __restrict__ delta, int P, float resolution,
                   float max_valid_dist2, const float* __restrict__ x, const float* __restrict__ label, int n,
                   float cert_thre, float label_max, int* __restrict__ flag, const long long* __restrict__ pool_counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (pool_counts) {  // x / label are the POOL arrays; this frame's samples are their last pool_counts[1] rows (device-side)
    const long long cur = pool_counts[1], first = pool_counts[0] - cur;
    if (i >= cur) {
      flag[i] = 0;
      return;
    }
    x += first * 3;
    label += first;
  }
  const float px = x[i * 3 + 0], py = x[i * 3 + 1], pz = x[i * 3 + 2];
  const int r0 = base_slot(px, py, pz, resolution, buffer_size);
  float best = 0.f;  // NeuralPoints.query_certainty (model/neural_points.py:1032-1051)
  for (int o = 0; o < P; ++o) {
    int slot = r0 + delta[o];
    if (slot >= buffer_size) slot -= buffer_size;
    const long long j = table[slot];
    if (j < 0) continue;
    const float ax = fsub(points[j * 3 + 0], px), ay = fsub(points[j * 3 + 1], py), az = fsub(points[j * 3 + 2], pz);
    const float d2 = fadd(fadd(fmul(ax, ax), fmul(ay, ay)), fmul(az, az));
    if (d2 > max_valid_dist2) continue;
    best = fmaxf(best, cert[j]);
  }
  flag[i] = (best < cert_thre && fabsf(label[i]) < label_max) ? 1 : 0;
}

__global__ void __launch_bounds__(256)
k_new_sample_list(const int* __restrict__ flag, const int* __restrict__ pos, int n, long long offset,
                  long long* __restrict__ idx_out, long long* __restrict__ count, const long long* __restrict__ pool_counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (pool_counts) offset = pool_counts[0] - pool_counts[1];
  if (i == n - 1) count[0] = (long long)pos[i] + flag[i];
  if (flag[i]) idx_out[pos[i]] = offset + i;
}

extern "C" int64_t clid_new_sample_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  return (int64_t)(2 * align256((size_t)n * 4) + align256(pool_scan_bytes(n)) + 256);
}

extern "C" int clid_new_sample_select(const int64_t* buffer_pt_index, int64_t buffer_size, const float* neural_points,
                                      const float* point_certainties, const int32_t* delta, int32_t P, float resolution,
                                      float max_valid_dist2, const float* x, const float* sdf_label, int64_t n,
                                      float certainty_thre, float label_max, int64_t index_offset, int64_t* idx_out,
                                      int64_t* count_out, const int64_t* pool_counts, void* workspace, void* stream) {
  if (n < 0 || n >= (1LL << 31) || buffer_size <= 0 || buffer_size >= (1LL << 30) || P <= 0 || !count_out || !workspace ||
      (n > 0 && (!buffer_pt_index || !neural_points || !point_certainties || !delta || !x || !sdf_label || !idx_out))) {
    clid_set_error("clid_new_sample_select: bad argument");
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  long long* count = reinterpret_cast<long long*>(count_out);
  if (n == 0) return hipMemsetAsync(count, 0, sizeof(long long), s) == hipSuccess ? CLID_OK : CLID_E_HIP;
  char* ws = static_cast<char*>(workspace);
  int* flag = reinterpret_cast<int*>(ws);
  int* pos = reinterpret_cast<int*>(ws + align256((size_t)n * 4));
  void* cub = ws + 2 * align256((size_t)n * 4);
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_new_sample_flags, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const long long*>(buffer_pt_index),
                     (int)buffer_size, neural_points, point_certainties, delta, P, resolution, max_valid_dist2, x, sdf_label,
                     (int)n, certainty_thre, label_max, flag, reinterpret_cast<const long long*>(pool_counts));
  scan_exclusive(flag, pos, (long long)n, cub, s);
  hipLaunchKernelGGL(k_new_sample_list, dim3(blocks), dim3(256), 0, s, flag, pos, (int)n, (long long)index_offset,
                     reinterpret_cast<long long*>(idx_out), count, reinterpret_cast<const long long*>(pool_counts));
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

// ---- local window ------------------------------------------------------------------------------------------------
extern "C" int64_t clid_local_window_workspace_bytes(int64_t n) {
  if (n <= 0) return 1024;
  return (int64_t)(align256((size_t)n) + 2 * align256((size_t)n * 4) + align256(pool_scan_bytes(n)) + 256);
}

extern "C" int clid_local_window(const float* neural_points, const int32_t* ts_create, const int32_t* ts_update,
                                 const float* travel_dist, int64_t n, int32_t cur_ts, int32_t use_mid_ts, int32_t temporal,
                                 int32_t use_travel_dist, float diff_travel, int32_t diff_ts_local, int32_t reboot_ts,
                                 int32_t reboot_map, const double* sensor_pos_host, double radius2, int32_t pos_is_f64,
                                 const float* point_orientations,
                                 const float* point_certainties, const float* geo_features, int64_t* local_ids_out,
                                 int64_t* global2local_out, uint8_t* local_mask_out, float* local_points_out,
                                 float* local_orient_out, float* local_cert_out, int32_t* local_ts_out, float* local_feat_out,
                                 int64_t* counts_out, void* workspace, const int64_t* n_extra_dev, int64_t n_upper,
                                 int64_t local_capacity, void* stream) {
  if (!n_extra_dev) n_upper = n;
  if (local_capacity < 0 || local_capacity > n_upper) local_capacity = n_upper;
  if (n < 0 || n_upper < n || n_upper >= (1LL << 31) || !sensor_pos_host || !counts_out || !workspace || !global2local_out || !local_mask_out ||
      !geo_features || !local_feat_out ||
      (n > 0 && (!neural_points || !ts_create || !ts_update || !point_orientations || !point_certainties || !local_ids_out ||
                 !local_points_out || !local_orient_out || !local_cert_out || !local_ts_out)) ||
      (temporal && use_travel_dist && !travel_dist)) {
    clid_set_error("clid_local_window: bad argument");
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  long long* counts = reinterpret_cast<long long*>(counts_out);
  if (hipMemsetAsync(counts, 0, 2 * sizeof(long long), s) != hipSuccess) return CLID_E_HIP;
  char* ws = static_cast<char*>(workspace);
  const long long nu = n_upper;  // grids, scan and workspace cover the upper bound; the kernels read the exact size
  unsigned char* bits = reinterpret_cast<unsigned char*>(ws);
  int* flag = reinterpret_cast<int*>(ws + align256((size_t)nu));
  int* pos = reinterpret_cast<int*>(ws + align256((size_t)nu) + align256((size_t)nu * 4));
  void* cub = ws + align256((size_t)nu) + 2 * align256((size_t)nu * 4);
  WindowArgs a{neural_points, ts_create, ts_update, travel_dist, n, reinterpret_cast<const long long*>(n_extra_dev), cur_ts,
               use_mid_ts, temporal, use_travel_dist, diff_ts_local, reboot_ts, reboot_map, diff_travel, sensor_pos_host[0],
               sensor_pos_host[1], sensor_pos_host[2], radius2, pos_is_f64};
  if (nu > 0) {
    const unsigned blocks = (unsigned)((nu + 255) / 256);
    hipLaunchKernelGGL(k_window_flags, dim3(blocks < kFlagBlocks ? blocks : kFlagBlocks), dim3(256), 0, s, a, bits, counts);
    hipLaunchKernelGGL(k_window_combine, dim3(blocks), dim3(256), 0, s, a, bits, nu, counts, flag);
    scan_exclusive(flag, pos, (long long)nu, cub, s);
  }
  WindowOut o{(long long)local_capacity, reinterpret_cast<long long*>(local_ids_out), reinterpret_cast<long long*>(global2local_out), local_mask_out,
              local_points_out, local_orient_out, local_cert_out, local_ts_out, local_feat_out, point_orientations,
              point_certainties, geo_features};
  hipLaunchKernelGGL(k_window_gather, dim3((unsigned)((nu + 1 + 255) / 256)), dim3(256), 0, s, a, flag, pos, o, counts);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

// ---- neural-point insertion -----------------------------------------------------------------------------------------
extern "C" int64_t clid_map_insert_workspace_bytes(int32_t n) {
  if (n <= 0) return 1024;
  return (int64_t)(3 * align256((size_t)n * 4) + align256((size_t)n * 8) + align256(pool_scan_bytes(n)) + 256);
}

extern "C" int clid_map_insert(const float* samples, int32_t n, int64_t* buffer_pt_index, int64_t buffer_size, float resolution,
                               float* neural_points, float* point_orientations, int32_t* ts_create, int32_t* ts_update,
                               float* certainties, int64_t base, const float* travel_dist, int32_t cur_ts, int32_t test_on,
                               int32_t temporal, float far_dist2, float diff_travel, int64_t* count_out, void* workspace,
                               const int64_t* sample_idx, const int64_t* n_dev, float* features_zero, void* stream) {
  if (n < 0 || !buffer_pt_index || buffer_size <= 0 || buffer_size >= (1LL << 30) || !count_out || !workspace || base < 0 ||
      (n > 0 && (!samples || !neural_points || !point_orientations || !ts_create || !ts_update || !certainties)) ||
      (test_on && temporal && !travel_dist)) {
    clid_set_error("clid_map_insert: bad argument");
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  long long* counts = reinterpret_cast<long long*>(count_out);
  if (n == 0) return hipMemsetAsync(counts, 0, sizeof(long long), s) == hipSuccess ? CLID_OK : CLID_E_HIP;
  char* ws = static_cast<char*>(workspace);
  int* phys = reinterpret_cast<int*>(ws);
  int* flag = reinterpret_cast<int*>(ws + align256((size_t)n * 4));
  int* pos = reinterpret_cast<int*>(ws + 2 * align256((size_t)n * 4));
  long long* held = reinterpret_cast<long long*>(ws + 3 * align256((size_t)n * 4));
  void* cub = ws + 3 * align256((size_t)n * 4) + align256((size_t)n * 8);
  InsertArgs a{samples, n, reinterpret_cast<const long long*>(sample_idx), reinterpret_cast<const long long*>(n_dev),
               reinterpret_cast<long long*>(buffer_pt_index), (int)buffer_size, neural_points, point_orientations,
               ts_create, ts_update, certainties, reinterpret_cast<float4*>(features_zero), travel_dist, (long long)base, test_on, temporal, cur_ts, resolution, far_dist2,
               diff_travel};
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_insert_probe, dim3(blocks), dim3(256), 0, s, a, phys, held, flag);
  scan_exclusive(flag, pos, (long long)n, cub, s);
  hipLaunchKernelGGL(k_insert_claim, dim3(blocks), dim3(256), 0, s, a, phys);
  hipLaunchKernelGGL(k_insert_commit, dim3(blocks), dim3(256), 0, s, a, phys, held, flag, pos, counts);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

// ---- raw-point map ----------------------------------------------------------------------------------------------------
extern "C" int64_t clid_cloud_workspace_bytes(int64_t n_total) {
  if (n_total <= 0) return 1024;
  return (int64_t)(2 * align256((size_t)n_total * 4) + align256(pool_scan_bytes(n_total)) + 256);
}

extern "C" int clid_cloud_update(const float* map_points, int64_t n_map, const float* samples, int64_t n_samples,
                                 const int64_t* table_old, int64_t* table_new, int64_t buffer_size, float resolution,
                                 const double* sensor_pos_host, double map_size, int32_t pos_is_f64, float* points_out,
                                 int64_t* counts_out, void* workspace, const int64_t* sample_idx, const int64_t* n_samples_dev,
                                 void* stream) {
  const long long n = n_map + n_samples;
  if (n_map < 0 || n_samples < 0 || n >= (1LL << 31) || !table_old || !table_new || table_old == table_new || buffer_size <= 0 ||
      buffer_size >= (1LL << 30) || !sensor_pos_host || !counts_out || !workspace || (n_map > 0 && !map_points) ||
      (n_samples > 0 && !samples) || (n > 0 && !points_out)) {
    clid_set_error("clid_cloud_update: bad argument");
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  long long* counts = reinterpret_cast<long long*>(counts_out);
  if (hipMemsetAsync(counts, 0, 2 * sizeof(long long), s) != hipSuccess ||
      hipMemsetAsync(table_new, 0xFF, (size_t)buffer_size * sizeof(int64_t), s) != hipSuccess) {
    clid_set_error("clid_cloud_update: memset failed");
    return CLID_E_HIP;
  }
  if (n == 0) return CLID_OK;
  char* ws = static_cast<char*>(workspace);
  int* flag = reinterpret_cast<int*>(ws);
  int* pos = reinterpret_cast<int*>(ws + align256((size_t)n * 4));
  void* cub = ws + 2 * align256((size_t)n * 4);
  CloudArgs a{map_points, n_map, samples, n_samples, reinterpret_cast<const long long*>(sample_idx),
              reinterpret_cast<const long long*>(n_samples_dev), reinterpret_cast<const long long*>(table_old),
              reinterpret_cast<long long*>(table_new), (int)buffer_size, resolution, sensor_pos_host[0], sensor_pos_host[1],
              sensor_pos_host[2], map_size, pos_is_f64};
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_cloud_flags, dim3(blocks < kFlagBlocks ? blocks : kFlagBlocks), dim3(256), 0, s, a, flag, counts);
  scan_exclusive(flag, pos, (long long)n, cub, s);
  hipLaunchKernelGGL(k_cloud_scatter, dim3(blocks), dim3(256), 0, s, a, flag, pos, points_out, counts);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

// ---- recreate_hash / prune_map entries ---------------------------------------------------------------------------------
extern "C" int clid_map_rehash(const float* points, const int64_t* idx, int32_t m, float resolution, int64_t* buffer_pt_index,
                               int64_t buffer_size, void* stream) {
  if (m < 0 || !(resolution > 0.f) || !buffer_pt_index || buffer_size <= 0 || buffer_size >= (1LL << 30) || (m > 0 && !points)) {
    clid_set_error("clid_map_rehash: bad argument");
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(buffer_pt_index, 0xFF, (size_t)buffer_size * 8, s) != hipSuccess) {  // -1 everywhere (:858-860)
    clid_set_error("clid_map_rehash: table reset failed");
    return CLID_E_HIP;
  }
  if (m == 0) return CLID_OK;
  const unsigned blocks = (unsigned)((m + 255) / 256);
  long long* table = reinterpret_cast<long long*>(buffer_pt_index);
  const long long* list = reinterpret_cast<const long long*>(idx);
  hipLaunchKernelGGL(k_rehash_claim, dim3(blocks), dim3(256), 0, s, points, list, m, resolution, table, (int)buffer_size);
  hipLaunchKernelGGL(k_rehash_commit, dim3(blocks), dim3(256), 0, s, points, list, m, resolution, table, (int)buffer_size);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

extern "C" int clid_map_gather(const int64_t* idx, int32_t m, int64_t pad_src_row, const float* points, const float* orient,
                               const int32_t* ts_create, const int32_t* ts_update, const float* cert, const float* feat,
                               float* points_out, float* orient_out, int32_t* ts_create_out, int32_t* ts_update_out,
                               float* cert_out, float* feat_out, void* stream) {
  if (m < 0 || pad_src_row < 0 || !feat || !feat_out ||
      (m > 0 && (!idx || !points || !orient || !ts_create || !ts_update || !cert || !points_out || !orient_out || !ts_create_out ||
                 !ts_update_out || !cert_out))) {
    clid_set_error("clid_map_gather: bad argument");
    return CLID_E_ARG;
  }
  MapRows a{points, reinterpret_cast<const float4*>(orient), ts_create, ts_update, cert, reinterpret_cast<const float4*>(feat)};
  MapRowsOut o{points_out, reinterpret_cast<float4*>(orient_out), ts_create_out, ts_update_out, cert_out,
               reinterpret_cast<float4*>(feat_out)};
  const long long threads = 2LL * ((long long)m + 1);
  hipLaunchKernelGGL(k_map_gather, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(idx), m, (long long)pad_src_row, a, o);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

extern "C" int64_t clid_map_prune_workspace_bytes(int64_t n) { return clid_new_sample_workspace_bytes(n); }

extern "C" int clid_map_prune_select(const int32_t* ts_update, const float* cert, int64_t n, const float* travel_dist,
                                     int32_t cur_ts, float certainty_thre, float diff_travel_dist, int32_t global_prune,
                                     int64_t* keep_idx_out, int64_t* count_out, void* workspace, void* stream) {
  if (n < 0 || n >= (1LL << 31) || !count_out || !workspace || cur_ts < 0 ||
      (n > 0 && (!ts_update || !cert || !keep_idx_out || (!global_prune && !travel_dist)))) {
    clid_set_error("clid_map_prune_select: bad argument");
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  long long* count = reinterpret_cast<long long*>(count_out);
  if (n == 0) return hipMemsetAsync(count, 0, sizeof(long long), s) == hipSuccess ? CLID_OK : CLID_E_HIP;
  char* ws = static_cast<char*>(workspace);
  int* flag = reinterpret_cast<int*>(ws);
  int* pos = reinterpret_cast<int*>(ws + align256((size_t)n * 4));
  void* cub = ws + 2 * align256((size_t)n * 4);
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_prune_flags, dim3(blocks), dim3(256), 0, s, ts_update, cert, (int)n, travel_dist, cur_ts, certainty_thre,
                     diff_travel_dist, global_prune, flag);
  scan_exclusive(flag, pos, (long long)n, cub, s);
  hipLaunchKernelGGL(k_new_sample_list, dim3(blocks), dim3(256), 0, s, flag, pos, (int)n, 0LL,
                     reinterpret_cast<long long*>(keep_idx_out), count, (const long long*)nullptr);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}
