// Micro-benchmark (developer tool): fp32 atomic-add throughput on gfx950 by memory scope and lane arrangement.
//   scope  agent      : global_atomic_add_f32 ... sc1  (performed at the memory side: every XCD sees it)
//          workgroup  : no sc1 -> performed in the issuing XCD's L2; only coherent inside that XCD, so each
//                       XCD accumulates into its own private copy (selected by HW_REG_XCC_ID) and the
//                       kernel-end write-back makes the copies visible to the next kernel
//   layout rows8      : 8 consecutive lanes on the 8 floats of one 32-byte row
//          half16     : lane (q = lane&15, g = lane>>4): lanes g=0/1 own floats 0-3 / 4-7 of query q's row, 4
//                       atomic instructions per lane (lanes of one row are 16 apart)
//          scatter    : every lane its own row
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench_atomic.hip -o tools/ubench_atomic.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s @%d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v;
}

template <int SCOPE>
__device__ __forceinline__ void add(float* p, float v) {
  if (SCOPE == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (SCOPE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// rows: number of 8-float rows per copy; copy stride = rows*8 floats
template <int SCOPE, int LAYOUT>
__global__ void k_atomic(float* g, unsigned rows, int iters, int per_xcd) {
  unsigned h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
  const int lane = threadIdx.x & 63;
  float* base = g + (per_xcd ? (size_t)xcc_id() * rows * 8 : 0);
  for (int i = 0; i < iters; ++i) {
    h = h * 1664525u + 1013904223u;
    if (LAYOUT == 0) {
      const unsigned hr = __shfl(h, lane & ~7, 64);
      const unsigned row = (hr >> 8) % rows;
      add<SCOPE>(&base[row * 8 + (lane & 7)], 1.0f);
    } else if (LAYOUT == 1) {
      const unsigned hr = __shfl(h, lane & 15, 64);
      const unsigned row = (hr >> 8) % rows;
      const int half = (lane >> 4) & 1;
      if ((lane >> 5) == (i & 1)) {  // only lanes g = 0,1 (or 2,3) are active, as in the tile kernel's first layout
#pragma unroll
        for (int e = 0; e < 4; ++e) add<SCOPE>(&base[row * 8 + half * 4 + e], 1.0f);
      }
    } else if (LAYOUT == 3 || LAYOUT == 4) {  // 64-byte rows (two 32-byte rows of the buffer): 9 or 16 of 16 lanes active
      const unsigned hr = __shfl(h, lane & ~15, 64);
      const unsigned row = ((hr >> 8) % (rows / 2)) * 2;
      if (LAYOUT == 4 || (lane & 15) < 9) add<SCOPE>(&base[row * 8 + (lane & 15)], 1.0f);
    } else {
      const unsigned row = (h >> 8) % rows;
      add<SCOPE>(&base[row * 8 + (i & 7)], 1.0f);
    }
  }
}

template <int SCOPE, int LAYOUT>
void run(const char* name, float* g, unsigned rows, int per_xcd) {
  const int blocks = 820, threads = 256, iters = 8;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(g, 0, (size_t)rows * 8 * 8 * sizeof(float)));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_atomic<SCOPE, LAYOUT>), dim3(blocks), dim3(threads), 0, 0, g, rows, iters, per_xcd);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best;
  }
  double n = (double)blocks * threads * iters;
  if (LAYOUT == 1) n = n / 2 * 4;
  if (LAYOUT == 3) n = n / 16 * 9;
  // correctness: the copies must sum to the number of atomics issued
  std::vector<float> host((size_t)rows * 8 * 8);
  CK(hipMemcpy(host.data(), g, host.size() * sizeof(float), hipMemcpyDeviceToHost));
  double tot = 0;
  for (float v : host) tot += v;
  printf("%-34s per_xcd=%d: %7.1f us, %7.1f G lane-atomics/s, sum %s (%.0f / %.0f)\n", name, per_xcd, best * 1e3,
         n / best / 1e6, tot == n ? "OK" : "MISMATCH", tot, n);
}

// LDS float atomics: every lane adds into a random slot of an `n`-float LDS array, `iters` times
__global__ void __launch_bounds__(256) k_lds_atomic(float* out, int n, int iters, int rtn) {
  extern __shared__ float sm[];
  for (int i = threadIdx.x; i < n; i += 256) sm[i] = 0.f;
  __syncthreads();
  unsigned h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    h = h * 1664525u + 1013904223u;
    const unsigned a = (h >> 8) % (unsigned)n;
    if (rtn) acc += __hip_atomic_fetch_add(&sm[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(&sm[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  float s = acc * 1e-30f;
  for (int i = threadIdx.x; i < n; i += 256) s += sm[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
static void lds_bench() {
  float* out;
  CK(hipMalloc(&out, 1024 * 256 * sizeof(float)));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int rtn = 0; rtn < 2; ++rtn)
    for (int n : {64, 1024, 8192}) {
      const int blocks = 1024, iters = 256;
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_lds_atomic, dim3(blocks), dim3(256), n * sizeof(float), 0, out, n, iters, rtn);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
      }
      const double ops = (double)blocks * 256 * iters;
      printf("LDS ds_add_f32 rtn=%d n=%5d: %7.1f us, %.2f lane-atomics/clk/CU (2.4 GHz, 256 CU)\n", rtn, n, best * 1e3,
             ops / (best * 1e-3) / 2.4e9 / 256);
    }
}

// every block adds a [n_lines x 16]-float vector into the SAME n_lines 64-byte lines (a decoder-gradient reduction by
// atomics): `copies` > 1 spreads the blocks over that many private copies (blockIdx %% copies)
__global__ void __launch_bounds__(256) k_same_lines(float* g, int n_lines, int copies) {
  float* base = g + (size_t)(blockIdx.x % copies) * n_lines * 16;
  for (int i = threadIdx.x; i < n_lines * 16; i += 256) atomicAdd(&base[i], 1.0f);
}
static void same_line_bench(float* g) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int copies : {1, 8, 32})
    for (int blocks : {410, 1640}) {
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(g, 0, 64 * 53 * 16 * sizeof(float)));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_same_lines, dim3(blocks), dim3(256), 0, 0, g, 53, copies);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
      }
      printf("same-line atomics: %4d blocks x 53 lines (64 B requests), %2d copies: %6.1f us\n", blocks, copies, best * 1e3);
    }
}

__global__ void k_xcc(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

int main() {
  const unsigned rows = 25002;
  float* g;
  CK(hipMalloc(&g, (size_t)rows * 8 * 8 * sizeof(float)));
  int* xo;
  CK(hipMalloc(&xo, 64 * sizeof(int)));
  hipLaunchKernelGGL(k_xcc, dim3(64), dim3(64), 0, 0, xo);
  int hx[64];
  CK(hipMemcpy(hx, xo, sizeof(hx), hipMemcpyDeviceToHost));
  printf("xcc of blocks 0..15:");
  for (int i = 0; i < 16; ++i) printf(" %d", hx[i]);
  printf("\n");
  run<0, 0>("agent rows8", g, rows, 0);
  run<0, 0>("agent rows8", g, rows, 1);
  run<1, 0>("workgroup rows8", g, rows, 1);
  run<2, 0>("wavefront rows8", g, rows, 1);
  run<1, 0>("workgroup rows8 (shared copy!)", g, rows, 0);
  run<0, 1>("agent half16", g, rows, 0);
  run<1, 1>("workgroup half16", g, rows, 1);
  run<0, 2>("agent scatter", g, rows, 0);
  run<0, 3>("agent rows16 (9 of 16 lanes)", g, rows, 0);
  run<0, 4>("agent rows16 (16 lanes, 64 B)", g, rows, 0);
  lds_bench();
  same_line_bench(g);
  run<1, 2>("workgroup scatter", g, rows, 1);
  return 0;
}
