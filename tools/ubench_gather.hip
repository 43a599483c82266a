// Micro-benchmark (developer tool): throughput of 16-byte gathers with 64 distinct cache lines per wave
// instruction out of an L2-resident table, and of coalesced / scattered fp32 atomics.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); exit(1);} } while (0)

// variant: only lanes with (lane % keep_mod == 0) load from random lines, the others read line 0
__global__ void k_gather_masked(const int4* __restrict__ tab, unsigned mask, int iters, int* out, int keep_mod, int skip) {
  unsigned h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
  const bool keep = (threadIdx.x % keep_mod) == 0;
  int acc = 0;
  for (int i = 0; i < iters; ++i) {
    int4 v[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      h = h * 1664525u + 1013904223u;
      v[t] = make_int4(0, 0, 0, 0);
      if (keep) v[t] = tab[(h >> 8) & mask];
      else if (!skip) v[t] = tab[0];
    }
#pragma unroll
    for (int t = 0; t < 6; ++t) acc += v[t].x + v[t].w;
    h += acc & 1;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int DEP>
__global__ void k_gather(const int4* __restrict__ tab, unsigned mask, int iters, int* out) {
  unsigned h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
  int acc = 0;
  for (int i = 0; i < iters; ++i) {
    int4 v[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      h = h * 1664525u + 1013904223u;
      v[t] = tab[(h >> 8) & mask];
    }
#pragma unroll
    for (int t = 0; t < 6; ++t) acc += v[t].x + v[t].w;
    if (DEP) h += acc & 1;  // dependent chain between batches
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ void k_atomic(float* g, unsigned rows, int iters, int coalesced) {
  unsigned h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
  const int lane = threadIdx.x & 63;
  for (int i = 0; i < iters; ++i) {
    h = h * 1664525u + 1013904223u;
    unsigned row;
    if (coalesced) {  // 8 consecutive lanes hit the 8 floats of one random row
      unsigned hr = __shfl(h, lane & ~7, 64);
      row = (hr >> 8) % rows;
      atomicAdd(&g[row * 8 + (lane & 7)], 1.0f);
    } else {
      row = (h >> 8) % rows;
      atomicAdd(&g[row * 8 + (i & 7)], 1.0f);
    }
  }
}

int main() {
  const unsigned entries = 1u << 16;  // 1 MB of int4
  int4* tab; int* out; float* g;
  CK(hipMalloc(&tab, entries * sizeof(int4)));
  CK(hipMemset(tab, 1, entries * sizeof(int4)));
  const int blocks = 820, threads = 256, iters = 4;
  CK(hipMalloc(&out, blocks * threads * sizeof(int)));
  const unsigned rows = 25002;
  CK(hipMalloc(&g, rows * 8 * sizeof(float)));
  CK(hipMemset(g, 0, rows * 8 * sizeof(float)));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int dep = 0; dep < 2; ++dep) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(a));
      if (dep) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(threads), 0, 0, tab, entries - 1, iters, out);
      else hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(threads), 0, 0, tab, entries - 1, iters, out);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      const double lanes = (double)blocks * threads * iters * 6;
      printf("gather dep=%d: %.1f us, %.2f G lane-loads/s, %.2f lane-loads/clk/CU (2.1GHz, 256 CU)\n", dep, ms * 1e3,
             lanes / ms / 1e6, lanes / (ms * 1e-3) / 2.1e9 / 256);
    }
  }
  for (int km = 1; km <= 8; km *= 2)
    for (int skip = 0; skip < 2; ++skip) {
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_gather_masked, dim3(blocks), dim3(threads), 0, 0, tab, entries - 1, iters, out, km, skip);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
      }
      printf("gather 1/%d lanes random, others %s: %.1f us\n", km, skip ? "masked off" : "read line 0", best * 1e3);
    }
  for (int co = 0; co < 2; ++co)
    for (int rep = 0; rep < 3; ++rep) {
      const int ait = 5;
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(threads), 0, 0, g, rows, ait, co);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      const double n = (double)blocks * threads * ait;
      printf("atomics coalesced=%d: %.1f us for %.0f atomics, %.1f G/s\n", co, ms * 1e3, n, n / ms / 1e6);
    }
  return 0;
}
