// Developer tool: calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on THIS repo's access patterns (MI355X_MICROARCH.md,
// HBM section: the x2 factor is established for wide coalesced 16 B/lane streams only; "other access widths and
// WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").  Working set 2 GiB,
// far past the 256 MiB Infinity Cache, every line touched at most ~once.
//   k_stream   coalesced 16 B/lane read of 1 GiB                      (anchor: the guide's 1/2 finding)
//   k_gather   2^24 lanes, ONE random 16-B load each from 2 GiB        (the decode / search kernels' reads)
//   k_scatter  2^24 lanes, fp32 atomicAdd, 8 lanes per random 32-B row (the decode kernel's gradient scatter)
// Run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/calib_fetch.bin   (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); exit(1);} } while (0)
__device__ __forceinline__ unsigned long long mix(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k;
}
__global__ void k_stream(const int4* __restrict__ t, size_t n, int* out) {
  int acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += t[i].x;
  if (acc == 12345) out[0] = acc;
}
__global__ void k_gather(const int4* __restrict__ t, unsigned long long mask, int* out) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int4 v = t[mix(i + 1) & mask];
  if (v.x == 12345) out[0] = v.w;
}
__global__ void k_scatter(float* g, unsigned long long row_mask) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long row = mix((i >> 3) + 7) & row_mask;  // 8 consecutive lanes share one 32-B row
  atomicAdd(&g[row * 8 + (i & 7)], 1.0f);
}
int main() {
  const size_t entries = (size_t)1 << 27;  // 2 GiB of int4
  int4* tab; int* out;
  CK(hipMalloc(&tab, entries * sizeof(int4)));
  CK(hipMemset(tab, 1, entries * sizeof(int4)));
  CK(hipMalloc(&out, 64));
  CK(hipDeviceSynchronize());
  const size_t n_stream = (size_t)1 << 26;  // 1 GiB read
  hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, tab, n_stream, out);
  const unsigned n_lanes = 1u << 24;
  hipLaunchKernelGGL(k_gather, dim3(n_lanes / 256), dim3(256), 0, 0, tab, (unsigned long long)entries - 1, out);
  hipLaunchKernelGGL(k_scatter, dim3(n_lanes / 256), dim3(256), 0, 0, reinterpret_cast<float*>(tab),
                     (unsigned long long)(entries * 16 / 32) - 1);
  CK(hipDeviceSynchronize());
  printf("k_stream  requested %.1f MiB (16 B/lane coalesced)\n", n_stream * 16.0 / 1048576.0);
  printf("k_gather  %u gathers x 16 B = %.1f MiB requested; x32 B = %.1f, x64 B = %.1f, x128 B = %.1f MiB of lines\n", n_lanes,
         n_lanes * 16.0 / 1048576.0, n_lanes * 32.0 / 1048576.0, n_lanes * 64.0 / 1048576.0, n_lanes * 128.0 / 1048576.0);
  printf("k_scatter %u atomics = %u rows x 32 B = %.1f MiB of rows; x64 B = %.1f MiB\n", n_lanes, n_lanes / 8,
         (n_lanes / 8) * 32.0 / 1048576.0, (n_lanes / 8) * 64.0 / 1048576.0);
  return 0;
}
