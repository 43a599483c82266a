// developer tool (GPU box): floor of a two-kernel dependent chain D(t) -> A(t) -> D(t+1) -> ...
//   serial  : both kernels on ONE stream (what clid_mapping_run does today): two launch boundaries per iteration
//   chained : D's on stream X, A's on stream Y, ordered by device-side counters only (release fence + atomic add by every
//             block, acquire after a bounded spin): the next D's dispatch and prologue run beside A
// Both variants carry a real data dependency (A reads what D wrote and vice versa, across XCDs) and verify the result.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_chain.hip -o tools/ubench_chain.bin ; run: tools/ubench_chain.bin [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr long long kSpinLimit = 1 << 20;

struct Chain {
  unsigned long long* wait_cnt;    // NULL: no wait
  unsigned long long wait_target;
  unsigned long long* sig_cnt;     // NULL: no signal
  int* err;
  int fence;                       // 0: every thread releases / acquires at agent scope; 1: one thread per block releases (the others
                                   // wait for their stores, workgroup scope), every thread acquires; 2: as 1, one wave acquires;
                                   // 3: no agent-scope fence at all (NOT coherent across XCDs: cost of counters + polling alone)
};

__device__ __forceinline__ void chain_wait(const Chain& c) {
  if (!c.wait_cnt) return;
  if (threadIdx.x == 0) {
    long long spins = 0;
    while (__hip_atomic_load(c.wait_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c.wait_target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > kSpinLimit) { atomicExch(c.err, 1); break; }
    }
  }
  __syncthreads();
  if (c.fence == 0 || c.fence == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  else if (c.fence == 2) {
    if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
  }
}
__device__ __forceinline__ void chain_signal(const Chain& c) {
  if (!c.sig_cnt) return;
  if (c.fence == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (c.fence == 1 || c.fence == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(c.sig_cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// D: n_d blocks x 256; thread i: g[i] = theta[perm(i)] + 1 (a "gather" from rows another XCD's A block wrote) ; work = extra dependent loads
__global__ void __launch_bounds__(256) k_d(const float* __restrict__ theta, float* __restrict__ g, int n, int work, Chain c) {
  chain_wait(c);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    int j = (int)(((long long)i * 7919 + 13) % n);
    float v = __builtin_nontemporal_load(&theta[j]);
    for (int w = 0; w < work; ++w) { j = (int)(((long long)j * 31 + (int)v) % n); v += __builtin_nontemporal_load(&theta[j]) * 0.f; }
    g[i] = v + 1.f;
  }
  chain_signal(c);
}
// A: n_a blocks x 256; thread i handles elements i, i + stride ...: theta[perm(i)] = g[i]
__global__ void __launch_bounds__(256) k_a(float* __restrict__ theta, const float* __restrict__ g, int n, Chain c) {
  chain_wait(c);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int j = (int)(((long long)i * 7919 + 13) % n);
    theta[j] = g[i];
  }
  chain_signal(c);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  const int n_d = 410, n_a = 237, n = n_d * 256;
  float *theta, *g;
  unsigned long long* cnt;  // [0] = D completions (blocks), [16] = A completions
  int* err;
  CK(hipMalloc(&theta, n * 4)); CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&cnt, 64 * 8)); CK(hipMalloc(&err, 4));
  hipStream_t X, Y;
  CK(hipStreamCreateWithFlags(&X, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&Y, hipStreamNonBlocking));
  for (int work : {0, 4}) {
   for (int fence = 0; fence < 4; ++fence)
    for (int mode = fence ? 1 : 0; mode < 3; ++mode) {  // 0 serial one stream, 1 chained two streams, 2 chained, both kernels on ONE stream (fences' own cost)
      double best = 1e30;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemsetAsync(theta, 0, n * 4, X)); CK(hipMemsetAsync(g, 0, n * 4, X)); CK(hipMemsetAsync(cnt, 0, 64 * 8, X)); CK(hipMemsetAsync(err, 0, 4, X));
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < iters; ++t) {
          Chain cd{nullptr, 0, nullptr, err, fence}, ca{nullptr, 0, nullptr, err, fence};
          if (mode >= 1) {
            cd = Chain{cnt + 16, (unsigned long long)t * n_a, cnt, err, fence};            // D(t) waits for A(t-1), signals cnt[0]
            ca = Chain{cnt, (unsigned long long)(t + 1) * n_d, cnt + 16, err, fence};      // A(t) waits for D(t), signals cnt[16]
          }
          hipLaunchKernelGGL(k_d, dim3(n_d), dim3(256), 0, X, theta, g, n, work, cd);
          hipLaunchKernelGGL(k_a, dim3(n_a), dim3(256), 0, (mode == 1 && t + 1 < iters) ? Y : X, theta, g, n, ca);
        }
        CK(hipStreamSynchronize(X)); CK(hipStreamSynchronize(Y));
        auto t1 = std::chrono::steady_clock::now();
        const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / iters;
        if (us < best) best = us;
        std::vector<float> h(n);
        int herr = 0;
        CK(hipMemcpy(h.data(), theta, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < n; ++i) bad += h[i] != (float)iters;
        if ((bad && fence != 3) || herr) printf("  !! mode %d work %d rep %d: %d wrong values (theta[0] = %g, want %d), err word %d\n", mode, work, rep, bad, h[0], iters, herr);
      }
      printf("{\"fence\": %d, \"mode\": \"%s\", \"dependent_loads_in_D\": %d, \"iters\": %d, \"us_per_iteration\": %.2f}\n",
             fence, mode == 0 ? "serial_one_stream" : (mode == 1 ? "chained_two_streams" : "flags_one_stream"), 1 + work, iters, best);
      fflush(stdout);
    }
  }
  return 0;
}
