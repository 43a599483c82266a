// Developer micro-benchmark: cost of a cooperative-groups grid barrier on MI355X at the decode kernel's launch shape
// (410 blocks x 512 threads, ~70 KB LDS per block), to decide whether folding k_adam_all into the decode launch
// behind a grid barrier could beat the ~3 us gap between two dependent launches.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_gridsync.hip -o tools/ubench_gridsync.bin && tools/ubench_gridsync.bin
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>

#include <cstdio>
namespace cg = cooperative_groups;

__global__ void __launch_bounds__(512, 4) k_sync(int n, float* out) {
  extern __shared__ float lds[];
  cg::grid_group g = cg::this_grid();
  float acc = 0.f;
  for (int i = 0; i < n; ++i) {
    acc += lds[(threadIdx.x + i) & 1023];
    g.sync();
  }
  if (acc == 123.f) out[0] = acc;
}

__global__ void k_empty(float* out) {
  if (out[0] == 123.f) out[1] = 1.f;
}

int main() {
  float* d;
  hipMalloc(&d, 1024);
  hipMemset(d, 0, 1024);
  hipStream_t s;
  hipStreamCreate(&s);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int blocks : {256, 410, 512}) {
    for (int n : {1, 101}) {
      void* args[] = {&n, &d};
      hipFuncSetAttribute((const void*)k_sync, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024);
      hipError_t e = hipLaunchCooperativeKernel((const void*)k_sync, dim3(blocks), dim3(512), args, 70 * 1024, s);
      if (e != hipSuccess) {
        printf("blocks %d: cooperative launch refused: %s\n", blocks, hipGetErrorString(e));
        break;
      }
      hipStreamSynchronize(s);
      hipEventRecord(a, s);
      for (int r = 0; r < 20; ++r) hipLaunchCooperativeKernel((const void*)k_sync, dim3(blocks), dim3(512), args, 70 * 1024, s);
      hipEventRecord(b, s);
      hipStreamSynchronize(s);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      printf("blocks %d, %3d grid syncs per launch: %.2f us per launch\n", blocks, n, 1e3f * ms / 20);
    }
  }
  hipEventRecord(a, s);
  for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(k_empty, dim3(410), dim3(512), 0, s, d);
  hipEventRecord(b, s);
  hipStreamSynchronize(s);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  printf("back-to-back dependent empty launches: %.2f us each\n", 1e3f * ms / 200);
  return 0;
}
