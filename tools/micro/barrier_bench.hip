// Cost of one workgroup-barrier round on gfx950, by workgroup size: N rounds of { s_waitcnt; s_barrier } per wave,
// one workgroup per CU (160 KB LDS requested), cycles per round from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/barrier_bench.hip -o /tmp/barrier_bench && /tmp/barrier_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void rounds(unsigned long long* out, int n, int work) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6;
  float acc = threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave < 8)
      for (int k = 0; k < work; ++k) acc = __builtin_fmaf(acc, 1.0001f, 0.5f);   // dependent VALU chain: ~4 cycles each
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 12345.678f) lds[threadIdx.x] = acc;
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 256 * sizeof(unsigned long long));
  hipFuncSetAttribute((const void*)rounds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int n = 2000;
  for (int work : {0, 64, 256}) {
    for (int waves : {1, 2, 4, 8, 11, 12, 16}) {
      for (int lds : {1024, 160 * 1024}) {
        hipLaunchKernelGGL(rounds, dim3(256), dim3(64 * waves), lds, 0, d, n, work);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256);
        hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += (double)v;
        printf("work %3d waves %2d lds %6d B: %8.1f cycles per round\n", work, waves, lds, s / 256 / n);
      }
    }
  }
  return 0;
}
