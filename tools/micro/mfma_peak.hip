// Sustained rate of v_mfma_f32_32x32x2_f32 on gfx950 from registers only (no LDS, no memory): what the matrix
// pipe delivers under a second of load, i.e. the ceiling a kernel can reach at the clock the chip actually holds.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
// Prints TFLOP/s for 1, 2, 3, 4 waves per SIMD and, from s_memtime (shader clock) against s_memrealtime (100 MHz),
// the average shader clock during the run.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(1024) void mfma_loop(float* out, unsigned long long* clk, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  float a = a0 + threadIdx.x * 1e-9f, b = b0;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[threadIdx.x] = s;
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = c1 - c0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <int NACC>
void run(int waves_per_simd, int iters, float* d, unsigned long long* dc) {
  const int threads = 64 * 4 * waves_per_simd, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(threads), 0, 0, d, dc, iters / 10, 1.0f, 0.0f);  // warm
  hipEventRecord(e0);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(threads), 0, 0, d, dc, iters, 1.0f, 0.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * blocks);
  hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
  double c = 0, r = 0;
  for (int i = 0; i < blocks; ++i) c += (double)h[2 * i], r += (double)h[2 * i + 1];
  const double flop = 2.0 * 32 * 32 * 2 * 8.0 * NACC * (double)iters * (threads / 64) * blocks;
  printf("acc %d waves/SIMD %d: %8.1f ms %7.1f TFLOP/s | memtime/memrealtime %.3f (x100 MHz = %.0f MHz if memtime counts shader clocks)\n",
         NACC, waves_per_simd, ms, flop / ms / 1e9, c / r, 100.0 * c / r);
}

int main() {
  float* d;
  unsigned long long* dc;
  hipMalloc(&d, 4096);
  hipMalloc(&dc, 2 * 256 * 8);
  for (int rep = 0; rep < 2; ++rep) {
    run<4>(1, 400000, d, dc);
    run<4>(2, 200000, d, dc);
    run<2>(4, 200000, d, dc);
    run<1>(4, 400000, d, dc);
  }
  return 0;
}
