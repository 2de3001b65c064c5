// How many workgroups of a given shape does a CU of this chip hold at once?  Every workgroup spins for ~20 us and records
// (start, end, XCC id, CU id); the host counts the maximum overlap per (XCC, CU).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/occupancy_census.hip -o /tmp/census && /tmp/census
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

template <int THREADS, int VREGS>
__global__ __launch_bounds__(THREADS) void census(unsigned long long* out, int spin) {
  extern __shared__ float lds[];
  float r[VREGS];
#pragma unroll
  for (int i = 0; i < VREGS; ++i) r[i] = threadIdx.x * 0.5f + i;
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long t = t0;
  while (t - t0 < (unsigned long long)spin) {
#pragma unroll
    for (int i = 0; i < VREGS; ++i) r[i] = r[i] * 1.0001f + 0.5f;
    t = __builtin_readcyclecounter();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VREGS; ++i) s += r[i];
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    lds[0] = s;
    out[4 * blockIdx.x + 0] = t0;
    out[4 * blockIdx.x + 1] = t;
    out[4 * blockIdx.x + 2] = ((unsigned long long)(xcc & 15) << 32) | hw;
    out[4 * blockIdx.x + 3] = (unsigned long long)lds[0];
  }
}

template <int THREADS, int VREGS>
void run(int blocks, int lds_bytes) {
  unsigned long long* d;
  hipMalloc(&d, 32ull * blocks);
  hipFuncSetAttribute(reinterpret_cast<const void*>(census<THREADS, VREGS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((census<THREADS, VREGS>), dim3(blocks), dim3(THREADS), lds_bytes, 0, d, 50000);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(4 * blocks);
  hipMemcpy(h.data(), d, 32ull * blocks, hipMemcpyDeviceToHost);
  std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
  for (int b = 0; b < blocks; ++b) {
    const unsigned long long id = h[4 * b + 2];
    const unsigned long long cu = (id >> 32) << 16 | ((id >> 8) & 0xf) << 4 | ((id >> 13) & 0x7) << 8;   // xcc, cu_id [11:8], se_id [15:13]
    ev[cu].push_back({h[4 * b], +1});
    ev[cu].push_back({h[4 * b + 1], -1});
  }
  int mx = 0;
  std::map<int, int> hist;
  for (auto& kv : ev) {
    auto& v = kv.second;
    std::sort(v.begin(), v.end());
    int cur = 0, m = 0;
    for (auto& e : v) { cur += e.second; m = std::max(m, cur); }
    hist[m]++;
    mx = std::max(mx, m);
  }
  printf("threads %4d vregs~%3d lds %6d blocks %5d: CUs seen %3zu, max resident per CU %d, histogram:", THREADS, VREGS, lds_bytes, blocks, ev.size(), mx);
  for (auto& kv : hist) printf(" %dx%d", kv.second, kv.first);
  printf("\n");
  hipFree(d);
}

int main() {
  run<512, 64>(1024, 41 * 1024);
  run<512, 64>(1024, 34 * 1024);
  run<512, 64>(1024, 1024);
  run<512, 24>(1024, 1024);
  run<256, 64>(2048, 1024);
  run<256, 64>(2048, 41 * 1024);
  run<768, 64>(1024, 1024);
  return 0;
}
