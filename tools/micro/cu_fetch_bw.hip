// What does ONE workgroup get from L2 / Infinity Cache when it streams a few hundred KB with 16-B loads?  The fused
// attention + output-projection launch of the PLM loop (csrc/hsp_mhaproj.hip) reads ~600 KB per workgroup (K, V of
// four heads + the whole 276 x 276 projection): its floor is this number, not MFMA time.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/cu_fetch_bw.hip -o /tmp/fetchbw && /tmp/fetchbw
// Variants: region shared by all workgroups (the projection weights: L2 hits after the first touch per XCD) or
// private per workgroup (K / V of one utterance); U loads in flight per lane; 4 / 8 / 16 waves; 64 ... 256 workgroups;
// `dep`: a second kernel runs first and dirties L2 (graph-like back-to-back launches: nothing of this launch's
// operands is in L2 when it starts).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ void fetch(const f4* __restrict__ base, size_t wg_stride_f4, int n_f4_per_wg, float* out) {
  const f4* p = base + (size_t)blockIdx.x * wg_stride_f4;
  const int nthr = blockDim.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < n_f4_per_wg; i += nthr * U) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int j = i + u * nthr;
      v[u] = p[j < n_f4_per_wg ? j : threadIdx.x];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

__global__ void dirty(float* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.f;
}

template <int U>
float run(const f4* buf, size_t stride_f4, int n_f4, float* out, int wgs, int threads, float* scratch, size_t nscr) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(dirty, dim3(1024), dim3(256), 0, 0, scratch, nscr);   // 64 MB through L2: evicts the operands
    hipEventRecord(e0);
    hipLaunchKernelGGL(fetch<U>, dim3(wgs), dim3(threads), 0, 0, buf, stride_f4, n_f4, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best * 1e3f;
}

int main() {
  const size_t total = 512ull << 20;
  float *buf, *out, *scr;
  hipMalloc(&buf, total);
  hipMalloc(&out, 64);
  const size_t nscr = 16ull << 20;
  hipMalloc(&scr, nscr * 4);
  hipMemset(buf, 0, total);
  hipMemset(scr, 0, nscr * 4);
  printf("%-8s %6s %5s %3s %8s | %8s %10s %10s\n", "region", "KB/wg", "wgs", "U", "threads", "us", "GB/s/wg", "TB/s all");
  for (int shared = 0; shared < 2; ++shared)
    for (int kb : {300, 600})
      for (int wgs : {64, 112, 208, 256, 512})
        for (int threads : {256, 512}) {
          const int n_f4 = kb * 1024 / 16;
          const size_t stride = shared ? 0 : (size_t)n_f4;
          float us4 = run<4>((const f4*)buf, stride, n_f4, out, wgs, threads, scr, nscr);
          float us8 = run<8>((const f4*)buf, stride, n_f4, out, wgs, threads, scr, nscr);
          float us16 = run<16>((const f4*)buf, stride, n_f4, out, wgs, threads, scr, nscr);
          for (auto pr : {std::make_pair(4, us4), std::make_pair(8, us8), std::make_pair(16, us16)})
            printf("%-8s %6d %5d %3d %8d | %8.2f %10.1f %10.2f\n", shared ? "shared" : "private", kb, wgs, pr.first, threads,
                   pr.second, kb * 1.024e-3 / (pr.second * 1e-6) * 1e-3, (double)kb * 1024 * wgs / (pr.second * 1e-6) * 1e-12);
        }
  return 0;
}
