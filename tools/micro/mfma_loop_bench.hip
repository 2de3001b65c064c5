// What a consumer-wave inner loop of the conv kernel can reach on gfx950: 4 x v_mfma_f32_32x32x2_f32 per step on a
// 2 x 2 register tile, with the step's operand traffic added piece by piece:
//   V0 MFMAs only | V1 + 4 ds_read_b32 (prefetch distance one step) + s_waitcnt | V2 + 2 VALU address adds
//   V3 + scalar bookkeeping and a branch (the tap wrap of the real loop) | V4 = V1 with a 3 x 2 tile (6 MFMAs, 5 reads)
//   V5 = V1 without the wait (wrong data: is it the wait?) | V6 = V1 with the reads between the MFMAs
//   V10 / V11 / V12 = the conv kernel's round-2 loop (trip of two steps, immediates, scalar walk) with 2 / 1 / 0
//   VALU address adds per trip | V13 / V14 = V10 with the adds between the MFMAs of a group
//   V8 = the same 16 B per lane as 2 x ds_read_b64 | V9 = as 1 x ds_read_b128
// at 1 and 2 consumer waves per SIMD, 256 workgroups (one per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_loop_bench.hip -o /tmp/mfma_loop_bench && /tmp/mfma_loop_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void rd(float& d, unsigned addr, int) { asm volatile("ds_read_b32 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
__device__ __forceinline__ void rd80(float& d, unsigned addr) { asm volatile("ds_read_b32 %0, %1 offset:0x80" : "=v"(d) : "v"(addr) : "memory"); }
__device__ __forceinline__ void rd100(float& d, unsigned addr) { asm volatile("ds_read_b32 %0, %1 offset:0x100" : "=v"(d) : "v"(addr) : "memory"); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void rd64(f32x2& d, unsigned addr) { asm volatile("ds_read_b64 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
__device__ __forceinline__ void rd128(f32x4& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }

template <int V>
__global__ __launch_bounds__(512) void loop(float* out, int iters, int wrap, int stride_a, int stride_b) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1.0f + i * 1e-7f;
  __syncthreads();
  f32x16 c00, c01, c10, c11, c20, c21;
#pragma unroll
  for (int r = 0; r < 16; ++r) c00[r] = c01[r] = c10[r] = c11[r] = c20[r] = c21[r] = 0.0f;
  const int lane = threadIdx.x & 63;
  unsigned pa = lane * 4, pb = 8192 + lane * 4;
  float a0 = 1.0f, a1 = 1.0f, a2 = 1.0f, b0 = 1.0f, b1 = 1.0f, na0, na1, na2, nb0, nb1;
  int tap = 0, ci = 0;
  if (V >= 1 && V < 10) {
    rd(a0, pa, 0); rd80(a1, pa); rd(b0, pb, 0); rd80(b1, pb);
    if (V == 4) rd100(a2, pa);
  }
  if (V == 13 || V == 14) {
    // V10 with the two VALU adds issued between the MFMAs of the first group (V13: after the first, V14: after the
    // third) instead of after the group: the adds then cost their own cycles, not a drain of the matrix pipe
    int offA = 0, pairB = 0, tapoff = 0;
    unsigned va = pa, vb = pb;
    rd(a0, va, 0); rd80(a1, va); rd(b0, vb, 0); rd80(b1, vb);
    for (int it = 0; it < iters; it += 2) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("ds_read_b32 %0, %1 offset:0x400" : "=v"(na0) : "v"(va) : "memory");
      asm volatile("ds_read_b32 %0, %1 offset:0x480" : "=v"(na1) : "v"(va) : "memory");
      asm volatile("ds_read_b32 %0, %1 offset:0x600" : "=v"(nb0) : "v"(vb) : "memory");
      asm volatile("ds_read_b32 %0, %1 offset:0x680" : "=v"(nb1) : "v"(vb) : "memory");
      int t;
      asm volatile(
          "s_add_i32 %[oa], %[oa], 0x800\n\t"
          "s_add_i32 %[pb], %[pb], 0xc00\n\t"
          "s_cmp_eq_u32 %[pb], %[span]\n\t"
          "s_cselect_b32 %[pb], 0, %[pb]\n\t"
          "s_cselect_b32 %[t], %[tap], 0\n\t"
          "s_add_i32 %[to], %[to], %[t]"
          : [oa] "+s"(offA), [pb] "+s"(pairB), [to] "+s"(tapoff), [t] "=&s"(t)
          : [span] "s"(stride_a * 24), [tap] "s"(wrap * 4)
          : "scc");
      const bool more = it + 2 < iters;
      const unsigned sa = (unsigned)((more ? offA : 0) & 4095), sb = (unsigned)((more ? pairB + tapoff : 0) & 4095);
      __builtin_amdgcn_sched_barrier(0);
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c00, 0, 0, 0);
      if (V == 13) {
        __builtin_amdgcn_sched_barrier(0);
        va = pa + sa; vb = pb + sb;
        __builtin_amdgcn_sched_barrier(0);
      }
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c10, 0, 0, 0);
      if (V == 14) {
        __builtin_amdgcn_sched_barrier(0);
        va = pa + sa; vb = pb + sb;
        __builtin_amdgcn_sched_barrier(0);
      }
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c11, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(na0), "+v"(na1), "+v"(nb0), "+v"(nb1)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      rd(a0, va, 0); rd80(a1, va); rd(b0, vb, 0); rd80(b1, vb);
      __builtin_amdgcn_sched_barrier(0);
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(na0, nb0, c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(na0, nb1, c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(na1, nb0, c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(na1, nb1, c11, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else
  if (V >= 10) {
    // the conv kernel's loop after round 2: a trip of two steps, second step through immediates, a branch-free
    // scalar walk per trip and NV = V - 10 ... VALU address adds per trip (2 = the kernel, 1, 0)
    int offA = 0, pairB = 0, tapoff = 0;
    unsigned va = pa, vb = pb;
    rd(a0, va, 0); rd80(a1, va); rd(b0, vb, 0); rd80(b1, vb);
    for (int it = 0; it < iters; it += 2) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("ds_read_b32 %0, %1 offset:0x400" : "=v"(na0) : "v"(va) : "memory");
      asm volatile("ds_read_b32 %0, %1 offset:0x480" : "=v"(na1) : "v"(va) : "memory");
      asm volatile("ds_read_b32 %0, %1 offset:0x600" : "=v"(nb0) : "v"(vb) : "memory");
      asm volatile("ds_read_b32 %0, %1 offset:0x680" : "=v"(nb1) : "v"(vb) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c11, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      int t;
      asm volatile(
          "s_add_i32 %[oa], %[oa], 0x800\n\t"
          "s_add_i32 %[pb], %[pb], 0xc00\n\t"
          "s_cmp_eq_u32 %[pb], %[span]\n\t"
          "s_cselect_b32 %[pb], 0, %[pb]\n\t"
          "s_cselect_b32 %[t], %[tap], 0\n\t"
          "s_add_i32 %[to], %[to], %[t]"
          : [oa] "+s"(offA), [pb] "+s"(pairB), [to] "+s"(tapoff), [t] "=&s"(t)
          : [span] "s"(stride_a * 24), [tap] "s"(wrap * 4)
          : "scc");
      const bool more = it + 2 < iters;
      if (V == 10) {
        va = pa + (unsigned)((more ? offA : 0) & 4095);
        vb = pb + (unsigned)((more ? pairB + tapoff : 0) & 4095);
      } else if (V == 11) {
        vb = pb + (unsigned)((more ? pairB + tapoff : 0) & 4095);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(na0), "+v"(na1), "+v"(nb0), "+v"(nb1)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      rd(a0, va, 0); rd80(a1, va); rd(b0, vb, 0); rd80(b1, vb);
      __builtin_amdgcn_sched_barrier(0);
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(na0, nb0, c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(na0, nb1, c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(na1, nb0, c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(na1, nb1, c11, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else
  if (V == 8 || V == 9) {
    unsigned wa = lane * 16, wb = 8192 + lane * 16;
    f32x2 x0 = {1.0f, 1.0f}, x1 = x0, y0 = x0, y1 = x0;
    f32x4 z0 = {1.0f, 1.0f, 1.0f, 1.0f}, z1 = z0;
#define WSTEP(X0, X1, Z, Y0, Y1, W)                                                        \
    {                                                                                      \
      if (V == 8) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X0), "+v"(X1)::"memory");     \
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Z)::"memory");                       \
      __builtin_amdgcn_sched_barrier(0);                                                   \
      if (V == 8) { rd64(Y0, wa); rd64(Y1, wb); } else rd128(W, wa);                       \
      __builtin_amdgcn_sched_barrier(0);                                                   \
      const float p0 = V == 8 ? X0[0] : Z[0], p1 = V == 8 ? X0[1] : Z[1];                  \
      const float q0 = V == 8 ? X1[0] : Z[2], q1 = V == 8 ? X1[1] : Z[3];                  \
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(p0, q0, c00, 0, 0, 0);                    \
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(p0, q1, c01, 0, 0, 0);                    \
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(p1, q0, c10, 0, 0, 0);                    \
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(p1, q1, c11, 0, 0, 0);                    \
    }
    for (int it = 0; it < iters; it += 2) {
      WSTEP(x0, x1, z0, y0, y1, z1)
      WSTEP(y0, y1, z1, x0, x1, z0)
    }
  } else
  // two steps per trip with the register sets swapped by hand (no copies), like the real loop
  {
#define STEP(A0, A1, A2, B0, B1, N0, N1, N2, M0, M1)                                                     \
  {                                                                                                      \
    if (V >= 1) {                                                                                        \
      if (V != 5) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A0), "+v"(A1), "+v"(B0), "+v"(B1), "+v"(A2)::"memory"); \
      unsigned qa = pa, qb = pb;                                                                         \
      if (V == 2 || V == 3) {                                                                            \
        if (V == 3) {                                                                                    \
          ++tap;                                                                                         \
          if (tap == wrap) { tap = 0; ++ci; }                                                            \
          qa = pa + ((ci * stride_a + tap * 64) & 4095);                                                 \
          qb = pb + ((ci * stride_b + tap * 4) & 4095);                                                  \
        } else {                                                                                         \
          qa = pa + ((it * stride_a) & 4095);                                                            \
          qb = pb + ((it * stride_b) & 4095);                                                            \
        }                                                                                                \
      }                                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
      if (V != 6) { rd(N0, qa, 0); rd80(N1, qa); rd(M0, qb, 0); rd80(M1, qb); }                          \
      if (V == 4) rd100(N2, qa);                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
    }                                                                                                    \
    c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B0, c00, 0, 0, 0);                                    \
    if (V == 6) { __builtin_amdgcn_sched_barrier(0); rd(N0, pa, 0); __builtin_amdgcn_sched_barrier(0); } \
    c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B1, c01, 0, 0, 0);                                    \
    if (V == 6) { __builtin_amdgcn_sched_barrier(0); rd80(N1, pa); __builtin_amdgcn_sched_barrier(0); }  \
    c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B0, c10, 0, 0, 0);                                    \
    if (V == 6) { __builtin_amdgcn_sched_barrier(0); rd(M0, pb, 0); __builtin_amdgcn_sched_barrier(0); } \
    c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B1, c11, 0, 0, 0);                                    \
    if (V == 6) { __builtin_amdgcn_sched_barrier(0); rd80(M1, pb); __builtin_amdgcn_sched_barrier(0); }  \
    if (V == 4) {                                                                                        \
      c20 = __builtin_amdgcn_mfma_f32_32x32x2f32(A2, B0, c20, 0, 0, 0);                                  \
      c21 = __builtin_amdgcn_mfma_f32_32x32x2f32(A2, B1, c21, 0, 0, 0);                                  \
    }                                                                                                    \
  }
  na0 = na1 = na2 = nb0 = nb1 = 1.0f;
  for (int it = 0; it < iters; it += 2) {
    STEP(a0, a1, a2, b0, b1, na0, na1, na2, nb0, nb1)
    STEP(na0, na1, na2, nb0, nb1, a0, a1, a2, b0, b1)
  }
  }
  float s = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += c00[r] + c01[r] + c10[r] + c11[r] + c20[r] + c21[r];
  if (s == 123.456f) out[threadIdx.x] = s;
}

template <int V>
void run(int waves_per_simd, int iters, float* d) {
  const int threads = 64 * 4 * waves_per_simd, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(loop<V>, dim3(blocks), dim3(threads), 65536, 0, d, iters / 10, 11, 256, 128);
  hipEventRecord(e0);
  hipLaunchKernelGGL(loop<V>, dim3(blocks), dim3(threads), 65536, 0, d, iters, 11, 256, 128);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = 2.0 * 32 * 32 * 2 * (V == 4 ? 6.0 : 4.0) * (double)iters * (threads / 64) * blocks;
  printf("V%d waves/SIMD %d: %8.1f ms %7.1f TFLOP/s (%.1f %% of 157.3)\n", V, waves_per_simd, ms, flop / ms / 1e9,
         flop / ms / 1e9 / 1.573);
}

int main() {
  float* d;
  hipMalloc(&d, 4096);
  hipFuncSetAttribute((const void*)loop<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)loop<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)loop<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)loop<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)loop<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)loop<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)loop<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)loop<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)loop<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)loop<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)loop<11>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)loop<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)loop<13>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)loop<14>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int w : {1, 2}) {
    const int iters = 800000 / w;
    run<0>(w, iters, d);
    run<0>(w, iters, d);
    run<1>(w, iters, d);
    run<2>(w, iters, d);
    run<3>(w, iters, d);
    run<4>(w, iters, d);
    run<5>(w, iters, d);
    run<6>(w, iters, d);
    run<8>(w, iters, d);
    run<9>(w, iters, d);
    run<10>(w, iters, d);
    run<11>(w, iters, d);
    run<12>(w, iters, d);
    run<13>(w, iters, d);
    run<14>(w, iters, d);
  }
  return 0;
}
