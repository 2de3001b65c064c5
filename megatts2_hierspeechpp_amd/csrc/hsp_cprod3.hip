// Channel product of a frequency-domain conv as a complex GEMM with THREE real products per bin (round 5):
// hsp_cprod3_f32, and the kernel that derives the per-bin weights from a conv's packed taps on the device
// (hsp_dftseg_weight_spectrum_f32).
//
// Between the two transforms of csrc/hsp_dftseg.hip every frequency bin multiplies the [C x C] complex matrix
// conj(rfft(w, 128))[bin] = a + i b into the spectrum X = Xr + i Xi of its Np segments
// (hierspeechpp_speechsynthesizer.py:340-392: the k = 7 / 11 convs of the AMP blocks).  Round 4 ran that as ONE real
// [2C x 2C] block matrix [[a, -b], [b, a]] per bin on the implicit-GEMM conv kernel: 8 C^2 flops per column.  The three-
// product form of a complex multiplication needs 6 C^2:
//     k1 = (a + b) Xr      k2 = a (Xi - Xr)      k3 = b (Xr + Xi)          Yr = k1 - k3      Yi = k1 + k2
// with (a + b), a, b packed once per conv (float64 -> fp32, hsp_dftseg_weight_spectrum_f32) -- 25 % fewer MFMAs and 25 %
// fewer weight bytes (3 C^2 floats per bin: 201 MB instead of 268 MB per conv at 512 channels).  The sums Xi - Xr and
// Xr + Xi are formed on the B fragments in registers (two VALU instructions per three MFMAs: the spectrum planes in HBM
// stay two) and the three accumulators of an output row meet in the epilogue (the product planes stay two as well).
// Bin 0 holds two REAL bins, DC and Nyquist, which must not mix: the forward transform writes (-E0, -O0) there instead of
// (E0 + O0, E0 - O0) (hsp_dftseg_args.prod3), so that Xi - Xr = Nyquist and Xr + Xi = -DC, and the bin's matrices are
// (0, W_nyquist, W_dc): Yr = -k3 = W_dc DC, Yi = k2 = W_nyquist Nyquist -- what the inverse transform expects, unchanged.
//
// Kernel: the conv kernel's structure (hsp_conv1d_mfma_kernel.h) specialised to this product.  One workgroup = 64 rows
// of C (128 output rows: Yr and Yi) x 128 columns of one bin, two workgroups per CU; waves 0-3 are consumers (a 32 x 64
// block each: 6 accumulators of 32 x 32), waves 4-7 producers that stage the next chunk of 16 input channels -- three
// [16][64] weight slabs and two [16][128] spectrum windows, 28 KB -- by LDS-DMA into the other half of a double buffer.
// The chunk depth is a compile-time constant, so every fragment address of a chunk is the lane's base plus an
// immediate: no address arithmetic between the MFMAs.
// Measured and not kept (same-box A/B of the 32 x 4 s step, profiles/r05_ab_cprod3_waveshape.json): consumer waves of
// 64 rows x 32 columns instead of 32 x 64 -- two VALU instructions per six MFMAs instead of four, but eight fragment reads
// instead of seven and every wave fetching all of the tile's A fragments: 58.96 against 58.72 ms per step; s_setprio 2 on
// the consumer waves: 59.04 against 58.65 (the producers' DMA issue falls behind), on the producer waves: 58.70 / 58.75; the
// epilogue through an LDS transpose with 16-B stores (a quarter of the store instructions): 58.99 against 58.59.
#include "hsp_conv1d_mfma_kernel.h"

namespace {
using namespace hspconv;

constexpr int CP_BM = 64, CP_BN = 128, CP_KC = 16;
constexpr int CP_WS = 3 * CP_KC * CP_BM;          // floats of the weight slabs of a chunk: [3][KC][BM]
constexpr int CP_XS = 2 * CP_KC * CP_BN;          // floats of the spectrum windows: [2][KC][BN]
constexpr int CP_BUF = CP_WS + CP_XS;             // one half of the double buffer: 28 KB
constexpr int CP_THREADS = 512;
// tuning build only (hsp_cprod3_args.debug, refused by the release library): 1 producers stage chunk 0 only, 2 no MFMAs,
// 16 no epilogue, 2048 no barriers, 4096 no narrow consumer for partial column tiles, 8192 no fragment sums, 16384 the plain block order -- results are
// then wrong
#define CP_BARRIER(a) do { if (!HSP_DBG(a, 2048)) lds_barrier(); } while (0)

template <int OFF>
__device__ __forceinline__ void cp_rd(float& dst, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read immediate offsets are 16 bits");
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

// fragment set of one k-step (two input channels): A of the three matrices, Re / Im of the NB column blocks of the wave
template <int NB>
struct CpFrag {
  float a[3], r[NB > 0 ? NB : 1], i[NB > 0 ? NB : 1];
};
template <int S, int NB>
__device__ __forceinline__ void cp_issue(CpFrag<NB>& f, unsigned va, unsigned vb) {
  constexpr int A0 = S * 2 * CP_BM * 4, B0 = S * 2 * CP_BN * 4;
  cp_rd<A0>(f.a[0], va);
  cp_rd<A0 + CP_KC * CP_BM * 4>(f.a[1], va);
  cp_rd<A0 + 2 * CP_KC * CP_BM * 4>(f.a[2], va);
  cp_rd<B0>(f.r[0], vb);
  if constexpr (NB > 1) cp_rd<B0 + 128>(f.r[1], vb);
  cp_rd<B0 + CP_KC * CP_BN * 4>(f.i[0], vb);
  if constexpr (NB > 1) cp_rd<B0 + CP_KC * CP_BN * 4 + 128>(f.i[1], vb);
}
// the set is valid behind the wait; re-defining its registers there gives every consumer a data dependency on the wait
// (hsp_conv1d_mfma_kernel.h: wait_frags)
template <int NB>
__device__ __forceinline__ void cp_bind(CpFrag<NB>& f) {
#pragma unroll
  for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(f.a[j]));
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    asm volatile("" : "+v"(f.r[n]));
    asm volatile("" : "+v"(f.i[n]));
  }
}
template <int NB>
__device__ __forceinline__ void cp_wait(CpFrag<NB>& f) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  cp_bind(f);
}

// The consumer role of one wave: rows m0 + 32 wm ... of C, NB = 1 or 2 valid 32-column blocks from n0 + 64 wn.  A wave
// whose second block lies wholly past Np (the last column tile of a bin: Np = 224 is 128 + 96) multiplies half as much;
// the MFMA pipe it leaves idle goes to the other workgroup's wave on its SIMD.
template <int NB>
__device__ __forceinline__ void cp_consume(const hsp_cprod3_args& a, float* const lds, const int bin, const int m0, const int n0,
                                           const int wm, const int wn, const int lane, const int nchunks) {
  const int C = a.C, Np = a.Np;
  const int l32 = lane & 31, half = lane >> 5;
  __builtin_assume(nchunks > 0);    // cp_check: C is a positive multiple of 64 (tools/check_isa.py: without this the compiler
                                    // keeps a "no chunk" exit that runs the epilogue under the first reads still in flight)
  f32x16 k1[NB], k2[NB], k3[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) k1[n][r] = k2[n][r] = k3[n][r] = 0.0f;
  const unsigned aA = lds_addr(lds + half * CP_BM + wm * 32 + l32);
  const unsigned aB = lds_addr(lds + CP_WS + half * CP_BN + wn * 64 + l32);
  CpFrag<NB> f0, f1;
  auto mma = [&](CpFrag<NB>& f) __attribute__((always_inline)) {
    if (HSP_DBG(a, 2)) return;
    // Xr + Xi and Xi - Xr on the fragments: counted VALU instructions between the wait and the first MFMA
    float s[NB], d[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      s[n] = HSP_DBG(a, 8192) ? f.i[n] : f.r[n] + f.i[n];       // (tuning bit 8192: no VALU between the MFMAs)
      d[n] = HSP_DBG(a, 8192) ? f.i[n] : f.i[n] - f.r[n];
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      k1[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[0], f.r[n], k1[n], 0, 0, 0);
      k2[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[1], d[n], k2[n], 0, 0, 0);
      k3[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[2], s[n], k3[n], 0, 0, 0);
    }
  };
  CP_BARRIER(a);                                                // chunk 0 staged
  unsigned va = aA, vb = aB;
  cp_issue<0>(f0, va, vb);
  for (int c = 0; c < nchunks; ++c) {
    // eight k-steps, fully unrolled: the reads of step S + 1 go out behind the wait that retires step S's and BEFORE
    // step S's MFMAs; the barrier that hands over chunk c + 1 sits between the last two MFMA groups of chunk c, and the
    // first reads of chunk c + 1 go out behind it (hsp_conv1d_mfma_kernel.h: the fragment pipeline runs across chunks)
    static_for<CP_KC / 2>([&](auto SS) __attribute__((always_inline)) {
      constexpr int S = decltype(SS)::value;
      CpFrag<NB>& cur = (S & 1) ? f1 : f0;
      CpFrag<NB>& nxt = (S & 1) ? f0 : f1;
      if constexpr (S + 1 < CP_KC / 2) {
        cp_wait(cur);
        __builtin_amdgcn_sched_barrier(0);
        cp_issue<S + 1>(nxt, va, vb);
      } else {
        CP_BARRIER(a);                                          // (waits lgkmcnt(0)) chunk c + 1 is staged, chunk c's buffer is free
        cp_wait(cur);
        if (c + 1 < nchunks) {
          const unsigned off = ((c + 1) & 1) ? (unsigned)(CP_BUF * 4) : 0u;
          va = aA + off;
          vb = aB + off;
          __builtin_amdgcn_sched_barrier(0);
          cp_issue<0>(nxt, va, vb);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      mma(cur);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  if (HSP_DBG(a, 16)) return;

  // ---- epilogue: Yr = k1 - k3 -> row m, Yi = k1 + k2 -> row C + m; stores straight from the accumulator layout (one
  // instruction = two rows x 128 B), uniform row pointer + one per-lane offset per column block
  float* const yb = a.yf + (int64_t)bin * a.yf_bs;
  const int lrow = m0 + wm * 32 + 4 * half;
  const int col0 = n0 + wn * 64 + l32;
  int voff[NB];
  bool cok[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    voff[n] = lrow * Np + col0 + 32 * n;
    cok[n] = col0 + 32 * n < Np;
  }
  const bool full = n0 + wn * 64 + 32 * NB <= Np;               // wave-uniform: every column of the wave's blocks exists
  auto store = [&](auto masked_tag) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    static_for<16>([&](auto rr) __attribute__((always_inline)) {
      constexpr int r = decltype(rr)::value;
      float* const yr = yb + (int64_t)HSP_ACC_ROW(r, 0) * Np;   // uniform
      float* const yi = yr + (int64_t)C * Np;
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        if (!MASKED || cok[n]) {
          yr[voff[n]] = k1[n][r] - k3[n][r];
          yi[voff[n]] = k1[n][r] + k2[n][r];
        }
      }
    });
  };
  if (full) store(std::false_type{});
  else store(std::true_type{});
}

__global__ __launch_bounds__(CP_THREADS, 4) void cprod3_kernel(const hsp_cprod3_args a, const int n_mt, const int n_nt) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][Ws | Xs]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int C = a.C, Np = a.Np;
  // Block order: every tile of a bin on ONE XCD (blocks b and b + 8 share an XCD -- observed dispatch order, speed only),
  // row tiles fastest.  The n_mt row tiles of a column window then run side by side behind the same L2 and fetch the
  // window once between them, and so do the column tiles of a weight slab: PMC traffic of the launches fell from 1.9 x to
  // ~1.2 x their algorithmic bytes (profiles/r05_traffic.json).  With the plain order (row tile = blockIdx % n_mt) the
  // row tiles of a window sat on n_mt DIFFERENT XCDs, each pulling its own copy from the Infinity Cache / HBM.
  int mt, nt, bin;
  {
    const int bid = blockIdx.x, T = n_mt * n_nt;
    if ((a.bins & 7) == 0 && !HSP_DBG(a, 16384)) {
      const int q = bid >> 3, t = q % T;
      bin = (q / T) * 8 + (bid & 7);
      mt = t % n_mt;
      nt = t / n_mt;
    } else {
      mt = bid % n_mt;
      nt = (bid / n_mt) % n_nt;
      bin = bid / T;
    }
  }
  const int m0 = mt * CP_BM, n0 = nt * CP_BN;
  const int nchunks = C / CP_KC;

  if (wave >= 4) {
    // ------------------------------------------------------------------------------------------------ producers
    // A DMA instruction moves 1 KB: four slab rows (64 floats each) or two window rows (128 floats).  A lane's byte
    // offset is computed once; the scalar unit walks the wave-uniform row base (hsp_conv1d_mfma_kernel.h: dma_w_fast).
    const int pw = wave - 4;
    const bool full = n0 + CP_BN <= Np;                         // workgroup-uniform: no column of the tile is past Np
    const float* const wb = a.w + (int64_t)bin * 3 * C * C + m0;
    const float* const xb = a.xf + (int64_t)bin * a.xf_bs + n0;
    const unsigned wofs = 4u * (unsigned)((lane >> 4) * C + (lane & 15) * 4);
    const unsigned xofs = 4u * (unsigned)((lane >> 5) * Np + (lane & 31) * 4);
    const bool colok = n0 + (lane & 31) * 4 < Np;               // Np % 4 == 0: a 16-B group is inside or outside as a whole
    auto stage = [&](int c0, float* buf) __attribute__((always_inline)) {
#pragma unroll
      for (int q0 = 0; q0 < 12; q0 += 4) {
        const int q = q0 + pw, j = q >> 2, g = q & 3;
        const char* base = reinterpret_cast<const char*>(wb + ((int64_t)j * C + c0 + 4 * g) * C);
        dma16(reinterpret_cast<const float*>(base + wofs), buf + (j * CP_KC + 4 * g) * CP_BM);
      }
#pragma unroll
      for (int q0 = 0; q0 < 16; q0 += 4) {
        const int q = q0 + pw, p = q >> 3, g = q & 7;
        const char* base = reinterpret_cast<const char*>(xb + ((int64_t)p * C + c0 + 2 * g) * Np);
        const float* src = reinterpret_cast<const float*>(base + xofs);
        if (!full) src = colok ? src : a.zeros;
        dma16(src, buf + CP_WS + (p * CP_KC + 2 * g) * CP_BN);
      }
    };
    stage(0, lds);
    wait_vm0();
    CP_BARRIER(a);
    for (int c = 0; c < nchunks; ++c) {
      if (c + 1 < nchunks && !HSP_DBG(a, 1)) {
        stage((c + 1) * CP_KC, lds + ((c + 1) & 1) * CP_BUF);
        wait_vm0();
      }
      CP_BARRIER(a);
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- consumers
  const int wm = wave >> 1, wn = wave & 1;
  const int left = Np - (n0 + wn * 64);                         // columns from this wave's first block to the end of the bin
  if (left > 32 || HSP_DBG(a, 4096)) {
    cp_consume<2>(a, lds, bin, m0, n0, wm, wn, lane, nchunks);
  } else if (left > 0) {
    cp_consume<1>(a, lds, bin, m0, n0, wm, wn, lane, nchunks);
  } else {
    for (int c = 0; c <= nchunks; ++c) CP_BARRIER(a);           // no column of this wave exists: keep the barrier count
  }
}

// ---------------------------------------------------------------------------------------------- weight spectrum
// One thread per (input channel, output row) of a packed conv weight w[k][Cin][M]: the 65 bins of the 128-point DFT of
// its k taps in float64 (twiddles from a float64 table the host generated), conjugated, rounded once to fp32 and written
// in the layout of the product kernel that will read it.
__global__ __launch_bounds__(256) void wspec_kernel(const float* __restrict__ w, const int k, const int C, const int w_ld,
                                                    const double* __restrict__ tw, float* __restrict__ out, const int form) {
  __shared__ double cs[256];
  cs[threadIdx.x] = tw[threadIdx.x];
  __syncthreads();
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= C * C) return;
  const int ci = idx / C, m = idx - ci * C;
  const float* wp = w + (int64_t)ci * w_ld + m;
  const int64_t tap = (int64_t)C * w_ld;
  const int64_t CC = (int64_t)C * C;
  double w0 = 0.0, w64 = 0.0;
  for (int j = 0; j < k; ++j) {
    const double v = (double)wp[j * tap];
    w0 += v;
    w64 += (j & 1) ? -v : v;
  }
  // bin 0: the two real bins DC (part 0 of the spectrum) and Nyquist (part 1)
  if (form == HSP_WSPEC_THREE) {
    out[0 * CC + (int64_t)ci * C + m] = 0.0f;
    out[1 * CC + (int64_t)ci * C + m] = (float)w64;
    out[2 * CC + (int64_t)ci * C + m] = (float)w0;
  } else {
    float* o = out + (int64_t)ci * 2 * C + m;
    o[0] = (float)w0;
    o[C] = 0.0f;
    o[CC * 2] = 0.0f;
    o[CC * 2 + C] = (float)w64;
  }
  for (int bin = 1; bin < 64; ++bin) {
    double re = 0.0, im = 0.0;                                  // rfft: sum_j w_j (cos - i sin)(2 pi bin j / 128)
    for (int j = 0; j < k; ++j) {
      const double v = (double)wp[j * tap];
      const int t = (bin * j) & 127;
      re += v * cs[t];
      im -= v * cs[128 + t];
    }
    // conj(W) = a + i b with a = re, b = -im
    if (form == HSP_WSPEC_THREE) {
      float* o = out + (int64_t)bin * 3 * CC + (int64_t)ci * C + m;
      o[0] = (float)(re - im);                                  // a + b
      o[CC] = (float)re;                                        // a
      o[2 * CC] = (float)(-im);                                 // b
    } else {
      float* o = out + (int64_t)bin * 4 * CC + (int64_t)ci * 2 * C + m;
      o[0] = (float)re;                                         // Yr += Wr Xr
      o[C] = (float)(-im);                                      // Yi -= Wi Xr
      o[CC * 2] = (float)im;                                    // Yr += Wi Xi
      o[CC * 2 + C] = (float)re;                                // Yi += Wr Xi
    }
  }
}

int cp_check(const hsp_cprod3_args& a) {
  if (!a.xf || !a.yf || !a.w || !a.zeros) return HSP_EINVAL;
  if (a.bins <= 0 || a.C <= 0 || a.Np <= 0) return HSP_EINVAL;
#ifndef HSP_TUNING
  if (a.debug) return HSP_EINVAL;
#endif
  if (a.C % CP_BM || (a.Np & 3)) return HSP_EINVAL;             // whole row tiles, 16-B column groups
  if (a.xf_bs < (int64_t)2 * a.C * a.Np || a.yf_bs < (int64_t)2 * a.C * a.Np || (a.xf_bs & 3) || (a.yf_bs & 3)) return HSP_EINVAL;
  if ((reinterpret_cast<uintptr_t>(a.xf) | reinterpret_cast<uintptr_t>(a.w)) & 15) return HSP_EINVAL;
  if ((int64_t)2 * a.C * a.Np > 0x7fffffffll / 4) return HSP_EINVAL;   // the lanes' byte offsets inside a bin are 32 bits
  const int64_t blocks = (int64_t)a.bins * (a.C / CP_BM) * ((a.Np + CP_BN - 1) / CP_BN);
  if (blocks > 0x7fffffff) return HSP_EINVAL;
  return 0;
}
}  // namespace

extern "C" int hsp_cprod3_supported(const hsp_cprod3_args* a) { return a && cp_check(*a) == 0 ? 1 : 0; }

extern "C" int hsp_cprod3_f32(const hsp_cprod3_args* ap, void* stream) {
  if (!ap) return HSP_EINVAL;
  const hsp_cprod3_args& a = *ap;
  if (int e = cp_check(a)) return e;
  const int n_mt = a.C / CP_BM, n_nt = (a.Np + CP_BN - 1) / CP_BN;
  const int64_t blocks = (int64_t)a.bins * n_mt * n_nt;
  constexpr int lds_bytes = 2 * CP_BUF * (int)sizeof(float);
  static hsp_lds_flags flags;
  if (int e = hsp_raise_lds_limit(reinterpret_cast<const void*>(cprod3_kernel), kMaxLdsBytes, flags)) return e;
  hipLaunchKernelGGL(cprod3_kernel, dim3((unsigned)blocks), dim3(CP_THREADS), lds_bytes, static_cast<hipStream_t>(stream), a, n_mt,
                     n_nt);
  return (int)hipGetLastError();
}

extern "C" int hsp_dftseg_weight_spectrum_f32(const float* w, int32_t k, int32_t C, int32_t w_ld, const double* tw, float* out,
                                              int32_t form, void* stream) {
  if (!w || !tw || !out || k < 1 || k > 64 || C < 1 || w_ld < C || (form != HSP_WSPEC_THREE && form != HSP_WSPEC_BLOCK))
    return HSP_EINVAL;
  const int64_t n = (int64_t)C * C;
  if (n > 0x7fffffff) return HSP_EINVAL;
  hipLaunchKernelGGL(wspec_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w, k, C, w_ld,
                     tw, out, form);
  return (int)hipGetLastError();
}
