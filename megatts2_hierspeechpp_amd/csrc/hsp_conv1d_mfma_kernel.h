// Fused Conv1d / ConvTranspose1d as an implicit GEMM on the fp32-input matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fmaf chain, same 157 TFLOP/s peak as
// the fp32 vector pipe -- MI355X_MICROARCH.md "Matrix cores").
//
//   acc[m, t] = sum_j sum_ci W[j][ci][m] * xin[ci, t + j*dil - pad]
//
// GEMM view: M = packed output rows, N = time, K = (taps x input channels).
// One workgroup owns a BM x BN output tile of one utterance and walks the input
// channels in chunks of KC.  Its waves are SPECIALISED:
//
//   waves 0-3   "consumers": each owns a (TM x TN) grid of 32x32 MFMA blocks and does
//               nothing but read A (weights) / B (shifted input window) fragments from
//               LDS -- one k-step ahead of the MFMAs that use them -- and issue MFMAs;
//   waves 4..   "producers": stage the NEXT chunk into the other half of a double
//               buffer -- the weight slab Ws[K][KC][BM] and the input window
//               Xa[KC][BN + (K-1)*dil] -- applying the PROLOGUE on the way: nothing,
//               leaky-ReLU, or the whole anti-aliased SnakeBeta activation (2x polyphase
//               up-sample -> snake -> 2x low-pass down-sample, replicate-padded at the
//               sequence ends exactly like alias_free_torch).  Each producer wave owns
//               whole channel rows, so the activation phases need only wave-local
//               ordering, and the activated tensor never exists in HBM.
//               Global loads are issued one chunk AHEAD into registers (issue early /
//               commit late), so their latency is covered by a whole chunk of MFMAs.
//
// Wave w and wave w+4 share a SIMD, so the producers' VALU/LDS/global work fills the
// issue slots between the consumers' 64-cycle MFMAs; the only workgroup-wide
// synchronisation is one barrier per chunk.  The EPILOGUE (bias, conditioning bias,
// gate / pointwise function, masks, per-channel scale, residual, running accumulation,
// ConvTranspose phase shuffle) runs on the accumulator registers of the consumers.
//
// Reference call sites: see include/hsp.h (hsp_conv1d_args).
//
// This header is the kernel template; hsp_conv1d_tile.hip instantiates it once per tile shape (one
// translation unit each, so the build parallelises) and hsp_conv1d_mfma.hip holds validation and the
// shape / epilogue selection.
#pragma once
#include <atomic>
#include <type_traits>
#include "hsp_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Tuning switches (hsp_conv1d_args.debug) exist only in the -DHSP_TUNING build (libhsp_tune.so); the
// release library refuses a non-zero debug word in validate().
#ifdef HSP_TUNING
#define HSP_DBG(a, bit) (((a).debug & (bit)) != 0)
#else
#define HSP_DBG(a, bit) false
#endif

// epilogue kinds (template parameter of the kernel: one epilogue per instantiation keeps the register
// allocation of the main loop free of the other variants' live ranges)
enum {
  HSP_EPI_INIT = 0,  // plain rows, no act / mask / cscale, scale == 1: bias + cbias + residual + running sum are
                     // loaded INTO the accumulators before the main loop; the epilogue is stores only
  HSP_EPI_VEC = 1,   // plain rows, any pointwise epilogue, 16-B addressable tensors: LDS transpose + float4
  HSP_EPI_GATE = 2,  // gated rows (WN / GLU)
  HSP_EPI_SHUF = 3,  // ConvTranspose phase shuffle, bias only
  HSP_EPI_GEN = 4    // everything else (scalar)
};

namespace hspconv {

template <int WM, int WN, int TM, int TN, int NPW_, int MINW_, int XSLACK_ = 64, int KS_ = 1>
struct Cfg {
  // KS_ > 1 (round 5): the K range of every chunk is split between KS_ groups of WM x WN consumer waves; their partial
  // sums meet in LDS behind the loop and group 0 runs the epilogue.  A short-sequence launch with fewer tiles than the chip
  // has CUs is bound by ONE tile's MFMA chain per SIMD; halving the tile and splitting its K range puts the same four SIMDs
  // of twice as many CUs on half the chain each (S64G2 below).
  static constexpr int kKS = KS_;
  static constexpr int NCW = WM * WN * KS_;        // consumer waves (4, or 8 = two per SIMD)
  static constexpr int kWM = WM, kWN = WN, kTM = TM, kTN = TN;
  static constexpr int BM = WM * TM * 32;
  static constexpr int BN = WN * TN * 32;
  static constexpr int NPW = NPW_;                 // producer waves
  static constexpr int NPT = NPW_ * 64;            // producer threads
  static constexpr int THREADS = 64 * (WM * WN * KS_ + NPW_);
  static constexpr int MINW = MINW_;               // waves per SIMD the register allocation must allow
  // LDS row pitch of the input window, a compile-time constant so that the consumer loop can reach the next
  // channel pair through ds_read immediates: BN columns + halo (K-1)*dil + 3 (the 16-B window DMA starts at the
  // aligned position below p0) must fit; a multiple of 64 (DMA instructions never straddle rows)
  static constexpr int XWP = BN + XSLACK_;
};

struct LdsPlan {
  int kc, lkc;  // chunk depth (power of two) and its log2
  int rpw;      // channel rows per producer wave and chunk
  int xw;       // activated window width  = BN + (K-1)*dil
  int xwp;      // its LDS row pitch (Cfg::XWP)
  int xrw;      // raw window width        = xw + 10        (ACT1D)
  int xrwp;     // raw row pitch in the producer scratch (multiple of 64)
  int a2w;      // 2x-rate window width    = 2*xw + 10      (ACT1D)
  int ws_sz, xa_sz, scr_sz;  // floats: one weight slab, one window buffer, one producer scratch
  int xa_off, scr_off, total;
};

template <class C>
__host__ __device__ inline LdsPlan make_plan(int K, int dil, int prologue, int lkc) {
  LdsPlan p;
  p.lkc = lkc;
  p.kc = 1 << lkc;
  p.rpw = (p.kc + C::NPW - 1) / C::NPW;
  p.xw = C::BN + (K - 1) * dil;
  p.xwp = C::XWP;  // valid only while xw + 3 <= XWP (pick_lkc checks)
  p.xrw = p.xw + 10;
  p.xrwp = (p.xrw + 63) & ~63;
  p.a2w = 2 * p.xw + 10;
  p.ws_sz = K * p.kc * C::BM;
  p.xa_sz = p.kc * p.xwp;
  // scratch of one producer wave: double-buffered raw rows + one 2x-rate row
  p.scr_sz = prologue == HSP_PRO_ACT1D ? 2 * p.rpw * p.xrwp + ((p.a2w + 3) & ~3) : 0;
  p.xa_off = 2 * p.ws_sz;
  p.scr_off = p.xa_off + 2 * p.xa_sz;
  p.total = p.scr_off + C::NPW * p.scr_sz;
  return p;
}

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// workgroup barrier that does NOT drain outstanding global loads: only this wave's LDS
// traffic has to be complete before the other role touches the buffer (__syncthreads()
// would add s_waitcnt vmcnt(0) and expose the latency of the prefetched chunk)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// tuning build: bit 2048 drops the chunk barriers (wrong results; what do they cost?  nothing measurable)
#define HSP_BARRIER(a) do { if (!HSP_DBG(a, 2048)) lds_barrier(); } while (0)

// 32-bit LDS byte address of a pointer into the workgroup's dynamic shared memory
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) float*)p;
}

// N fragment registers from consecutive 32-float blocks: f[i] = lds[addr + i * 128 B]
template <int N, int I = 0>
__device__ __forceinline__ void ds_read_frags(float (&f)[N], unsigned addr) {
  if constexpr (I < N) {
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f[I]) : "v"(addr), "n"(I * 128));
    ds_read_frags<N, I + 1>(f, addr);
  }
}

// the same from a compile-time byte offset: f[i] = lds[addr + OFF + i * 128 B]
template <int N, int OFF, int I = 0>
__device__ __forceinline__ void ds_read_frags_at(float (&f)[N], unsigned addr) {
  if constexpr (I < N) {
    static_assert(OFF + I * 128 < 65536, "ds_read immediate offsets are 16 bits");
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f[I]) : "v"(addr), "n"(OFF + I * 128));
    ds_read_frags_at<N, OFF, I + 1>(f, addr);
  }
}

// The fragment registers written by the asynchronous inline-asm ds_reads above become valid at an s_waitcnt.  The
// compiler does not know that: nothing but scheduling barriers kept it from moving an MFMA that reads them above a
// bare `asm volatile("s_waitcnt")` (hsp_bgemm.hip met the failure: stale fragments in some waves).  Here every
// fragment register is re-defined by an (empty) asm statement right behind the wait -- volatile asm statements keep
// their order -- so each consumer of a fragment carries a DATA dependency on the wait.  No instruction is emitted
// for the re-definitions: the ISA of the consumer loop is unchanged.
template <int N, int I = 0>
__device__ __forceinline__ void bind_frags(float (&f)[N]) {
  if constexpr (I < N) {
    asm volatile("" : "+v"(f[I]));
    bind_frags<N, I + 1>(f);
  }
}
template <int NA, int NB>
__device__ __forceinline__ void wait_frags(float (&fa)[NA], float (&fb)[NB]) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  bind_frags<NA>(fa);
  bind_frags<NB>(fb);
}

// wave-local LDS ordering: all earlier LDS ops of this wave are complete and the
// compiler may not move memory accesses across this point
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ------------------------------------------------------------------- producer side
// Only the fields the producers touch, held by value (a reference to the by-value kernel
// argument inside an aggregate makes the compiler spill the whole struct to scratch).
struct ProdArgs {
  const float* w;
  const float* zeros;      // >= 16 B of zeros: source of every out-of-range DMA lane
  const float* alpha_exp;
  const float* beta_inv;
  const float* filt;
  int K, Cin, Lin, M, w_ld, x_cs, x_ts, prologue;
  float slope;
};

// LDS-DMA (global_load_lds): each lane supplies a global address, the data lands at
// lds_base + lane * BYTES with no register staging; counted in vmcnt like a load.
#define HSP_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define HSP_LPTR(p) ((__attribute__((address_space(3))) void*)(p))
__device__ __forceinline__ void dma16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds(HSP_GPTR(g), HSP_LPTR(l), 16, 0, 0);
}
__device__ __forceinline__ void dma4(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds(HSP_GPTR(g), HSP_LPTR(l), 4, 0, 0);
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <class C>
struct Prod {
  static constexpr int CPR = C::BM / 4;                       // float4 columns per slab row
  static constexpr int RPI = (64 / CPR) > 0 ? (64 / CPR) : 1;  // slab rows per DMA instruction

  // ---- weight slab of chunk c0 -> Ws[j][kc][BM]: rows (j, c0+kc) of the packed matrix
  static __device__ __forceinline__ void dma_w(const ProdArgs& a, const LdsPlan& P, float* Ws, int c0, int m0, int pw,
                                               int lane) {
    const int r_in = lane / CPR, col = (lane % CPR) * 4;
    const int lipj = P.lkc - __builtin_ctz(RPI);  // log2(instructions per tap); KC >= RPI
    const int ninstr = a.K << lipj;
    const bool colok = m0 + col < a.M;
    const float* wcol = a.w + m0 + col;
    for (int q = pw; q < ninstr; q += C::NPW) {
      const int j = q >> lipj, g = q & ((1 << lipj) - 1);
      const int ci = c0 + g * RPI + r_in;
      const float* src = (colok && ci < a.Cin) ? wcol + (j * a.Cin + ci) * a.w_ld : a.zeros;
      dma16(src, Ws + ((j << P.lkc) + g * RPI) * C::BM);
    }
  }

  // ---- plain window rows -> Xa[kc][xwp] (zero outside [0, Lin) and beyond Cin)
  static __device__ __forceinline__ void dma_x(const ProdArgs& a, const LdsPlan& P, float* Xa, const float* xb, int c0,
                                               int p0, int pw, int lane) {
    const int npr = P.xwp >> 6;
    for (int kc = pw; kc < P.kc; kc += C::NPW) {
      const int ci = c0 + kc;
      const float* xc = xb + ci * a.x_cs;
      for (int i = 0; i < npr; ++i) {
        const int p = p0 + lane + 64 * i;
        const float* src = (ci < a.Cin && p >= 0 && p < a.Lin) ? xc + p * a.x_ts : a.zeros;
        dma4(src, Xa + kc * P.xwp + 64 * i);
      }
    }
  }
  // same with 16-B lanes: the window starts at p0a = p0 rounded down to a multiple of 4 and the
  // tensor is 16-B addressable with Lin % 4 == 0, so every 4-group is fully inside or outside
  static __device__ __forceinline__ void dma_x16(const ProdArgs& a, const LdsPlan& P, float* Xa, const float* xb,
                                                 int c0, int p0a, int pw, int lane) {
    const int ngrp = (P.xw + 3 + 3) >> 2;
    for (int kc = pw; kc < P.kc; kc += C::NPW) {
      const int ci = c0 + kc;
      const float* xc = xb + ci * a.x_cs;
      for (int g0 = 0; g0 < ngrp; g0 += 64) {
        const int g = g0 + lane, p = p0a + 4 * g;
        if (g < ngrp) {
          const float* src = (ci < a.Cin && p >= 0 && p < a.Lin) ? xc + p : a.zeros;
          dma16(src, Xa + kc * P.xwp + 4 * g0);
        }
      }
    }
  }
  // ---- fast staging: no per-instruction vector arithmetic.  The fp32 MFMAs of the consumer wave on the same
  // SIMD run on the VALU's own FMA lanes, so every address computation above (a 32-bit multiply, 64-bit adds,
  // selects: ~10 VALU instructions per DMA) is paid in matrix time (measured: 12 % of a k = 3 launch).  Here a
  // lane's byte offset is computed ONCE per tile and each instruction adds it to a wave-uniform base that the
  // scalar unit walks.  Preconditions (wave-uniform, checked by the kernel): the row tile lies inside [0, M),
  // Cin % KC == 0 (no channel tail), 16-B addressable window.
  static __device__ __forceinline__ void dma_w_fast(const float* wtile, int w_ld, int Cin, int K, const LdsPlan& P,
                                                    float* Ws, int c0, int pw, unsigned wofs) {
    const int lipj = P.lkc - __builtin_ctz(RPI);
    const int ninstr = K << lipj;
    for (int q = pw; q < ninstr; q += C::NPW) {
      const int j = q >> lipj, g = q & ((1 << lipj) - 1);
      const char* base = reinterpret_cast<const char*>(wtile + (size_t)(j * Cin + c0 + g * RPI) * (size_t)w_ld);  // uniform
      dma16(reinterpret_cast<const float*>(base + wofs), Ws + ((j << P.lkc) + g * RPI) * C::BM);
    }
  }
  // window rows: `xtile` = row 0 of the utterance + p0a (may point before the row: such lanes are masked).
  // Lanes whose 4-group lies outside [0, Lin) take no part (EXEC-masked DMA lanes write nothing); the kernel
  // zero-fills those LDS positions once per tile (zero_x_rows).
  static __device__ __forceinline__ void dma_x16_fast(const float* xtile, int x_cs, int Lin, const LdsPlan& P, float* Xa,
                                                      int c0, int p0a, int pw, int lane) {
    const int ngrp = (P.xw + 3 + 3) >> 2;
    for (int g0 = 0; g0 < ngrp; g0 += 64) {
      const int g = g0 + lane, p = p0a + 4 * g;
      if (g < ngrp && p >= 0 && p < Lin) {
        const unsigned xofs = 16u * (unsigned)lane;
        for (int kc = pw; kc < P.kc; kc += C::NPW) {
          const char* base = reinterpret_cast<const char*>(xtile + (size_t)(c0 + kc) * (size_t)x_cs + 4 * g0);  // uniform
          dma16(reinterpret_cast<const float*>(base + xofs), Xa + kc * P.xwp + 4 * g0);
        }
      }
    }
  }
  // rows this wave stages (kc = pw, pw + NPW, ...) of BOTH window buffers <- 0 (edge tiles only)
  static __device__ __forceinline__ void zero_x_rows(const LdsPlan& P, float* Xa0, int pw, int lane) {
    for (int bsel = 0; bsel < 2; ++bsel)
      for (int kc = pw; kc < P.kc; kc += C::NPW)
        for (int s = lane; s < P.xwp; s += 64) Xa0[bsel * P.xa_sz + kc * P.xwp + s] = 0.0f;
    wave_lds_fence();  // the zeros have landed before this wave's DMA may write the same rows
  }
  static __device__ __forceinline__ void lrelu_x(const ProdArgs& a, const LdsPlan& P, float* Xa, int pw, int lane) {
    for (int kc = pw; kc < P.kc; kc += C::NPW)
      for (int s = lane; s < P.xwp; s += 64) {
        const float v = Xa[kc * P.xwp + s];
        Xa[kc * P.xwp + s] = v > 0.0f ? v : v * a.slope;
      }
  }

  // ---- ACT1D: raw rows (replicate padding = index clamp) -> this wave's scratch
  static __device__ __forceinline__ void dma_raw(const ProdArgs& a, const LdsPlan& P, float* rawbuf, const float* xb,
                                                 int c0, int p0, int pw, int lane) {
    const int npr = P.xrwp >> 6;
    for (int rs = 0; rs < P.rpw; ++rs) {
      const int ci = c0 + pw + rs * C::NPW;
      const float* xc = xb + ci * a.x_cs;
      for (int i = 0; i < npr; ++i) {
        const float* src = ci < a.Cin ? xc + hsp_clampi(p0 - 5 + lane + 64 * i, 0, a.Lin - 1) : a.zeros;
        dma4(src, rawbuf + rs * P.xrwp + 64 * i);
      }
    }
  }

  // one channel row: raw (LDS) -> 2x-rate snake signal (LDS) -> activated window row.
  // VALU cycles here are stolen from the fp32 MFMAs of the consumer wave on the same SIMD
  // (they share the fp32 datapath), so the arithmetic is kept minimal: packed fp32 FMAs
  // (v_pk_fma_f32: an even and an odd output sample per lane) and hardware cosine.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ void act_row(const ProdArgs& a, const LdsPlan& P, const float* raw, float* a2,
                                                 float* xa, int ci, int p0, int lane) {
    const int L = a.Lin;
    const int mlo = 2 * p0 - 5;
    const bool interior = mlo >= 0 && mlo + P.a2w <= 2 * L;  // no index clamp needed at the 2x rate
    const float* hu = a.filt;
    const float* hd = a.filt + 12;
    const int cic = ci < a.Cin ? ci : a.Cin - 1;
    const float kf = a.alpha_exp[cic] * 0.318309886183790672f, kb = 0.5f * a.beta_inv[cic];
    // phase B: a[m] = snake(2 * up[m]);  up[2q]   = sum_i x[q-3+i] * hu[11-2i],
    //                                    up[2q+1] = sum_i x[q-2+i] * hu[10-2i]   (i = 0..5)
    if (interior) {
      // slot 2i+1 <-> m = 2q (q = p0-2+i), slot 2i+2 <-> m = 2q+1: both read raw[i .. i+6]
      const int npair = (P.a2w - 1) >> 1;
      for (int i = lane; i < npair; i += 64) {
        float xv[7];
#pragma unroll
        for (int t = 0; t < 7; ++t) xv[t] = raw[i + t];
        f32x2 u = {0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < 6; ++t) {
          const f32x2 xx = {xv[t], xv[t + 1]};
          const f32x2 hh = {hu[11 - 2 * t], hu[10 - 2 * t]};
          u = __builtin_elementwise_fma(xx, hh, u);
        }
        u = u * 2.0f;
        a2[2 * i + 1] = hsp_snake_hw(u.x, kf, kb);
        a2[2 * i + 2] = hsp_snake_hw(u.y, kf, kb);
      }
      if (lane < 2) {  // slot 0 (odd m, q = p0-3, raw[0..5]) and the last slot (even m, raw[xw+4..xw+9])
        const float* xr_ = lane ? raw + P.xw + 4 : raw;
        float uu = 0.0f;
#pragma unroll
        for (int t = 0; t < 6; ++t) uu = fmaf(xr_[t], lane ? hu[11 - 2 * t] : hu[10 - 2 * t], uu);
        a2[lane ? P.a2w - 1 : 0] = hsp_snake_hw(2.0f * uu, kf, kb);
      }
    } else {
      for (int s = lane; s < P.a2w; s += 64) {
        const int m = hsp_clampi(mlo + s, 0, 2 * L - 1);
        const int q = m >> 1, odd = m & 1;
        const float* xr_ = raw + (q - 3 + odd) - (p0 - 5);
        float u = 0.0f;
#pragma unroll
        for (int t = 0; t < 6; ++t) u = fmaf(xr_[t], odd ? hu[10 - 2 * t] : hu[11 - 2 * t], u);
        a2[s] = hsp_snake_hw(2.0f * u, kf, kb);
      }
    }
    wave_lds_fence();
    // phase C: y[p] = sum_k hd[k] * a[clamp(2p+k-5)], zero outside [0, L) (conv zero padding)
    for (int s = lane; s < P.xwp; s += 64) {
      const int p = p0 + s;
      f32x2 v = {0.0f, 0.0f};
      if (p >= 0 && p < L && s < P.xw) {
        const f32x2* ar = reinterpret_cast<const f32x2*>(a2 + 2 * s);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const f32x2 hh = {hd[2 * k], hd[2 * k + 1]};
          v = __builtin_elementwise_fma(ar[k], hh, v);
        }
      }
      xa[s] = v.x + v.y;
    }
    wave_lds_fence();  // a2 is reused by the next row
  }

  static __device__ __forceinline__ void act_rows(const ProdArgs& a, const LdsPlan& P, const float* rawbuf, float* a2,
                                                  float* Xa, int c0, int p0, int pw, int lane) {
    for (int rs = 0; rs < P.rpw; ++rs) {
      const int row = pw + rs * C::NPW;
      if (row < P.kc) act_row(a, P, rawbuf + rs * P.xrwp, a2, Xa + row * P.xwp, c0 + row, p0, lane);
    }
  }
};

// ------------------------------------------------------------------------ kernel
// C/D map of the 32x32 MFMA forms: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
#define HSP_ACC_ROW(r, half) (((r) & 3) + 8 * ((r) >> 2) + 4 * (half))

// The consumer role of one wave: a (TM x TN) grid of 32 x 32 MFMA blocks at rows m0 + wm * TM * 32, columns
// t0 + wn * TN * 32 of the tile.  TM / TN are the shape's own (C::kTM, C::kTN) except in a narrow tail tile
// (has_tail_path): the slab and window layouts in LDS belong to the SHAPE (C::BM, C::XWP), only the part of the
// tile a wave multiplies changes.
template <class C, int EPI, bool ACT, int TM, int TN>
__device__ __forceinline__ void conv_consume(const hsp_conv1d_args& a, const LdsPlan& P, float* const lds, const int b,
                                             const int m0, const int t0, const int p0, const int wm, const int wn,
                                             const int lane, const int nchunks, const int lkc, const int xvec,
                                             const int ks = 0) {
  constexpr int BM = C::BM;
  const int KC = P.kc;
  // validate() refuses Cin < 1.  Said to the compiler as well: without it the "no chunk at all" exit it generates runs the
  // epilogue on registers that the first fragment reads -- issued before the loop -- are still going to write
  // (tools/check_isa.py reports exactly that path).
  __builtin_assume(nchunks > 0);
  const int wave = wm * C::kWN + wn;               // only the VEC epilogue's staging slot uses it (full tiles)
  const int l32 = lane & 31, half = lane >> 5;
  const int mw = m0 + wm * (TM * 32);               // first packed row of this wave
  const int tw = t0 + wn * (TN * 32) + l32;         // this lane's column in block n = 0
  f32x16 acc[TM][TN];

  // (a plain-row tile with a K split -- 64 x 64, two groups of four waves -- was built and measured in round 5: single
  // requests 7.70 / 9.09 ms with the gated split tile alone, 7.68 / 9.11 with both; not kept, and with it went the
  // accumulator-init handling a second K group needs: it would have to start from zero)
  static_assert(C::kKS == 1 || EPI != HSP_EPI_INIT, "a K split with the accumulator-init epilogue would add the residual once per group");
  if constexpr (EPI == HSP_EPI_INIT) {
    // The accumulators START at bias + cbias + residual (+ the running sum of `accumulate`), loaded in
    // the MFMA C layout while the producers' first chunk is still in flight: the loads' latency hides
    // under the staging latency every tile pays anyway, and the epilogue has no loads left -- nothing
    // there ever waits on vmcnt.  (Each load instruction covers two rows x 128 B.)  Out-of-range rows /
    // columns read a clamped address and are never stored.
    // Whole passes sit under ONE wave-uniform branch each, so that the loads of a pass issue back to back
    // (a branch per load makes hipcc wait on vmcnt(0) after every one of them).  In a wave whose rows all lie
    // inside [0, Cout) an address is a uniform (scalar) per-register row pointer plus one per-lane offset per
    // column block; otherwise (Cout % 32 != 0) the row is clamped per lane.  Lanes past the last column read a
    // clamped column.  A wave wholly past Cout starts from zero and stores nothing.
    const int rcs = (int)a.res_cs, ycs = (int)a.y_cs;
    const int lrow = mw + 4 * half;                // this lane's row for register 0 of block i = 0
    int tcl[TN];
#pragma unroll
    for (int n = 0; n < TN; ++n) tcl[n] = min(tw + n * 32, a.ncols - 1);
    auto init = [&](auto full_tag) __attribute__((always_inline)) {
      constexpr bool FULL = decltype(full_tag)::value;
      // element offset of (row of register r in block i, column block n) for a tensor with channel stride cs
      auto vec1 = [&](const float* p, int c) __attribute__((always_inline)) -> float {
        if constexpr (FULL) return (p + c)[lrow];
        else return p[min(lrow + c, a.Cout - 1)];
      };
      auto mat = [&](const float* p, int c, int cs, int n) __attribute__((always_inline)) -> float {
        if constexpr (FULL) return (p + c * cs)[lrow * cs + tcl[n]];
        else return p[min(lrow + c, a.Cout - 1) * cs + tcl[n]];
      };
      float add[TM][16];
      if (a.bias) {
        static_for<TM>([&](auto ii) __attribute__((always_inline)) {
          constexpr int i = decltype(ii)::value;
          static_for<16>([&](auto rr) __attribute__((always_inline)) {
            constexpr int r = decltype(rr)::value;
            add[i][r] = vec1(a.bias, i * 32 + HSP_ACC_ROW(r, 0));
          });
        });
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) add[i][r] = 0.0f;
      }
      if (a.res) {
        const float* const rb = a.res + (int64_t)b * a.res_bs;
        static_for<TM>([&](auto ii) __attribute__((always_inline)) {
          constexpr int i = decltype(ii)::value;
          static_for<16>([&](auto rr) __attribute__((always_inline)) {
            constexpr int r = decltype(rr)::value;
            static_for<TN>([&](auto nn) __attribute__((always_inline)) {
              constexpr int n = decltype(nn)::value;
              acc[i][n][r] = mat(rb, i * 32 + HSP_ACC_ROW(r, 0), rcs, n);
            });
          });
        });
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.0f;
      }
      static_for<TM>([&](auto ii) __attribute__((always_inline)) {
        constexpr int i = decltype(ii)::value;
        static_for<TN>([&](auto nn) __attribute__((always_inline)) {
          constexpr int n = decltype(nn)::value;
          static_for<16>([&](auto rr) __attribute__((always_inline)) {
            constexpr int r = decltype(rr)::value;
            acc[i][n][r] += add[i][r];
          });
        });
      });
      if (a.cbias) {
        const float* const cb = a.cbias + (int64_t)b * a.cbias_bs;
        static_for<TM>([&](auto ii) __attribute__((always_inline)) {
          constexpr int i = decltype(ii)::value;
          static_for<16>([&](auto rr) __attribute__((always_inline)) {
            constexpr int r = decltype(rr)::value;
            const float c = vec1(cb, i * 32 + HSP_ACC_ROW(r, 0));
            static_for<TN>([&](auto nn) __attribute__((always_inline)) {
              constexpr int n = decltype(nn)::value;
              acc[i][n][r] += c;
            });
          });
        });
      }
      if (a.accumulate) {
        // one 32 x 32 block at a time: its 16 loads issue together into temporaries, then one wait
        const float* const yb = a.y + (int64_t)b * a.y_bs;
        static_for<TM>([&](auto ii) __attribute__((always_inline)) {
          constexpr int i = decltype(ii)::value;
          static_for<TN>([&](auto nn) __attribute__((always_inline)) {
            constexpr int n = decltype(nn)::value;
            float tmp[16];
            static_for<16>([&](auto rr) __attribute__((always_inline)) {
              constexpr int r = decltype(rr)::value;
              tmp[r] = mat(yb, i * 32 + HSP_ACC_ROW(r, 0), ycs, n);
            });
            __builtin_amdgcn_sched_barrier(0);
            static_for<16>([&](auto rr) __attribute__((always_inline)) {
              constexpr int r = decltype(rr)::value;
              acc[i][n][r] += tmp[r];
            });
            __builtin_amdgcn_sched_barrier(0);
          });
        });
      }
    };
    if (mw + TM * 32 <= a.Cout) {
      init(std::true_type{});
    } else if (mw < a.Cout) {
      init(std::false_type{});
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.0f;
    }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  }

  const int wlane = half * BM + wm * (TM * 32) + l32;
  const int xlane = half * P.xwp + wn * (TN * 32) + l32 + (xvec ? (p0 & 3) : 0);
  // steps of a chunk (even: KC >= 4, so a tap has an even number of channel pairs); with a K split this group's share,
  // [ks * nsteps, (ks + 1) * nsteps) of them (pick_lkc: the share is even, so a trip never straddles a tap)
  const int nsteps = ((a.K * KC) >> 1) / C::kKS;

  HSP_BARRIER(a);  // chunk 0 staged
  // k-steps ordered tap outer, channel pair inner.  The weight slab is [tap][channel][row], so the A side is
  // one linear walk (1 KB per step for BM = 128); the B side moves one pair (2 window rows) per step and one tap
  // (dil columns) per KC / 2 steps.  A trip is two steps of the same tap: the second step's addresses are the
  // first's plus compile-time immediates (Cfg::XWP is a constant of the shape), so a trip costs two VALU adds and
  // six scalar instructions, no branch.  With fp32
  // MFMAs sharing the VALU datapath that matters: tools/micro/mfma_loop_bench.hip loses 5 % of the matrix pipe to
  // two VALU adds per step and 11 % more to a branchy tap wrap (what this loop looked like in round 1).
  // Fragment reads are hand-placed (inline asm + explicit lgkmcnt): the reads of step s+1 are issued right
  // after the wait that retires the reads of step s and BEFORE the MFMAs of step s, so an LDS round trip never
  // sits between two MFMA groups (left to itself hipcc sinks the prefetch under the MFMAs and waits at once).
  {
    constexpr int kStepA = 2 * BM * 4;     // bytes: next channel pair in the weight slab
    constexpr int rowB = 2 * C::XWP * 4;   // bytes: next channel pair in the window
    const int spanB = (KC >> 1) * rowB;    // all pairs of one tap
    const int tapB = 4 * a.dil;
    float fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    auto mma_set = [&](const float (&fa)[TM], const float (&fb)[TN]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n)
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[n], acc[i][n], 0, 0, 0);
    };
    // The fragment pipeline runs ACROSS chunks: the barrier that hands over chunk c+1 sits between the last two
    // MFMA groups of chunk c (all LDS reads of chunk c are retired by then - its last fragments are in registers),
    // and the first fragments of chunk c+1 are requested right behind it, under the last MFMA group of chunk c.
    // With the barrier after the last group the first reads of every chunk were exposed (~3 % at 22-28 steps per
    // chunk).  Barrier count is unchanged: one before the loop, one per chunk.
    unsigned aA = lds_addr(lds + wlane);                 // per-lane LDS byte addresses of the chunk's buffers
    unsigned aB = lds_addr(lds + P.xa_off + xlane);
    // where this K group's share of a chunk starts: step s0 = tap s0 / (KC / 2), channel pair s0 % (KC / 2)
    int offA0 = 0, pairB0 = 0, tapoff0 = 0;
    if constexpr (C::kKS > 1) {
      const int s0 = ks * nsteps;
      offA0 = s0 * kStepA;
      pairB0 = (s0 & ((KC >> 1) - 1)) * rowB;
      tapoff0 = (s0 >> (lkc - 1)) * tapB;
    }
    unsigned va = aA + (unsigned)offA0, vb = aB + (unsigned)(pairB0 + tapoff0);
    const bool run = !HSP_DBG(a, 2);
    if (run) {
      ds_read_frags<TM>(fa0, va);
      ds_read_frags<TN>(fb0, vb);
    }
    for (int c = 0; c < nchunks; ++c) {
      if (!run) {
        HSP_BARRIER(a);
        continue;
      }
      int offA = offA0, pairB = pairB0, tapoff = tapoff0;  // scalar byte offsets of the current trip
      // The hazard recogniser wants one COUNTED instruction between the re-definition of the fragments in wait_frags
      // and the first MFMA that reads them, and it does not count inline asm.  So the scalar walk of a trip sits in its
      // first slot, followed by the one scalar add hipcc emits itself (window offset = pair + tap), and the two VALU
      // address adds sit in the second slot behind its wait: no s_nop, and as many instructions per trip as before.
      int offB = 0;
      auto advance = [&]() __attribute__((always_inline)) {
        int t;
        asm volatile(
            "s_add_i32 %[oa], %[oa], %[sa]\n\t"
            "s_add_i32 %[pb], %[pb], %[sb]\n\t"
            "s_cmp_eq_u32 %[pb], %[span]\n\t"
            "s_cselect_b32 %[pb], 0, %[pb]\n\t"
            "s_cselect_b32 %[t], %[tap], 0\n\t"
            "s_add_i32 %[to], %[to], %[t]"
            : [oa] "+s"(offA), [pb] "+s"(pairB), [to] "+s"(tapoff), [t] "=&s"(t)
            : [sa] "n"(2 * kStepA), [sb] "n"(2 * rowB), [span] "s"(spanB), [tap] "s"(tapB)
            : "scc");
        offB = pairB + tapoff;
      };
      // first slot of a trip: the second step of the same tap, through immediates (+ the scalar walk to the next trip)
      auto slot_a = [&](bool walk) __attribute__((always_inline)) {
        wait_frags(fa0, fb0);               // the first step's fragments have landed (and only now are they valid)
        __builtin_amdgcn_sched_barrier(0);
        if (walk) advance();
        ds_read_frags_at<TM, kStepA>(fa1, va);
        ds_read_frags_at<TN, rowB>(fb1, vb);
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs (true double buffer)
        mma_set(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);  // or the MFMAs sink below the next wait, which then follows its reads at once
      };
      for (int s = 0; s + 2 < nsteps; s += 2) {   // every trip but the last
        slot_a(true);
        wait_frags(fa1, fb1);               // the second step's fragments have landed
        __builtin_amdgcn_sched_barrier(0);
        va = aA + (unsigned)offA;           // two VALU adds behind the wait: counted instructions before the MFMAs
        vb = aB + (unsigned)offB;
        ds_read_frags<TM>(fa0, va);
        ds_read_frags<TN>(fb0, vb);
        __builtin_amdgcn_sched_barrier(0);
        mma_set(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
      }
      slot_a(false);                               // last trip, first step
      HSP_BARRIER(a);                              // (waits lgkmcnt(0) first) chunk c+1 is staged, chunk c's buffer is free
      bind_frags<TM>(fa1);                         // the barrier's wait is what makes the last step's fragments valid
      bind_frags<TN>(fb1);
      if (c + 1 < nchunks) {
        const int nb = (c + 1) & 1;
        aA = lds_addr(lds + nb * P.ws_sz + wlane);
        aB = lds_addr(lds + P.xa_off + nb * P.xa_sz + xlane);
        va = aA + (unsigned)offA0;
        vb = aB + (unsigned)(pairB0 + tapoff0);
        __builtin_amdgcn_sched_barrier(0);
        ds_read_frags<TM>(fa0, va);                // chunk c+1, step 0
        ds_read_frags<TN>(fb0, vb);
      }
      __builtin_amdgcn_sched_barrier(0);
      mma_set(fa1, fb1);                           // last step of chunk c, over the barrier's wake-up and the new reads
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  if constexpr (C::kKS > 1) {
    // The K groups' partial sums meet in LDS: the chunk buffers are free behind the loop's last barrier (every consumer
    // retired its fragment reads there), the producers have left, so the barrier below counts the consumer waves only.
    static_assert(C::kKS == 2, "one partner group per wave");
    // (behind the VEC epilogue's per-wave staging slots: a group-0 wave that has finished adding may already be
    // transposing its tile there while its neighbour still reads its partner's partial sums)
    float* const red = lds + C::kWM * C::kWN * (32 * 36) + (wm * C::kWN + wn) * (TM * TN * 16 * 64) + lane;
    if (ks > 0) {
      static_for<TM>([&](auto ii) __attribute__((always_inline)) {
        constexpr int i = decltype(ii)::value;
        static_for<TN>([&](auto nn) __attribute__((always_inline)) {
          constexpr int n = decltype(nn)::value;
          static_for<16>([&](auto rr) __attribute__((always_inline)) {
            constexpr int r = decltype(rr)::value;
            red[((i * TN + n) * 16 + r) * 64] = acc[i][n][r];
          });
        });
      });
    }
    lds_barrier();
    if (ks > 0) return;
    static_for<TM>([&](auto ii) __attribute__((always_inline)) {
      constexpr int i = decltype(ii)::value;
      static_for<TN>([&](auto nn) __attribute__((always_inline)) {
        constexpr int n = decltype(nn)::value;
        static_for<16>([&](auto rr) __attribute__((always_inline)) {
          constexpr int r = decltype(rr)::value;
          acc[i][n][r] += red[((i * TN + n) * 16 + r) * 64];
        });
      });
    });
  }

  // ---- epilogue.  static_for keeps every accumulator index a compile-time constant: a runtime
  //      index into acc[][] would demote the whole accumulator file to scratch memory.
  if (HSP_DBG(a, 16)) return;
  if constexpr (EPI == HSP_EPI_INIT) {
    // stores only, straight from the accumulator layout: one instruction = two rows x 128 B
    float* const yb = a.y + (int64_t)b * a.y_bs;
    const int ycs = (int)a.y_cs;
    const bool full = mw + TM * 32 <= a.Cout && t0 + wn * (TN * 32) + TN * 32 <= a.ncols;  // wave-uniform
    const float ps = a.post_scale;
    if (full) {
      const int lrow = mw + 4 * half;
      int voff[TN];
#pragma unroll
      for (int n = 0; n < TN; ++n) voff[n] = lrow * ycs + tw + n * 32;
      static_for<TM>([&](auto ii) __attribute__((always_inline)) {
        constexpr int i = decltype(ii)::value;
        static_for<16>([&](auto rr) __attribute__((always_inline)) {
          constexpr int r = decltype(rr)::value;
          float* const yp = yb + (i * 32 + HSP_ACC_ROW(r, 0)) * ycs;  // uniform
          static_for<TN>([&](auto nn) __attribute__((always_inline)) {
            constexpr int n = decltype(nn)::value;
            // tuning bit 65536: non-temporal stores -- measured: k = 3 launches +2-4 % in isolation, step unchanged
            if (HSP_DBG(a, 65536)) __builtin_nontemporal_store(acc[i][n][r] * ps, yp + voff[n]);
            else yp[voff[n]] = acc[i][n][r] * ps;
          });
        });
      });
    } else {
      static_for<TM>([&](auto ii) __attribute__((always_inline)) {
        constexpr int i = decltype(ii)::value;
        static_for<TN>([&](auto nn) __attribute__((always_inline)) {
          constexpr int n = decltype(nn)::value;
          const int t = tw + n * 32;
          static_for<16>([&](auto rr) __attribute__((always_inline)) {
            constexpr int r = decltype(rr)::value;
            const int row = mw + i * 32 + HSP_ACC_ROW(r, half);
            if (row < a.Cout && t < a.ncols) yb[row * ycs + t] = acc[i][n][r] * ps;
          });
        });
      });
    }
  } else if constexpr (EPI == HSP_EPI_VEC) {
    // Vector epilogue (plain rows, 16-B aligned tensors, any pointwise tail): each 32x32 accumulator
    // block is transposed through this wave's LDS staging area (the weight buffers are free after the
    // last barrier) so that a lane owns 4 consecutive time steps of 4 rows -> float4 residual /
    // accumulate loads and float4 stores.
    constexpr int ESTR = 36;
    float* const stage = lds + wave * (32 * ESTR);
    const int er = lane >> 3, ec = (lane & 7) * 4;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    static_for<TM>([&](auto ii) __attribute__((always_inline)) {
      constexpr int i = decltype(ii)::value;
      float add[4], cs[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = mw + i * 32 + er + 8 * q;
        add[q] = 0.0f;
        cs[q] = a.scale;
        if (co < a.Cout) {
          if (a.bias) add[q] = a.bias[co];
          if (a.cbias) add[q] += a.cbias[(int64_t)b * a.cbias_bs + co];
          if (a.cscale) cs[q] *= a.cscale[(int64_t)b * a.cscale_bs + co];
        }
      }
      static_for<TN>([&](auto nn) __attribute__((always_inline)) {
        constexpr int n = decltype(nn)::value;
        const int t = t0 + wn * (TN * 32) + n * 32 + ec;
        const bool tok = t < a.ncols;
        float4 rs[4], yo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = mw + i * 32 + er + 8 * q;
          const bool ok = tok && co < a.Cout;
          rs[q] = z4;
          yo[q] = z4;
          if (ok && a.res) rs[q] = *reinterpret_cast<const float4*>(a.res + (int64_t)b * a.res_bs + (int64_t)co * a.res_cs + t);
          if (ok && a.accumulate) yo[q] = *reinterpret_cast<const float4*>(a.y + (int64_t)b * a.y_bs + (int64_t)co * a.y_cs + t);
        }
        float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
        if (a.mask_mode != HSP_MASK_NONE && tok) mk = *reinterpret_cast<const float4*>(a.mask + (int64_t)b * a.mask_bs + t);
        static_for<16>([&](auto rr) __attribute__((always_inline)) {
          constexpr int r = decltype(rr)::value;
          stage[HSP_ACC_ROW(r, half) * ESTR + l32] = acc[i][n][r];
        });
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = mw + i * 32 + er + 8 * q;
          if (tok && co < a.Cout) {
            const float4 v = *reinterpret_cast<const float4*>(stage + (er + 8 * q) * ESTR + ec);
            float e[4] = {v.x, v.y, v.z, v.w};
            const float r4[4] = {rs[q].x, rs[q].y, rs[q].z, rs[q].w};
            const float y4[4] = {yo[q].x, yo[q].y, yo[q].z, yo[q].w};
            const float m4[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              float x = hsp_apply_act(e[u] + add[q], a.act);
              if (a.mask_mode & HSP_MASK_PRE) x *= m4[u];
              x = fmaf(x, cs[q], r4[u]);
              if (a.mask_mode & HSP_MASK_POST) x *= m4[u];
              e[u] = (x + y4[u]) * a.post_scale;
            }
            *reinterpret_cast<float4*>(a.y + (int64_t)b * a.y_bs + (int64_t)co * a.y_cs + t) =
                make_float4(e[0], e[1], e[2], e[3]);
          }
        }
        wave_lds_fence();  // the staging area is rewritten by the next block
      });
    });
  } else if constexpr (EPI == HSP_EPI_GATE) {
    static_assert(TM % 2 == 0, "gated rows need both halves of a channel in one wave");
    const int H = a.gate_half;
    static_for<TM / 2>([&](auto ih) __attribute__((always_inline)) {
      constexpr int i = 2 * decltype(ih)::value;
      const int mpair = mw + i * 32;  // packed row of the 'a' block; multiple of 64
      static_for<TN>([&](auto nn) __attribute__((always_inline)) {
        constexpr int n = decltype(nn)::value;
        const int t = tw + n * 32;
        if (mpair < a.M && t < a.ncols) {
          static_for<16>([&](auto rr) __attribute__((always_inline)) {
            constexpr int r = decltype(rr)::value;
            const int co = (mpair >> 6) * 32 + HSP_ACC_ROW(r, half);
            if (co < H) {
              float va = acc[i][n][r], vb = acc[i + 1][n][r];
              if (a.bias) { va += a.bias[co]; vb += a.bias[H + co]; }
              if (a.cbias) {
                va += a.cbias[(int64_t)b * a.cbias_bs + co];
                vb += a.cbias[(int64_t)b * a.cbias_bs + H + co];
              }
              const float v = (a.rows == HSP_ROWS_GATE_WN ? hsp_tanh(va) : va) * hsp_sigmoid(vb);
              hsp_epilogue_store(a, b, co, t, v);
            }
          });
        }
      });
    });
  } else if constexpr (EPI == HSP_EPI_SHUF) {
    // ConvTranspose fast path: row m = co*up + phase writes y[co, up*t + phase - pad].  Row
    // constants (channel, phase, bias, row pointer) are resolved once per accumulator row; the
    // store loop has no loads, so nothing ever waits on vmcnt.  The `up` stores of one channel
    // (registers r&3 for up = 4, r&1 for up = 2) interleave into full lines in L2.
    // (round 4) stride 4 / stride 2 with a whole number of channels per register group: the `up` phases of a channel
    // sit in ADJACENT accumulator registers of one lane (rows r & 3 of a group of four) and land on `up` consecutive
    // output samples, so they leave as ONE 16-B / 8-B store per lane -- lanes are consecutive t, a store instruction
    // covers 512 / 256 contiguous bytes -- instead of `up` scalar stores at a 16-B / 8-B lane stride that only merge
    // into full lines in L2 (4 x / 2 x the store instructions; the stride-2 launches of the Generator ran at 57 and 79
    // TFLOP/s on them).  The first / last output group of a row (phases shifted outside [0, Lout) by the transposed
    // conv's padding) takes scalar stores.
    if ((a.up == 4 || a.up == 2) && !HSP_DBG(a, 262144)) {   // tuning bit 262144: the scalar phase stores of round 3
      typedef float st4 __attribute__((ext_vector_type(4), aligned(4)));
      typedef float st2 __attribute__((ext_vector_type(2), aligned(4)));
      const float sc = a.scale * a.post_scale;
      const int upl = a.up == 4 ? 2 : 1;               // log2(up)
      static_for<TM>([&](auto ii) __attribute__((always_inline)) {
        constexpr int i = decltype(ii)::value;
        // register group q = r >> 2 holds rows mw + i * 32 + 8 q + 4 half + (0..3): one channel (up = 4) or two (up = 2)
        float* yrow[4][2];
        float bz[4][2];
        bool rok[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2) {
            const int m = mw + i * 32 + 8 * q + 4 * half + 2 * c2;      // up = 2: channel c2 of the group; up = 4: c2 = 0 only
            const int co = m >> upl;
            rok[q][c2] = m < a.M && co < a.Cout;
            yrow[q][c2] = a.y + (int64_t)b * a.y_bs + (int64_t)(rok[q][c2] ? co : 0) * a.y_cs;
            bz[q][c2] = (rok[q][c2] && a.bias) ? a.bias[co] : 0.0f;
          }
        static_for<TN>([&](auto nn) __attribute__((always_inline)) {
          constexpr int n = decltype(nn)::value;
          const int t = tw + n * 32;
          if (t < a.ncols) {
            const int to0 = a.up * t - a.shuf_pad;
            const bool inside = to0 >= 0 && to0 + a.up <= a.Lout;
            static_for<4>([&](auto qq) __attribute__((always_inline)) {
              constexpr int q = decltype(qq)::value;
              if (a.up == 4) {
                if (rok[q][0]) {
                  const st4 v = {(acc[i][n][4 * q] + bz[q][0]) * sc, (acc[i][n][4 * q + 1] + bz[q][0]) * sc,
                                 (acc[i][n][4 * q + 2] + bz[q][0]) * sc, (acc[i][n][4 * q + 3] + bz[q][0]) * sc};
                  if (inside) *reinterpret_cast<st4*>(yrow[q][0] + to0) = v;
                  else {
#pragma unroll
                    for (int ph = 0; ph < 4; ++ph)
                      if (to0 + ph >= 0 && to0 + ph < a.Lout) yrow[q][0][to0 + ph] = v[ph];
                  }
                }
              } else {
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                  if (rok[q][c2]) {
                    const st2 v = {(acc[i][n][4 * q + 2 * c2] + bz[q][c2]) * sc, (acc[i][n][4 * q + 2 * c2 + 1] + bz[q][c2]) * sc};
                    if (inside) *reinterpret_cast<st2*>(yrow[q][c2] + to0) = v;
                    else {
#pragma unroll
                      for (int ph = 0; ph < 2; ++ph)
                        if (to0 + ph >= 0 && to0 + ph < a.Lout) yrow[q][c2][to0 + ph] = v[ph];
                    }
                  }
                }
              }
            });
          }
        });
      });
      return;
    }
    const float up_inv = 1.0f / (float)a.up;
    static_for<TM>([&](auto ii) __attribute__((always_inline)) {
      constexpr int i = decltype(ii)::value;
      float* yrow[16];
      float bz[16];
      int toff[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mw + i * 32 + HSP_ACC_ROW(r, half);
        const int co = (int)(((float)m + 0.5f) * up_inv);
        const bool ok = m < a.M && co < a.Cout;
        toff[r] = ok ? (m - co * a.up) - a.shuf_pad : -(1 << 30);
        yrow[r] = a.y + (int64_t)b * a.y_bs + (int64_t)(ok ? co : 0) * a.y_cs;
        bz[r] = (ok && a.bias) ? a.bias[co] : 0.0f;
      }
      static_for<TN>([&](auto nn) __attribute__((always_inline)) {
        constexpr int n = decltype(nn)::value;
        const int t = tw + n * 32;
        if (t < a.ncols) {
          static_for<16>([&](auto rr) __attribute__((always_inline)) {
            constexpr int r = decltype(rr)::value;
            const int to = a.up * t + toff[r];
            if (to >= 0 && to < a.Lout) yrow[r][to] = (acc[i][n][r] + bz[r]) * a.scale * a.post_scale;
          });
        }
      });
    });
  } else {
    const float up_inv = a.rows == HSP_ROWS_SHUFFLE ? 1.0f / (float)a.up : 1.0f;
    static_for<TM>([&](auto ii) __attribute__((always_inline)) {
      constexpr int i = decltype(ii)::value;
      static_for<TN>([&](auto nn) __attribute__((always_inline)) {
        constexpr int n = decltype(nn)::value;
        const int t = tw + n * 32;
        if (t < a.ncols) {
          static_for<16>([&](auto rr) __attribute__((always_inline)) {
            constexpr int r = decltype(rr)::value;
            const int m = mw + i * 32 + HSP_ACC_ROW(r, half);
            int co = m, to = t;
            bool ok = m < a.M;
            if (a.rows == HSP_ROWS_SHUFFLE) {
              co = (int)(((float)m + 0.5f) * up_inv);  // m / up, exact for m < 2^22 (no integer divide)
              to = a.up * t + (m - co * a.up) - a.shuf_pad;
              ok = ok && to >= 0 && to < a.Lout;
            }
            ok = ok && co < a.Cout;
            if (ok) {
              float v = acc[i][n][r];
              if (a.bias) v += a.bias[co];
              if (a.cbias) v += a.cbias[(int64_t)b * a.cbias_bs + co];
              v = hsp_apply_act(v, a.act);
              hsp_epilogue_store(a, b, co, to, v);
            }
          });
        }
      });
    });
  }
}

// does this instantiation carry the narrow-tail consumer (a last column tile with at most 32 valid columns is
// computed as 4 waves x (32 rows x 32 columns) instead of 4 waves x (64 x 64): a quarter of the MFMAs)?
template <class C, int EPI, bool ACT>
constexpr bool has_tail_path() {
  return EPI == HSP_EPI_INIT && !ACT && C::kWM * C::kWN == 4 && C::kTM * C::kWM == 4 && C::kTN * C::kWN == 4;  // 128 x 128
}

// `sched` = g | tail << 8 (host: launch_one).
//   g = 0: blockIdx.x = mt + n_mt * ct, row tiles fastest: the blocks that round-robin onto one XCD keep hitting the
//          same weight slab in that XCD's L2 -- but the input window of a column tile is then fetched once per row
//          tile, by a different XCD each time.
//   g > 0: XCD-grouped.  Blocks b and b + 8 share an XCD (observed dispatch order; speed only, never correctness).
//          XCD x = blockIdx.x % 8 owns row group (x % (n_mt / g)) = g consecutive row tiles, whose weight slabs fit
//          its L2 together, and walks its column tiles with those g row tiles adjacent in time: the window is
//          fetched from HBM / Infinity Cache once per XCD group instead of once per row tile.
//   tail:  the last column tile of every utterance holds <= 32 valid columns; those tiles are enumerated LAST
//          (short jobs at the end of the grid) and take the narrow consumer.
template <class C, int EPI, bool ACT>
__global__ __launch_bounds__(C::THREADS, C::MINW) void conv1d_mfma_kernel(const hsp_conv1d_args a, const int n_mt,
                                                                          const int n_nt, const int lkc,
                                                                          const int xvec, const int sched) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int BM = C::BM, BN = C::BN, TM = C::kTM, TN = C::kTN;
  const LdsPlan P = make_plan<C>(a.K, a.dil, ACT ? HSP_PRO_ACT1D : HSP_PRO_NONE, lkc);
  const int KC = P.kc;

  int mt, ct;
  {
    const int g = sched & 255;
    const int bid = blockIdx.x;
    if (g == 0) {
      mt = bid % n_mt;
      ct = bid / n_mt;
    } else {
      const int x = bid & 7, q = bid >> 3;
      const int groups_m = n_mt / g, nx = 8 / groups_m;
      mt = (x % groups_m) * g + q % g;
      ct = (q / g) * nx + x / groups_m;
      if (ct >= n_nt * a.B) return;  // grid padding (whole workgroup, before any barrier)
    }
  }
  int nt, b;
  bool tail_tile = false;
  if (sched >> 8) {
    const int n_main = (n_nt - 1) * a.B;
    if (ct < n_main) {
      nt = ct % (n_nt - 1);
      b = ct / (n_nt - 1);
    } else {
      nt = n_nt - 1;
      b = ct - n_main;
      tail_tile = true;
    }
  } else {
    nt = ct % n_nt;
    b = ct / n_nt;
  }
  const int m0 = mt * BM, t0 = nt * BN;
  const int p0 = t0 - a.pad;  // first activated-input position of the window

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int nchunks = (a.Cin + KC - 1) >> lkc;

  if (wave >= C::NCW) {
    // ------------------------------------------------------------ producers
    const int pw = wave - C::NCW;
    const float* const wb = a.w + (int64_t)b * a.w_bs;   // per-"utterance" weights (frequency bins, hsp.h: w_bs), else w_bs == 0
    const ProdArgs pa{wb, a.zeros, a.alpha_exp, a.beta_inv, a.filt, a.K, a.Cin, a.Lin, a.M, a.w_ld, (int)a.x_cs,
                      (int)a.x_ts, a.prologue, a.slope};
    const float* xb = a.x + (int64_t)b * a.x_bs;
    using PR = Prod<C>;
    float* const Ws0 = lds;
    float* const Xa0 = lds + P.xa_off;
    if constexpr (!ACT) {
      const int p0a = p0 & ~3;
      const bool fast = xvec && m0 + BM <= a.M && (a.Cin & (KC - 1)) == 0 && !HSP_DBG(a, 512);  // wave-uniform
      if (fast) {
        const float* const wtile = wb + m0;
        const float* const xtile = xb + p0a;
        const int xcs = (int)a.x_cs;
        const unsigned wofs = 4u * (unsigned)((lane / PR::CPR) * a.w_ld + (lane % PR::CPR) * 4);
        if (p0a < 0 || p0a + 4 * ((P.xw + 3 + 3) >> 2) > a.Lin) PR::zero_x_rows(P, Xa0, pw, lane);
        PR::dma_w_fast(wtile, a.w_ld, a.Cin, a.K, P, Ws0, 0, pw, wofs);
        PR::dma_x16_fast(xtile, xcs, a.Lin, P, Xa0, 0, p0a, pw, lane);
        wait_vm0();
        if (a.prologue == HSP_PRO_LRELU) PR::lrelu_x(pa, P, Xa0, pw, lane);
        HSP_BARRIER(a);
        for (int c = 0; c < nchunks; ++c) {
          const int nb = (c + 1) & 1;
          if (c + 1 < nchunks && !HSP_DBG(a, 1)) {
            PR::dma_w_fast(wtile, a.w_ld, a.Cin, a.K, P, Ws0 + nb * P.ws_sz, (c + 1) << lkc, pw, wofs);
            PR::dma_x16_fast(xtile, xcs, a.Lin, P, Xa0 + nb * P.xa_sz, (c + 1) << lkc, p0a, pw, lane);
            wait_vm0();
            if (a.prologue == HSP_PRO_LRELU) PR::lrelu_x(pa, P, Xa0 + nb * P.xa_sz, pw, lane);
          }
          HSP_BARRIER(a);
        }
        return;
      }
      PR::dma_w(pa, P, Ws0, 0, m0, pw, lane);
      if (xvec) PR::dma_x16(pa, P, Xa0, xb, 0, p0a, pw, lane); else PR::dma_x(pa, P, Xa0, xb, 0, p0, pw, lane);
      wait_vm0();
      if (a.prologue == HSP_PRO_LRELU) PR::lrelu_x(pa, P, Xa0, pw, lane);
      HSP_BARRIER(a);
      for (int c = 0; c < nchunks; ++c) {
        const int nb = (c + 1) & 1;
        if (c + 1 < nchunks && !HSP_DBG(a, 1)) {
          PR::dma_w(pa, P, Ws0 + nb * P.ws_sz, (c + 1) << lkc, m0, pw, lane);
          if (xvec) PR::dma_x16(pa, P, Xa0 + nb * P.xa_sz, xb, (c + 1) << lkc, p0a, pw, lane);
          else PR::dma_x(pa, P, Xa0 + nb * P.xa_sz, xb, (c + 1) << lkc, p0, pw, lane);
          wait_vm0();
          if (a.prologue == HSP_PRO_LRELU) PR::lrelu_x(pa, P, Xa0 + nb * P.xa_sz, pw, lane);
        }
        HSP_BARRIER(a);
      }
    } else {
      float* const scr = lds + P.scr_off + pw * P.scr_sz;   // raw[2][rpw][xrwp], a2[a2w]
      float* const a2 = scr + 2 * P.rpw * P.xrwp;
      const int rsz = P.rpw * P.xrwp;
      PR::dma_raw(pa, P, scr, xb, 0, p0, pw, lane);
      PR::dma_w(pa, P, Ws0, 0, m0, pw, lane);
      wait_vm0();
      PR::act_rows(pa, P, scr, a2, Xa0, 0, p0, pw, lane);
      if (nchunks > 1) PR::dma_raw(pa, P, scr + rsz, xb, KC, p0, pw, lane);
      wait_vm0();
      HSP_BARRIER(a);
      for (int c = 0; c < nchunks; ++c) {
        const int nb = (c + 1) & 1;
        if (c + 1 < nchunks && !HSP_DBG(a, 1)) {
          PR::dma_w(pa, P, Ws0 + nb * P.ws_sz, (c + 1) << lkc, m0, pw, lane);
          if (c + 2 < nchunks) PR::dma_raw(pa, P, scr + (c & 1) * rsz, xb, (c + 2) << lkc, p0, pw, lane);
          PR::act_rows(pa, P, scr + nb * rsz, a2, Xa0 + nb * P.xa_sz, (c + 1) << lkc, p0, pw, lane);
          wait_vm0();
        }
        HSP_BARRIER(a);
      }
    }
    return;
  }

  // -------------------------------------------------------------- consumers
  if constexpr (has_tail_path<C, EPI, ACT>()) {
    if (tail_tile) {
      // <= 32 valid columns: wave w takes rows 32 w .. 32 w + 31 of the tile and the first 32 columns
      conv_consume<C, EPI, ACT, 1, 1>(a, P, lds, b, m0, t0, p0, wave, 0, lane, nchunks, lkc, xvec);
      return;
    }
  }
  if constexpr (C::kKS > 1) {
    const int ks = wave / (C::kWM * C::kWN), w2 = wave % (C::kWM * C::kWN);
    conv_consume<C, EPI, ACT, TM, TN>(a, P, lds, b, m0, t0, p0, w2 / C::kWN, w2 % C::kWN, lane, nchunks, lkc, xvec, ks);
  } else {
    conv_consume<C, EPI, ACT, TM, TN>(a, P, lds, b, m0, t0, p0, wave / C::kWN, wave % C::kWN, lane, nchunks, lkc, xvec);
  }
}

// -------------------------------------------------------------------- host side
constexpr int kMaxLdsBytes = 160 * 1024;
constexpr int kLdsTarget = 80 * 1024;  // two workgroups per CU when possible

// largest chunk depth (log2) this tile shape can stage for the launch, or -1
template <class C>
int pick_lkc(const hsp_conv1d_args& a, int lds_limit) {
  const bool act = a.prologue == HSP_PRO_ACT1D;
  if (C::BN + (a.K - 1) * a.dil + 3 > C::XWP) return -1;  // halo beyond this shape's window pitch
  int cin_p2 = 2;
  while ((1 << cin_p2) < a.Cin && cin_p2 < 6) ++cin_p2;  // no point staging past Cin
  for (int lkc = cin_p2; lkc >= 2; --lkc) {
    const LdsPlan P = make_plan<C>(a.K, a.dil, a.prologue, lkc);
    if (P.total * (int)sizeof(float) > lds_limit) continue;
    if (P.kc < Prod<C>::RPI) continue;   // a weight DMA instruction must stay inside one tap
    if (C::kKS > 1 && (((a.K * P.kc) >> 1) % (2 * C::kKS) != 0 || P.total < C::kWM * C::kWN * (32 * 36 + C::kTM * C::kTN * 16 * 64))) continue;  // even shares; room for the partial sums
    if (act && P.rpw > 2) continue;      // activation work per producer wave and chunk
    return lkc;
  }
  return -1;
}

// Launch one instantiation.  EPI / ACT must already match the arguments (select_epilogue()).
template <class C, int EPI, bool ACT>
int launch_one(const hsp_conv1d_args& a, hipStream_t s, int32_t* plan_out) {
  // a shape whose register budget admits two workgroups per CU keeps its LDS under half the
  // CU's; the others take the deepest chunk the whole 160 KB can stage (fewer barriers, and the
  // DMA of a chunk gets a longer MFMA phase to land under)
  // Round 3: the short-sequence shapes (64 x 64, 64 x 128-gated, 32 x 128: <= 87 VGPRs, one workgroup per CU with the deep
  // chunk) take the half-CU chunk as well once the launch has more tiles than the chip has CUs: the FFN fc1 of an
  // 8-utterance front group is 384 tiles -- two rounds of lone workgroups at 52 us, one round of pairs at ~40.
  const int64_t tiles = (int64_t)((a.M + C::BM - 1) / C::BM) * ((a.ncols + C::BN - 1) / C::BN) * a.B;
  // (same box: 42.4 -> 39.3 us per launch, step 81.05 -> 80.77 ms; beyond 512 tiles the deep chunk wins again: 67.8 vs 69.7)
  // (tuning bit 1 << 26: the half-CU chunk for ANY launch of up to 512 tiles, so that launches of different streams can share a CU)
  // (tuning bit 1 << 29, round 6 / VERDICT r05 item 8: a k <= 3 conv takes the deepest chunk the whole CU can stage -- twice
  // the MFMAs per barrier and window DMA, ONE workgroup per CU; measured in profiles/r06_k3_deep_chunk.txt)
  const bool deep_k3 = HSP_DBG(a, 1 << 29) && a.K <= 3;
  const bool two_per_cu = !deep_k3 && (C::MINW * 256 >= 2 * C::THREADS || ((tiles > 256 || HSP_DBG(a, 1 << 26)) && tiles <= 512 && !HSP_DBG(a, 1 << 21)));
  int lkc = pick_lkc<C>(a, two_per_cu ? kLdsTarget : kMaxLdsBytes);
  if (lkc < 0) lkc = pick_lkc<C>(a, kMaxLdsBytes);
  if (lkc < 0) return HSP_EINVAL;
  int lds_bytes = make_plan<C>(a.K, a.dil, a.prologue, lkc).total * (int)sizeof(float);
  if (plan_out) {
    plan_out[0] = C::BM; plan_out[1] = C::BN; plan_out[2] = 1 << lkc; plan_out[3] = lds_bytes;
    return 0;
  }
  const int n_mt = (a.M + C::BM - 1) / C::BM;
  const int n_nt = (a.ncols + C::BN - 1) / C::BN;
  // block schedule (see the kernel): row tiles per XCD group, narrow tail tiles
  const int64_t n_ct = (int64_t)n_nt * a.B;
  // g > 0 (XCD-grouped row tiles) is a measured LOSS and exists in the tuning build only (debug bit 16384):
  // k = 3 convs run 4-5 % slower with it (C = 512 / L = 800: 114.5 vs 119.9 TFLOP/s, C = 256 / L = 4000: 120.9 vs
  // 126.2), k = 7 / 11 are unchanged (profiles/r03_sched_sweep.txt) -- the blocks of one XCD sharing ONE weight slab
  // matters more than fetching a window once.
  int g = 0;
  if (HSP_DBG(a, 16384) && n_mt >= 1 && n_mt <= 8 && (8 % n_mt) == 0) {
    // the largest group of row tiles whose weight slabs share one XCD's 4-MB L2 with room for the streamed windows
    const int64_t slab = (int64_t)a.K * a.Cin * C::BM * (int64_t)sizeof(float);
    g = 1;
    for (int cand = n_mt; cand > 1; cand >>= 1)
      if (n_mt % cand == 0 && (int64_t)cand * slab <= (int64_t)3200 * 1024) { g = cand; break; }
  }
  const int rem = a.ncols - (n_nt - 1) * C::BN;
  const int tail = (has_tail_path<C, EPI, ACT>() && n_nt >= 2 && rem <= 32 && !HSP_DBG(a, 32768)) ? 1 : 0;
  int64_t blocks = (int64_t)n_mt * n_ct;
  if (g > 0) {
    const int nx = 8 / (n_mt / g);
    blocks = 8 * (int64_t)g * ((n_ct + nx - 1) / nx);
  }
  if (blocks <= 0 || blocks > 0x7fffffff) return HSP_EINVAL;
  const int sched = g | (tail << 8);
  auto kern = conv1d_mfma_kernel<C, EPI, ACT>;
  // raise the kernel's dynamic-LDS cap once per device (the attribute is per device; kept out of the
  // launch path afterwards so that launches are legal inside a hipGraph stream capture)
  static hsp_lds_flags flags;
  if (lds_bytes > 32 * 1024)
    if (int e = hsp_raise_lds_limit(reinterpret_cast<const void*>(kern), kMaxLdsBytes, flags)) return e;
  if (EPI == HSP_EPI_VEC && lds_bytes < C::NCW * 32 * 36 * 4) lds_bytes = C::NCW * 32 * 36 * 4;  // staging area
  // 16-B window DMA: plain prologue on a 16-B addressable input whose length is a multiple of 4
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool xvec = !ACT && a.x_ts == 1 && (a.Lin & 3) == 0 && (a.x_cs & 3) == 0 && (a.x_bs & 3) == 0 && al16(a.x) &&
                    !HSP_DBG(a, 64);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::THREADS), lds_bytes, s, a, n_mt, n_nt, lkc, xvec ? 1 : 0, sched);
  return (int)hipGetLastError();
}

// the epilogue kind the arguments ask for
inline int select_epilogue(const hsp_conv1d_args& a) {
  if (a.rows == HSP_ROWS_GATE_WN || a.rows == HSP_ROWS_GATE_GLU) return HSP_EPI_GATE;
  const bool pointwise = a.act != HSP_ACT_NONE || a.mask_mode != HSP_MASK_NONE || a.cscale;
  if (a.rows == HSP_ROWS_SHUFFLE)
    return (!pointwise && !a.cbias && !a.res && !a.accumulate) ? HSP_EPI_SHUF : HSP_EPI_GEN;
  if (!pointwise && a.scale == 1.0f && !HSP_DBG(a, 32)) return HSP_EPI_INIT;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool vec = (a.ncols & 3) == 0 && al16(a.y) && (a.y_bs & 3) == 0 && (a.y_cs & 3) == 0 &&
                   (!a.res || (al16(a.res) && (a.res_bs & 3) == 0 && (a.res_cs & 3) == 0)) &&
                   (a.mask_mode == HSP_MASK_NONE || (al16(a.mask) && (a.mask_bs & 3) == 0));
  return vec ? HSP_EPI_VEC : HSP_EPI_GEN;
}

// tile shapes <WM, WN, TM, TN, producer waves, min waves per SIMD>
using M128 = Cfg<2, 2, 2, 2, 4, 4>;    // 128 x 128, two workgroups per CU
using M64 = Cfg<1, 4, 2, 2, 8, 3>;     //  64 x 256, eight producer waves (fused activation prologue)
using M32 = Cfg<1, 4, 1, 4, 8, 3>;     //  32 x 512, likewise
using M64P = Cfg<1, 4, 2, 2, 4, 4>;    //  64 x 256, plain prologue: four producer waves, two workgroups per CU
using M32P = Cfg<1, 4, 1, 4, 4, 4>;    //  32 x 512, likewise
using S64 = Cfg<2, 2, 1, 1, 4, 2>;     //  64 x 64   short sequences: many small, deep-chunk tiles
using S64G = Cfg<1, 4, 2, 1, 4, 2>;    //  64 x 128  short sequences, gated rows (needs TM even)
using S32 = Cfg<1, 4, 1, 1, 4, 2>;     //  32 x 128
using S64W = Cfg<2, 2, 1, 1, 4, 2, 128>;   //  64 x 64 with a 192-float window pitch: halos of 62 ... 125 columns
using S64GW = Cfg<1, 4, 2, 1, 4, 2, 128>;  //  64 x 128 gated rows with a 256-float pitch (WN with dilation_rate > 1)
using S64G2 = Cfg<1, 2, 2, 1, 4, 2, 64, 2>;  // 64 x 64 gated rows, the K range of a chunk split between two wave pairs (round 5)

}  // namespace hspconv

// One function per tile shape (hsp_conv1d_tile.hip, compiled once per shape): launches the instantiation
// for (epi, act) or returns HSP_EINVAL when the shape does not carry that combination.
#define HSP_TILE_LIST(X) X(M128) X(M64) X(M32) X(M64P) X(M32P) X(S64) X(S64G) X(S32) X(S64W) X(S64GW) X(S64G2)
#define HSP_TILE_DECL(name) \
  int hsp_conv_tile_##name(const hsp_conv1d_args& a, int epi, bool act, hipStream_t s, int32_t* plan_out);
HSP_TILE_LIST(HSP_TILE_DECL)
#undef HSP_TILE_DECL
