// Fused Conv1d / ConvTranspose1d as an implicit GEMM on the fp32-input matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fmaf chain, same 157 TFLOP/s peak as
// the fp32 vector pipe -- MI355X_MICROARCH.md "Matrix cores").
//
//   acc[m, t] = sum_j sum_ci W[j][ci][m] * xin[ci, t + j*dil - pad]
//
// GEMM view: M = packed output rows, N = time, K = (taps x input channels).
// One 512-thread workgroup owns a BM x BN output tile of one utterance and walks the
// input channels in chunks of KC.  Its 8 waves are SPECIALISED:
//
//   waves 0-3  "consumers": each owns a (TM x TN) grid of 32x32 MFMA blocks and does
//              nothing but read A (weights) / B (shifted input window) fragments from
//              LDS and issue MFMAs;
//   waves 4-7  "producers": stage the NEXT chunk into the other half of a double
//              buffer -- the weight slab Ws[K][KC][BM] and the input window
//              Xa[KC][BN + (K-1)*dil] -- applying the PROLOGUE on the way: nothing,
//              leaky-ReLU, or the whole anti-aliased SnakeBeta activation (2x polyphase
//              up-sample -> snake -> 2x low-pass down-sample, replicate-padded at the
//              sequence ends exactly like alias_free_torch).  Each producer wave owns
//              whole channel rows, so the three activation phases need only wave-local
//              ordering, and the activated tensor never exists in HBM.
//
// Wave w and wave w+4 share a SIMD, so the producer's VALU/LDS/global work fills the
// issue slots between the consumer's 64-cycle MFMAs; the only workgroup-wide
// synchronisation is one barrier per chunk.  The EPILOGUE (bias, conditioning bias,
// gate / pointwise function, masks, per-channel scale, residual, running accumulation,
// ConvTranspose phase shuffle) runs on the accumulator registers of the consumers.
//
// Reference call sites: see include/hsp.h (hsp_conv1d_args).
#include <atomic>
#include <type_traits>
#include "hsp_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int NCW = 4;               // consumer waves
constexpr int NPW = 4;               // producer waves
constexpr int THREADS = 64 * (NCW + NPW);

template <int WM, int WN, int TM, int TN, int KC>
struct Cfg {
  static_assert(WM * WN == NCW, "four consumer waves");
  static constexpr int kWM = WM, kWN = WN, kTM = TM, kTN = TN, kKC = KC;
  static constexpr int BM = WM * TM * 32;
  static constexpr int BN = WN * TN * 32;
};

struct LdsPlan {
  int xw;    // activated window width  = BN + (K-1)*dil
  int xrw;   // raw window width        = xw + 10        (ACT1D)
  int a2w;   // 2x-rate window width    = 2*xw + 10      (ACT1D)
  int ws_sz, xa_sz, scr_sz;  // floats: one weight slab, one window buffer, one producer scratch
  int ws_off, xa_off, scr_off, total;
};

template <class C>
__host__ __device__ inline LdsPlan make_plan(int K, int dil, int prologue) {
  LdsPlan p;
  p.xw = C::BN + (K - 1) * dil;
  p.xrw = p.xw + 10;
  p.a2w = 2 * p.xw + 10;
  p.ws_sz = K * C::kKC * C::BM;
  p.xa_sz = (C::kKC * p.xw + 3) & ~3;
  p.scr_sz = prologue == HSP_PRO_ACT1D ? ((p.xrw + p.a2w + 3) & ~3) : 0;
  p.ws_off = 0;
  p.xa_off = 2 * p.ws_sz;
  p.scr_off = p.xa_off + 2 * p.xa_sz;
  p.total = p.scr_off + NPW * p.scr_sz;
  return p;
}

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// wave-local LDS ordering: all earlier LDS ops of this wave are complete and the
// compiler may not move memory accesses across this point
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <class C>
__device__ __forceinline__ void produce_chunk(const hsp_conv1d_args& a, const LdsPlan& P, float* __restrict__ Ws,
                                              float* __restrict__ Xa, float* __restrict__ scr, const float* xb,
                                              int c0, int m0, int p0, int pw, int lane) {
  constexpr int BM = C::BM, KC = C::kKC;
  const int L = a.Lin;
  // ---- weight slab: Ws[j][kc][mm] = w[j][c0+kc][m0+mm]
  {
    const int nvec = a.K * KC * (BM / 4);
    for (int v = pw * 64 + lane; v < nvec; v += NPW * 64) {
      const int mm4 = v % (BM / 4);
      const int rest = v / (BM / 4);
      const int kc = rest % KC, j = rest / KC;
      const int ci = c0 + kc, m = m0 + mm4 * 4;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ci < a.Cin && m < a.M) val = *reinterpret_cast<const float4*>(a.w + ((int64_t)j * a.Cin + ci) * a.w_ld + m);
      *reinterpret_cast<float4*>(Ws + (j * KC + kc) * BM + mm4 * 4) = val;
    }
  }
  // ---- input window rows owned by this producer wave
  if (a.prologue != HSP_PRO_ACT1D) {
    for (int kc = pw; kc < KC; kc += NPW) {
      const int ci = c0 + kc;
      const float* xc = xb + (int64_t)ci * a.x_cs;
      for (int s = lane; s < P.xw; s += 64) {
        const int p = p0 + s;
        float v = 0.0f;
        if (ci < a.Cin && p >= 0 && p < L) {
          v = xc[(int64_t)p * a.x_ts];
          if (a.prologue == HSP_PRO_LRELU) v = v > 0.0f ? v : v * a.slope;
        }
        Xa[kc * P.xw + s] = v;
      }
    }
    return;
  }
  float* const raw = scr;          // [xrw]  x[clamp(p0-5+s)]
  float* const a2 = scr + P.xrw;   // [a2w]  snake(up2x)[clamp(2*p0-5+s)]
  const int mlo = 2 * p0 - 5;
  for (int kc = pw; kc < KC; kc += NPW) {
    const int ci = c0 + kc;
    float* const xa = Xa + kc * P.xw;
    if (ci >= a.Cin) {
      for (int s = lane; s < P.xw; s += 64) xa[s] = 0.0f;
      continue;
    }
    const float* xc = xb + (int64_t)ci * a.x_cs;
    const float ea = a.alpha_exp[ci], binv = a.beta_inv[ci];
    // phase A: raw samples, replicate-padded (index clamp) -- F.pad(mode='replicate')
    for (int s = lane; s < P.xrw; s += 64) raw[s] = xc[hsp_clampi(p0 - 5 + s, 0, L - 1)];
    wave_lds_fence();
    // phase B: a[m] = snake(2 * up[m]); even m=2q: taps h[11],h[9],..,h[1] on x[q-3..q+2],
    //          odd m=2q+1: taps h[10],h[8],..,h[0] on x[q-2..q+3]
    for (int s = lane; s < P.a2w; s += 64) {
      const int m = hsp_clampi(mlo + s, 0, 2 * L - 1);
      const int q = m >> 1, odd = m & 1;
      const float* xr = raw + (q - 3 + odd) - (p0 - 5);
      float u = 0.0f;
#pragma unroll
      for (int i = 0; i < 6; ++i) u = fmaf(xr[i], odd ? a.filt[10 - 2 * i] : a.filt[11 - 2 * i], u);
      a2[s] = hsp_snake(2.0f * u, ea, binv);
    }
    wave_lds_fence();
    // phase C: y[p] = sum_k hd[k] * a[clamp(2p+k-5)], zero outside [0, L) (conv zero padding)
    for (int s = lane; s < P.xw; s += 64) {
      const int p = p0 + s;
      float v = 0.0f;
      if (p >= 0 && p < L) {
        const float* ar = a2 + 2 * s;
#pragma unroll
        for (int k = 0; k < 12; ++k) v = fmaf(a.filt[12 + k], ar[k], v);
      }
      xa[s] = v;
    }
    wave_lds_fence();  // raw / a2 are reused by the next row
  }
}

template <class C>
__global__ __launch_bounds__(THREADS) void conv1d_mfma_kernel(const hsp_conv1d_args a, const int n_mt,
                                                              const int n_nt) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int BM = C::BM, BN = C::BN, KC = C::kKC, TM = C::kTM, TN = C::kTN;
  const LdsPlan P = make_plan<C>(a.K, a.dil, a.prologue);

  // blockIdx.x = mt + n_mt * (nt + n_nt * b): row tiles fastest, so the blocks that
  // round-robin onto one XCD keep hitting the same weight slab in that XCD's L2.
  int bid = blockIdx.x;
  const int mt = bid % n_mt;
  bid /= n_mt;
  const int nt = bid % n_nt;
  const int b = bid / n_nt;
  const int m0 = mt * BM, t0 = nt * BN;
  const int p0 = t0 - a.pad;  // first activated-input position of the window

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int nchunks = (a.Cin + KC - 1) / KC;

  if (wave >= NCW) {
    // ------------------------------------------------------------ producers
    const int pw = wave - NCW;
    const float* xb = a.x + (int64_t)b * a.x_bs;
    float* scr = lds + P.scr_off + pw * P.scr_sz;
    produce_chunk<C>(a, P, lds + P.ws_off, lds + P.xa_off, scr, xb, 0, m0, p0, pw, lane);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
      const int nb = (c + 1) & 1;
      if (c + 1 < nchunks)
        produce_chunk<C>(a, P, lds + P.ws_off + nb * P.ws_sz, lds + P.xa_off + nb * P.xa_sz, scr, xb, (c + 1) * KC, m0,
                         p0, pw, lane);
      __syncthreads();
    }
    return;
  }

  // -------------------------------------------------------------- consumers
  const int wm = wave / C::kWN, wn = wave % C::kWN;
  const int l32 = lane & 31, half = lane >> 5;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  __syncthreads();  // chunk 0 staged
  for (int c = 0; c < nchunks; ++c) {
    const int cb = c & 1;
    const float* wbase = lds + P.ws_off + cb * P.ws_sz + half * BM + wm * (TM * 32) + l32;
    const float* xbase = lds + P.xa_off + cb * P.xa_sz + half * P.xw + wn * (TN * 32) + l32;
    for (int j = 0; j < a.K; ++j) {
      const float* wj = wbase + j * (KC * BM);
      const float* xj = xbase + j * a.dil;
#pragma unroll
      for (int kk = 0; kk < KC / 2; ++kk) {
        float fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = wj[(2 * kk) * BM + i * 32];
#pragma unroll
        for (int i = 0; i < TN; ++i) fb[i] = xj[(2 * kk) * P.xw + i * 32];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int n = 0; n < TN; ++n)
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[n], acc[i][n], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue.  C/D map of the 32x32 forms: col = lane & 31,
  //      row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
  // static_for keeps every accumulator index a compile-time constant: a runtime index
  // into acc[][] would demote the whole accumulator file to scratch memory.
  const int mw = m0 + wm * (TM * 32);
  const int tw = t0 + wn * (TN * 32) + l32;
  if (a.rows == HSP_ROWS_GATE_WN || a.rows == HSP_ROWS_GATE_GLU) {
    if constexpr (TM % 2 == 0) {
      const int H = a.gate_half;
      static_for<TM / 2>([&](auto ih) {
        constexpr int i = 2 * decltype(ih)::value;
        const int mpair = mw + i * 32;  // packed row of the 'a' block; multiple of 64
        static_for<TN>([&](auto nn) {
          constexpr int n = decltype(nn)::value;
          const int t = tw + n * 32;
          if (mpair < a.M && t < a.ncols) {
            static_for<16>([&](auto rr) {
              constexpr int r = decltype(rr)::value;
              const int co = (mpair >> 6) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
              if (co < H) {
                float va = acc[i][n][r], vb = acc[i + 1][n][r];
                if (a.bias) { va += a.bias[co]; vb += a.bias[H + co]; }
                if (a.cbias) {
                  va += a.cbias[(int64_t)b * a.cbias_bs + co];
                  vb += a.cbias[(int64_t)b * a.cbias_bs + H + co];
                }
                const float v = (a.rows == HSP_ROWS_GATE_WN ? tanhf(va) : va) * hsp_sigmoid(vb);
                hsp_epilogue_store(a, b, co, t, v);
              }
            });
          }
        });
      });
    }
  } else {
    static_for<TM>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      static_for<TN>([&](auto nn) {
        constexpr int n = decltype(nn)::value;
        const int t = tw + n * 32;
        if (t < a.ncols) {
          static_for<16>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            const int m = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            int co = m, to = t;
            bool ok = m < a.M;
            if (a.rows == HSP_ROWS_SHUFFLE) {
              co = m / a.up;
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

constexpr int kMaxLdsBytes = 160 * 1024;

template <class C>
int lds_bytes_of(const hsp_conv1d_args& a) {
  return make_plan<C>(a.K, a.dil, a.prologue).total * (int)sizeof(float);
}

template <class C>
int launch(const hsp_conv1d_args& a, hipStream_t s, int32_t* plan_out) {
  const int lds_bytes = lds_bytes_of<C>(a);
  if (plan_out) {
    plan_out[0] = C::BM; plan_out[1] = C::BN; plan_out[2] = C::kKC; plan_out[3] = lds_bytes;
    return 0;
  }
  if (lds_bytes > kMaxLdsBytes) return HSP_EINVAL;
  const int n_mt = (a.M + C::BM - 1) / C::BM;
  const int n_nt = (a.ncols + C::BN - 1) / C::BN;
  const int64_t blocks = (int64_t)n_mt * n_nt * a.B;
  if (blocks <= 0 || blocks > 0x7fffffff) return HSP_EINVAL;
  auto kern = conv1d_mfma_kernel<C>;
  // raise the kernel's dynamic-LDS cap once per size (idempotent; kept out of the launch
  // path afterwards so that launches are legal inside a hipGraph stream capture)
  static std::atomic<int> lds_cap{32 * 1024};
  if (lds_bytes > lds_cap.load(std::memory_order_relaxed)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes);
    if (e != hipSuccess) return (int)e;
    lds_cap.store(kMaxLdsBytes, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), lds_bytes, s, a, n_mt, n_nt);
  return (int)hipGetLastError();
}

int validate(const hsp_conv1d_args& a) {
  if (!a.x || !a.w || !a.y) return HSP_EINVAL;
  if (a.B <= 0 || a.Cin <= 0 || a.Lin <= 0 || a.K <= 0 || a.M <= 0 || a.Cout <= 0 || a.Lout <= 0 || a.ncols <= 0)
    return HSP_EINVAL;
  if (a.stride != 1 || (a.M & 3) || (a.w_ld & 3) || a.w_ld < a.M || a.dil < 1) return HSP_EINVAL;
  if ((reinterpret_cast<uintptr_t>(a.w) & 15) != 0) return HSP_EINVAL;
  if (a.prologue == HSP_PRO_ACT1D && (!a.alpha_exp || !a.beta_inv || !a.filt || a.x_ts != 1)) return HSP_EINVAL;
  if (a.prologue != HSP_PRO_NONE && a.prologue != HSP_PRO_LRELU && a.prologue != HSP_PRO_ACT1D) return HSP_EINVAL;
  if (a.mask_mode != HSP_MASK_NONE && !a.mask) return HSP_EINVAL;
  if (a.rows == HSP_ROWS_GATE_WN || a.rows == HSP_ROWS_GATE_GLU) {
    if (a.gate_half <= 0 || (a.gate_half & 31) || a.M != 2 * a.gate_half || a.Cout != a.gate_half) return HSP_EINVAL;
  } else if (a.rows == HSP_ROWS_SHUFFLE) {
    if (a.up <= 0 || a.M != a.Cout * a.up) return HSP_EINVAL;
  } else if (a.rows != HSP_ROWS_PLAIN) {
    return HSP_EINVAL;
  }
  return 0;
}

// tile configurations <WM, WN, TM, TN, KC>; the chunk depth KC is picked per launch so
// that the double-buffered weight slab (2 * K * KC * BM floats) fits LDS
template <int KC> using M256 = Cfg<2, 2, 4, 2, KC>;  // 256 x 128
template <int KC> using M128 = Cfg<2, 2, 2, 2, KC>;  // 128 x 128
template <int KC> using M64 = Cfg<1, 4, 2, 2, KC>;   //  64 x 256
template <int KC> using M32 = Cfg<1, 4, 1, 4, KC>;   //  32 x 512
template <int KC> using M64S = Cfg<1, 4, 2, 1, KC>;  //  64 x 128  (short sequences)
template <int KC> using M32S = Cfg<1, 4, 1, 1, KC>;  //  32 x 128  (short sequences)

constexpr int kLdsTarget = 80 * 1024;  // two workgroups per CU when possible

template <template <int> class T>
int launch_kc(const hsp_conv1d_args& a, hipStream_t s, int32_t* plan_out) {
  if (lds_bytes_of<T<16>>(a) <= kLdsTarget) return launch<T<16>>(a, s, plan_out);
  if (lds_bytes_of<T<8>>(a) <= kLdsTarget) return launch<T<8>>(a, s, plan_out);
  if (lds_bytes_of<T<4>>(a) <= kLdsTarget) return launch<T<4>>(a, s, plan_out);
  if (lds_bytes_of<T<8>>(a) <= kMaxLdsBytes) return launch<T<8>>(a, s, plan_out);
  return launch<T<4>>(a, s, plan_out);
}

int dispatch(const hsp_conv1d_args& a, hipStream_t s, int32_t* plan_out) {
  const bool short_seq = a.ncols <= 1024;
  if (a.M > 128) return launch_kc<M256>(a, s, plan_out);
  if (a.M > 64) return launch_kc<M128>(a, s, plan_out);
  if (a.M > 32) return short_seq ? launch_kc<M64S>(a, s, plan_out) : launch_kc<M64>(a, s, plan_out);
  return short_seq ? launch_kc<M32S>(a, s, plan_out) : launch_kc<M32>(a, s, plan_out);
}

}  // namespace

extern "C" int hsp_conv1d_mfma_f32(const hsp_conv1d_args* a, void* stream) {
  if (!a) return HSP_EINVAL;
  if (int e = validate(*a)) return e;
  return dispatch(*a, static_cast<hipStream_t>(stream), nullptr);
}

extern "C" int hsp_conv1d_mfma_plan(const hsp_conv1d_args* a, int32_t out4[4]) {
  if (!a || !out4) return HSP_EINVAL;
  if (int e = validate(*a)) return e;
  return dispatch(*a, nullptr, out4);
}
