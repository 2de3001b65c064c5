// Fused Conv1d / ConvTranspose1d as an implicit GEMM on the fp32-input matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fmaf chain, same peak as the fp32
// vector pipe -- MI355X_MICROARCH.md "Matrix cores").
//
//   acc[m, t] = sum_j sum_ci W[j][ci][m] * xin[ci, t + j*dil - pad]
//
// GEMM view: M = packed output rows, N = time, K = (taps x input channels).
// One workgroup owns a BM x BN output tile of one utterance and walks the input
// channels in chunks of KC.  Per chunk it stages
//   * the weight slab  Ws[K][KC][BM]                       (global -> LDS, float4)
//   * the input window Xa[KC][BN + (K-1)*dil]              (global -> LDS)
//     with the PROLOGUE applied on the way: nothing, leaky-ReLU, or the whole
//     anti-aliased SnakeBeta activation (2x polyphase up-sample -> snake -> 2x
//     low-pass down-sample, replicate-padded at the sequence ends exactly like
//     alias_free_torch) computed in LDS so the activated tensor never exists in HBM,
// then every wave issues its TM x TN grid of 32x32x2 MFMAs per (tap, channel-pair),
// reading A (weights) and B (shifted input window) fragments straight from LDS.
// The EPILOGUE (bias, conditioning bias, gate / pointwise function, masks, per-channel
// scale, residual, running accumulation, ConvTranspose phase shuffle) runs on the
// accumulator registers.
//
// Reference call sites: see include/hsp.h (hsp_conv1d_args).
#include "hsp_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

template <int WM, int WN, int TM, int TN, int KC>
struct Cfg {
  static constexpr int kWM = WM, kWN = WN, kTM = TM, kTN = TN, kKC = KC;
  static constexpr int BM = WM * TM * 32;
  static constexpr int BN = WN * TN * 32;
  static constexpr int THREADS = WM * WN * 64;
};

struct LdsPlan {
  int xw;       // activated window width  = BN + (K-1)*dil
  int xrw;      // raw window width        = xw + 10        (ACT1D)
  int a2w;      // 2x-rate window width    = 2*xw + 10      (ACT1D)
  int ws_off, r1_off, r2_off, total;  // in floats
};

template <class C>
__host__ __device__ inline LdsPlan make_plan(int K, int dil, int prologue) {
  LdsPlan p;
  p.xw = C::BN + (K - 1) * dil;
  p.xrw = p.xw + 10;
  p.a2w = 2 * p.xw + 10;
  p.ws_off = 0;
  int ws = K * C::kKC * C::BM;
  p.r1_off = ws;
  int r1 = C::kKC * (prologue == HSP_PRO_ACT1D ? p.xrw : p.xw);
  r1 = (r1 + 3) & ~3;
  p.r2_off = p.r1_off + r1;
  int r2 = prologue == HSP_PRO_ACT1D ? C::kKC * p.a2w : 0;
  p.total = p.r2_off + r2;
  return p;
}

template <class C>
__global__ __launch_bounds__(C::THREADS) void conv1d_mfma_kernel(const hsp_conv1d_args a, const int n_mt,
                                                                  const int n_nt) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int BM = C::BM, BN = C::BN, KC = C::kKC, TM = C::kTM, TN = C::kTN, NT = C::THREADS, NW = C::THREADS / 64;
  const LdsPlan P = make_plan<C>(a.K, a.dil, a.prologue);
  float* const Ws = lds + P.ws_off;
  float* const R1 = lds + P.r1_off;  // raw window (ACT1D) then activated window
  float* const R2 = lds + P.r2_off;  // 2x-rate snake signal (ACT1D)

  // blockIdx.x = mt + n_mt * (nt + n_nt * b): row tiles fastest, so the blocks that
  // round-robin onto one XCD keep hitting the same weight slab in that XCD's L2.
  int bid = blockIdx.x;
  const int mt = bid % n_mt;
  bid /= n_mt;
  const int nt = bid % n_nt;
  const int b = bid / n_nt;
  const int m0 = mt * BM, t0 = nt * BN;
  const int p0 = t0 - a.pad;  // first activated-input position of the window
  const int L = a.Lin;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / C::kWN, wn = wave % C::kWN;
  const int l32 = lane & 31, half = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const float* xb = a.x + (int64_t)b * a.x_bs;

  for (int c0 = 0; c0 < a.Cin; c0 += KC) {
    __syncthreads();  // previous chunk's MFMA phase has finished reading Ws / R1
    // ---- weight slab: Ws[j][kc][mm] = w[j][c0+kc][m0+mm]
    {
      const int nvec = a.K * KC * (BM / 4);
      for (int v = tid; v < nvec; v += NT) {
        const int mm4 = v % (BM / 4);
        const int rest = v / (BM / 4);
        const int kc = rest % KC, j = rest / KC;
        const int ci = c0 + kc, m = m0 + mm4 * 4;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ci < a.Cin && m < a.M) val = *reinterpret_cast<const float4*>(a.w + ((int64_t)j * a.Cin + ci) * a.w_ld + m);
        *reinterpret_cast<float4*>(Ws + (j * KC + kc) * BM + mm4 * 4) = val;
      }
    }
    // ---- input window
    if (a.prologue != HSP_PRO_ACT1D) {
      for (int kc = wave; kc < KC; kc += NW) {
        const int ci = c0 + kc;
        const float* xc = xb + (int64_t)ci * a.x_cs;
        for (int s = lane; s < P.xw; s += 64) {
          const int p = p0 + s;
          float v = 0.0f;
          if (ci < a.Cin && p >= 0 && p < L) {
            v = xc[(int64_t)p * a.x_ts];
            if (a.prologue == HSP_PRO_LRELU) v = v > 0.0f ? v : v * a.slope;
          }
          R1[kc * P.xw + s] = v;
        }
      }
    } else {
      // phase A: raw samples, replicate-padded (index clamp) -- F.pad(mode='replicate')
      {
        for (int kc = wave; kc < KC; kc += NW) {
          const int ci = c0 + kc;
          const float* xc = xb + (int64_t)ci * a.x_cs;
          for (int s = lane; s < P.xrw; s += 64) {
            float v = 0.0f;
            if (ci < a.Cin) v = xc[hsp_clampi(p0 - 5 + s, 0, L - 1)];
            R1[kc * P.xrw + s] = v;
          }
        }
      }
      __syncthreads();
      // phase B: a[m] = snake(up2x[m]) at the 2x rate; slot s holds a[clamp(2*p0-5+s)]
      {
        float hu[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) hu[i] = a.filt[i];
        const int mlo = 2 * p0 - 5;
        for (int kc = wave; kc < KC; kc += NW) {
          const int ci = c0 + kc;
          const bool live = ci < a.Cin;
          const float ea = live ? a.alpha_exp[ci] : 0.0f, binv = live ? a.beta_inv[ci] : 0.0f;
          for (int s = lane; s < P.a2w; s += 64) {
            float v = 0.0f;
            if (live) {
              const int m = hsp_clampi(mlo + s, 0, 2 * L - 1);
              const int q = m >> 1, odd = m & 1;
              // even m=2q: taps h[11],h[9],..,h[1] on x[q-3..q+2]; odd: h[10],..,h[0] on x[q-2..q+3]
              const float* xr = R1 + kc * P.xrw + (q - 3 + odd) - (p0 - 5);
              float u = 0.0f;
#pragma unroll
              for (int i = 0; i < 6; ++i) u = fmaf(xr[i], odd ? hu[10 - 2 * i] : hu[11 - 2 * i], u);
              u *= 2.0f;
              v = hsp_snake(u, ea, binv);
            }
            R2[kc * P.a2w + s] = v;
          }
        }
      }
      __syncthreads();
      // phase C: y[p] = sum_k hd[k] * a[clamp(2p+k-5)], zero outside [0, L) (conv zero padding)
      {
        float hd[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) hd[i] = a.filt[12 + i];
        for (int kc = wave; kc < KC; kc += NW) {
          const bool live = c0 + kc < a.Cin;
          for (int s = lane; s < P.xw; s += 64) {
            const int p = p0 + s;
            float v = 0.0f;
            if (live && p >= 0 && p < L) {
              const float* ar = R2 + kc * P.a2w + 2 * s;
#pragma unroll
              for (int k = 0; k < 12; ++k) v = fmaf(hd[k], ar[k], v);
            }
            R1[kc * P.xw + s] = v;
          }
        }
      }
    }
    __syncthreads();
    // ---- MFMA phase
    {
      const float* wbase = Ws + half * BM + wm * (TM * 32) + l32;
      const float* xbase = R1 + half * P.xw + wn * (TN * 32) + l32;
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
    }
  }

  // ---- epilogue.  C/D map of the 32x32 forms: col = lane & 31,
  //      row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int mw = m0 + wm * (TM * 32);
  const int tw = t0 + wn * (TN * 32) + l32;
  if (a.rows == HSP_ROWS_GATE_WN || a.rows == HSP_ROWS_GATE_GLU) {
    if constexpr (TM % 2 == 0) {
      const int H = a.gate_half;
#pragma unroll
      for (int i = 0; i < TM; i += 2) {
        const int mpair = mw + i * 32;  // packed row of the 'a' block; multiple of 64
        if (mpair >= a.M) continue;
#pragma unroll
        for (int n = 0; n < TN; ++n) {
          const int t = tw + n * 32;
          if (t >= a.ncols) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = (mpair >> 6) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co >= H) continue;
            float va = acc[i][n][r], vb = acc[i + 1][n][r];
            if (a.bias) { va += a.bias[co]; vb += a.bias[H + co]; }
            if (a.cbias) {
              va += a.cbias[(int64_t)b * a.cbias_bs + co];
              vb += a.cbias[(int64_t)b * a.cbias_bs + H + co];
            }
            const float v = (a.rows == HSP_ROWS_GATE_WN ? tanhf(va) : va) * hsp_sigmoid(vb);
            hsp_epilogue_store(a, b, co, t, v);
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        const int t = tw + n * 32;
        if (t >= a.ncols) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (m >= a.M) continue;
          int co = m, to = t;
          if (a.rows == HSP_ROWS_SHUFFLE) {
            co = m / a.up;
            to = a.up * t + (m - co * a.up) - a.shuf_pad;
            if (to < 0 || to >= a.Lout) continue;
          }
          if (co >= a.Cout) continue;
          float v = acc[i][n][r];
          if (a.bias) v += a.bias[co];
          if (a.cbias) v += a.cbias[(int64_t)b * a.cbias_bs + co];
          v = hsp_apply_act(v, a.act);
          hsp_epilogue_store(a, b, co, to, v);
        }
      }
    }
  }
}

// tile configurations: <WM, WN, TM, TN, KC>
using CfgM128 = Cfg<2, 2, 2, 2, 8>;  // 128 x 128
using CfgM64 = Cfg<1, 4, 2, 2, 8>;   //  64 x 256
using CfgM32 = Cfg<1, 4, 1, 4, 8>;   //  32 x 512
using CfgM32S = Cfg<1, 4, 1, 1, 8>;  //  32 x 128  (short sequences)
using CfgM64S = Cfg<1, 4, 2, 1, 8>;  //  64 x 128  (short sequences)

constexpr int kMaxLdsBytes = 160 * 1024;

template <class C>
int launch(const hsp_conv1d_args& a, hipStream_t s, int32_t* plan_out) {
  const LdsPlan P = make_plan<C>(a.K, a.dil, a.prologue);
  const int lds_bytes = P.total * (int)sizeof(float);
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
  if (lds_bytes > 32 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::THREADS), lds_bytes, s, a, n_mt, n_nt);
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

int dispatch(const hsp_conv1d_args& a, hipStream_t s, int32_t* plan_out) {
  const bool gated = a.rows == HSP_ROWS_GATE_WN || a.rows == HSP_ROWS_GATE_GLU;
  const bool short_seq = a.ncols <= 1024;
  if (a.M > 64 || (gated && a.M > 64)) return launch<CfgM128>(a, s, plan_out);
  if (gated) return short_seq ? launch<CfgM64S>(a, s, plan_out) : launch<CfgM64>(a, s, plan_out);
  if (a.M > 32) return short_seq ? launch<CfgM64S>(a, s, plan_out) : launch<CfgM64>(a, s, plan_out);
  return short_seq ? launch<CfgM32S>(a, s, plan_out) : launch<CfgM32>(a, s, plan_out);
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
