// VALU Conv1d for the shapes that are not GEMM-like: one input channel
// (pre_filter: Conv1d(1,192,k9,s4), hierspeechpp_speechsynthesizer.py:187,196), one
// output channel (conv_post: :283,307,419,449), and the L = 1 "Linear on the style
// vector" cases (cond / cond_layer / cond_block / adaLN_modulation:
// hierspeechpp_speechsynthesizer.py:72-73,285,423; modules.py:127,402-405).
// One thread per output sample, time fastest (coalesced x reads and y writes; the
// weight read w[j][ci][co] is wave-uniform when a wave covers one channel, and
// coalesced over co when L == 1).
#include <type_traits>
#include "hsp_device.h"

namespace {

__global__ __launch_bounds__(256) void conv1d_direct_kernel(const hsp_conv1d_args a) {
  const int64_t total = (int64_t)a.B * a.Cout * a.Lout;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(idx % a.Lout);
    const int64_t r = idx / a.Lout;
    const int co = (int)(r % a.Cout);
    const int b = (int)(r / a.Cout);
    const float* xb = a.x + (int64_t)b * a.x_bs;
    float acc = 0.0f;
    for (int j = 0; j < a.K; ++j) {
      const int p = t * a.stride + j * a.dil - a.pad;
      if (p < 0 || p >= a.Lin) continue;
      const float* wj = a.w + (int64_t)j * a.Cin * a.w_ld + co;
      const float* xp = xb + (int64_t)p * a.x_ts;
      if (a.prologue == HSP_PRO_NONE) {
        // independent loads, 8 in flight: a "Linear on the style vector" is a pure latency chain otherwise
        float acc4[4] = {0.f, 0.f, 0.f, 0.f};
        int ci = 0;
        for (; ci + 8 <= a.Cin; ci += 8) {
          float wv[8], xv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            wv[u] = wj[(int64_t)(ci + u) * a.w_ld];
            xv[u] = xp[(int64_t)(ci + u) * a.x_cs];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc4[u & 3] = fmaf(wv[u], xv[u], acc4[u & 3]);
        }
        for (; ci < a.Cin; ++ci) acc4[0] = fmaf(wj[(int64_t)ci * a.w_ld], xp[(int64_t)ci * a.x_cs], acc4[0]);
        acc += (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
      } else {
        for (int ci = 0; ci < a.Cin; ++ci) {
          float xv = xp[(int64_t)ci * a.x_cs];
          if (a.prologue == HSP_PRO_LRELU) xv = xv > 0.0f ? xv : xv * a.slope;
          else if (a.prologue == HSP_PRO_SILU) xv = xv * hsp_sigmoid(xv);
          acc = fmaf(wj[(int64_t)ci * a.w_ld], xv, acc);
        }
      }
    }
    float v = acc;
    if (a.bias) v += a.bias[co];
    if (a.cbias) v += a.cbias[(int64_t)b * a.cbias_bs + co];
    v = hsp_apply_act(v, a.act);
    hsp_epilogue_store(a, b, co, t, v);
  }
}

// One output channel, unit stride / dilation, K <= 9, 16-B addressable rows with L % 4 == 0: conv_post of both
// Generators (C -> 1, k = 7, + tanh: hierspeechpp_speechsynthesizer.py:449-450, speechsr48k/speechsr.py:106-107) and of
// the SourceNetwork.  The one-thread-per-output kernel above ran it at 3.6 TFLOP/s (0.25 ms per 32 x 4 s step: a
// dependent chain of C x K scalar loads per output).  Here a thread owns FOUR consecutive outputs: per channel it
// loads the three aligned float4 that cover t - 4 .. t + 7 (neighbouring lanes overlap: those re-reads hit the L1),
// all channels' loads independent, and the K x 4 FMAs run from registers; the C x K weights sit in LDS (broadcast
// reads).  HBM-bound on the input: ~5 TB/s.
constexpr int C1_MAXW = 4096;   // C * K floats of weights in LDS

template <int N, class F>
__device__ __forceinline__ void c1_static_for(F&& f) {
  if constexpr (N > 0) {
    c1_static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <int K>   // odd, <= 9: the window t - 4 .. t + 7 covers every tap of the four outputs
__global__ __launch_bounds__(256) void conv1d_cout1_kernel(const hsp_conv1d_args a, int n_tiles) {
  constexpr int H = (K - 1) / 2, CU = 4;   // CU channels' loads in flight at a time
  __shared__ float ws[C1_MAXW];
  for (int e = threadIdx.x; e < a.Cin * K; e += 256) {
    const int c = e / K, j = e - c * K;
    ws[e] = a.w[((int64_t)j * a.Cin + c) * a.w_ld];
  }
  __syncthreads();
  const int tile = blockIdx.x % n_tiles, b = blockIdx.x / n_tiles;
  const int t = tile * 1024 + 4 * threadIdx.x;          // first of this thread's four outputs
  if (t >= a.Lout) return;
  const int L = a.Lin;
  const float* xb = a.x + (int64_t)b * a.x_bs + t;
  const int64_t xcs = a.x_cs;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  const bool lo = t - 4 >= 0, hi = t + 4 < L;           // t .. t + 3 is inside: L % 4 == 0
  // every index below is a compile-time constant (c1_static_for): a runtime index would put the staged
  // float4s in scratch memory.  Loads are unconditional, from clamped (always legal) addresses, and the zero
  // padding is selected afterwards: a load in a divergent block gets an s_waitcnt vmcnt(0) behind it and the
  // round trips of a group serialise.
  const int offl = lo ? -4 : 0, offh = hi ? 4 : 0;
  const int ngroups = a.Cin / CU;
  for (int g = 0; g < ngroups; ++g) {
    const float* xc = xb + (int64_t)(g * CU) * xcs;
    float4 va[CU], vb[CU], vc[CU];
    c1_static_for<CU>([&](auto uu) __attribute__((always_inline)) {
      constexpr int u = decltype(uu)::value;
      va[u] = *reinterpret_cast<const float4*>(xc + u * xcs + offl);
      vb[u] = *reinterpret_cast<const float4*>(xc + u * xcs);
      vc[u] = *reinterpret_cast<const float4*>(xc + u * xcs + offh);
    });
    c1_static_for<CU>([&](auto uu) __attribute__((always_inline)) {
      constexpr int u = decltype(uu)::value;
      const float w_[12] = {lo ? va[u].x : 0.f, lo ? va[u].y : 0.f, lo ? va[u].z : 0.f, lo ? va[u].w : 0.f,
                            vb[u].x, vb[u].y, vb[u].z, vb[u].w,
                            hi ? vc[u].x : 0.f, hi ? vc[u].y : 0.f, hi ? vc[u].z : 0.f, hi ? vc[u].w : 0.f};
      const float* wc = ws + (g * CU + u) * K;
      c1_static_for<K>([&](auto jj) __attribute__((always_inline)) {
        constexpr int j = decltype(jj)::value;
        const float wj = wc[j];
        acc0 = fmaf(wj, w_[4 + j - H], acc0);       // x[t + q + j - H], q = 0 .. 3
        acc1 = fmaf(wj, w_[5 + j - H], acc1);
        acc2 = fmaf(wj, w_[6 + j - H], acc2);
        acc3 = fmaf(wj, w_[7 + j - H], acc3);
      });
    });
  }
  for (int c = ngroups * CU; c < a.Cin; ++c) {          // channel tail (Cin % CU)
    const float* xc = xb + (int64_t)c * xcs;
    const float4 va = *reinterpret_cast<const float4*>(xc + offl);
    const float4 vb = *reinterpret_cast<const float4*>(xc);
    const float4 vc = *reinterpret_cast<const float4*>(xc + offh);
    const float w_[12] = {lo ? va.x : 0.f, lo ? va.y : 0.f, lo ? va.z : 0.f, lo ? va.w : 0.f, vb.x, vb.y, vb.z, vb.w,
                          hi ? vc.x : 0.f, hi ? vc.y : 0.f, hi ? vc.z : 0.f, hi ? vc.w : 0.f};
    const float* wc = ws + c * K;
    c1_static_for<K>([&](auto jj) __attribute__((always_inline)) {
      constexpr int j = decltype(jj)::value;
      const float wj = wc[j];
      acc0 = fmaf(wj, w_[4 + j - H], acc0);
      acc1 = fmaf(wj, w_[5 + j - H], acc1);
      acc2 = fmaf(wj, w_[6 + j - H], acc2);
      acc3 = fmaf(wj, w_[7 + j - H], acc3);
    });
  }
  const float bz = a.bias ? a.bias[0] : 0.0f;
  const float sc = a.scale * a.post_scale;
  float4 o;
  o.x = hsp_apply_act(acc0 + bz, a.act) * sc;
  o.y = hsp_apply_act(acc1 + bz, a.act) * sc;
  o.z = hsp_apply_act(acc2 + bz, a.act) * sc;
  o.w = hsp_apply_act(acc3 + bz, a.act) * sc;
  *reinterpret_cast<float4*>(a.y + (int64_t)b * a.y_bs + t) = o;
}

// "Linear on the style vector": K = 1, Lin = Lout = 1 (cond / cond_layer / cond_block / adaLN_modulation on [B, C, 1]).
// The one-thread-per-output kernel above walks all Cin channels serially (8 loads in flight): 20-46 us for a 192 x 768
// or 3072 x 256 matrix, on the critical path of every front group.  Here a workgroup of 8 waves owns 64 outputs of up to
// 8 utterances: lanes along the output channel (the packed weight row is coalesced), the waves split Cin -- every
// wave's weight loads are independent, so its whole share is in flight at once -- the input vectors sit in LDS
// (broadcast reads), and the eight partial sums meet in LDS.
constexpr int LV_WAVES = 8, LV_BB = 8, LV_MAXC = 1024;   // Cin <= 1024: 32 KB of staged inputs + 16 KB of partial sums

__global__ __launch_bounds__(64 * LV_WAVES) void linear_vec_kernel(const hsp_conv1d_args a) {
  __shared__ float xs[LV_BB][LV_MAXC];
  __shared__ float part[LV_WAVES][LV_BB][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int co = blockIdx.x * 64 + lane;
  const int b0 = blockIdx.y * LV_BB, nb = min(LV_BB, a.B - b0);
  for (int e = tid; e < nb * a.Cin; e += 64 * LV_WAVES) {
    const int bb = e / a.Cin, ci = e - bb * a.Cin;
    float v = a.x[(int64_t)(b0 + bb) * a.x_bs + (int64_t)ci * a.x_cs];
    if (a.prologue == HSP_PRO_LRELU) v = v > 0.0f ? v : v * a.slope;
    else if (a.prologue == HSP_PRO_SILU) v = v * hsp_sigmoid(v);
    xs[bb][ci] = v;
  }
  __syncthreads();
  const int per = (a.Cin + LV_WAVES - 1) / LV_WAVES;
  const int c0 = wave * per, c1 = min(c0 + per, a.Cin);
  const float* wc = a.w + min(co, a.M - 1);
  float acc[LV_BB];
#pragma unroll
  for (int bb = 0; bb < LV_BB; ++bb) acc[bb] = 0.0f;
  int ci = c0;
  for (; ci + 8 <= c1; ci += 8) {
    float wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) wv[u] = wc[(int64_t)(ci + u) * a.w_ld];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int bb = 0; bb < LV_BB; ++bb) acc[bb] = fmaf(wv[u], xs[bb][ci + u], acc[bb]);
  }
  for (; ci < c1; ++ci) {
    const float wv = wc[(int64_t)ci * a.w_ld];
#pragma unroll
    for (int bb = 0; bb < LV_BB; ++bb) acc[bb] = fmaf(wv, xs[bb][ci], acc[bb]);
  }
#pragma unroll
  for (int bb = 0; bb < LV_BB; ++bb) part[wave][bb][lane] = acc[bb];
  __syncthreads();
  if (co >= a.Cout) return;
  for (int bb = wave; bb < nb; bb += LV_WAVES) {            // wave w finalises utterance b0 + w
    float v = 0.0f;
#pragma unroll
    for (int w = 0; w < LV_WAVES; ++w) v += part[w][bb][lane];   // fixed order: deterministic
    const int b = b0 + bb;
    if (a.bias) v += a.bias[co];
    if (a.cbias) v += a.cbias[(int64_t)b * a.cbias_bs + co];
    v = hsp_apply_act(v, a.act);
    hsp_epilogue_store(a, b, co, 0, v);
  }
}

bool linear_vec_fast(const hsp_conv1d_args& a) {
  return a.K == 1 && a.Lin == 1 && a.Lout == 1 && a.stride == 1 && a.pad == 0 && a.Cin >= 64 && a.Cin <= LV_MAXC && a.Cout >= 64;
}

bool cout1_fast(const hsp_conv1d_args& a) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return a.Cout == 1 && a.stride == 1 && a.dil == 1 && a.K <= 9 && (a.K & 1) && a.pad == (a.K - 1) / 2 && a.x_ts == 1 &&
         a.Lin == a.Lout && (a.Lin & 3) == 0 && (a.x_cs & 3) == 0 && (a.x_bs & 3) == 0 && (a.y_bs & 3) == 0 && al16(a.x) &&
         al16(a.y) && a.prologue == HSP_PRO_NONE && !a.cbias && !a.cscale && !a.res && !a.accumulate &&
         a.mask_mode == HSP_MASK_NONE && (int64_t)a.Cin * a.K <= C1_MAXW;
}

}  // namespace

extern "C" int hsp_conv1d_direct_f32(const hsp_conv1d_args* ap, void* stream) {
  if (!ap) return HSP_EINVAL;
  const hsp_conv1d_args& a = *ap;
  if (!a.x || !a.w || !a.y) return HSP_EINVAL;
  if (a.B <= 0 || a.Cin <= 0 || a.Lin <= 0 || a.K <= 0 || a.M <= 0 || a.Cout <= 0 || a.Lout <= 0) return HSP_EINVAL;
  if (a.stride < 1 || a.dil < 1 || a.rows != HSP_ROWS_PLAIN || a.Cout > a.M || a.w_ld < a.M) return HSP_EINVAL;
  if (a.prologue != HSP_PRO_NONE && a.prologue != HSP_PRO_LRELU && a.prologue != HSP_PRO_SILU) return HSP_EINVAL;
  if (a.mask_mode != HSP_MASK_NONE && !a.mask) return HSP_EINVAL;
  if (a.res && a.res_ts > 1) return HSP_EINVAL;   // strided residual: register-path token GEMM only
  if (a.w_bs) return HSP_EINVAL;                  // per-batch weights (frequency bins): the implicit-GEMM kernel only
  if (a.ln_c1 || a.split_row) return HSP_EINVAL;  // fused LayerNorm / second output: token-GEMM path only
  if (linear_vec_fast(a)) {
    const unsigned gx = (unsigned)((a.Cout + 63) / 64), gy = (unsigned)((a.B + LV_BB - 1) / LV_BB);
    if (gy > 65535) return HSP_EINVAL;
    hipLaunchKernelGGL(linear_vec_kernel, dim3(gx, gy), dim3(64 * LV_WAVES), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
  }
  if (cout1_fast(a)) {
    const int n_tiles = (a.Lout + 1023) / 1024;
    const int64_t nb = (int64_t)n_tiles * a.B;
    if (nb > 0x7fffffff) return HSP_EINVAL;
    const hipStream_t st = static_cast<hipStream_t>(stream);
    switch (a.K) {
      case 1: hipLaunchKernelGGL(conv1d_cout1_kernel<1>, dim3((unsigned)nb), dim3(256), 0, st, a, n_tiles); break;
      case 3: hipLaunchKernelGGL(conv1d_cout1_kernel<3>, dim3((unsigned)nb), dim3(256), 0, st, a, n_tiles); break;
      case 5: hipLaunchKernelGGL(conv1d_cout1_kernel<5>, dim3((unsigned)nb), dim3(256), 0, st, a, n_tiles); break;
      case 7: hipLaunchKernelGGL(conv1d_cout1_kernel<7>, dim3((unsigned)nb), dim3(256), 0, st, a, n_tiles); break;
      default: hipLaunchKernelGGL(conv1d_cout1_kernel<9>, dim3((unsigned)nb), dim3(256), 0, st, a, n_tiles); break;
    }
    return (int)hipGetLastError();
  }
  const int64_t total = (int64_t)a.B * a.Cout * a.Lout;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 262144) blocks = 262144;
  hipLaunchKernelGGL(conv1d_direct_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}
