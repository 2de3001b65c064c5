// VALU Conv1d for the shapes that are not GEMM-like: one input channel
// (pre_filter: Conv1d(1,192,k9,s4), hierspeechpp_speechsynthesizer.py:187,196), one
// output channel (conv_post: :283,307,419,449), and the L = 1 "Linear on the style
// vector" cases (cond / cond_layer / cond_block / adaLN_modulation:
// hierspeechpp_speechsynthesizer.py:72-73,285,423; modules.py:127,402-405).
// One thread per output sample, time fastest (coalesced x reads and y writes; the
// weight read w[j][ci][co] is wave-uniform when a wave covers one channel, and
// coalesced over co when L == 1).
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

}  // namespace

extern "C" int hsp_conv1d_direct_f32(const hsp_conv1d_args* ap, void* stream) {
  if (!ap) return HSP_EINVAL;
  const hsp_conv1d_args& a = *ap;
  if (!a.x || !a.w || !a.y) return HSP_EINVAL;
  if (a.B <= 0 || a.Cin <= 0 || a.Lin <= 0 || a.K <= 0 || a.M <= 0 || a.Cout <= 0 || a.Lout <= 0) return HSP_EINVAL;
  if (a.stride < 1 || a.dil < 1 || a.rows != HSP_ROWS_PLAIN || a.Cout > a.M || a.w_ld < a.M) return HSP_EINVAL;
  if (a.prologue != HSP_PRO_NONE && a.prologue != HSP_PRO_LRELU && a.prologue != HSP_PRO_SILU) return HSP_EINVAL;
  if (a.mask_mode != HSP_MASK_NONE && !a.mask) return HSP_EINVAL;
  if (a.ln_c1 || a.split_row) return HSP_EINVAL;  // fused LayerNorm / second output: token-GEMM path only
  const int64_t total = (int64_t)a.B * a.Cout * a.Lout;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 262144) blocks = 262144;
  hipLaunchKernelGGL(conv1d_direct_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}
