// Multi-head attention on channel-major (B, H*D, T) tensors, fp32.
//
// Covers timm 0.6.13 Attention as used by DiTConVBlock (modules.py:397,409: no key
// mask, scale D^-0.5), and attentions.MultiHeadAttention (attentions.py:157-188:
// masked_fill(mask == 0, -1e4), optional relative-position window on keys and values).
//
// One workgroup = 16 queries of one (batch, head).  Scores for all keys are kept in
// LDS (16 x Tk), soft-maxed by rows, then multiplied with V, which is streamed through
// LDS in 64-key slabs transposed so that lanes run along the head dimension.
// Round-1 implementation on the vector pipe: T <= a few hundred on this path (50 Hz
// frames), the whole attention work is ~10 % of the vocoder FLOPs.
#include <atomic>
#include "hsp_device.h"

namespace {

constexpr int QT = 16;       // queries per workgroup
constexpr int ATT_THREADS = 256;

__global__ __launch_bounds__(ATT_THREADS) void mha_kernel(const hsp_mha_args a, int n_qt, int dpad, int spad) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Qs = lds;                       // [D][QT]
  float* S = Qs + a.D * QT;              // [QT][spad]
  float* Vs = S + QT * spad;             // [64][dpad]
  int bid = blockIdx.x;
  const int qt = bid % n_qt;
  bid /= n_qt;
  const int h = bid % a.H;
  const int b = bid / a.H;
  const int i0 = qt * QT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = a.D, Tq = a.Tq, Tk = a.Tk;
  const float* qh = a.q + (int64_t)b * a.q_bs + (int64_t)h * D * Tq;
  const float* kh = a.k + (int64_t)b * a.k_bs + (int64_t)h * D * Tk;
  const float* vh = a.v + (int64_t)b * a.v_bs + (int64_t)h * D * Tk;
  float* oh = a.o + (int64_t)b * a.o_bs + (int64_t)h * D * Tq;

  // ---- Q tile, pre-scaled (attentions.py:164 divides the query; timm scales the product)
  for (int e = tid; e < D * QT; e += ATT_THREADS) {
    const int i = e % QT, d = e / QT;
    Qs[e] = (i0 + i < Tq) ? qh[(int64_t)d * Tq + i0 + i] * a.qk_scale : 0.0f;
  }
  __syncthreads();

  // ---- scores: wave w owns queries 4w..4w+3, lanes run along keys
  {
    const int iq = wave * 4;
    for (int j = lane; j < Tk; j += 64) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (int d = 0; d < D; ++d) {
        const float kv = kh[(int64_t)d * Tk + j];
        const float4 qv = *reinterpret_cast<const float4*>(Qs + d * QT + iq);
        s0 = fmaf(qv.x, kv, s0);
        s1 = fmaf(qv.y, kv, s1);
        s2 = fmaf(qv.z, kv, s2);
        s3 = fmaf(qv.w, kv, s3);
      }
      float sv[4] = {s0, s1, s2, s3};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + iq + u;
        float s = sv[u];
        if (a.rel_k) {
          const int r = j - i;
          if (r >= -a.window && r <= a.window && i < Tq) {
            const float* ek = a.rel_k + (int64_t)(r + a.window) * D;
            float t = 0.0f;
            for (int d = 0; d < D; ++d) t = fmaf(Qs[d * QT + iq + u], ek[d], t);
            s += t;
          }
        }
        if (a.mask_q && i < Tq) {
          if (a.mask_q[(int64_t)b * Tq + i] * a.mask_k[(int64_t)b * Tk + j] == 0.0f) s = -1e4f;
        }
        S[(iq + u) * spad + j] = s;
      }
    }
  }
  __syncthreads();

  // ---- row softmax (4 rows per wave)
  for (int u = 0; u < 4; ++u) {
    float* row = S + (wave * 4 + u) * spad;
    float mx = -3.0e38f;
    for (int j = lane; j < Tk; j += 64) mx = fmaxf(mx, row[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.0f;
    for (int j = lane; j < Tk; j += 64) {
      const float e = expf(row[j] - mx);
      row[j] = e;
      sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.0f / sum;
    for (int j = lane; j < Tk; j += 64) row[j] *= inv;
  }

  // ---- O = P V : thread = (head-dim d, query group of 8)
  const int d = tid & 127, ig = tid >> 7;
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.0f;
  for (int d0 = 0; d0 < D; d0 += 128) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.0f;
    for (int j0 = 0; j0 < Tk; j0 += 64) {
      __syncthreads();  // softmax rows complete (first pass) / previous slab consumed
      for (int dd = wave; dd < 128 && d0 + dd < D; dd += 4)
        Vs[lane * dpad + dd] = (j0 + lane < Tk) ? vh[(int64_t)(d0 + dd) * Tk + j0 + lane] : 0.0f;
      __syncthreads();
      if (d0 + d < D) {
        const int jn = min(64, Tk - j0);
        for (int j = 0; j < jn; ++j) {
          const float vv = Vs[j * dpad + d];
#pragma unroll
          for (int u = 0; u < 8; ++u) acc[u] = fmaf(S[(ig * 8 + u) * spad + j0 + j], vv, acc[u]);
        }
      }
    }
    if (d0 + d < D) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + ig * 8 + u;
        if (i >= Tq) continue;
        float v = acc[u];
        if (a.rel_v) {
          for (int r = -a.window; r <= a.window; ++r) {
            const int j = i + r;
            if (j >= 0 && j < Tk) v = fmaf(S[(ig * 8 + u) * spad + j], a.rel_v[(int64_t)(r + a.window) * D + d0 + d], v);
          }
        }
        oh[(int64_t)(d0 + d) * Tq + i] = v;
      }
    }
  }
}

}  // namespace

extern "C" int hsp_mha_f32(const hsp_mha_args* ap, void* stream) {
  if (!ap) return HSP_EINVAL;
  const hsp_mha_args& a = *ap;
  if (!a.q || !a.k || !a.v || !a.o || a.B <= 0 || a.H <= 0 || a.D <= 0 || a.Tq <= 0 || a.Tk <= 0) return HSP_EINVAL;
  if ((a.mask_q == nullptr) != (a.mask_k == nullptr)) return HSP_EINVAL;
  if ((a.rel_k || a.rel_v) && (a.window <= 0 || a.Tq != a.Tk)) return HSP_EINVAL;
  const int n_qt = (a.Tq + QT - 1) / QT;
  const int dpad = (a.D < 128 ? a.D : 128) | 1;
  const int spad = a.Tk + 1;
  const int64_t lds_bytes = ((int64_t)a.D * QT + (int64_t)QT * spad + 64 * dpad) * (int64_t)sizeof(float);
  if (lds_bytes > 160 * 1024) return HSP_EINVAL;
  static std::atomic<int> lds_cap{32 * 1024};
  if (lds_bytes > lds_cap.load(std::memory_order_relaxed)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mha_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    lds_cap.store(160 * 1024, std::memory_order_relaxed);
  }
  const int64_t blocks = (int64_t)n_qt * a.H * a.B;
  hipLaunchKernelGGL(mha_kernel, dim3((unsigned)blocks), dim3(ATT_THREADS), (size_t)lds_bytes,
                     static_cast<hipStream_t>(stream), a, n_qt, dpad, spad);
  return (int)hipGetLastError();
}
