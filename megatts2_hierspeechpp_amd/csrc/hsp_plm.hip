// Kernels specific to the Mega-TTS2 prosody language model loop (SURVEY.md row A18,
// ttv_v1/t2w2v_transformer.py:702-718): the per-step input assembly and the greedy argmax
// that feeds the next step.  Both read/write the code buffer in device memory so that the
// whole T-step loop is a chain of launches without a host round trip (hipGraph-capturable).
#include "hsp_device.h"

namespace {

// x[b, c, j] = (c < Dtc ? tc[b, c, j] : emb[codes[b, j], c - Dtc]) + alpha * pe_t[c, j],  j < n
// (torch.cat([tc_latent[:, :t+1], pc_embedding(p_code)], -1) then SinePositionalEmbedding.forward,
//  t2w2v_transformer.py:711-713,510-514; x_scale = 1).  pe_t is the sinusoid table transposed
// to [D][P] so that lanes (running along j) read it coalesced.
__global__ __launch_bounds__(256) void plm_embed_kernel(const float* __restrict__ tc, int64_t tc_bs, int64_t tc_cs,
                                                        int Dtc, const int64_t* __restrict__ codes, int64_t codes_bs,
                                                        const float* __restrict__ emb, int Demb, int n_emb,
                                                        const float* __restrict__ pe_t, int P,
                                                        const float* __restrict__ alpha, float* __restrict__ x,
                                                        int64_t x_bs, int64_t x_cs, int B, int n) {
  const int D = Dtc + Demb;
  const int64_t total = (int64_t)B * D * n;
  const float al = alpha[0];
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int j = (int)(e % n);
    const int64_t bc = e / n;
    const int c = (int)(bc % D), b = (int)(bc / D);
    float v;
    if (c < Dtc) {
      v = tc[b * tc_bs + c * tc_cs + j];
    } else {
      int64_t id = codes[b * codes_bs + j];
      id = id < 0 ? 0 : (id >= n_emb ? n_emb - 1 : id);  // a corrupted code must not fault the GPU
      v = emb[id * Demb + (c - Dtc)];
    }
    x[b * x_bs + c * x_cs + j] = fmaf(al, pe_t[(int64_t)c * P + j], v);
  }
  // side-by-side layout (x_bs == n) with the row padded to x_cs > B*n columns: zero the padding
  if (x_bs == n && x_cs > (int64_t)B * n) {
    const int padc = (int)(x_cs - (int64_t)B * n);
    for (int e = blockIdx.x * 256 + threadIdx.x; e < D * padc; e += gridDim.x * 256)
      x[(int64_t)(e / padc) * x_cs + (int64_t)B * n + e % padc] = 0.0f;
  }
}

// The same with the greedy choice of the PREVIOUS step folded in (one launch per step less): workgroup (b, 0) first
// takes codes[b, n - 1] = argmax_c logits[b * l_bs + c * l_cs] (ties -> lowest index, as argmax_kernel), stores it and
// then writes the code-embedding rows of utterance b; the workgroups (b, 1 ...) write the tc_latent rows, which do
// not depend on any code.  Grid (B, 1 + ceil(Dtc * n / 1024)).
__global__ __launch_bounds__(256) void plm_embed_step_kernel(const float* __restrict__ tc, int64_t tc_bs, int64_t tc_cs,
                                                             int Dtc, int64_t* __restrict__ codes, int64_t codes_bs,
                                                             const float* __restrict__ emb, int Demb, int n_emb,
                                                             const float* __restrict__ pe_t, int P,
                                                             const float* __restrict__ alpha, float* __restrict__ x,
                                                             int64_t x_bs, int64_t x_cs, int B, int n,
                                                             const float* __restrict__ logits, int64_t l_bs, int64_t l_cs,
                                                             int n_logits) {
  __shared__ float smax[4];
  __shared__ int sidx[4];
  __shared__ int s_code;
  const int b = blockIdx.x, part = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float al = alpha[0];
  const int D = Dtc + Demb;
  if (part == 0) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    const float* row = logits + (int64_t)b * l_bs;
    for (int c = tid; c < n_logits; c += 256) {
      const float v = row[(int64_t)c * l_cs];
      if (v > best || (v == best && c < bi)) best = v, bi = c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) best = ov, bi = oi;
    }
    if (lane == 0) smax[wave] = best, sidx[wave] = bi;
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w)
        if (smax[w] > best || (smax[w] == best && sidx[w] < bi)) best = smax[w], bi = sidx[w];
      bi = bi == 0x7fffffff ? 0 : bi;
      codes[(int64_t)b * codes_bs + n - 1] = bi;
      s_code = bi;
    }
    __syncthreads();
    const int newest = s_code;
    for (int e = tid; e < Demb * n; e += 256) {
      const int j = e % n, c = Dtc + e / n;
      int64_t id = j == n - 1 ? (int64_t)newest : codes[(int64_t)b * codes_bs + j];
      id = id < 0 ? 0 : (id >= n_emb ? n_emb - 1 : id);   // a corrupted code must not fault the GPU
      x[b * x_bs + (int64_t)c * x_cs + j] = fmaf(al, pe_t[(int64_t)c * P + j], emb[id * Demb + (c - Dtc)]);
    }
    // side-by-side layout (x_bs == n) with the row padded to x_cs > B*n columns: zero the padding
    if (b == 0 && x_bs == n && x_cs > (int64_t)B * n) {
      const int padc = (int)(x_cs - (int64_t)B * n);
      for (int e = tid; e < D * padc; e += 256) x[(int64_t)(e / padc) * x_cs + (int64_t)B * n + e % padc] = 0.0f;
    }
  } else {
    const int total = Dtc * n;
    for (int e = (part - 1) * 1024 + tid; e < min(total, part * 1024); e += 256) {
      const int j = e % n, c = e / n;
      x[b * x_bs + (int64_t)c * x_cs + j] = fmaf(al, pe_t[(int64_t)c * P + j], tc[b * tc_bs + c * tc_cs + j]);
    }
  }
}

// out[b * out_bs] = argmax_c logits[b * l_bs + c * l_cs]; ties -> lowest index (torch.argmax on CPU returns the
// first maximal element).  One workgroup per row.
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, int64_t l_bs, int64_t l_cs, int N,
                                                     int64_t* __restrict__ out, int64_t out_bs) {
  __shared__ float smax[4];
  __shared__ int sidx[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (int64_t)b * l_bs;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = tid; c < N; c += 256) {
    const float v = row[(int64_t)c * l_cs];
    if (v > best || (v == best && c < bi)) best = v, bi = c;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) best = ov, bi = oi;
  }
  if (lane == 0) smax[wave] = best, sidx[wave] = bi;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (smax[w] > best || (smax[w] == best && sidx[w] < bi)) best = smax[w], bi = sidx[w];
    out[(int64_t)b * out_bs] = bi == 0x7fffffff ? 0 : bi;
  }
}

// y[b, c, t] (contiguous) = x[b * s_bs + c * s_cs + t * s_ts]
__global__ __launch_bounds__(256) void copy_strided_kernel(const float* __restrict__ x, int64_t s_bs, int64_t s_cs,
                                                           int64_t s_ts, float* __restrict__ y, int B, int C, int T) {
  const int64_t total = (int64_t)B * C * T;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int t = (int)(e % T);
    const int64_t bc = e / T;
    y[e] = x[(bc / C) * s_bs + (bc % C) * s_cs + t * s_ts];
  }
}

}  // namespace

#define HSP_STREAM static_cast<hipStream_t>(stream)

extern "C" int hsp_plm_embed_f32(const float* tc, int64_t tc_bs, int64_t tc_cs, int32_t Dtc, const int64_t* codes,
                                 int64_t codes_bs, const float* emb, int32_t Demb, int32_t n_emb, const float* pe_t,
                                 int32_t P, const float* alpha, float* x, int64_t x_bs, int64_t x_cs, int32_t B,
                                 int32_t n, void* stream) {
  if (!tc || !codes || !emb || !pe_t || !alpha || !x) return HSP_EINVAL;
  if (B <= 0 || n <= 0 || n > P || Dtc <= 0 || Demb <= 0 || n_emb <= 0 || x_bs < 0 || x_cs < n) return HSP_EINVAL;
  const int64_t total = (int64_t)B * (Dtc + Demb) * n;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(plm_embed_kernel, dim3((unsigned)blocks), dim3(256), 0, HSP_STREAM, tc, tc_bs, tc_cs, Dtc, codes,
                     codes_bs, emb, Demb, n_emb, pe_t, P, alpha, x, x_bs, x_cs, B, n);
  return (int)hipGetLastError();
}

extern "C" int hsp_plm_embed_step_f32(const float* tc, int64_t tc_bs, int64_t tc_cs, int32_t Dtc, int64_t* codes,
                                      int64_t codes_bs, const float* emb, int32_t Demb, int32_t n_emb, const float* pe_t,
                                      int32_t P, const float* alpha, float* x, int64_t x_bs, int64_t x_cs, int32_t B,
                                      int32_t n, const float* logits, int64_t l_bs, int64_t l_cs, int32_t n_logits,
                                      void* stream) {
  if (!tc || !codes || !emb || !pe_t || !alpha || !x || !logits) return HSP_EINVAL;
  if (B <= 0 || n < 1 || n > P || Dtc <= 0 || Demb <= 0 || n_emb <= 0 || x_bs < 0 || x_cs < n) return HSP_EINVAL;   // n == 1: hsp.h
  if (n_logits <= 0 || l_cs <= 0 || l_bs < 0) return HSP_EINVAL;
  const int parts = 1 + (int)(((int64_t)Dtc * n + 1023) / 1024);
  if (B > 65535 || parts > 65535) return HSP_EINVAL;
  hipLaunchKernelGGL(plm_embed_step_kernel, dim3((unsigned)B, (unsigned)parts), dim3(256), 0, HSP_STREAM, tc, tc_bs, tc_cs,
                     Dtc, codes, codes_bs, emb, Demb, n_emb, pe_t, P, alpha, x, x_bs, x_cs, B, n, logits, l_bs, l_cs,
                     n_logits);
  return (int)hipGetLastError();
}

extern "C" int hsp_argmax_f32(const float* logits, int64_t l_bs, int64_t l_cs, int32_t B, int32_t N, int64_t* out,
                              int64_t out_bs, void* stream) {
  if (!logits || !out || B <= 0 || N <= 0 || l_cs <= 0) return HSP_EINVAL;
  hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)B), dim3(256), 0, HSP_STREAM, logits, l_bs, l_cs, N, out, out_bs);
  return (int)hipGetLastError();
}

extern "C" int hsp_copy_strided_f32(const float* x, int64_t s_bs, int64_t s_cs, int64_t s_ts, float* y, int32_t B,
                                    int32_t C, int32_t T, void* stream) {
  if (!x || !y || B <= 0 || C <= 0 || T <= 0 || s_bs < 0 || s_cs < 0 || s_ts < 0) return HSP_EINVAL;
  int64_t blocks = ((int64_t)B * C * T + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(copy_strided_kernel, dim3((unsigned)blocks), dim3(256), 0, HSP_STREAM, x, s_bs, s_cs, s_ts, y, B, C,
                     T);
  return (int)hipGetLastError();
}
