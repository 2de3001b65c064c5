// Kernels of the text -> wav2vec front-end (ttv_v1/t2w2v_transformer.py:937-994):
// embedding sums, the bidirectional LSTM recurrence of the duration / range predictors, the
// duration rounding and the Gaussian upsampling.  Everything is per utterance: a batch is B
// independent sequences with their own lengths (the reference runs B = 1).
#include "hsp_device.h"

namespace {

// out[b, c, t] = ((tab0[id0] + tab1[id1]) + tab2[id2])[c], each term scaled first
// (TextEncoder.forward, t2w2v_transformer.py:127-131; one table + scale 1 = codebook lookup,
//  core_vq.py:188-190).
__global__ __launch_bounds__(256) void embedding_sum_kernel(const int64_t* __restrict__ id0, const int64_t* __restrict__ id1,
                                                            const int64_t* __restrict__ id2, const float* __restrict__ t0,
                                                            const float* __restrict__ t1, const float* __restrict__ t2,
                                                            int n0, int n1, int n2, float scale, float* __restrict__ out,
                                                            int64_t o_bs, int64_t o_cs, int B, int C, int T) {
  const int64_t total = (int64_t)B * C * T;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int t = (int)(e % T);
    const int64_t bc = e / T;
    const int c = (int)(bc % C), b = (int)(bc / C);
    const int64_t p = (int64_t)b * T + t;
    float v = t0[(int64_t)hsp_clampi((int)id0[p], 0, n0 - 1) * C + c] * scale;
    if (t1) v += t1[(int64_t)hsp_clampi((int)id1[p], 0, n1 - 1) * C + c] * scale;
    if (t2) v += t2[(int64_t)hsp_clampi((int)id2[p], 0, n2 - 1) * C + c] * scale;
    out[b * o_bs + c * o_cs + t] = v;
  }
}

// Bidirectional LSTM recurrence for one layer.  grid = (B, 2): workgroup (b, dir) walks its
// utterance's len[b] steps (dir 1 from the last valid step backwards, which is what a packed
// sequence / a B = 1 run does).  xp = x W_ih^T + b_ih for both directions, [B][2][4H][N]
// channel-major (one 1x1 GEMM); thread j owns gate row j (torch order i, f, g, o), reads
// W_hh^T[k][j] coalesced across j and the hidden state from LDS.  Steps >= len[b] are zero.
__global__ __launch_bounds__(1024) void lstm_bidir_kernel(const float* __restrict__ xp, int64_t xp_bs,
                                                          const float* __restrict__ whh_t, const float* __restrict__ bhh,
                                                          const int64_t* __restrict__ len, float* __restrict__ out,
                                                          int64_t o_bs, int64_t o_cs, int H, int N) {
  __shared__ float hs[256];
  __shared__ float gs[1024];
  const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
  const int G = 4 * H;
  const int L = (int)min((int64_t)N, max((int64_t)0, len[b]));
  const float* xb = xp + (int64_t)b * xp_bs + (int64_t)dir * G * N;
  const float* W = whh_t + (int64_t)dir * H * G;
  const float bj = j < G ? bhh[dir * G + j] : 0.0f;
  float* ob = out + (int64_t)b * o_bs + (int64_t)dir * H * o_cs;
  float c = 0.0f;
  if (j < H) hs[j] = 0.0f;
  for (int t = L + j; t < N; t += 1024)          // padded steps: zero output (pad_packed_sequence)
    for (int k = 0; k < H; ++k) ob[(int64_t)k * o_cs + t] = 0.0f;
  __syncthreads();
  for (int s = 0; s < L; ++s) {
    const int t = dir ? L - 1 - s : s;
    if (j < G) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      const float xv = xb[(int64_t)j * N + t];
#pragma unroll 4
      for (int k = 0; k < H; k += 4) {
        a0 = fmaf(W[(int64_t)(k + 0) * G + j], hs[k + 0], a0);
        a1 = fmaf(W[(int64_t)(k + 1) * G + j], hs[k + 1], a1);
        a2 = fmaf(W[(int64_t)(k + 2) * G + j], hs[k + 2], a2);
        a3 = fmaf(W[(int64_t)(k + 3) * G + j], hs[k + 3], a3);
      }
      gs[j] = xv + (((a0 + a1) + (a2 + a3)) + bj);
    }
    __syncthreads();
    if (j < H) {
      const float ig = hsp_sigmoid(gs[j]), fg = hsp_sigmoid(gs[H + j]), gg = tanhf(gs[2 * H + j]),
                  og = hsp_sigmoid(gs[3 * H + j]);
      c = fg * c + ig * gg;
      const float h = og * tanhf(c);
      hs[j] = h;
      ob[(int64_t)j * o_cs + t] = h;
    }
    __syncthreads();
  }
}

// dur[b, n] = n < len[b] ? ceil(exp(logw[b, n]) * length_scale) : 0 ; frames[b] = sum_n dur[b, n]
// (t2w2v_transformer.py:955-957,972-975).  One workgroup per utterance.
__global__ __launch_bounds__(256) void duration_kernel(const float* __restrict__ logw, int64_t lw_bs,
                                                       const int64_t* __restrict__ len, float length_scale,
                                                       float* __restrict__ dur, int64_t d_bs, float* __restrict__ frames,
                                                       int N, int from_logw) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int L = (int)min((int64_t)N, max((int64_t)0, len[b]));
  float s = 0.0f;
  for (int n = tid; n < N; n += 256) {
    float d;
    if (from_logw) d = n < L ? ceilf(expf(logw[b * lw_bs + n]) * length_scale) : 0.0f;
    else d = n < L ? dur[b * d_bs + n] : 0.0f;   // caller-supplied durations: only clear the padding
    dur[b * d_bs + n] = d;
    s += d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) frames[b] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Gaussian upsampling (ttv_v1/Gaussian.py:35-69) with the range clamp of
// t2w2v_transformer.py:961-963 folded in.  One workgroup = 32 output frames of one utterance:
//   c_n = cumsum(dur)_n - dur_n / 2,  v_n = max(min(rng_n, 2 dur_n), 1e-5)
//   w[n][t] = -0.5 (log 2pi + log v_n + (t - c_n)^2 / v_n),  -1e15 for n >= len;  softmax over n
//   out[:, t] = sum_n p[n][t] x[:, n];   frames t >= frames[b] are written as zeros.
constexpr int GT = 32;
__global__ __launch_bounds__(256) void gaussian_upsample_kernel(const float* __restrict__ x, int64_t x_bs, int64_t x_cs,
                                                                const float* __restrict__ dur, int64_t d_bs,
                                                                const float* __restrict__ rng, int64_t r_bs,
                                                                const int64_t* __restrict__ len,
                                                                const float* __restrict__ frames, float* __restrict__ out,
                                                                int C, int N, int T, int n_tt) {
  extern __shared__ float lds[];
  float* P = lds;               // [N][GT + 1]
  float* cen = P + N * (GT + 1);  // [N]
  float* var = cen + N;           // [N]
  __shared__ float red[8][GT + 1];
  const int tt = blockIdx.x % n_tt, b = blockIdx.x / n_tt;
  const int tid = threadIdx.x, tl = tid & 31, grp = tid >> 5;
  const int L = (int)min((int64_t)N, max((int64_t)0, len[b]));
  const int t = tt * GT + tl;
  if (tid == 0) {  // sequential prefix sum like torch.cumsum (exact: durations are small integers)
    float run = 0.0f;
    for (int n = 0; n < N; ++n) {
      const float d = dur[b * d_bs + n];
      run += d;
      cen[n] = run - 0.5f * d;
      var[n] = fmaxf(fminf(rng[b * r_bs + n], d * 2.0f), 1e-5f);
    }
  }
  __syncthreads();
  const float ft = (float)t;
  float mx = -3.0e38f;
  for (int n = grp; n < N; n += 8) {
    const float df = ft - cen[n];
    float w = -0.5f * (1.8378770664093453f + logf(var[n]) + df * df / var[n]);
    if (n >= L) w = -1e15f;
    P[n * (GT + 1) + tl] = w;
    mx = fmaxf(mx, w);
  }
  red[grp][tl] = mx;
  __syncthreads();
  mx = red[0][tl];
#pragma unroll
  for (int g = 1; g < 8; ++g) mx = fmaxf(mx, red[g][tl]);
  __syncthreads();
  float sum = 0.0f;
  for (int n = grp; n < N; n += 8) {
    const float e = expf(P[n * (GT + 1) + tl] - mx);
    P[n * (GT + 1) + tl] = e;
    sum += e;
  }
  red[grp][tl] = sum;
  __syncthreads();
  sum = 0.0f;
#pragma unroll
  for (int g = 0; g < 8; ++g) sum += red[g][tl];
  const float inv = 1.0f / sum;
  const bool live = t < T && ft < frames[b];
  __syncthreads();
  for (int n = grp; n < N; n += 8) P[n * (GT + 1) + tl] *= inv;
  __syncthreads();
  const float* xb = x + (int64_t)b * x_bs;
  float* ob = out + (int64_t)b * C * T;
  for (int c0 = grp; c0 < C; c0 += 32) {   // 4 channels per pass: independent accumulators
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < L; ++n) {
      const float p = P[n * (GT + 1) + tl];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + 8 * u;
        if (c < C) acc[u] = fmaf(p, xb[(int64_t)c * x_cs + n], acc[u]);
      }
    }
    if (t < T) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + 8 * u;
        if (c < C) ob[(int64_t)c * T + t] = live ? acc[u] : 0.0f;
      }
    }
  }
}

// y[b, c, t] = x[b, c, t] + cb[b, c]   (`x + self.cond(g)` with g a per-utterance vector)
__global__ __launch_bounds__(256) void add_cbias_kernel(const float* __restrict__ x, int64_t x_bs, int64_t x_cs,
                                                        const float* __restrict__ cb, int64_t cb_bs,
                                                        float* __restrict__ y, int B, int C, int T) {
  const int64_t total = (int64_t)B * C * T;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int t = (int)(e % T);
    const int64_t bc = e / T;
    const int c = (int)(bc % C), b = (int)(bc / C);
    y[e] = x[b * x_bs + c * x_cs + t] + cb[b * cb_bs + c];
  }
}

// y = x < thr ? 0 : x   (`pitch[pitch < log(55)] = 0`, inference_plm.py:166)
__global__ __launch_bounds__(256) void zero_below_kernel(const float* __restrict__ x, float thr, float* __restrict__ y,
                                                         int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float v = x[e];
    y[e] = v < thr ? 0.0f : v;
  }
}

// out[b, i] = (int16)(x[b, i] / max_j |x[b, j]| * 32767 * gain), j over the first len[b] samples; samples
// past len[b] are 0.  Same operation order as inference_plm.py:186 and numpy's truncating astype(int16).
__global__ __launch_bounds__(1024) void peak_int16_kernel(const float* __restrict__ x, int64_t x_bs,
                                                          const int64_t* __restrict__ len, float gain,
                                                          int16_t* __restrict__ out, int64_t o_bs, int64_t n) {
  __shared__ float red[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t L = len ? min(n, max((int64_t)0, len[b])) : n;
  const float* xb = x + b * x_bs;
  float mx = 0.0f;
  for (int64_t i = tid; i < L; i += 1024) mx = fmaxf(mx, fabsf(xb[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
  int16_t* ob = out + b * o_bs;
  for (int64_t i = tid; i < n; i += 1024) {
    float v = 0.0f;
    if (i < L) v = xb[i] / mx * 32767.0f * gain;
    ob[i] = (int16_t)fminf(fmaxf(v, -32768.0f), 32767.0f);
  }
}


// ---- legacy (non-PLM) prosody path of SynthesizerTrn.infer: MaxPool1d(k, stride k) and the nearest-code search of
//      the prompt's 20-dim prosody track (ttv_v1/t2w2v_transformer.py:1041-1054, core_vq.py:175-183)
__global__ void maxpool1d_kernel(const float* __restrict__ x, int64_t x_bs, int64_t x_cs, float* __restrict__ y, int B,
                                 int C, int Lout, int k) {
  const int64_t n = (int64_t)B * C * Lout;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % Lout);
    const int c = (int)((i / Lout) % C);
    const int b = (int)(i / ((int64_t)Lout * C));
    const float* p = x + (int64_t)b * x_bs + (int64_t)c * x_cs + (int64_t)t * k;
    float m = p[0];
    for (int j = 1; j < k; ++j) m = fmaxf(m, p[j]);
    y[i] = m;
  }
}

// one thread per (b, t): dist[e] = -(|x|^2 - 2 x.e + |e|^2) as core_vq.py:177-181 writes it, first maximum wins
// (torch.max's tie rule); the code is written `rep` times: codes[b, rep t + r], r < rep, while rep t + r < Tout
__global__ void vq_nearest_kernel(const float* __restrict__ x, int64_t x_bs, int64_t x_cs, const float* __restrict__ embed,
                                  int64_t* __restrict__ codes, int64_t c_bs, int B, int D, int T, int bins, int rep,
                                  int Tout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T) return;
  const int b = i / T, t = i - b * T;
  const float* xp = x + (int64_t)b * x_bs + t;
  float xx = 0.0f;
  for (int d = 0; d < D; ++d) xx = fmaf(xp[(int64_t)d * x_cs], xp[(int64_t)d * x_cs], xx);
  float best = -INFINITY;
  int arg = 0;
  for (int e = 0; e < bins; ++e) {
    const float* ep = embed + (int64_t)e * D;
    float xe = 0.0f, ee = 0.0f;
    for (int d = 0; d < D; ++d) {
      xe = fmaf(xp[(int64_t)d * x_cs], ep[d], xe);
      ee = fmaf(ep[d], ep[d], ee);
    }
    const float dist = -((xx - 2.0f * xe) + ee);
    if (dist > best) { best = dist; arg = e; }
  }
  for (int r = 0; r < rep; ++r)
    if (rep * t + r < Tout) codes[(int64_t)b * c_bs + rep * t + r] = arg;
}

}  // namespace

#define HSP_STREAM static_cast<hipStream_t>(stream)

extern "C" int hsp_embedding_sum_f32(const int64_t* id0, const int64_t* id1, const int64_t* id2, const float* tab0,
                                     const float* tab1, const float* tab2, int32_t n0, int32_t n1, int32_t n2,
                                     float scale, float* out, int64_t o_bs, int64_t o_cs, int32_t B, int32_t C,
                                     int32_t T, void* stream) {
  if (!id0 || !tab0 || !out || B <= 0 || C <= 0 || T <= 0 || n0 <= 0) return HSP_EINVAL;
  if ((tab1 && (!id1 || n1 <= 0)) || (tab2 && (!id2 || n2 <= 0 || !tab1))) return HSP_EINVAL;
  int64_t blocks = ((int64_t)B * C * T + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(embedding_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, HSP_STREAM, id0, id1, id2, tab0, tab1,
                     tab2, n0, n1, n2, scale, out, o_bs, o_cs, B, C, T);
  return (int)hipGetLastError();
}

extern "C" int hsp_lstm_bidir_f32(const float* xproj, int64_t xp_bs, const float* whh_t, const float* bhh,
                                  const int64_t* lengths, float* out, int64_t o_bs, int64_t o_cs, int32_t B, int32_t H,
                                  int32_t N, void* stream) {
  if (!xproj || !whh_t || !bhh || !lengths || !out || B <= 0 || N <= 0) return HSP_EINVAL;
  if (H <= 0 || H > 256 || (H & 3)) return HSP_EINVAL;
  hipLaunchKernelGGL(lstm_bidir_kernel, dim3((unsigned)B, 2), dim3(1024), 0, HSP_STREAM, xproj, xp_bs, whh_t, bhh,
                     lengths, out, o_bs, o_cs, H, N);
  return (int)hipGetLastError();
}

extern "C" int hsp_duration_f32(const float* logw, int64_t lw_bs, const int64_t* lengths, float length_scale, float* dur,
                                int64_t d_bs, float* frames, int32_t B, int32_t N, void* stream) {
  if (!lengths || !dur || !frames || B <= 0 || N <= 0) return HSP_EINVAL;
  hipLaunchKernelGGL(duration_kernel, dim3((unsigned)B), dim3(256), 0, HSP_STREAM, logw, lw_bs, lengths, length_scale,
                     dur, d_bs, frames, N, logw ? 1 : 0);
  return (int)hipGetLastError();
}

extern "C" int hsp_gaussian_upsample_f32(const float* x, int64_t x_bs, int64_t x_cs, const float* dur, int64_t d_bs,
                                         const float* rng, int64_t r_bs, const int64_t* lengths, const float* frames,
                                         float* out, int32_t B, int32_t C, int32_t N, int32_t T, void* stream) {
  if (!x || !dur || !rng || !lengths || !frames || !out || B <= 0 || C <= 0 || N <= 0 || T <= 0) return HSP_EINVAL;
  const size_t lds_bytes = ((size_t)N * (GT + 1) + 2 * (size_t)N) * sizeof(float);
  if (lds_bytes > 60 * 1024) return HSP_EINVAL;   // N <= ~440 phones
  const int n_tt = (T + GT - 1) / GT;
  hipLaunchKernelGGL(gaussian_upsample_kernel, dim3((unsigned)(n_tt * B)), dim3(256), lds_bytes, HSP_STREAM, x, x_bs, x_cs,
                     dur, d_bs, rng, r_bs, lengths, frames, out, C, N, T, n_tt);
  return (int)hipGetLastError();
}

extern "C" int hsp_add_cbias_f32(const float* x, int64_t x_bs, int64_t x_cs, const float* cb, int64_t cb_bs, float* y,
                                 int32_t B, int32_t C, int32_t T, void* stream) {
  if (!x || !cb || !y || B <= 0 || C <= 0 || T <= 0) return HSP_EINVAL;
  int64_t blocks = ((int64_t)B * C * T + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(add_cbias_kernel, dim3((unsigned)blocks), dim3(256), 0, HSP_STREAM, x, x_bs, x_cs, cb, cb_bs, y, B, C,
                     T);
  return (int)hipGetLastError();
}

extern "C" int hsp_zero_below_f32(const float* x, float thr, float* y, int64_t n, void* stream) {
  if (!x || !y || n <= 0) return HSP_EINVAL;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(zero_below_kernel, dim3((unsigned)blocks), dim3(256), 0, HSP_STREAM, x, thr, y, n);
  return (int)hipGetLastError();
}

extern "C" int hsp_peak_int16(const float* x, int64_t x_bs, const int64_t* lengths, float gain, int16_t* out,
                              int64_t o_bs, int32_t B, int64_t n, void* stream) {
  if (!x || !out || B <= 0 || n <= 0) return HSP_EINVAL;
  hipLaunchKernelGGL(peak_int16_kernel, dim3((unsigned)B), dim3(1024), 0, HSP_STREAM, x, x_bs, lengths, gain, out, o_bs, n);
  return (int)hipGetLastError();
}

extern "C" int hsp_maxpool1d_f32(const float* x, int64_t x_bs, int64_t x_cs, float* y, int32_t B, int32_t C, int32_t L,
                                 int32_t k, void* stream) {
  if (!x || !y || B <= 0 || C <= 0 || L <= 0 || k <= 0 || L / k <= 0) return HSP_EINVAL;
  const int Lout = L / k;                      // torch MaxPool1d(kernel_size = stride = k), floor mode
  int64_t blocks = ((int64_t)B * C * Lout + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(maxpool1d_kernel, dim3((unsigned)blocks), dim3(256), 0, HSP_STREAM, x, x_bs, x_cs, y, B, C, Lout, k);
  return (int)hipGetLastError();
}

extern "C" int hsp_vq_nearest_f32(const float* x, int64_t x_bs, int64_t x_cs, const float* embed, int64_t* codes,
                                  int64_t c_bs, int32_t B, int32_t D, int32_t T, int32_t bins, int32_t rep, int32_t Tout,
                                  void* stream) {
  if (!x || !embed || !codes || B <= 0 || D <= 0 || T <= 0 || bins <= 0 || rep <= 0 || Tout <= 0 || Tout > rep * T)
    return HSP_EINVAL;
  const int blocks = (B * T + 63) / 64;
  hipLaunchKernelGGL(vq_nearest_kernel, dim3((unsigned)blocks), dim3(64), 0, HSP_STREAM, x, x_bs, x_cs, embed, codes, c_bs, B,
                     D, T, bins, rep, Tout);
  return (int)hipGetLastError();
}
