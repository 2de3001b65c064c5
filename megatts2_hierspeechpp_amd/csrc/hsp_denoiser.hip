// Kernels of the prompt denoiser (MP-SENet, reference denoiser/: SURVEY.md §8f N4) that are not convolutions or
// GEMMs: the spectrogram front / back end of denoiser/infer.py and the normalisation / gating glue of
// denoiser/generator.py and conformer.py.  Everything here is pointwise or a small reduction over a prompt of a few
// seconds (a [64, T, 201] tensor is 40 MB at 5 s): latency, not throughput.  Reference call sites are listed next
// to each entry point in include/hsp.h.
#include "hsp_device.h"

namespace {

// one thread per element, no grid-stride loop in these kernels: the grid covers every element (the x dimension
// of a grid holds 2^31 - 1 blocks) or the launcher refuses (dn_fits)
constexpr int64_t DN_MAX_BLOCKS = 0x7fffffff;
inline unsigned dn_grid(int64_t n, int threads) {
  int64_t b = (n + threads - 1) / threads;
  if (b < 1) b = 1;
  return (unsigned)(b > DN_MAX_BLOCKS ? DN_MAX_BLOCKS : b);
}
inline bool dn_fits(int64_t n, int threads) { return (n + threads - 1) / threads <= DN_MAX_BLOCKS; }

// block-wide sum of a double (1024 threads at most); every thread receives the total
__device__ __forceinline__ double dn_block_sum(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < nw; ++w) t += sh[w];
  return t;
}

__global__ __launch_bounds__(1024) void sum_sq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  __shared__ double sh[16];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += (double)x[i] * (double)x[i];
  s = dn_block_sum(s, sh);
  if (threadIdx.x == 0) out[0] = (float)s;
}

// spec rows [0, nf) real, [nf, 2 nf) imaginary, pitch s_ld  ->  mag[f][t] = |z|^c, pha[f][t] = angle(z)
__global__ __launch_bounds__(256) void mag_pha_kernel(const float* __restrict__ spec, int64_t s_ld, float* __restrict__ mag,
                                                      float* __restrict__ pha, int nf, int T, float c) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)nf * T) return;
  const int f = (int)(i / T), t = (int)(i % T);
  const float re = spec[(int64_t)f * s_ld + t];
  // a real FFT returns exactly +0 for the imaginary part of the DC and Nyquist bins (torch.stft does); the DFT
  // product leaves rounding noise there, whose sign would turn a phase of pi into -pi
  const float im = (f == 0 || f == nf - 1) ? 0.0f : spec[(int64_t)(nf + f) * s_ld + t];
  mag[i] = powf(hypotf(re, im), c);
  pha[i] = atan2f(im, re);
}

// InstanceNorm2d(affine) + PReLU over the N contiguous values of each channel plane (one utterance), in place
__global__ __launch_bounds__(1024) void instnorm_prelu_kernel(float* __restrict__ x, int64_t cs, int64_t N,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ slope, float eps) {
  __shared__ double sh[16];
  const int c = blockIdx.x;
  float* p = x + (int64_t)c * cs;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < N; i += blockDim.x) s += (double)p[i];
  const double mean = dn_block_sum(s, sh) / (double)N;
  double q = 0.0;
  for (int64_t i = threadIdx.x; i < N; i += blockDim.x) {
    const double d = (double)p[i] - mean;
    q += d * d;
  }
  const double var = dn_block_sum(q, sh) / (double)N;   // biased, as F.instance_norm
  const float inv = (float)(1.0 / sqrt(var + (double)eps));
  const float m = (float)mean, g = gamma[c], b = beta[c], sl = slope[c];
  for (int64_t i = threadIdx.x; i < N; i += blockDim.x) {
    const float y = (p[i] - m) * inv * g + b;
    p[i] = y > 0.0f ? y : sl * y;
  }
}

// depthwise Conv1d (same padding, odd K) + BatchNorm1d in eval mode + SiLU over [B, C, N]
__global__ __launch_bounds__(256) void dwconv_bn_silu_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const float* __restrict__ bn_w,
                                                             const float* __restrict__ bn_b, const float* __restrict__ bn_mean,
                                                             const float* __restrict__ bn_var, float bn_eps,
                                                             float* __restrict__ y, int C, int N, int K, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int n = (int)(i % N);
  const int c = (int)((i / N) % C);
  const float* row = x + (i - n);
  const float* wc = w + (int64_t)c * K;
  const int half = K >> 1;
  float acc = 0.0f;
  for (int j = 0; j < K; ++j) {
    const int m = n + j - half;
    if (m >= 0 && m < N) acc = fmaf(wc[j], row[m], acc);
  }
  acc += bias[c];
  const float alpha = bn_w[c] / sqrtf(bn_var[c] + bn_eps);
  const float v = acc * alpha + (bn_b[c] - bn_mean[c] * alpha);
  y[i] = v / (1.0f + expf(-v));
}

// out[t][f] = mag[t][f] * beta * sigmoid(slope[f] * m[t][f])
__global__ __launch_bounds__(256) void lsigmoid_mul_kernel(const float* __restrict__ m, const float* __restrict__ slope,
                                                           float beta, const float* __restrict__ mag, float* __restrict__ out,
                                                           int F, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float s = slope[i % F] * m[i];
  out[i] = mag[i] * (beta / (1.0f + expf(-s)));
}

__global__ __launch_bounds__(256) void atan2_kernel(const float* __restrict__ yy, const float* __restrict__ xx,
                                                    float* __restrict__ out, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < total) out[i] = atan2f(yy[i], xx[i]);
}

// re = mag^p cos(pha), im = mag^p sin(pha)
__global__ __launch_bounds__(256) void polar_kernel(const float* __restrict__ mag, const float* __restrict__ pha, float p,
                                                    float* __restrict__ re, int64_t re_ld, float* __restrict__ im,
                                                    int64_t im_ld, int T, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t f = i / T, t = i % T;
  const float m = p == 1.0f ? mag[i] : powf(mag[i], p);
  re[f * re_ld + t] = m * cosf(pha[i]);
  im[f * im_ld + t] = m * sinf(pha[i]);
}

// torch.istft after the inverse DFT: out[n] = sum_t frames[n + N/2 - t hop][t] w[.] / sum_t w[.]^2, n < hop (T - 1)
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, int64_t f_ld,
                                                        const float* __restrict__ window, float* __restrict__ out, int n_fft,
                                                        int hop, int T, int64_t L, float scale) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= L) return;
  const int64_t pos = n + (n_fft >> 1);
  int64_t t_hi = pos / hop;
  if (t_hi > T - 1) t_hi = T - 1;
  int64_t t_lo = (pos - n_fft + hop) / hop;   // smallest t with pos - t hop <= n_fft - 1
  if (pos - n_fft + 1 <= 0) t_lo = 0;
  float num = 0.0f, den = 0.0f;
  for (int64_t t = t_lo; t <= t_hi; ++t) {
    const int k = (int)(pos - t * hop);
    if (k < 0 || k >= n_fft) continue;
    const float w = window[k];
    num += frames[(int64_t)k * f_ld + t] * w;
    den += w * w;
  }
  out[n] = num / den * scale;
}

}  // namespace

#define HSP_STREAM static_cast<hipStream_t>(stream)

extern "C" int hsp_sum_sq_f32(const float* x, int64_t n, float* out, void* stream) {
  if (!x || !out || n <= 0) return HSP_EINVAL;
  hipLaunchKernelGGL(sum_sq_kernel, dim3(1), dim3(1024), 0, HSP_STREAM, x, n, out);
  return (int)hipGetLastError();
}

extern "C" int hsp_mag_pha_f32(const float* spec, int64_t s_ld, float* mag, float* pha, int32_t n_freqs, int32_t T,
                               float compress, void* stream) {
  if (!spec || !mag || !pha || n_freqs < 2 || T <= 0 || s_ld < T) return HSP_EINVAL;
  const int64_t total = (int64_t)n_freqs * T;
  if (!dn_fits(total, 256)) return HSP_EINVAL;
  hipLaunchKernelGGL(mag_pha_kernel, dim3(dn_grid(total, 256)), dim3(256), 0, HSP_STREAM, spec, s_ld, mag, pha, n_freqs, T,
                     compress);
  return (int)hipGetLastError();
}

extern "C" int hsp_instnorm_prelu_f32(float* x, int64_t x_cs, int32_t C, int64_t N, const float* gamma, const float* beta,
                                      const float* slope, float eps, void* stream) {
  if (!x || !gamma || !beta || !slope || C <= 0 || N <= 0 || x_cs < N) return HSP_EINVAL;
  hipLaunchKernelGGL(instnorm_prelu_kernel, dim3((unsigned)C), dim3(1024), 0, HSP_STREAM, x, x_cs, N, gamma, beta, slope, eps);
  return (int)hipGetLastError();
}

extern "C" int hsp_dwconv_bn_silu_f32(const float* x, const float* w, const float* bias, const float* bn_weight,
                                      const float* bn_bias, const float* bn_mean, const float* bn_var, float bn_eps, float* y,
                                      int32_t B, int32_t C, int32_t N, int32_t K, void* stream) {
  if (!x || !w || !bias || !bn_weight || !bn_bias || !bn_mean || !bn_var || !y) return HSP_EINVAL;
  if (B <= 0 || C <= 0 || N <= 0 || K <= 0 || (K & 1) == 0) return HSP_EINVAL;
  const int64_t total = (int64_t)B * C * N;
  if (!dn_fits(total, 256)) return HSP_EINVAL;
  hipLaunchKernelGGL(dwconv_bn_silu_kernel, dim3(dn_grid(total, 256)), dim3(256), 0, HSP_STREAM, x, w, bias, bn_weight, bn_bias,
                     bn_mean, bn_var, bn_eps, y, C, N, K, total);
  return (int)hipGetLastError();
}

extern "C" int hsp_lsigmoid_mul_f32(const float* m, const float* slope, float beta, const float* mag, float* out, int32_t T,
                                    int32_t F, void* stream) {
  if (!m || !slope || !mag || !out || T <= 0 || F <= 0) return HSP_EINVAL;
  const int64_t total = (int64_t)T * F;
  if (!dn_fits(total, 256)) return HSP_EINVAL;
  hipLaunchKernelGGL(lsigmoid_mul_kernel, dim3(dn_grid(total, 256)), dim3(256), 0, HSP_STREAM, m, slope, beta, mag, out, F, total);
  return (int)hipGetLastError();
}

extern "C" int hsp_atan2_f32(const float* y, const float* x, float* out, int64_t n, void* stream) {
  if (!y || !x || !out || n <= 0) return HSP_EINVAL;
  if (!dn_fits(n, 256)) return HSP_EINVAL;
  hipLaunchKernelGGL(atan2_kernel, dim3(dn_grid(n, 256)), dim3(256), 0, HSP_STREAM, y, x, out, n);
  return (int)hipGetLastError();
}

extern "C" int hsp_polar_f32(const float* mag, const float* pha, float power, float* re, int64_t re_ld, float* im,
                             int64_t im_ld, int32_t F, int32_t T, void* stream) {
  if (!mag || !pha || !re || !im || F <= 0 || T <= 0 || re_ld < T || im_ld < T) return HSP_EINVAL;
  const int64_t total = (int64_t)F * T;
  if (!dn_fits(total, 256)) return HSP_EINVAL;
  hipLaunchKernelGGL(polar_kernel, dim3(dn_grid(total, 256)), dim3(256), 0, HSP_STREAM, mag, pha, power, re, re_ld, im, im_ld, T,
                     total);
  return (int)hipGetLastError();
}

extern "C" int hsp_istft_ola_f32(const float* frames, int64_t f_ld, const float* window, float* out, int32_t n_fft,
                                 int32_t hop, int32_t T, float scale, void* stream) {
  if (!frames || !window || !out || n_fft <= 0 || (n_fft & 1) || hop <= 0 || hop > n_fft || T < 2 || f_ld < T) return HSP_EINVAL;
  const int64_t L = (int64_t)hop * (T - 1);
  if (!dn_fits(L, 256)) return HSP_EINVAL;
  hipLaunchKernelGGL(istft_ola_kernel, dim3(dn_grid(L, 256)), dim3(256), 0, HSP_STREAM, frames, f_ld, window, out, n_fft, hop, T,
                     L, scale);
  return (int)hipGetLastError();
}
