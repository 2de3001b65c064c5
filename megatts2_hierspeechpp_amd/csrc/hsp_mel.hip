// Prompt front-end: the two HBM-bound steps around the DFT GEMM of
// MelSpectrogramFixed (Mels_preprocess.py:8-18).  Declarations and semantics: include/hsp.h.
#include "hsp_device.h"

namespace {

// frames[b][n][t] = w[n] * x[b][reflect(t * hop + n - n_fft / 2)].  One workgroup = 64 frames x 256
// window positions of one utterance; lanes run along t (the GEMM's column axis, time fastest like
// every activation of the path), so the stores are coalesced and the strided reads of x (lane stride
// = hop) hit lines that the 4 following n of the same thread re-use from L1.
__global__ __launch_bounds__(256) void stft_frames_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ frames, int L, int n_fft, int hop,
                                                          int T, int f_ld) {
  const int b = blockIdx.z;
  const int t = blockIdx.x * 64 + (threadIdx.x & 63);
  const int n0 = blockIdx.y * 256 + (threadIdx.x >> 6) * 64;
  if (t >= f_ld) return;
  const float* xb = x + (int64_t)b * L;
  float* fb = frames + (int64_t)b * n_fft * f_ld;
  const int base = t * hop - (n_fft >> 1);
  for (int n = n0; n < min(n0 + 64, n_fft); ++n) {
    int i = base + n;
    i = i < 0 ? -i : i;
    i = i >= L ? 2 * (L - 1) - i : i;
    fb[(int64_t)n * f_ld + t] = t < T ? w[n] * xb[i] : 0.0f;   // columns [T, f_ld) are row-pitch padding
  }
}

// out[b][m][t] = log(sum_{f in [lo[m], hi[m])} fb[f][m] * (re^2 + im^2) + eps); lanes along t.
__global__ __launch_bounds__(64) void power_mel_log_kernel(const float* __restrict__ spec, int64_t s_bs, int s_ld,
                                                           const float* __restrict__ fbank, const int* __restrict__ f_lo,
                                                           const int* __restrict__ f_hi, float* __restrict__ out,
                                                           int n_freqs, int n_mels, int T_out, float eps) {
  const int b = blockIdx.z, m = blockIdx.y;
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= T_out) return;
  const float* re = spec + (int64_t)b * s_bs + t;
  const float* im = re + (int64_t)n_freqs * s_ld;
  const int lo = f_lo[m], hi = f_hi[m];
  float acc = 0.0f;
  for (int f = lo; f < hi; ++f) {
    const float r = re[(int64_t)f * s_ld], i = im[(int64_t)f * s_ld];
    acc = fmaf(fbank[f * n_mels + m], fmaf(r, r, i * i), acc);
  }
  out[((int64_t)b * n_mels + m) * T_out + t] = logf(acc + eps);
}

}  // namespace

#define HSP_STREAM static_cast<hipStream_t>(stream)

extern "C" int hsp_stft_frames_f32(const float* x, const float* window, float* frames, int32_t B, int32_t L,
                                   int32_t n_fft, int32_t hop, int32_t T, int32_t f_ld, void* stream) {
  if (!x || !window || !frames || B <= 0 || n_fft <= 0 || hop <= 0 || T <= 0 || f_ld < T) return HSP_EINVAL;
  if (L <= n_fft / 2) return HSP_EINVAL;                       // reflect padding needs L > n_fft / 2 (as torch.stft)
  if ((int64_t)(T - 1) * hop > (int64_t)L) return HSP_EINVAL;  // the last frame's centre lies inside the signal
  if (B > 65535 || (n_fft + 255) / 256 > 65535) return HSP_EINVAL;
  hipLaunchKernelGGL(stft_frames_kernel, dim3((f_ld + 63) / 64, (n_fft + 255) / 256, B), dim3(256), 0, HSP_STREAM, x,
                     window, frames, L, n_fft, hop, T, f_ld);
  return (int)hipGetLastError();
}

extern "C" int hsp_power_mel_log_f32(const float* spec, int64_t s_bs, int32_t s_ld, const float* fb,
                                     const int32_t* f_lo, const int32_t* f_hi, float* out, int32_t B, int32_t n_freqs,
                                     int32_t n_mels, int32_t T_out, float eps, void* stream) {
  if (!spec || !fb || !f_lo || !f_hi || !out || B <= 0 || n_freqs <= 0 || n_mels <= 0 || T_out <= 0) return HSP_EINVAL;
  if (s_ld < T_out || B > 65535 || n_mels > 65535) return HSP_EINVAL;
  hipLaunchKernelGGL(power_mel_log_kernel, dim3((T_out + 63) / 64, n_mels, B), dim3(64), 0, HSP_STREAM, spec, s_bs,
                     s_ld, fb, f_lo, f_hi, out, n_freqs, n_mels, T_out, eps);
  return (int)hipGetLastError();
}
