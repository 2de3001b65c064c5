// Device-side helpers shared by the libhsp kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hsp.h"

#define HSP_WAVE 64

// Raise a kernel's dynamic-LDS limit to `bytes` once per device.  hipFuncSetAttribute applies to the current
// device only, so the "already raised" flag is kept per device (`flags`: one zero-initialised static array per
// kernel instantiation).  Idempotent and safe from several host threads; kept out of the launch path afterwards
// so that launches stay legal inside a hipGraph stream capture.  Returns 0 or the hipError_t / HSP_EINVAL.
#include <atomic>
constexpr int HSP_MAX_DEVICES = 32;
struct hsp_lds_flags { std::atomic<int> raised[HSP_MAX_DEVICES]; };
inline int hsp_raise_lds_limit(const void* kernel, int bytes, hsp_lds_flags& flags) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= HSP_MAX_DEVICES) return HSP_EINVAL;
  if (flags.raised[dev].load(std::memory_order_acquire)) return 0;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  flags.raised[dev].store(1, std::memory_order_release);
  return 0;
}

__device__ __forceinline__ float hsp_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp2 (v_exp_f32) and a true division: ~12 instructions
// instead of ocml tanhf's ~100 (which dominated the GELU / WN-gate epilogues).  Absolute error < 2e-7
// (exp2 is accurate to 1 ulp; the subtraction from 1 bounds the error by ulp(1)); saturates to +-1 cleanly
// for large |x| (exp -> inf / 0).
__device__ __forceinline__ float hsp_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);   // exp(2x) = 2^(2x log2 e)
  return 1.0f - 2.0f / (e + 1.0f);
}

__device__ __forceinline__ float hsp_apply_act(float v, int act) {
  switch (act) {
    case HSP_ACT_TANH: return tanhf(v);   // final waveform tanh: full precision
    case HSP_ACT_GELU_TANH: {
      const float k0 = 0.7978845608028654f, k1 = 0.044715f;
      return 0.5f * v * (1.0f + hsp_tanh(k0 * (v + k1 * v * v * v)));
    }
    case HSP_ACT_RELU: return fmaxf(v, 0.0f);
    case HSP_ACT_MISH: {
      float sp = v > 20.0f ? v : log1pf(expf(v));
      return v * tanhf(sp);
    }
    case HSP_ACT_SILU: return v * hsp_sigmoid(v);
    case HSP_ACT_SOFTPLUS: return v > 20.0f ? v : log1pf(expf(v));
    case HSP_ACT_GELU_ERF: return 0.5f * v * (1.0f + erff(v * 0.7071067811865476f));
    default: return v;
  }
}

// SnakeBeta on one sample: x + binv * sin(x * ea)^2   (activations.py:118)
// sin(y) for the snake argument.  ocml's sinf carries a Payne-Hanek slow path that costs
// 320 B of scratch per lane; the activation argument x*exp(alpha) is O(1..1e3), so a
// three-constant Cody-Waite reduction by pi (k*PI_HI exact for |k| < 2^15) followed by
// the degree-11 odd Taylor polynomial on [-pi/2, pi/2] (remainder < 6e-8) is used
// instead.  sin(y) = (-1)^k sin(r); only sin^2 is consumed, so the sign is dropped.
__device__ __forceinline__ float hsp_sin_abs(float y) {
  const float k = rintf(y * 0.318309886183790672f);
  float r = fmaf(k, -3.140625f, y);
  r = fmaf(k, -9.67502593994140625e-4f, r);
  r = fmaf(k, -1.509957990978376432e-7f, r);
  const float r2 = r * r;
  float p = -2.50521083854417188e-8f;
  p = fmaf(p, r2, 2.75573192239858907e-6f);
  p = fmaf(p, r2, -1.98412698412698413e-4f);
  p = fmaf(p, r2, 8.33333333333333333e-3f);
  p = fmaf(p, r2, -1.66666666666666667e-1f);
  return fmaf(r * r2, p, r);
}

__device__ __forceinline__ float hsp_snake(float x, float ea, float binv) {
  const float s = hsp_sin_abs(x * ea);
  return fmaf(binv, s * s, x);
}

// Hardware-trig form used inside the fused conv prologue, where every VALU cycle is taken
// from the fp32 MFMAs running on the same SIMD:  sin^2(y) = (1 - cos 2y) / 2 with
// v_cos_f32 (argument in revolutions, reduced by v_fract_f32):
//   snake(x) = (x + kb) - kb * cos(2*pi * fract(x * kf)),   kf = exp(alpha)/pi, kb = binv/2
// 5 VALU ops instead of 14.  v_cos_f32 is accurate to ~1e-6 absolute on [0, 1).
__device__ __forceinline__ float hsp_snake_hw(float x, float kf, float kb) {
  const float c = __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(x * kf));
  return fmaf(-kb, c, x + kb);
}

__device__ __forceinline__ int hsp_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Shared epilogue of both conv kernels for one output element in PLAIN/SHUFFLE row
// modes (GATE modes combine two accumulators before calling with act = NONE).
struct hsp_epi_ctx {
  const hsp_conv1d_args* a;
};

__device__ __forceinline__ void hsp_epilogue_store(const hsp_conv1d_args& a, int b, int co, int t, float v) {
  // v already holds act(acc + bias + cbias)
  float mk = 1.0f;
  if (a.mask_mode != HSP_MASK_NONE) mk = a.mask[(int64_t)b * a.mask_bs + t];
  if (a.mask_mode & HSP_MASK_PRE) v *= mk;
  if (a.cscale) v *= a.cscale[(int64_t)b * a.cscale_bs + co];
  v *= a.scale;
  if (a.res) v += a.res[(int64_t)b * a.res_bs + (int64_t)co * a.res_cs + t];
  if (a.mask_mode & HSP_MASK_POST) v *= mk;
  float* yp = a.y + (int64_t)b * a.y_bs + (int64_t)co * a.y_cs + t;
  if (a.accumulate) v += *yp;
  *yp = v * a.post_scale;
}
