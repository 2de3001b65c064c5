// Frequency-domain form of a long dilated Conv1d (round 4): hsp_dftseg_fwd_f32 / hsp_dftseg_inv_f32.
//
// The Generator's AMP blocks (hierspeechpp_speechsynthesizer.py:340-392: convs1 k in {3, 7, 11}, dilation (1, 3, 5),
// convs2 dilation 1, C -> C channels) spend 2 k C^2 flops per output sample: 39 of the 78 ms of the 32 x 4 s step are the
// k = 7 and k = 11 convs.  Overlap-save with a 128-point transform computes the same correlation with
//   a 128-point real DFT of every 128 - (k - 1) = 118 (122) output samples per channel,
//   ONE complex C x C product per frequency bin (64 bins: DC and Nyquist share a slot) and
//   the inverse transform,
// i.e. 4 * 65 / 118 = 2.2 real multiply-adds per output and channel pair instead of k = 11 (7): 5.0 x (3.3 x) fewer in
// the channel-mixing product, which stays on the implicit-GEMM conv kernel -- a 1x1 "conv" whose batch index is the BIN
// (hsp_conv1d_args.w_bs: one [2C][2C] real block matrix [[Wr, Wi], [-Wi, Wr]] = conj(W) per bin).  fp32 throughout;
// measured against float64 the result is as close as the direct fp32 sum (1.2-2.3e-6 against 1.4-2.9e-6 at C = 128 ... 512,
// tools/fft_conv_err.py): a 128-term DFT sum and a C-term product per output instead of a k C-term one.
//
// Dilation d = d interleaved unit-dilation problems (polyphase): phase p of the zero-padded input, q_p[i] = xpad[p + d i],
// yields the outputs t = p + d i.  Segment s of phase p reads q_p[s hop .. s hop + 127] and produces the outputs
// i in [s hop, s hop + hop).  Spectrum layout: xf[bin][part * C + c][n], n = (b * d + p) * nseg + s -- for every bin a
// [2C][Np] matrix, what the conv kernel takes as one "utterance".
//
// Both transforms are products with a constant 128 x 128 matrix on v_mfma_f32_32x32x2_f32 (exact fp32): a wave keeps
// its 32 rows of the matrix in registers (64 VGPRs) for the whole launch.  That is 2 x 282 flops per sample and channel
// -- as much as the channel product at C = 128, a quarter of it at C = 512 -- and it needs no butterfly network; a real
// FFT in registers would remove most of it.
//   forward: the input rows of a channel group are staged in LDS once (zero padding applied there), B fragments are
//            strided LDS reads (lane = segment), the 128 spectrum rows of a segment block leave as 128-B runs;
//   inverse: B fragments are coalesced global loads from the 128 spectrum rows, the time samples are scattered into an
//            LDS image of the output rows (all phases), then bias / residual / running sum are applied in one
//            coalesced pass (the epilogue of the conv this replaces: hsp_conv1d_args bias, res, accumulate, post_scale).
#include "hsp_device.h"

namespace {
typedef float ds_f32x16 __attribute__((ext_vector_type(16)));
constexpr int DS_N = 128;
#define DS_ACC_ROW(r, half) (((r) & 3) + 8 * ((r) >> 2) + 4 * (half))

// A workgroup owns `cg` channel rows of one utterance.  Its GEMM columns are the (channel, phase, segment) triples of
// those rows, flattened -- col = (ch * d + p) * nseg + s -- so that a 32-column MFMA block is full whatever nseg is
// (L = 800 at dilation 5 has TWO segments per phase: one block per (channel, phase) ran the matrix cores 16 x idle).
struct DsCol {
  int ch, p, s;
  bool ok;
};
__device__ __forceinline__ DsCol ds_col(int col, int ncols, int d, int nseg) {
  DsCol c;
  c.ok = col < ncols;
  const int cc = c.ok ? col : ncols - 1;
  const int per = d * nseg;
  c.ch = cc / per;
  const int rem = cc - c.ch * per;
  c.p = rem / nseg;
  c.s = rem - c.p * nseg;
  return c;
}

__global__ __launch_bounds__(256) void dftseg_fwd_kernel(const hsp_dftseg_args a, int cg, int pitch, int ngrp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [cg][pitch]: zero-padded input rows
  const int b = blockIdx.x / ngrp, c0 = (blockIdx.x % ngrp) * cg;
  const int ncg = min(cg, a.C - c0);                            // channel rows of this workgroup
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int hop = DS_N - (a.k - 1), d = a.dil;
  // A fragments: rows 32 wave + l32 of the forward matrix, taps 2 ks + half
  float fa[64];
  {
    const float* frow = a.dft + (32 * wave + l32) * DS_N + half;
#pragma unroll
    for (int ks = 0; ks < 64; ++ks) fa[ks] = frow[2 * ks];
  }
  for (int ch = 0; ch < ncg; ++ch) {
    const float* xr = a.x + (int64_t)b * a.x_bs + (int64_t)(c0 + ch) * a.x_cs;
    float* row = lds + ch * pitch;
    for (int t = tid; t < pitch; t += 256) {
      const int ti = t - a.pad;
      const float v = xr[min(max(ti, 0), a.L - 1)];           // unconditional load, selected below
      row[t] = (ti >= 0 && ti < a.L) ? v : 0.0f;
    }
  }
  __syncthreads();
  const int ncols = ncg * d * a.nseg;
  const int step = 2 * d;                                      // floats between the taps of consecutive k-steps
  for (int cb = 0; cb < ncols; cb += 32) {
    const DsCol q = ds_col(cb + l32, ncols, d, a.nseg);
    const float* bp = lds + q.ch * pitch + q.p + d * (q.s * hop + half);
    ds_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 64; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks], bp[ks * step], acc, 0, 0, 0);
    if (q.ok) {
      float* op = a.xf + (int64_t)(c0 + q.ch) * a.Np + (b * d + q.p) * a.nseg + q.s;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * wave + DS_ACC_ROW(r, half);        // 0..63: Re(bin), 64..127: Im(bin - 64) (64: Nyquist)
        op[(int64_t)(row & 63) * a.xf_bs + (int64_t)(row >> 6) * a.C * a.Np] = acc[r];
      }
    }
  }
}

__global__ __launch_bounds__(256) void dftseg_inv_kernel(const hsp_dftseg_args a, int cg, int pitch, int ngrp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [cg][pitch]: the output rows
  const int b = blockIdx.x / ngrp, c0 = (blockIdx.x % ngrp) * cg;
  const int ncg = min(cg, a.C - c0);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int hop = DS_N - (a.k - 1), d = a.dil;
  // A fragments: rows (time samples of a segment) 32 wave + l32 of the inverse matrix, spectrum rows 2 ks + half
  float fa[64];
  {
    const float* frow = a.dft + (32 * wave + l32) * DS_N + half;
#pragma unroll
    for (int ks = 0; ks < 64; ++ks) fa[ks] = frow[2 * ks];
  }
  const int ncols = ncg * d * a.nseg;
  // spectrum row r = 2 ks + half: bin r & 63, part r >> 6 -> k-steps 0..31 read the real parts, 32..63 the imaginary.
  // The 64 loads of a column block are requested one block AHEAD of the MFMAs that consume them.
  auto request = [&](int cb, float (&fb)[64]) __attribute__((always_inline)) {
    const DsCol q = ds_col(cb + l32, ncols, d, a.nseg);
    const float* re = a.xf + (int64_t)half * a.xf_bs + (int64_t)(c0 + q.ch) * a.Np + (b * d + q.p) * a.nseg + q.s;
    const float* im = re + (int64_t)a.C * a.Np;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
      fb[ks] = re[(int64_t)(2 * ks) * a.xf_bs];
      fb[32 + ks] = im[(int64_t)(2 * ks) * a.xf_bs];
    }
  };
  float f0[64], f1[64];
  request(0, f0);
  auto consume = [&](int cb, const float (&fb)[64]) __attribute__((always_inline)) {
    ds_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 64; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks], fb[ks], acc, 0, 0, 0);
    const DsCol q = ds_col(cb + l32, ncols, d, a.nseg);
    if (q.ok) {
      float* row = lds + q.ch * pitch;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = 32 * wave + DS_ACC_ROW(r, half);          // time inside the segment: the first hop are valid
        const int t = q.p + d * (q.s * hop + i);
        if (i < hop && t < a.L) row[t] = acc[r];
      }
    }
  };
  for (int cb = 0; cb < ncols; cb += 64) {
    if (cb + 32 < ncols) request(cb + 32, f1);
    consume(cb, f0);
    if (cb + 32 < ncols) {
      if (cb + 64 < ncols) request(cb + 64, f0);
      consume(cb + 32, f1);
    }
  }
  __syncthreads();
  // ---- epilogue of the conv this replaces: y = ((corr + bias + res) [+ y]) * post_scale, one coalesced pass
  const bool vec = ((a.L | (int)a.y_bs | (int)a.y_cs | (int)a.res_bs | (int)a.res_cs) & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(a.y) | reinterpret_cast<uintptr_t>(a.res)) & 15) == 0;   // workgroup-uniform
  for (int ch = 0; ch < ncg; ++ch) {
    const int c = c0 + ch;
    const float bz = a.bias ? a.bias[c] : 0.0f;
    const float* row = lds + ch * pitch;
    float* yr = a.y + (int64_t)b * a.y_bs + (int64_t)c * a.y_cs;
    const float* rr = a.res ? a.res + (int64_t)b * a.res_bs + (int64_t)c * a.res_cs : nullptr;
    if (vec) {
      for (int t = 4 * tid; t < a.L; t += 1024) {
        float4 v = *reinterpret_cast<const float4*>(row + t);
        v.x += bz, v.y += bz, v.z += bz, v.w += bz;
        if (rr) {
          const float4 r4 = *reinterpret_cast<const float4*>(rr + t);
          v.x += r4.x, v.y += r4.y, v.z += r4.z, v.w += r4.w;
        }
        if (a.accumulate) {
          const float4 o4 = *reinterpret_cast<const float4*>(yr + t);
          v.x += o4.x, v.y += o4.y, v.z += o4.z, v.w += o4.w;
        }
        v.x *= a.post_scale, v.y *= a.post_scale, v.z *= a.post_scale, v.w *= a.post_scale;
        *reinterpret_cast<float4*>(yr + t) = v;
      }
    } else {
      for (int t = tid; t < a.L; t += 256) {
        float v = row[t] + bz;
        if (rr) v += rr[t];
        if (a.accumulate) v += yr[t];
        yr[t] = v * a.post_scale;
      }
    }
  }
}

int ds_check(const hsp_dftseg_args& a) {
  if (!a.xf || !a.dft || a.B <= 0 || a.C <= 0 || a.L <= 0 || a.k < 2 || a.k > 64 || a.dil < 1 || a.dil > 8) return HSP_EINVAL;
  const int hop = DS_N - (a.k - 1);
  const int per_phase = (a.L + a.dil - 1) / a.dil;
  if (a.nseg != (per_phase + hop - 1) / hop || a.pad < 0 || a.pad > (a.k - 1) * a.dil) return HSP_EINVAL;
  if ((int64_t)a.B * a.dil * a.nseg > a.Np || a.xf_bs < (int64_t)2 * a.C * a.Np) return HSP_EINVAL;
  return 0;
}

// channel rows per workgroup: as many as fit 64 KB of LDS (at least one: up to 40 000 floats = the whole 160 KB)
int ds_group(int pitch, int C) {
  int cg = 16384 / pitch;
  cg = cg < 1 ? 1 : (cg > 16 ? 16 : cg);
  return cg > C ? C : cg;
}
}  // namespace

extern "C" int hsp_dftseg_fwd_f32(const hsp_dftseg_args* ap, void* stream) {
  if (!ap || !ap->x) return HSP_EINVAL;
  const hsp_dftseg_args& a = *ap;
  if (int e = ds_check(a)) return e;
  const int hop = DS_N - (a.k - 1);
  const int pitch = a.dil * (a.nseg * hop + DS_N);             // every tap of every segment of every phase is inside
  if (pitch > 40000) return HSP_EINVAL;
  const int cg = ds_group(pitch, a.C), ngrp = (a.C + cg - 1) / cg;
  const size_t lds_bytes = (size_t)cg * pitch * sizeof(float);
  static hsp_lds_flags flags;
  if (lds_bytes > 32 * 1024)
    if (int e = hsp_raise_lds_limit(reinterpret_cast<const void*>(dftseg_fwd_kernel), 160 * 1024, flags)) return e;
  hipLaunchKernelGGL(dftseg_fwd_kernel, dim3((unsigned)(a.B * ngrp)), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), a,
                     cg, pitch, ngrp);
  return (int)hipGetLastError();
}

extern "C" int hsp_dftseg_inv_f32(const hsp_dftseg_args* ap, void* stream) {
  if (!ap || !ap->y) return HSP_EINVAL;
  const hsp_dftseg_args& a = *ap;
  if (int e = ds_check(a)) return e;
  const int pitch = (a.L + 3) & ~3;
  if (pitch > 40000) return HSP_EINVAL;
  const int cg = ds_group(pitch, a.C), ngrp = (a.C + cg - 1) / cg;
  const size_t lds_bytes = (size_t)cg * pitch * sizeof(float);
  static hsp_lds_flags flags;
  if (lds_bytes > 32 * 1024)
    if (int e = hsp_raise_lds_limit(reinterpret_cast<const void*>(dftseg_inv_kernel), 160 * 1024, flags)) return e;
  hipLaunchKernelGGL(dftseg_inv_kernel, dim3((unsigned)(a.B * ngrp)), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), a,
                     cg, pitch, ngrp);
  return (int)hipGetLastError();
}
