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

// A workgroup owns `cg` channel rows of one utterance over a CHUNK of S consecutive segments (of every phase): the input
// samples those segments read -- padded-time range [d s0 hop, d ((s0 + S - 1) hop + 128)) -- are one contiguous stretch
// of every row, and so are the outputs they produce, [d s0 hop, d (s0 + S) hop).  Its GEMM columns are the (channel,
// phase, segment) triples, flattened -- col = (ch * d + p) * S + s -- so that a 32-column MFMA block is full whatever
// the segment count is (L = 800 at dilation 5 has TWO segments per phase).  Any length: the chunk bounds the LDS.
struct DsCol {
  int ch, p, s;
  bool ok;
};
__device__ __forceinline__ DsCol ds_col(int col, int ncols, int d, int S) {
  DsCol c;
  c.ok = col < ncols;
  const int cc = c.ok ? col : ncols - 1;
  const int per = d * S;
  c.ch = cc / per;
  const int rem = cc - c.ch * per;
  c.p = rem / S;
  c.s = rem - c.p * S;
  return c;
}

struct DsGeom {
  int cg, S, pitch, ngrp, nchunk;
};

__global__ __launch_bounds__(256) void dftseg_fwd_kernel(const hsp_dftseg_args a, const DsGeom G) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [cg][pitch]: zero-padded input stretch of every row
  const int id = blockIdx.x;
  const int b = id / G.ngrp, c0 = (id % G.ngrp) * G.cg;
  const int ncg = min(G.cg, a.C - c0);                          // channel rows of this workgroup
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int hop = DS_N - (a.k - 1), d = a.dil, pitch = G.pitch;
  // A fragments: rows 32 wave + l32 of the forward matrix, taps 2 ks + half -- fetched once, used for every chunk of the
  // rows (a workgroup per chunk re-fetched these 64 KB for three column blocks of work: measured 1.4 x slower)
  float fa[64];
  {
    const float* frow = a.dft + (32 * wave + l32) * DS_N + half;
#pragma unroll
    for (int ks = 0; ks < 64; ++ks) fa[ks] = frow[2 * ks];
  }
  for (int sc = 0; sc < G.nchunk; ++sc) {
  const int s0 = sc * G.S, S = min(G.S, a.nseg - s0);           // this chunk's segments
  if (sc) __syncthreads();                                      // the previous chunk's stretch has been read
  // ---- stage the stretch: row[j] = xpad[t0 + j], xpad = the conv's zero-padded input.  Sixteen loads per thread in
  // flight (a load per loop iteration waits one L2 round trip each: that was most of the first version's time).
  const int t0 = d * s0 * hop - a.pad;                          // input index of row[0]
  const int len = d * ((S - 1) * hop + DS_N);
  const bool vec = ((a.L | (int)a.x_bs | (int)a.x_cs) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;   // uniform
  for (int ch = 0; ch < ncg; ++ch) {
    const float* xr = a.x + (int64_t)b * a.x_bs + (int64_t)(c0 + ch) * a.x_cs;
    float* row = lds + ch * pitch;
    if (vec) {
      // aligned 16-B loads of the input groups that overlap the stretch (its start t0 is odd: four scalar LDS writes
      // each); the padding on either side is zero-filled
      for (int j = tid; j < min(-t0, len); j += 256) row[j] = 0.0f;
      for (int j = max(a.L - t0, 0) + tid; j < len; j += 256) row[j] = 0.0f;
      const int g_lo = max(t0, 0) >> 2, g_hi = (min(t0 + len, a.L) + 3) >> 2;       // input 4-groups [g_lo, g_hi)
      for (int g0 = g_lo; g0 < g_hi; g0 += 256 * 16) {
        float4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float4*>(xr + 4 * min(g0 + tid + 256 * u, g_hi - 1));
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int gq = g0 + tid + 256 * u;
          if (gq < g_hi) {
            const int j = 4 * gq - t0;
            const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (j + i >= 0 && j + i < len) row[j + i] = e[i];
          }
        }
      }
    } else {
      for (int j0 = 0; j0 < len; j0 += 256 * 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = xr[min(max(t0 + j0 + tid + 256 * u, 0), a.L - 1)];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int j = j0 + tid + 256 * u, ti = t0 + j;
          if (j < len) row[j] = (ti >= 0 && ti < a.L) ? v[u] : 0.0f;
        }
      }
    }
  }
  __syncthreads();
  const int ncols = ncg * d * S;
  const int step = 2 * d;                                      // floats between the taps of consecutive k-steps
  for (int cb = 0; cb < ncols; cb += 32) {
    const DsCol q = ds_col(cb + l32, ncols, d, S);
    const float* bp = lds + q.ch * pitch + q.p + d * (q.s * hop + half);
    ds_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 64; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks], bp[ks * step], acc, 0, 0, 0);
    if (q.ok) {
      float* op = a.xf + (int64_t)(c0 + q.ch) * a.Np + (b * d + q.p) * a.nseg + s0 + q.s;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * wave + DS_ACC_ROW(r, half);        // 0..63: Re(bin), 64..127: Im(bin - 64) (64: Nyquist)
        op[(int64_t)(row & 63) * a.xf_bs + (int64_t)(row >> 6) * a.C * a.Np] = acc[r];
      }
    }
  }
  }   // chunks
}

__global__ __launch_bounds__(256) void dftseg_inv_kernel(const hsp_dftseg_args a, const DsGeom G, int nbuf) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [cg][pitch] output stretch of every row | B buffers
  const int id = blockIdx.x;
  const int b = id / G.ngrp, c0 = (id % G.ngrp) * G.cg;
  const int ncg = min(G.cg, a.C - c0);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int hop = DS_N - (a.k - 1), d = a.dil, pitch = G.pitch;
  // A fragments: rows (time samples of a segment) 32 wave + l32 of the inverse matrix, spectrum rows 2 ks + half
  float fa[64];
  {
    const float* frow = a.dft + (32 * wave + l32) * DS_N + half;
#pragma unroll
    for (int ks = 0; ks < 64; ++ks) fa[ks] = frow[2 * ks];
  }
  for (int sc = 0; sc < G.nchunk; ++sc) {
  const int s0 = sc * G.S, S = min(G.S, a.nseg - s0);
  if (sc) __syncthreads();                                      // the previous chunk's outputs have left the stretch
  const int tb = d * s0 * hop;                                  // output index of row[0] (a multiple of 4: S is even)
  const int tl = min(a.L - tb, d * S * hop);                    // outputs of this chunk
  const int ncols = ncg * d * S;
  // B operand: the 128 spectrum rows (row r: bin r & 63, part r >> 6) x 32 columns of a column block.  All four waves
  // multiply the SAME fragments (they differ in their rows of the inverse matrix), so the block is staged through LDS
  // once -- wave w fetches the rows 32 w .. 32 w + 31, one row pair per instruction, two 128-B runs each.
  float* const bbuf = lds + G.cg * pitch;                       // [nbuf][128][32]
  auto fetch = [&](int cb, float (&v)[16]) __attribute__((always_inline)) {
    const DsCol q = ds_col(cb + l32, ncols, d, S);
    const float* src = a.xf + (int64_t)(c0 + q.ch) * a.Np + (b * d + q.p) * a.nseg + s0 + q.s;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int r = 32 * wave + 2 * u + half;                   // spectrum row
      v[u] = src[(int64_t)(r & 63) * a.xf_bs + (int64_t)(r >> 6) * a.C * a.Np];
    }
  };
  auto stash = [&](int buf, const float (&v)[16]) __attribute__((always_inline)) {
    float* dst = bbuf + buf * (128 * 32) + l32;
#pragma unroll
    for (int u = 0; u < 16; ++u) dst[(32 * wave + 2 * u + half) * 32] = v[u];
  };
  auto consume = [&](int cb, int buf) __attribute__((always_inline)) {
    const float* bp = bbuf + buf * (128 * 32) + half * 32 + l32;
    ds_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 64; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks], bp[ks * 64], acc, 0, 0, 0);
    const DsCol q = ds_col(cb + l32, ncols, d, S);
    if (q.ok) {
      float* row = lds + q.ch * pitch;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = 32 * wave + DS_ACC_ROW(r, half);          // time inside the segment: the first hop are valid
        const int j = q.p + d * (q.s * hop + i);                // index inside the chunk's stretch
        if (i < hop && j < tl) row[j] = acc[r];
      }
    }
  };
  {
    float v[16];
    fetch(0, v);
    stash(0, v);
    __syncthreads();
    int buf = 0;
    for (int cb = 0; cb < ncols; cb += 32) {
      const bool more = cb + 32 < ncols;                        // workgroup-uniform
      if (more) fetch(cb + 32, v);                              // in flight under this block's MFMAs
      consume(cb, buf);
      if (nbuf == 1) __syncthreads();                           // one buffer: everyone has read it
      if (more) stash(nbuf == 1 ? 0 : buf ^ 1, v);
      __syncthreads();                                          // the next block is staged
      buf = nbuf == 1 ? 0 : buf ^ 1;
    }
  }
  // ---- epilogue of the conv this replaces: y = ((corr + bias + res) [+ y]) * post_scale, one coalesced pass
  const bool vec = ((a.L | (int)a.y_bs | (int)a.y_cs | (int)a.res_bs | (int)a.res_cs | tb) & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(a.y) | reinterpret_cast<uintptr_t>(a.res)) & 15) == 0;   // workgroup-uniform
  for (int ch = 0; ch < ncg; ++ch) {
    const int c = c0 + ch;
    const float bz = a.bias ? a.bias[c] : 0.0f;
    const float* row = lds + ch * pitch;
    float* yr = a.y + (int64_t)b * a.y_bs + (int64_t)c * a.y_cs + tb;
    const float* rr = a.res ? a.res + (int64_t)b * a.res_bs + (int64_t)c * a.res_cs + tb : nullptr;
    if (vec) {
      // eight 16-B groups per thread at a time: the residual / running-sum loads of all of them in flight together
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const int n4 = tl >> 2;                                   // (L and tb are multiples of 4: so is tl)
      for (int j0 = 0; j0 < n4; j0 += 256 * 8) {
        float4 r4[8], o4[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = min(j0 + tid + 256 * u, n4 - 1);
          r4[u] = rr ? *reinterpret_cast<const float4*>(rr + 4 * j) : z4;
          o4[u] = a.accumulate ? *reinterpret_cast<const float4*>(yr + 4 * j) : z4;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + tid + 256 * u;
          if (j < n4) {
            float4 v = *reinterpret_cast<const float4*>(row + 4 * j);
            v.x = (v.x + bz + r4[u].x + o4[u].x) * a.post_scale;
            v.y = (v.y + bz + r4[u].y + o4[u].y) * a.post_scale;
            v.z = (v.z + bz + r4[u].z + o4[u].z) * a.post_scale;
            v.w = (v.w + bz + r4[u].w + o4[u].w) * a.post_scale;
            *reinterpret_cast<float4*>(yr + 4 * j) = v;
          }
        }
      }
    } else {
      for (int t = tid; t < tl; t += 256) {
        float v = row[t] + bz;
        if (rr) v += rr[t];
        if (a.accumulate) v += yr[t];
        yr[t] = v * a.post_scale;
      }
    }
  }
  }   // chunks
}

int ds_check(const hsp_dftseg_args& a) {
  if (!a.xf || !a.dft || a.B <= 0 || a.C <= 0 || a.L <= 0 || a.k < 2 || a.k > 64 || a.dil < 1 || a.dil > 8) return HSP_EINVAL;
  const int hop = DS_N - (a.k - 1);
  const int per_phase = (a.L + a.dil - 1) / a.dil;
  if (a.nseg != (per_phase + hop - 1) / hop || a.pad < 0 || a.pad > (a.k - 1) * a.dil) return HSP_EINVAL;
  if ((int64_t)a.B * a.dil * a.nseg > a.Np || a.xf_bs < (int64_t)2 * a.C * a.Np) return HSP_EINVAL;
  return 0;
}

// Chunk geometry: the LARGEST even segment count whose stretch fits the LDS budget (the whole row at the Generator's
// lengths: one staging, no chunk barriers -- smaller chunks were measured 1.1-2 x slower), then as many channel rows as
// still fit.  Budget: 68 KB of input stretch (forward) / 64 KB of output stretch (inverse, next to one or two 16-KB B
// buffers): two workgroups per CU either way.
DsGeom ds_geom(const hsp_dftseg_args& a, bool inverse) {
  const int hop = DS_N - (a.k - 1), d = a.dil;
  const int budget = inverse ? 16384 : 17408;
  auto row_of = [&](int S) { return inverse ? ((d * S * hop + 3) & ~3) : d * ((S - 1) * hop + DS_N); };
  int S = a.nseg;
  if (row_of(S) > budget) {
    S = inverse ? budget / (d * hop) : (budget / d - DS_N) / hop + 1;
    S &= ~1;
    if (S < 2) S = 2;
  }
  DsGeom G;
  G.S = S;
  G.pitch = row_of(S);
  int cg = budget / G.pitch;
  cg = cg < 1 ? 1 : (cg > 16 ? 16 : cg);
  G.cg = cg > a.C ? a.C : cg;
  G.nchunk = (a.nseg + S - 1) / S;
  G.ngrp = (a.C + G.cg - 1) / G.cg;
  return G;
}
}  // namespace

extern "C" int hsp_dftseg_fwd_f32(const hsp_dftseg_args* ap, void* stream) {
  if (!ap || !ap->x) return HSP_EINVAL;
  const hsp_dftseg_args& a = *ap;
  if (int e = ds_check(a)) return e;
  const DsGeom G = ds_geom(a, false);
  const size_t lds_bytes = (size_t)G.cg * G.pitch * sizeof(float);
  const int64_t blocks = (int64_t)a.B * G.ngrp;
  if (lds_bytes > 160 * 1024 || blocks > 0x7fffffff) return HSP_EINVAL;
  static hsp_lds_flags flags;
  if (lds_bytes > 32 * 1024)
    if (int e = hsp_raise_lds_limit(reinterpret_cast<const void*>(dftseg_fwd_kernel), 160 * 1024, flags)) return e;
  hipLaunchKernelGGL(dftseg_fwd_kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), a, G);
  return (int)hipGetLastError();
}

extern "C" int hsp_dftseg_inv_f32(const hsp_dftseg_args* ap, void* stream) {
  if (!ap || !ap->y) return HSP_EINVAL;
  const hsp_dftseg_args& a = *ap;
  if (int e = ds_check(a)) return e;
  const DsGeom G = ds_geom(a, true);
  const int nbuf = ((size_t)G.cg * G.pitch + 2 * 128 * 32) * sizeof(float) <= 80 * 1024 ? 2 : 1;   // two workgroups per CU
  const size_t lds_bytes = ((size_t)G.cg * G.pitch + nbuf * 128 * 32) * sizeof(float);
  const int64_t blocks = (int64_t)a.B * G.ngrp;
  if (lds_bytes > 160 * 1024 || blocks > 0x7fffffff) return HSP_EINVAL;
  static hsp_lds_flags flags;
  if (lds_bytes > 32 * 1024)
    if (int e = hsp_raise_lds_limit(reinterpret_cast<const void*>(dftseg_inv_kernel), 160 * 1024, flags)) return e;
  hipLaunchKernelGGL(dftseg_inv_kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), a, G,
                     nbuf);
  return (int)hipGetLastError();
}
