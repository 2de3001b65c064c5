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
// Both transforms are products with constant matrices on v_mfma_f32_32x32x2_f32 (exact fp32), split ONCE radix-2: the
// 128-point real DFT of a segment is E[k] + W^k O[k] with E, O the 64-point real DFTs of its even / odd samples, and
// (E[k], O[k]) also give bin 64 - k = conj(E[k] - W^k O[k]).  So the MFMA work is two 64 x 64 real products per segment
// (half of the 128 x 128 one; the first version of this file did that: 1.3-1.5 x slower transforms) and the recombination
// is 8 VALU operations per bin on the accumulators -- the rows of the 64-point matrix are ordered so that a lane holds
// Re and Im of the SAME bin of both E and O (accumulator register r and r + 8).  The inverse runs the same split
// backwards: E^[k] = X[k] + conj(X[64 - k]), O^[k] = (X[k] - conj(X[64 - k])) W^-k, formed by the threads that stage the
// spectrum in LDS, then even / odd output samples from the 64-point inverse matrix.  A wave keeps its 32 rows of the
// 64 x 64 matrix in registers (32 VGPRs) for the whole launch.  hsp_dftseg_tables_f32 fills both tables.
//   forward: the input rows of a channel group are staged in LDS once (zero padding applied there), B fragments are
//            strided LDS reads (lane = segment), the 128 spectrum rows of a segment block leave as 128-B runs;
//   inverse: the spectrum rows arrive as coalesced global loads, are recombined and staged through LDS as the B operand
//            of all four waves, the time samples are scattered into an LDS image of the output rows (all phases), then
//            bias / residual / running sum are applied in one coalesced pass (the epilogue of the conv this replaces:
//            hsp_conv1d_args bias, res, accumulate, post_scale).
//
// Both kernels are PERSISTENT (two 4-wave workgroups per CU, each walking its share of the items): the table fetch and
// the launch ramp are paid once per workgroup, not once per row, and the forward kernel loads the NEXT item's input into
// registers before the MFMAs of this one, so the HBM round trip of the staging is off the critical path.  The spectrum
// stores of a column block are issued one per k-step under the next block's MFMAs: as a burst behind the loop they
// cost 2500 cycles per block (every CU bursts at the same time, and a wave that cannot issue its stores cannot issue
// MFMAs either).  Tried and measured slower in round 4: waves split by role (four MFMA + four memory waves per CU over
// two LDS stretches: one MFMA wave per SIMD ran at half the MFMA rate and the transform stayed at ~2.9 TB/s), smaller
// LDS budgets for more workgroups per CU, non-temporal loads / stores, an odd LDS lane stride (no bank-conflict gain).
// The fused activation (below) was also tried with the three stages of consecutive 112-sample segments in one step (one
// LDS round trip per step instead of three per segment): 40 % MORE VALU instructions per output (56-59 of 64 lanes busy,
// per-step bookkeeping) and the forward launches went from 5.2 to 7.3 ms per step -- the phase is bound by VALU issue,
// which the fp32 MFMA shares, not by LDS latency.  A start delay for the second workgroup of a CU (to put its
// activation phase under the first one's MFMAs) changed nothing for the same reason.
#include "hsp_device.h"
#include <algorithm>
#include <cmath>

namespace {
typedef float ds_f32x16 __attribute__((ext_vector_type(16)));
typedef float ds_f32x4 __attribute__((ext_vector_type(4)));
constexpr int DS_N = 128, DS_H = 64;
constexpr int DS_MEM = 256;                                     // threads of a workgroup
#define DS_ACC_ROW(r, half) (((r) & 3) + 8 * ((r) >> 2) + 4 * (half))

// An item = `cg` channel rows of one utterance over a CHUNK of S consecutive segments (of every phase): the input
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
  int cg, S, pitch, ngrp, nchunk, bufsz;                   // bufsz: floats of one stretch ([cg][pitch], 16-B multiple)
};
struct DsItem {
  int b, c0, ncg, s0, S;
};
__device__ __forceinline__ DsItem ds_item(int item, const hsp_dftseg_args& a, const DsGeom& G) {
  DsItem I;
  const int id = item / G.nchunk, sc = item - id * G.nchunk;
  I.b = id / G.ngrp;
  I.c0 = (id - I.b * G.ngrp) * G.cg;
  I.ncg = min(G.cg, a.C - I.c0);
  I.s0 = sc * G.S;
  I.S = min(G.S, a.nseg - I.s0);
  return I;
}
// Role barrier: LDS traffic of this wave has landed; global loads / stores stay in flight across it (__syncthreads
// drains vmcnt: the MFMA waves would wait for their spectrum stores, the inverse's for the next block's fetch).
__device__ __forceinline__ void ds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// B operands come out of LDS through inline-asm reads in a ring of three register groups, two groups ahead of the MFMAs
// that consume them: ONE MFMA wave per SIMD has nobody to hide its LDS latency behind, and the compiler's own schedule
// recycled two registers (read, one MFMA, wait: 1.5-2 x the MFMA time).  A group is valid behind ds_wait<N>, which
// re-defines its registers so that the consumers carry a data dependency on the wait (hsp_conv1d_mfma_kernel.h).
__device__ __forceinline__ unsigned ds_lds_addr(const float* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) float*)p;
}
__device__ __forceinline__ void ds_rd(float& dst, unsigned addr) { asm volatile("ds_read_b32 %0, %1" : "=v"(dst) : "v"(addr)); }
template <int OFF>
__device__ __forceinline__ void ds_rd_at(float& dst, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read immediate offsets are 16 bits");
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void ds_wait(float (&r)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(r[i]));
}

// ---------------------------------------------------------------------------------------------------------- forward
// Staging of an item: row[j + sh] = xpad[t0 + j], xpad = the conv's zero-padded input, as a flat list of 16-B input
// groups -- element e = ch * ng + gi is group g_lo + gi of row ch -- sixteen per thread and batch.
struct DsStage {
  int t0, len, sh, g_lo, ng, tot;
};
__device__ __forceinline__ DsStage ds_stage_of(const hsp_dftseg_args& a, const DsItem& I) {
  const int hop = DS_N - (a.k - 1), d = a.dil;
  DsStage s;
  s.t0 = d * I.s0 * hop - a.pad;                                // input index of the stretch's sample 0
  s.len = d * ((I.S - 1) * hop + DS_N);
  s.sh = 4 + (s.t0 & 3);                                        // input 4-groups land 16-B aligned in LDS
  s.g_lo = max(s.t0, 0) >> 2;
  s.ng = ((min(s.t0 + s.len, a.L) + 3) >> 2) - s.g_lo;          // input 4-groups [g_lo, g_lo + ng) overlap the stretch
  s.tot = I.ncg * s.ng;
  return s;
}
// e / ng for 0 <= e < 2^22 (a per-lane integer division is ~25 VALU instructions, and the staging has 32 of them per
// item): the float quotient is off by at most one
__device__ __forceinline__ int ds_div(int e, int ng, float inv_ng) {
  int q = (int)((float)e * inv_ng);
  const int r = e - q * ng;
  q += r >= ng ? 1 : 0;
  q -= r < 0 ? 1 : 0;
  return q;
}
// col -> (channel row, phase, segment) with float-reciprocal divisions (columns of an item: < 2^13)
__device__ __forceinline__ DsCol ds_col_f(int col, int ncols, int d, int S, float inv_per, float inv_S) {
  DsCol c;
  c.ok = col < ncols;
  const int cc = c.ok ? col : ncols - 1;
  const int per = d * S;
  c.ch = ds_div(cc, per, inv_per);
  const int rem = cc - c.ch * per;
  c.p = d == 1 ? 0 : ds_div(rem, S, inv_S);
  c.s = rem - c.p * S;
  return c;
}

__device__ __forceinline__ void ds_stage_load(const hsp_dftseg_args& a, const DsItem& I, const DsStage& s, int e0, int t,
                                              ds_f32x4 (&v)[16]) {
  const float* xb = a.x + (int64_t)I.b * a.x_bs + (int64_t)I.c0 * a.x_cs;
  const float inv_ng = 1.0f / (float)s.ng;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int e = min(e0 + t + DS_MEM * u, s.tot - 1), ch = I.ncg == 1 ? 0 : ds_div(e, s.ng, inv_ng), gi = e - ch * s.ng;
    v[u] = *reinterpret_cast<const ds_f32x4*>(xb + (int64_t)ch * a.x_cs + 4 * (s.g_lo + gi));
  }
}
__device__ __forceinline__ void ds_stage_write(const DsGeom& G, const DsStage& s, int e0, int t, float* buf, const ds_f32x4 (&v)[16]) {
  const float inv_ng = 1.0f / (float)s.ng;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int e = e0 + t + DS_MEM * u, ch = s.tot == s.ng ? 0 : ds_div(min(e, s.tot - 1), s.ng, inv_ng), gi = e - ch * s.ng;
    // a group may hang over either end of the stretch by up to three samples: the rows have that slack
    if (e < s.tot) *reinterpret_cast<ds_f32x4*>(buf + ch * G.pitch + s.sh - s.t0 + 4 * (s.g_lo + gi)) = v[u];
  }
}

// ---- the anti-aliased SnakeBeta in front of the conv (Activation1d, hierspeechpp_speechsynthesizer.py:340-392: every
// AMP conv reads act(x)), fused into the staging: the stretch receives act(x) and the activation's own launch -- one read
// and one write of the tensor -- goes away.  The arithmetic is act1d_seg_kernel's (hsp_pointwise.hip), on segments of
// DA_SEG outputs: a WAVE owns a segment and its own LDS slice (raw window | 2x-rate snake signal), phases separated by
// compiler fences only (a wave's LDS instructions retire in order).
//   raw[j] = x[clamp(p0 - 8 + j)]                 j in [0, 256): one aligned float4 per lane
//   a2[j]  = a[clamp(2 p0 - 5 + j, 0, 2 L - 1)]   a = snake(2 * upsampled x)
//   y[p0 + s] = sum_k hd[k] a2[2 s + k]
constexpr int DA_SEG = 240, DA_RAW = 256, DA_A2 = 496, DA_SLICE = DA_RAW + DA_A2;
typedef float ds_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void da_fence() { asm volatile("" ::: "memory"); }
struct DaItem {
  int pa, pb, pa4, nsg, nsegs;                                  // outputs [pa, pb) of every row, segments of DA_SEG from pa4
};
__device__ __forceinline__ DaItem da_item(const hsp_dftseg_args& a, const DsItem& I, const DsStage& s) {
  DaItem A;
  A.pa = max(s.t0, 0);
  A.pb = min(s.t0 + s.len, a.L);
  A.pa4 = A.pa & ~3;
  A.nsg = (A.pb - A.pa4 + DA_SEG - 1) / DA_SEG;
  A.nsegs = I.ncg * A.nsg;
  return A;
}
// the raw window of segment sI (clamped to the item's last one: always a legal address, no branch around the load)
__device__ __forceinline__ ds_f32x4 da_load(const hsp_dftseg_args& a, const DsItem& I, const DaItem& A, int sI, int lane) {
  const int q = min(sI, A.nsegs - 1), ch = q / A.nsg, p0 = A.pa4 + DA_SEG * (q - ch * A.nsg);
  const float* xr = a.x + (int64_t)I.b * a.x_bs + (int64_t)(I.c0 + ch) * a.x_cs;
  return *reinterpret_cast<const ds_f32x4*>(xr + hsp_clampi(p0 - 8 + 4 * lane, 0, a.L - 4));
}
// one segment: raw window in registers -> act -> the stretch (row[j + sh] = act(x)[t0 + j] for t0 + j in [pa, pb))
__device__ __forceinline__ void da_segment(const hsp_dftseg_args& a, const DsGeom& G, const DsItem& I, const DsStage& s,
                                           const DaItem& A, int sI, int lane, ds_f32x4 rv, float* slice, float* buf,
                                           const float* flt, float add = 0.0f) {
  const int ch = sI / A.nsg, p0 = A.pa4 + DA_SEG * (sI - ch * A.nsg);
  const int n_out = min(DA_SEG, a.L - p0);                      // a multiple of 4
  const int c = I.c0 + ch;
  const float kf = a.act_alpha_exp[c] * 0.318309886183790672f, kb = 0.5f * a.act_beta_inv[c];
  float* raw = slice;
  float* a2 = slice + DA_RAW;
  // the taps come out of LDS for every segment (24 registers held across the MFMA loop cost the kernel its occupancy)
  ds_f32x2 hu[6], hd[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    hu[i] = *reinterpret_cast<const ds_f32x2*>(flt + 2 * i);
    hd[i] = *reinterpret_cast<const ds_f32x2*>(flt + 12 + 2 * i);
  }
  // ---- phase A: registers -> LDS (replicate padding: the clamped address was 0 or L - 4)
  {
    const int idx = p0 - 8 + 4 * lane;
    ds_f32x4 t = rv + add;                                      // (the pair kernel: the first conv's bias)
    t = idx < 0 ? ds_f32x4{t.x, t.x, t.x, t.x} : t;
    t = idx >= a.L ? ds_f32x4{t.w, t.w, t.w, t.w} : t;
    *reinterpret_cast<ds_f32x4*>(raw + 4 * lane) = t;
  }
  da_fence();
  // ---- phase B: lane pair pi owns a[2q + 1 .. 2q + 4], q = p0 - 3 + 2 pi; its inputs x[q - 2 .. q + 4] = raw[2 pi + 3 ..]
  const int npairs = (2 * n_out + 13) >> 2;                     // slots [0, 2 n_out + 10)
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int pi = lane + 64 * it;
    if (pi < npairs) {
      const ds_f32x2* rp = reinterpret_cast<const ds_f32x2*>(raw + 2 * pi + 2);
      const ds_f32x2 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
      const float xv[7] = {r0.y, r1.x, r1.y, r2.x, r2.y, r3.x, r3.y};
      ds_f32x2 u0 = {0.0f, 0.0f}, u1 = {0.0f, 0.0f};
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        u0 = __builtin_elementwise_fma(ds_f32x2{xv[i], xv[i]}, hu[i], u0);
        u1 = __builtin_elementwise_fma(ds_f32x2{xv[i + 1], xv[i + 1]}, hu[i], u1);
      }
      *reinterpret_cast<ds_f32x4*>(a2 + 4 * pi) =
          ds_f32x4{hsp_snake_hw(u0.x, kf, kb), hsp_snake_hw(u0.y, kf, kb), hsp_snake_hw(u1.x, kf, kb), hsp_snake_hw(u1.y, kf, kb)};
    }
  }
  da_fence();
  // replicate padding of the 2x-rate signal: a[-5 .. -1] = a[0], a[2L .. 2L + 4] = a[2L - 1]
  if (p0 == 0 || p0 + n_out == a.L) {
    if (p0 == 0 && lane < 5) a2[lane] = a2[5];
    if (p0 + n_out == a.L && lane >= 8 && lane < 13) a2[2 * n_out + lane - 3] = a2[2 * n_out + 4];
    da_fence();
  }
  // ---- phase C: two outputs per lane, into the stretch
  float* row = buf + ch * G.pitch + s.sh - s.t0;                // row[p] = sample p of the padded input
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int l2 = lane + 64 * it;
    if (2 * l2 < n_out) {
      const ds_f32x4* ap = reinterpret_cast<const ds_f32x4*>(a2 + 4 * l2);
      const ds_f32x4 q0 = ap[0], q1 = ap[1], q2 = ap[2];
      const ds_f32x2 q3 = *reinterpret_cast<const ds_f32x2*>(a2 + 4 * l2 + 12);
      const ds_f32x2 aw[7] = {{q0.x, q0.y}, {q0.z, q0.w}, {q1.x, q1.y}, {q1.z, q1.w}, {q2.x, q2.y}, {q2.z, q2.w}, q3};
      ds_f32x2 s0 = {0.0f, 0.0f}, s1 = {0.0f, 0.0f};
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        s0 = __builtin_elementwise_fma(aw[k], hd[k], s0);
        s1 = __builtin_elementwise_fma(aw[k + 1], hd[k], s1);
      }
      const int p = p0 + 2 * l2;
      if (p >= A.pa && p < A.pb) row[p] = s0.x + s0.y;
      if (p + 1 >= A.pa && p + 1 < A.pb) row[p + 1] = s1.x + s1.y;
    }
  }
  da_fence();                                                   // raw / a2 are rewritten by the next segment
}

template <bool ACT>
__global__ __launch_bounds__(DS_MEM) __attribute__((amdgpu_waves_per_eu(2, 2))) void dftseg_fwd_kernel(const hsp_dftseg_args a, const DsGeom G) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [cg][pitch]: zero-padded input stretch of every row
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int hop = DS_N - (a.k - 1), d = a.dil, pitch = G.pitch;
  const int nitems = a.B * G.ngrp * G.nchunk;
  const int nmine = (nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // >= 1: the grid is <= nitems
  // Two waves share a 32-column block: wave half `wh` owns the bins 16 wh .. 16 wh + 15 of E and O (rows 32 wh + l32 of
  // the 64 x 64 table: Re of those bins in tile rows 0..15, Im in 16..31), i.e. the output bins k and 64 - k; the wave
  // pairs (0, 1) and (2, 3) work on alternate column blocks.  A fragments: taps 2 ks + half, fetched once per launch.
  // Accumulator register r (and r + 8) holds bin 16 wh + DS_ACC_ROW(r, half); its twiddle W^k comes out of LDS.
  const int wh = wave & 1, pairw = wave >> 1;
  float fa[32];
  {
    const float* frow = a.dft + (32 * wh + l32) * DS_H + half;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) fa[ks] = frow[2 * ks];
  }
  float* const twl = lds + G.bufsz;                             // (cos | sin)(2 pi k / 128), k < 32, behind the stretch
  if (tid < 64) twl[tid] = a.dft[DS_H * DS_H + tid];
  // the previous column block's outputs of this lane (bin register r: X[kb] Re, Im, X[64 - kb] Re, Im) and where they go:
  // 32-bit byte offsets into the spectrum (the host checks that it is below 4 GB), lane part in two registers -- bins
  // kb = lanek + c_r count up from lb1, bins 64 - kb down from lb2 -- and the register's part c_r xf_bs uniform (the
  // 32 full 64-bit offsets, hoisted out of the item loop by the compiler, cost 64 registers and with them the second
  // workgroup of the CU).
  float ov[32];
  bool pok = false;
  unsigned lb1 = 0, lb2 = 0, lbz = 0;
  const unsigned rowb = 4u * (unsigned)a.xf_bs, imb = 4u * (unsigned)(a.C * a.Np);   // bytes of a bin plane / to part 1 (Im)
  const int lanek = 16 * wh + 4 * half;
  auto put = [&](int i) __attribute__((always_inline)) {
    const int r = i >> 2, w = i & 3;
    const unsigned cr = (unsigned)((r & 3) + 8 * (r >> 2)) * rowb;   // uniform
    unsigned off = w < 2 ? lb1 + cr : lb2 - cr;
    if (r == 0 && w >= 2) off = lanek == 0 ? lbz : off;         // bin 0's lane: X[32] where the others put bin 64 - kb
    if (w & 1) off += imb;
    if (pok) *reinterpret_cast<float*>(reinterpret_cast<char*>(a.xf) + off) = ov[i];
  };
  const bool vec = ((a.L | (int)a.x_bs | (int)a.x_cs) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;   // uniform
  // Persistent workgroups, two per CU.  The first sixteen 16-B groups per thread of the NEXT item (all of it at the
  // Generator's shapes) are loaded into registers before this item's MFMAs and written to LDS behind them: the HBM round
  // trip of the staging is off the critical path.
  constexpr bool act = ACT;                                     // (the host asks for it on 16-B addressable rows only)
  float* const flt = lds + G.bufsz + 64;                        // upsampling taps (x2 gain folded in, reversed) | downsampling taps
  float* const slice = flt + 32 + wave * DA_SLICE;              // this wave's activation scratch
  if (act && tid < 24) {
    const int i = tid >> 1;                                     // hu[i] = 2 (filt[10 - 2 i], filt[11 - 2 i]), hd[i] = (filt[12 + 2 i], filt[13 + 2 i])
    flt[tid] = tid < 12 ? 2.0f * a.act_filt[10 - 2 * i + (tid & 1)] : a.act_filt[tid];
  }
  if (act) __syncthreads();                                     // the first item's staging reads the taps
  DsItem I = ds_item(blockIdx.x, a, G);
  DsStage sg = ds_stage_of(a, I);
  DaItem da = da_item(a, I, sg);
  ds_f32x4 pv[16];
  auto prefetch = [&]() __attribute__((always_inline)) {
    if (act) {
#pragma unroll
      for (int i = 0; i < 16; ++i) pv[i] = da_load(a, I, da, wave + 4 * i, lane);   // this wave's first sixteen segments
    } else if (vec) {
      ds_stage_load(a, I, sg, 0, tid, pv);
    }
  };
  prefetch();
  for (int it = 0; it < nmine; ++it) {
    if (it) ds_barrier();                                       // the previous item's stretch has been read
    // ---- finish the staging of item `it`
    if (act) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (wave + 4 * i < da.nsegs) da_segment(a, G, I, sg, da, wave + 4 * i, lane, pv[i], slice, lds, flt);
      for (int sI = wave + 64; sI < da.nsegs; sI += 4)          // beyond the prefetch depth
        da_segment(a, G, I, sg, da, sI, lane, da_load(a, I, da, sI, lane), slice, lds, flt);
      for (int ch = 0; ch < I.ncg; ++ch) {                      // the conv's zero padding on either side
        float* row = lds + ch * pitch + sg.sh;
        for (int j = tid; j < min(-sg.t0, sg.len); j += DS_MEM) row[j] = 0.0f;
        for (int j = max(a.L - sg.t0, 0) + tid; j < sg.len; j += DS_MEM) row[j] = 0.0f;
      }
    } else if (vec) {
      ds_stage_write(G, sg, 0, tid, lds, pv);
      for (int e0 = DS_MEM * 16; e0 < sg.tot; e0 += DS_MEM * 16) {
        ds_f32x4 v[16];
        ds_stage_load(a, I, sg, e0, tid, v);
        ds_stage_write(G, sg, e0, tid, lds, v);
      }
      for (int ch = 0; ch < I.ncg; ++ch) {                      // the conv's zero padding on either side
        float* row = lds + ch * pitch + sg.sh;
        for (int j = tid; j < min(-sg.t0, sg.len); j += DS_MEM) row[j] = 0.0f;
        for (int j = max(a.L - sg.t0, 0) + tid; j < sg.len; j += DS_MEM) row[j] = 0.0f;
      }
    } else {
      for (int ch = 0; ch < I.ncg; ++ch) {
        const float* xr = a.x + (int64_t)I.b * a.x_bs + (int64_t)(I.c0 + ch) * a.x_cs;
        float* row = lds + ch * pitch + sg.sh;
        for (int j0 = 0; j0 < sg.len; j0 += DS_MEM * 16) {
          float v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = xr[min(max(sg.t0 + j0 + tid + DS_MEM * u, 0), a.L - 1)];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const int j = j0 + tid + DS_MEM * u, ti = sg.t0 + j;
            if (j < sg.len) row[j] = (ti >= 0 && ti < a.L) ? v[u] : 0.0f;
          }
        }
      }
    }
    ds_barrier();
    const DsItem Ic = I;
    const int shc = sg.sh;
    if (it + 1 < nmine) {                                       // the next item's loads go out now
      I = ds_item(blockIdx.x + (it + 1) * gridDim.x, a, G);
      sg = ds_stage_of(a, I);
      da = da_item(a, I, sg);
      prefetch();
    }
    // ---- the transform of item `it`
    const int S = Ic.S, ncols = Ic.ncg * d * S;
    const float ipc = 1.0f / (float)(d * S), isc = 1.0f / (float)S;   // (round 6: column decode by float reciprocals)
    const float* buf = lds + shc;                               // sample 0 of row 0
    const int step = 4 * d;                                     // floats between the even (odd) taps of consecutive k-steps
    for (int cb = 32 * pairw; cb < ncols; cb += 64) {
      const DsCol q = ds_col_f(cb + l32, ncols, d, S, ipc, isc);
      const float* bp = buf + q.ch * pitch + q.p + d * (q.s * hop + 2 * half);
      ds_f32x16 ae, ao;                                         // E = F64 x[even], O = F64 x[odd]
#pragma unroll
      for (int r = 0; r < 16; ++r) ae[r] = ao[r] = 0.0f;
      // groups of four k-steps: (even, odd) taps 4 (4 g + j) + 2 half (+ 1) of the segment, d floats apart
      float rb[2][8];
      unsigned pe = ds_lds_addr(bp), po = pe + 4u * d;
      const unsigned st = 4u * step;
      auto issue = [&](float (&r)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ds_rd(r[2 * j], pe);
          ds_rd(r[2 * j + 1], po);
          pe += st;
          po += st;
        }
      };
      issue(rb[0]);
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        ds_wait<0>(rb[g & 1]);
        if (g + 1 < 8) issue(rb[(g + 1) & 1]);                  // in flight under this group's eight MFMAs
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ae = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[4 * g + j], rb[g & 1][2 * j], ae, 0, 0, 0);
          ao = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[4 * g + j], rb[g & 1][2 * j + 1], ao, 0, 0, 0);
          put(4 * g + j);                                       // one of the previous block's 32 stores
        }
      }
      // X[kb] = E + W^kb O, X[64 - kb] = conj(E - W^kb O) for the eight bins of this lane; bin 0's lane holds (E[0],
      // E[32]) and (O[0], O[32]), all real: DC = E0 + O0 and Nyquist = E0 - O0 share bin slot 0, and X[32] = E[32] -
      // i O[32] goes where the other lanes put bin 64 - kb.  The values wait in registers: their stores are issued one
      // per k-step under the NEXT block's MFMAs (as a burst behind the loop they cost 2500 cycles per block -- every
      // CU bursts at the same time, and a wave that cannot issue its stores cannot issue its MFMAs either).
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float er = ae[r], ei = ae[r + 8], orr = ao[r], oi = ao[r + 8];
        const float wc = twl[lanek + (r & 3) + 8 * (r >> 2)], ws = twl[32 + lanek + (r & 3) + 8 * (r >> 2)];
        const float tr = wc * orr + ws * oi, ti = wc * oi - ws * orr;
        const bool z = r == 0 && wh == 0 && half == 0;
        const bool z3 = z && a.prod3;                           // slot 0 for the three-product form: (-E0, -O0), hsp.h prod3
        ov[4 * r + 0] = z3 ? -er : er + tr;
        ov[4 * r + 1] = z ? (a.prod3 ? -tr : er - tr) : ei + ti;
        ov[4 * r + 2] = z ? ei : er - tr;
        ov[4 * r + 3] = z ? -oi : ti - ei;
      }
      {
        const unsigned colb = 4u * (unsigned)((Ic.c0 + q.ch) * a.Np + (Ic.b * d + q.p) * a.nseg + Ic.s0 + q.s);
        pok = q.ok;
        lb1 = colb + (unsigned)lanek * rowb;
        lb2 = colb + (unsigned)(64 - lanek) * rowb;
        lbz = colb + 32u * rowb;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 32; ++i) put(i);                          // the last block's outputs
}

// ---------------------------------------------------------------------------------------------------------- inverse
// (round 6)  The index arithmetic of this kernel used to cost 14.7 vector instructions per MFMA (profiles/r05_dftseg_pmc.txt)
// -- and the fp32 MFMA shares its issue with the VALU, so every one of them was paid in matrix time.  What was there and
// where it went:
//   * sixteen 64-bit addresses per thread and block for the spectrum loads  ->  four wave-uniform 64-bit bases (the
//     planes of this wave's first bin, Re / Im, X[k] / X[64 - k]: scalar registers) + a 32-bit byte offset per load = the
//     column's offset + a per-thread constant (eight v_add per block; the host bounds the spectrum of a launch to 4 GB);
//   * the column decode col -> (channel row, phase, segment), two integer divisions, done twice per block (fetch, scatter)
//     ->  float-reciprocal divisions (ds_div), once: the fetch hands its decode to the scatter two steps later with the
//     register set;
//   * a bounds test per scattered sample  ->  none: a row's stretch holds d S hop floats, every index the scatter can form
//     is inside it, and the epilogue reads [0, tl) only (the samples past the tensor's end land in the row's own slack);
//     the `i < hop` test is compile-time true except for the last four accumulator registers of the odd wave halves;
//   * an epilogue that divided every strip index by the strips per row and multiplied 64-bit strides per element  ->
//     a wave owns UNITS of 64 U consecutive vectors of one row: row base addresses on the scalar unit, one 32-bit byte
//     offset per vector.
// Same arithmetic in the same order as before: results are bit-identical to round 5's kernel.

// Epilogue of the conv this replaces, y = ((corr + bias + res) [+ y]) * post_scale, out of the LDS stretch of an item.
// A UNIT = 64 U consecutive W-float vectors of one row; wave w of nw takes units w, w + nw, ...: the row's base addresses
// (y, res) are wave-uniform 64-bit values, a lane adds one 32-bit byte offset per vector.
// KEEP: the finished samples also go back into the stretch (the pair kernel's pass-through form: the activation of the
// next conv reads them there)
template <int W, int U, bool RES, bool ACC, bool KEEP = false>
__device__ __forceinline__ void ds_inv_epilogue_t(const hsp_dftseg_args& a, const DsGeom& G, const DsItem& I, const float* buf,
                                                  int w, int nw, int lane, int tb, int tl) {
  typedef float vec_t __attribute__((ext_vector_type(W)));
  const int nel = tl / W;                                       // vectors per row
  const int R = (nel + 64 * U - 1) / (64 * U);                  // units per row
  const float ps = a.post_scale;
  int ch = 0, rr = w;
  while (rr >= R) rr -= R, ++ch;                                // (uniform)
  while (ch < I.ncg) {
    const int c = I.c0 + ch;
    const float bz = a.bias ? a.bias[c] : 0.0f;
    char* const yb = reinterpret_cast<char*>(a.y + (int64_t)I.b * a.y_bs + (int64_t)c * a.y_cs + tb);
    const char* const rb = RES ? reinterpret_cast<const char*>(a.res + (int64_t)I.b * a.res_bs + (int64_t)c * a.res_cs + tb) : nullptr;
    const float* const row = buf + ch * G.pitch;
    const int j0 = rr * (64 * U) + lane;
    vec_t r4[U], o4[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned off = (unsigned)(min(j0 + 64 * u, nel - 1) * (W * 4));
      if constexpr (RES) r4[u] = *reinterpret_cast<const vec_t*>(rb + off);
      if constexpr (ACC) o4[u] = *reinterpret_cast<const vec_t*>(yb + off);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + 64 * u;
      if (j < nel) {
        // ((corr + bias + res) + y) * post_scale in this order; a term that is not asked for is skipped (x + 0 and
        // x * 1 are x: same values as the general form)
        vec_t v = *reinterpret_cast<const vec_t*>(row + W * j) + bz;
        if constexpr (RES) v += r4[u];
        if constexpr (ACC) v += o4[u];
        if (ps != 1.0f) v *= ps;
        *reinterpret_cast<vec_t*>(yb + (unsigned)(j * (W * 4))) = v;
        if constexpr (KEEP) *reinterpret_cast<vec_t*>(const_cast<float*>(row) + W * j) = v;
      }
    }
    rr += nw;
    while (rr >= R) rr -= R, ++ch;
  }
}
template <int W, int U>
__device__ __forceinline__ void ds_inv_epilogue(const hsp_dftseg_args& a, const DsGeom& G, const DsItem& I, const float* buf,
                                                int w, int nw, int lane, int tb, int tl) {
  const bool has_res = a.res != nullptr, acc = a.accumulate != 0;   // (uniform: one of four instantiations runs)
  if (has_res && acc) ds_inv_epilogue_t<W, U, true, true>(a, G, I, buf, w, nw, lane, tb, tl);
  else if (has_res) ds_inv_epilogue_t<W, U, true, false>(a, G, I, buf, w, nw, lane, tb, tl);
  else if (acc) ds_inv_epilogue_t<W, U, false, true>(a, G, I, buf, w, nw, lane, tb, tl);
  else ds_inv_epilogue_t<W, U, false, false>(a, G, I, buf, w, nw, lane, tb, tl);
}

// The spectrum side of a thread: it recombines the bins kf(u) = 8 w + 4 half + u, u < 4, of one column, i.e. it loads
// X[kf] and X[64 - kf] (bin 0's thread: (DC | Nyquist) and X[32]), Re and Im.  Byte offset of a load = column offset +
// c1[u] (c2[u]) from the uniform base b1 (b2), + `imb` for the Im part.
struct DsInvSrc {
  const char *b1, *b1i, *b2, *b2i;                              // wave-uniform
  unsigned c1[4], c2[4];
};
__device__ __forceinline__ DsInvSrc ds_inv_src(const float* xf, int64_t xf_bs, int C, int Np, int w, int half) {
  DsInvSrc s;
  const unsigned rowb = 4u * (unsigned)xf_bs, imb = 4u * (unsigned)(C * Np);
  const int kb = 8 * w;                                         // first bin of this wave
  const int B2 = w == 0 ? 32 : 57 - kb;                         // 64 - kf = B2 + (m2 - 4 half - u), m2 - 4 half - u >= 0
  const int m2 = w == 0 ? 32 : 7;
  s.b1 = reinterpret_cast<const char*>(xf) + (size_t)kb * rowb;
  s.b1i = s.b1 + imb;
  s.b2 = reinterpret_cast<const char*>(xf) + (size_t)B2 * rowb;
  s.b2i = s.b2 + imb;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    s.c1[u] = (unsigned)(4 * half + u) * rowb;
    s.c2[u] = (unsigned)(m2 - 4 * half - u) * rowb;
  }
  if (w == 0 && half == 0) s.c2[0] = 0;                         // bin 0's thread: X[32] where the others read X[64 - kf]
  return s;
}
__device__ __forceinline__ void ds_inv_fetch(const DsInvSrc& s, unsigned colb, float (&v)[16]) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const unsigned o1 = colb + s.c1[u], o2 = colb + s.c2[u];
    v[4 * u + 0] = *reinterpret_cast<const float*>(s.b1 + o1);
    v[4 * u + 1] = *reinterpret_cast<const float*>(s.b1i + o1);
    v[4 * u + 2] = *reinterpret_cast<const float*>(s.b2 + o2);
    v[4 * u + 3] = *reinterpret_cast<const float*>(s.b2i + o2);
  }
}
// E^ = X[k] + conj(X[64 - k]), O^ = (X[k] - conj(X[64 - k])) W^-k into the operand buffer [E^ | O^][64 slots][32 columns]
// (the factor 1/2 is in the table); dst = the buffer + this lane's column
__device__ __forceinline__ void ds_inv_stash(float* dst, int kf0, const float (&v)[16], const float (&twc)[4], const float (&tws)[4]) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int kf = kf0 + u;
    const float xr = v[4 * u], xi = v[4 * u + 1], yr = v[4 * u + 2], yi = v[4 * u + 3];
    const float dr = xr - yr, di = xi + yi;                     // X[k] - conj(X[64 - k])
    const bool z = u == 0 && kf == 0;                           // (DC, Nyquist, Re X[32], Im X[32]): E^0 E^32 O^0 O^32 real
    dst[kf * 32] = z ? xr + xi : xr + yr;                       // Re E^[k]
    dst[(32 + kf) * 32] = z ? 2.0f * yr : xi - yi;              // Im E^[k]             (slot 32: E^[32] = 2 Re X[32])
    dst[(64 + kf) * 32] = z ? xr - xi : dr * twc[u] - di * tws[u];       // Re O^[k]
    dst[(96 + kf) * 32] = z ? -2.0f * yi : dr * tws[u] + di * twc[u];    // Im O^[k]  (slot 32: O^[32] = -2 Im X[32])
  }
}
// Scatter of a block's time samples into the LDS image of the output rows: accumulator register r of wave (eo, wh) is
// sample i = 2 (32 wh + DS_ACC_ROW(r, half)) + eo of the column's segment; the first `hop` samples of a segment are
// valid.  rp = the address of this lane's sample of register 0; no test against the tensor's end (see above).
template <int D>
__device__ __forceinline__ void ds_inv_scatter_d(float* rp, int d, int i0, int hop, int wh, const ds_f32x16& acc, const ds_f32x16& acc2) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int cr = (r & 3) + 8 * (r >> 2);
    // sample 2 (32 wh + cr + 4 half) + eo of the segment: <= 63 < hop for every wh = 0 wave (hop >= 65: k <= 64); wh = 1:
    // <= 95 for r < 8 (cr <= 11), <= 111 for r < 12 (cr <= 19), <= 127 beyond -- the uniform tests settle the registers a
    // hop leaves whole, the per-lane compare the rest (k = 7 / 11: hop 122 / 118, the last four registers)
    if (wh == 0 || (r < 8 && hop >= 96) || (r < 12 && hop >= 112) || i0 + 2 * cr < hop) rp[2 * (D ? D : d) * cr] = acc[r] + acc2[r];
  }
}
// (the dilations of the AMP blocks -- 1, 3, 5 -- with the sample stride as an immediate offset of the LDS store)
__device__ __forceinline__ void ds_inv_scatter(float* rp, int d, int i0, int hop, int wh, const ds_f32x16& acc, const ds_f32x16& acc2) {
  if (d == 1) ds_inv_scatter_d<1>(rp, d, i0, hop, wh, acc, acc2);
  else if (d == 3) ds_inv_scatter_d<3>(rp, d, i0, hop, wh, acc, acc2);
  else if (d == 5) ds_inv_scatter_d<5>(rp, d, i0, hop, wh, acc, acc2);
  else ds_inv_scatter_d<0>(rp, d, i0, hop, wh, acc, acc2);
}

__global__ __launch_bounds__(DS_MEM) void dftseg_inv_kernel(const hsp_dftseg_args a, const DsGeom G, int nbuf) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [cg][pitch] output stretch of every row | B buffers
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int hop = DS_N - (a.k - 1), d = a.dil, pitch = G.pitch;
  const int nitems = a.B * G.ngrp * G.nchunk;                   // persistent workgroups, as the forward kernel
  // All four waves work on ONE 32-column block: wave (eo, wh) produces the even (eo = 0) or odd time samples
  // 2 (32 wh + row) + eo of its segments from E^ (O^).  A fragments: rows 32 wh + l32 of the 64 x 64 inverse table
  // (time j x packed spectrum slot: slot t < 32 = Re of bin t, slot 32 = bin 32 (real), slot 32 + t = Im of bin t).
  const int wh = wave & 1, eo = wave >> 1;
  float fa[32], twc[4], tws[4];
  {
    const float* frow = a.dft + (32 * wh + l32) * DS_H + half;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) fa[ks] = frow[2 * ks];
    // staging: this thread recombines the bins kf(u) = 8 wave + 4 half + u, u < 4, of column l32
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      twc[u] = a.dft[DS_H * DS_H + 8 * wave + 4 * half + u];
      tws[u] = a.dft[DS_H * DS_H + 32 + 8 * wave + 4 * half + u];
    }
  }
  float* const bbuf = lds + G.bufsz;                            // [nbuf][E^ | O^][64 slots][32 columns]
  const DsInvSrc src = ds_inv_src(a.xf, a.xf_bs, a.C, a.Np, wave, half);
  const int kf0 = 8 * wave + 4 * half;
  // the epilogue runs in the scalar form unless every row it touches is 16-B addressable (workgroup-uniform)
  const bool vec0 = ((a.L | (int)a.y_bs | (int)a.y_cs | (int)a.res_bs | (int)a.res_cs) & 3) == 0 &&
                    ((reinterpret_cast<uintptr_t>(a.y) | reinterpret_cast<uintptr_t>(a.res)) & 15) == 0;
  // B operand of a column block: [E^ | O^][64 slots][32 columns], the same for all four waves, so it is staged through
  // LDS once.  A thread fetches X[k] and X[64 - k] (bin 0's thread: DC | Nyquist and X[32]) of four bins of one column
  // -- every load instruction two 128-B runs -- and stores E^, O^ (ds_inv_stash).  The fetches run TWO column blocks
  // ahead of the MFMAs, across item boundaries (the first block of the next row is in flight under this row's epilogue):
  // a block is 32 MFMAs per wave, 1 us, less than one HBM round trip.
  const int nmine = (nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  auto item_of = [&](int it) __attribute__((always_inline)) { return ds_item(blockIdx.x + it * gridDim.x, a, G); };
  int it_f = 0, blk_f = 0;                                      // the position the next fetch reads
  DsItem If = item_of(0);
  int ncols_f = If.ncg * d * If.S;
  float ipf = 1.0f / (float)(d * If.S), isf = 1.0f / (float)If.S;
  bool more_f = true;
  // fetch the next block of the fetch stream into v; q = its column decode (handed to the scatter two steps later)
  auto fetch_next = [&](float (&v)[16], DsCol& q) __attribute__((always_inline)) {
    q = ds_col_f(32 * blk_f + l32, ncols_f, d, If.S, ipf, isf);
    ds_inv_fetch(src, 4u * (unsigned)((If.c0 + q.ch) * a.Np + (If.b * d + q.p) * a.nseg + If.s0 + q.s), v);
    // advance (behind the last block of the last item the position stays: two harmless repeats)
    if (32 * (blk_f + 1) < ncols_f) {
      ++blk_f;
    } else if (more_f && it_f + 1 < nmine) {
      ++it_f;
      blk_f = 0;
      If = item_of(it_f);
      ncols_f = If.ncg * d * If.S;
      ipf = 1.0f / (float)(d * If.S);
      isf = 1.0f / (float)If.S;
    } else {
      more_f = false;
    }
  };
  // the position the MFMAs are at
  int it_c = 0, blk_c = 0, buf = 0;
  DsItem I = item_of(0);
  int ncols = I.ncg * d * I.S;
  bool done = false;
  const int i0 = 2 * (32 * wh + 4 * half) + eo;                 // sample (inside its segment) of accumulator register 0
  // One step = one column block: stage its operand (fetched two steps ago), refill the same registers with the block
  // two steps ahead, multiply, scatter; behind the last block of an item, its epilogue.  Called alternately with the
  // two register sets, so no set is ever copied (a copy would wait for the loads it copies).
  auto step = [&](float (&v)[16], DsCol& qv) __attribute__((always_inline)) {
    const int nblk = (ncols + 31) >> 5;
    // (the barrier behind the first stash of an item also says: everyone has left the previous item's epilogue)
    float* const ob = bbuf + buf * (128 * 32);
    ds_inv_stash(ob + l32, kf0, v, twc, tws);
    const DsCol q = qv;                                         // this block's decode, before the refill overwrites it
    ds_barrier();
    fetch_next(v, qv);                                          // in flight under this block's and the next one's MFMAs
    {
      const float* bp = ob + eo * (64 * 32) + half * 32 + l32;
      ds_f32x16 acc, acc2;                                      // two chains: even / odd k-steps
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 32; ks += 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks], bp[ks * 64], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks + 1], bp[ks * 64 + 64], acc2, 0, 0, 0);
      }
      if (q.ok) ds_inv_scatter(lds + q.ch * pitch + q.p + d * (q.s * hop + i0), d, i0, hop, wh, acc, acc2);
    }
    const bool last = blk_c + 1 == nblk;
    if (nbuf == 1 || last) ds_barrier();                        // the buffer is free again / the item is scattered
    buf = nbuf == 1 ? 0 : buf ^ 1;
    if (!last) {
      ++blk_c;
      return;
    }
    const int tb = d * I.s0 * hop;                              // output index of row[0]
    const int tl = min(a.L - tb, d * I.S * hop);                // outputs of this chunk
    if (vec0 && ((tb | tl) & 3) == 0) ds_inv_epilogue<4, 8>(a, G, I, lds, wave, 4, lane, tb, tl);
    else ds_inv_epilogue<1, 4>(a, G, I, lds, wave, 4, lane, tb, tl);
    if (++it_c < nmine) {
      blk_c = 0;
      I = item_of(it_c);
      ncols = I.ncg * d * I.S;
    } else {
      done = true;
    }
  };
  float v1[16], v2[16];
  DsCol q1, q2;
  fetch_next(v1, q1);
  fetch_next(v2, q2);
  while (true) {
    step(v1, q1);
    if (done) break;
    step(v2, q2);
    if (done) break;
  }
}

// ------------------------------------------------------------------------------- inverse -> activation -> forward
// The two convs of an AMP pair (hierspeechpp_speechsynthesizer.py:380-384: xt = c1(a1(x)); xt = c2(a2(xt)); x = xt + x)
// in their frequency-domain forms meet here: the inverse transform of c1's product, c1's bias, the activation a2 and the
// forward transform of c2's input in ONE launch -- xt is never in HBM (a write and a read of the tensor, the staging
// of the forward kernel and the epilogue of the inverse one go away).  One persistent 8-wave workgroup per CU, an item =
// `cg` whole channel rows of one utterance (both convs unchunked), three phases separated by workgroup barriers:
//   1. inverse of c1 into stretch A: the waves form two groups of four, each group one column block per step (fetches
//      two steps ahead, across items, as in dftseg_inv_kernel);
//   2. activation A -> stretch B (the forward stretch of c2: zero padding, 16-B shift): a wave per 240-sample segment,
//      its raw window a 16-B LDS load of A instead of a global one, then exactly da_segment;
//   3. forward of c2 out of B: four wave pairs on alternate column blocks.
// LDS: A | B | (two 16-KB operand buffers of phase 1 = the eight activation slices of phase 2) | twiddles, taps.
struct DpGeom {
  int cg, ngrp, S1, pitchA, S2, pitchB, offB, offX, offT;
};

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void dftseg_pair_kernel(
    const hsp_dftseg_args ai, const hsp_dftseg_args af, const DpGeom G) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int hop1 = DS_N - (ai.k - 1), d1 = ai.dil, hop2 = DS_N - (af.k - 1), d2 = af.dil;
  float* const sA = lds;
  float* const sB = lds + G.offB;
  float* const twl = lds + G.offT;                              // forward twiddles (cos | sin)(2 pi k / 128)
  float* const flt = twl + 64;                                  // activation taps as da_segment wants them
  // phase-1 role: group grp of four waves, wave (eo, wh) of it; phase-3 role: pair pw, bin half wh3
  const int grp = wave >> 2, w4 = wave & 3, wh1 = w4 & 1, eo = w4 >> 1;
  const int pw = wave >> 1, wh3 = wave & 1;
  float* const bbuf = lds + G.offX + grp * (128 * 32);
  float* const slice = lds + G.offX + wave * DA_SLICE;
  float fa1[32], fa3[32], twc[4], tws[4];
  {
    const float* frow = ai.dft + (32 * wh1 + l32) * DS_H + half;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) fa1[ks] = frow[2 * ks];
    const float* grow = af.dft + (32 * wh3 + l32) * DS_H + half;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) fa3[ks] = grow[2 * ks];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      twc[u] = ai.dft[DS_H * DS_H + 8 * w4 + 4 * half + u];
      tws[u] = ai.dft[DS_H * DS_H + 32 + 8 * w4 + 4 * half + u];
    }
  }
  if (tid < 64) twl[tid] = af.dft[DS_H * DS_H + tid];
  if (tid >= 64 && tid < 88) {
    const int j = tid - 64, i = j >> 1;
    flt[j] = j < 12 ? 2.0f * af.act_filt[10 - 2 * i + (j & 1)] : af.act_filt[j];
  }
  const int nitems = ai.B * G.ngrp;
  const int nmine = (nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  auto item_of = [&](int it) __attribute__((always_inline)) {
    DsItem I;
    const int id = blockIdx.x + it * gridDim.x;
    I.b = id / G.ngrp;
    I.c0 = (id - I.b * G.ngrp) * G.cg;
    I.ncg = min(G.cg, ai.C - I.c0);
    I.s0 = 0;
    I.S = G.S1;
    return I;
  };
  // ---- phase-1 fetch stream of this group: blocks grp, grp + 2, ... of every item (a block behind the item's last one
  // has no valid column: clamped addresses, nothing scattered), two steps ahead
  int it_f = 0, st_f = 0;
  DsItem If = item_of(0);
  int ncols_f = If.ncg * d1 * G.S1;
  bool more_f = true;
  const float ip1 = 1.0f / (float)(d1 * G.S1), is1 = 1.0f / (float)G.S1;
  const DsInvSrc src = ds_inv_src(ai.xf, ai.xf_bs, ai.C, ai.Np, w4, half);
  const int kf0 = 8 * w4 + 4 * half;
  const int i0 = 2 * (32 * wh1 + 4 * half) + eo;                // sample (inside its segment) of accumulator register 0
  // (round 6: addresses, column decode and scatter as in dftseg_inv_kernel -- four uniform bases + 32-bit offsets, float-
  // reciprocal decode handed from the fetch to the scatter, no per-sample bounds test)
  auto fetch_next = [&](float (&v)[16], DsCol& q) __attribute__((always_inline)) {
    q = ds_col_f(32 * (2 * st_f + grp) + l32, ncols_f, d1, G.S1, ip1, is1);
    ds_inv_fetch(src, 4u * (unsigned)((If.c0 + q.ch) * ai.Np + (If.b * d1 + q.p) * ai.nseg + q.s), v);
    if (64 * (st_f + 1) < ncols_f) {
      ++st_f;
    } else if (more_f && it_f + 1 < nmine) {
      ++it_f;
      st_f = 0;
      If = item_of(it_f);
      ncols_f = If.ncg * d1 * G.S1;
    } else {
      more_f = false;
    }
  };
  const DsGeom Gf = {G.cg, G.S2, G.pitchB, G.ngrp, 1, 0};        // what da_segment reads of the forward geometry: pitch
  float v1[16], v2[16];
  DsCol q1, q2;
  fetch_next(v1, q1);
  fetch_next(v2, q2);
  __syncthreads();                                              // tables in LDS
  bool odd = false;                                             // which register set the next step stages
  for (int it = 0; it < nmine; ++it) {
    const DsItem I = item_of(it);
    const int ncols1 = I.ncg * d1 * G.S1, nst = (ncols1 + 63) >> 6;
    // ---------------- phase 1
    auto step1 = [&](float (&v)[16], DsCol& qv) __attribute__((always_inline)) {
      ds_inv_stash(bbuf + l32, kf0, v, twc, tws);
      const DsCol q = qv;
      ds_barrier();
      fetch_next(v, qv);
      const float* bp = bbuf + eo * (64 * 32) + half * 32 + l32;
      ds_f32x16 acc, acc2;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 32; ks += 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[ks], bp[ks * 64], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[ks + 1], bp[ks * 64 + 64], acc2, 0, 0, 0);
      }
      // (c1's bias joins in phase 2: a load in this branch would wait for the fetches just issued -- vmcnt(0).  A block
      // behind the item's last one has no valid column.  pitchA >= d1 S1 hop1: every index the scatter forms is inside
      // its row, and phases 2 / 3 read [0, L) only.)
      if (q.ok) ds_inv_scatter(sA + q.ch * G.pitchA + q.p + d1 * (q.s * hop1 + i0), d1, i0, hop1, wh1, acc, acc2);
      ds_barrier();                                             // the operand buffer is free / A is complete
    };
    for (int st = 0; st < nst; ++st) {
      if (odd) step1(v2, q2);
      else step1(v1, q1);
      odd = !odd;
    }
    // ---------------- phase 1.5 (pass-through form, round 6: ai.y given): the first conv's whole epilogue -- bias + residual --
    // happens here, its output leaves for HBM (it is the residual of the NEXT iteration: x = xt + x,
    // hierspeechpp_speechsynthesizer.py:384) and stays in A for the activation of the next conv
    const bool through = ai.y != nullptr;
    if (through) {
      const DsGeom Ga = {G.cg, G.S1, G.pitchA, G.ngrp, 1, 0};
      if (ai.res) ds_inv_epilogue_t<4, 4, true, false, true>(ai, Ga, I, sA, wave, 8, lane, 0, ai.L);
      else ds_inv_epilogue_t<4, 4, false, false, true>(ai, Ga, I, sA, wave, 8, lane, 0, ai.L);
      ds_barrier();
    }
    // ---------------- phase 2: A -> act -> B
    {
      DsStage sg;
      sg.t0 = -af.pad;
      sg.len = d2 * ((G.S2 - 1) * hop2 + DS_N);
      sg.sh = 4 + (sg.t0 & 3);
      sg.g_lo = 0;
      sg.ng = 0;
      sg.tot = 0;
      const DaItem da = da_item(af, I, sg);
      for (int sI = wave; sI < da.nsegs; sI += 8) {
        const int ch = sI / da.nsg, p0 = da.pa4 + DA_SEG * (sI - ch * da.nsg);
        const ds_f32x4 rv = *reinterpret_cast<const ds_f32x4*>(sA + ch * G.pitchA + hsp_clampi(p0 - 8 + 4 * lane, 0, af.L - 4));
        da_segment(af, Gf, I, sg, da, sI, lane, rv, slice, sB, flt, (ai.bias && !through) ? ai.bias[I.c0 + ch] : 0.0f);
      }
      for (int ch = 0; ch < I.ncg; ++ch) {                      // the conv's zero padding on either side
        float* row = sB + ch * G.pitchB + sg.sh;
        for (int j = tid; j < min(-sg.t0, sg.len); j += 512) row[j] = 0.0f;
        for (int j = max(af.L - sg.t0, 0) + tid; j < sg.len; j += 512) row[j] = 0.0f;
      }
      ds_barrier();
      // ---------------- phase 3: forward of c2 out of B
      const int ncols2 = I.ncg * d2 * G.S2;
      const float* buf = sB + sg.sh;
      const int step = 4 * d2;
      const unsigned rowb = 4u * (unsigned)af.xf_bs, imb = 4u * (unsigned)(af.C * af.Np);
      const int lanek = 16 * wh3 + 4 * half;
      const float ip2 = 1.0f / (float)(d2 * G.S2), is2 = 1.0f / (float)G.S2;
      for (int cb = 32 * pw; cb < ncols2; cb += 128) {
        const DsCol q = ds_col_f(cb + l32, ncols2, d2, G.S2, ip2, is2);
        const float* bp = buf + q.ch * G.pitchB + q.p + d2 * (q.s * hop2 + 2 * half);
        ds_f32x16 ae, ao;
#pragma unroll
        for (int r = 0; r < 16; ++r) ae[r] = ao[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
          ae = __builtin_amdgcn_mfma_f32_32x32x2f32(fa3[ks], bp[ks * step], ae, 0, 0, 0);
          ao = __builtin_amdgcn_mfma_f32_32x32x2f32(fa3[ks], bp[ks * step + d2], ao, 0, 0, 0);
        }
        if (q.ok) {
          const unsigned colb = 4u * (unsigned)((I.c0 + q.ch) * af.Np + (I.b * d2 + q.p) * af.nseg + q.s);
          char* const base = reinterpret_cast<char*>(af.xf);
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int kr = (r & 3) + 8 * (r >> 2);
            const float er = ae[r], ei = ae[r + 8], orr = ao[r], oi = ao[r + 8];
            const float wc = twl[lanek + kr], ws = twl[32 + lanek + kr];
            const float tr = wc * orr + ws * oi, ti = wc * oi - ws * orr;
            const bool z = r == 0 && lanek == 0;
            const bool z3 = z && af.prod3;                      // (-E0, -O0) in slot 0: hsp.h prod3
            const unsigned o1 = colb + (unsigned)(lanek + kr) * rowb;
            const unsigned o2 = colb + (z ? 32u : (unsigned)(64 - lanek - kr)) * rowb;
            *reinterpret_cast<float*>(base + o1) = z3 ? -er : er + tr;
            *reinterpret_cast<float*>(base + o1 + imb) = z ? (af.prod3 ? -tr : er - tr) : ei + ti;
            *reinterpret_cast<float*>(base + o2) = z ? ei : er - tr;
            *reinterpret_cast<float*>(base + o2 + imb) = z ? -oi : ti - ei;
          }
        }
      }
    }
    // (the first barrier of the next item's phase 1 stands between these reads of B and anything that writes it)
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

// the persistent grid: the workgroups the device keeps resident at this LDS footprint
int64_t ds_resident(size_t lds_bytes) {
  static int cus = 0;
  if (!cus) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cus = n;
  }
  const int per = (int)std::min<size_t>(2, std::max<size_t>(1, (160 * 1024) / std::max<size_t>(lds_bytes, 1)));
  return (int64_t)cus * per;
}

// Chunk geometry: the LARGEST even segment count whose stretch fits the LDS budget (the whole row at the Generator's
// lengths: one staging per row -- smaller chunks were measured 1.1-2 x slower), then as many channel rows as still fit,
// preferring a count that fills the last MFMA column block (112 columns of 16 rows x 7 segments leave an eighth of
// the fourth block empty, 18 rows fill it).  Budget: 68 KB of input stretch (forward) / 64 KB of output stretch
// (inverse, next to one or two 16-KB B buffers): two workgroups per CU either way.
DsGeom ds_geom(const hsp_dftseg_args& a, bool inverse) {
  const int hop = DS_N - (a.k - 1), d = a.dil;
  const int budget = inverse ? 16384 : 17408;
  auto row_of = [&](int S) { return inverse ? ((d * S * hop + 3) & ~3) : ((d * ((S - 1) * hop + DS_N) + 12 + 3) & ~3); };
  int S = a.nseg;
  if (row_of(S) > budget) {
    int smax = inverse ? budget / (d * hop) : ((budget - 15) / d - DS_N) / hop + 1;
    smax &= ~1;
    if (smax < 2) smax = 2;
    const int nch = (a.nseg + smax - 1) / smax;                 // equal chunks (28 segments: 14 + 14, not 26 + 2)
    S = ((a.nseg + nch - 1) / nch + 1) & ~1;
    if (S > smax) S = smax;
  }
  DsGeom G;
  G.S = S;
  G.pitch = row_of(S);
  int cmax = budget / G.pitch;
  cmax = cmax < 1 ? 1 : (cmax > 32 ? 32 : cmax);
  cmax = cmax > a.C ? a.C : cmax;
  const int quantum = inverse ? 32 : 64;                        // columns of one round of the MFMA waves
  // (round 6) ... and a count whose items come out in whole ROUNDS of the persistent grid: 512 channels x 800 samples
  // with 19 rows per item are 864 items on 512 resident workgroups -- a third of them walks one item while the others
  // walk two (84 % of two rounds); 16 rows per item are 1 024 = two rounds exactly, for an eighth of the last column
  // block left empty.  Score = column fill x round balance.
  const int nchunk = (a.nseg + S - 1) / S;
  int best = cmax;
  double best_score = 0.0;
  for (int cg = cmax; cg >= (cmax + 1) / 2; --cg) {
    const int ncols = cg * d * S;
    const double fill = (double)ncols / (quantum * ((ncols + quantum - 1) / quantum));
    const size_t lds = inverse ? ((size_t)cg * G.pitch + 128 * 32) * sizeof(float) : ((size_t)cg * G.pitch + 64 + 32 + 4 * DA_SLICE) * sizeof(float);
    const int64_t res = ds_resident(lds), items = (int64_t)a.B * ((a.C + cg - 1) / cg) * nchunk;
    const double balance = (double)items / (double)(((items + res - 1) / res) * res);
    const double score = fill * balance;
    if (score > best_score + 1e-9) best_score = score, best = cg;
  }
  G.cg = best;
  G.bufsz = (G.cg * G.pitch + 3) & ~3;
  G.nchunk = (a.nseg + S - 1) / S;
  G.ngrp = (a.C + G.cg - 1) / G.cg;
  return G;
}

// The pair kernel's geometry; cg == 0: the pair cannot be fused (the caller runs the two launches)
DpGeom dp_geom(const hsp_dftseg_args& ai, const hsp_dftseg_args& af) {
  DpGeom G = {};
  if (ds_check(ai) || ds_check(af)) return G;
  if (ai.B != af.B || ai.C != af.C || ai.L != af.L || ai.accumulate || ai.post_scale != 1.0f) return G;
  if (ai.res && !ai.y) return G;                                // a residual belongs to the pass-through form (ai.y given)
  if (ai.y) {                                                   // ... whose epilogue moves 16-B vectors
    if (((ai.L | (int)ai.y_bs | (int)ai.y_cs) & 3) || (reinterpret_cast<uintptr_t>(ai.y) & 15)) return G;
    if (ai.res && ((((int)ai.res_bs | (int)ai.res_cs) & 3) || (reinterpret_cast<uintptr_t>(ai.res) & 15))) return G;
  }
  if (!af.act_alpha_exp || !af.act_beta_inv || !af.act_filt || (af.L & 3)) return G;
  if (af.xf_bs * 64 * 4 > 0xffffffffll) return G;
  const int hop1 = DS_N - (ai.k - 1), hop2 = DS_N - (af.k - 1);
  G.S1 = ai.nseg;
  G.S2 = af.nseg;
  G.pitchA = (ai.dil * G.S1 * hop1 + 3) & ~3;
  G.pitchB = (af.dil * ((G.S2 - 1) * hop2 + DS_N) + 12 + 3) & ~3;
  const int avail = 160 * 1024 / 4 - 2 * 128 * 32 - 96;        // floats left for the two stretches
  int cg = avail / (G.pitchA + G.pitchB);
  cg = cg > 32 ? 32 : cg;
  cg = cg > ai.C ? ai.C : cg;
  if (cg < 1) return G;
  {
    // (round 6) whole rounds of the one-workgroup-per-CU grid, as in ds_geom: 864 items of 19 rows on 256 workgroups are
    // four rounds at 84 %, 1 024 items of 16 rows four rounds exactly
    const int64_t res = ds_resident(160 * 1024);
    int best = cg;
    double best_score = 0.0;
    for (int c = cg; c >= (cg + 1) / 2; --c) {
      const int n1 = c * ai.dil * G.S1, n2 = c * af.dil * G.S2;
      const double fill = 0.5 * ((double)n1 / (64 * ((n1 + 63) / 64)) + (double)n2 / (128 * ((n2 + 127) / 128)));
      const int64_t items = (int64_t)ai.B * ((ai.C + c - 1) / c);
      const double score = fill * (double)items / (double)(((items + res - 1) / res) * res);
      if (score > best_score + 1e-9) best_score = score, best = c;
    }
    cg = best;
  }
  G.cg = cg;
  G.ngrp = (ai.C + cg - 1) / cg;
  G.offB = cg * G.pitchA;
  G.offX = G.offB + cg * G.pitchB;
  G.offT = G.offX + 2 * 128 * 32;
  return G;
}
// what hsp_dftseg_fwd_f32 / hsp_dftseg_inv_f32 refuse beyond ds_check, from the geometry alone (hsp_dftseg_supported)
int ds_limits(const hsp_dftseg_args& a, bool act) {
  if (a.xf_bs * 64 * 4 > 0xffffffffll) return HSP_EINVAL;       // the kernels address the spectrum with 32-bit byte offsets
  const DsGeom Gf = ds_geom(a, false), Gi = ds_geom(a, true);
  const size_t lds_f = ((size_t)Gf.bufsz + 64 + (act ? 32 + 4 * DA_SLICE : 0)) * sizeof(float);
  const size_t lds_i = ((size_t)Gi.bufsz + 128 * 32) * sizeof(float);
  if (lds_f > 160 * 1024 || lds_i > 160 * 1024) return HSP_EINVAL;
  if ((int64_t)a.B * Gf.ngrp * Gf.nchunk > 0x7fffffff || (int64_t)a.B * Gi.ngrp * Gi.nchunk > 0x7fffffff) return HSP_EINVAL;
  return 0;
}
}  // namespace

extern "C" int hsp_dftseg_supported(const hsp_dftseg_args* ap) {
  if (!ap) return 0;
  hsp_dftseg_args a = *ap;
  static const float dummy = 0.0f;                              // ds_check wants non-null pointers; none is read
  a.xf = const_cast<float*>(&dummy);
  a.dft = &dummy;
  return ds_check(a) == 0 && ds_limits(a, true) == 0 ? 1 : 0;
}

extern "C" int hsp_dftseg_tables_f32(float* fwd, float* inv) {
  if (!fwd || !inv) return HSP_EINVAL;
  const double w = 2.0 * 3.14159265358979323846 / DS_H;
  for (int wh = 0; wh < 2; ++wh)
    for (int j = 0; j < 32; ++j) {
      const int k = 16 * wh + (j & 15);                         // forward row 32 wh + j: Re (j < 16) / Im of bin k of the
      float* row = fwd + (32 * wh + j) * DS_H;                  // 64-point transform; Im of bin 0 is replaced by bin 32
      for (int n = 0; n < DS_H; ++n)
        row[n] = (float)(j < 16 ? cos(w * k * n) : (k == 0 ? ((n & 1) ? -1.0 : 1.0) : -sin(w * k * n)));
    }
  for (int j = 0; j < DS_H; ++j)                                // inverse: time j x slot (half of the usual scale: the
    for (int t = 0; t < DS_H; ++t) {                            // staged operand is twice E^ / O^)
      const int k = t & 31;
      double v;
      if (k == 0) v = (t == 0 ? 1.0 : ((j & 1) ? -1.0 : 1.0)) / 128.0;
      else v = (t < 32 ? cos(w * k * j) : -sin(w * k * j)) / 64.0;
      inv[j * DS_H + t] = (float)v;
    }
  for (int k = 0; k < 32; ++k) {                                // W^k of the 128-point transform: (cos, sin)(2 pi k / 128)
    fwd[DS_H * DS_H + k] = inv[DS_H * DS_H + k] = (float)cos(w * k / 2);
    fwd[DS_H * DS_H + 32 + k] = inv[DS_H * DS_H + 32 + k] = (float)sin(w * k / 2);
  }
  return 0;
}

extern "C" int hsp_dftseg_fwd_f32(const hsp_dftseg_args* ap, void* stream) {
  if (!ap || !ap->x) return HSP_EINVAL;
  const hsp_dftseg_args& a = *ap;
  if (int e = ds_check(a)) return e;
  const DsGeom G = ds_geom(a, false);
  const bool act = a.act_alpha_exp != nullptr;
  if (act && (!a.act_beta_inv || !a.act_filt || ((a.L | (int)a.x_bs | (int)a.x_cs) & 3) || (reinterpret_cast<uintptr_t>(a.x) & 15)))
    return HSP_EINVAL;                                          // the fused activation wants 16-B addressable rows
  const size_t lds_bytes = ((size_t)G.bufsz + 64 + (act ? 32 + 4 * DA_SLICE : 0)) * sizeof(float);
  const int64_t items = (int64_t)a.B * G.ngrp * G.nchunk;
  if (lds_bytes > 160 * 1024 || items > 0x7fffffff) return HSP_EINVAL;
  if (a.xf_bs * 64 * 4 > 0xffffffffll) return HSP_EINVAL;      // the kernel addresses the spectrum with 32-bit byte offsets
  const int64_t blocks = std::min<int64_t>(items, ds_resident(lds_bytes));
  static hsp_lds_flags flags[2];
  const void* kern = act ? reinterpret_cast<const void*>(dftseg_fwd_kernel<true>) : reinterpret_cast<const void*>(dftseg_fwd_kernel<false>);
  if (lds_bytes > 32 * 1024)
    if (int e = hsp_raise_lds_limit(kern, 160 * 1024, flags[act])) return e;
  if (act)
    hipLaunchKernelGGL(dftseg_fwd_kernel<true>, dim3((unsigned)blocks), dim3(DS_MEM), lds_bytes, static_cast<hipStream_t>(stream), a, G);
  else
    hipLaunchKernelGGL(dftseg_fwd_kernel<false>, dim3((unsigned)blocks), dim3(DS_MEM), lds_bytes, static_cast<hipStream_t>(stream), a, G);
  return (int)hipGetLastError();
}

extern "C" int hsp_dftseg_inv_f32(const hsp_dftseg_args* ap, void* stream) {
  if (!ap || !ap->y) return HSP_EINVAL;
  const hsp_dftseg_args& a = *ap;
  if (int e = ds_check(a)) return e;
  const DsGeom G = ds_geom(a, true);
  const int nbuf = ((size_t)G.bufsz + 2 * 128 * 32) * sizeof(float) <= 80 * 1024 ? 2 : 1;   // two workgroups per CU
  const size_t lds_bytes = ((size_t)G.bufsz + nbuf * 128 * 32) * sizeof(float);
  const int64_t items = (int64_t)a.B * G.ngrp * G.nchunk;
  if (lds_bytes > 160 * 1024 || items > 0x7fffffff) return HSP_EINVAL;
  const int64_t blocks = std::min<int64_t>(items, ds_resident(lds_bytes));
  static hsp_lds_flags flags;
  if (lds_bytes > 32 * 1024)
    if (int e = hsp_raise_lds_limit(reinterpret_cast<const void*>(dftseg_inv_kernel), 160 * 1024, flags)) return e;
  hipLaunchKernelGGL(dftseg_inv_kernel, dim3((unsigned)blocks), dim3(DS_MEM), lds_bytes, static_cast<hipStream_t>(stream), a, G,
                     nbuf);
  return (int)hipGetLastError();
}

extern "C" int hsp_dftseg_pair_supported(const hsp_dftseg_args* inv, const hsp_dftseg_args* fwd) {
  return inv && fwd && dp_geom(*inv, *fwd).cg > 0 ? 1 : 0;
}

extern "C" int hsp_dftseg_pair_f32(const hsp_dftseg_args* inv, const hsp_dftseg_args* fwd, void* stream) {
  if (!inv || !fwd) return HSP_EINVAL;
  const DpGeom G = dp_geom(*inv, *fwd);
  if (G.cg < 1) return HSP_EINVAL;
  const size_t lds_bytes = (size_t)(G.offT + 96) * sizeof(float);
  const int64_t items = (int64_t)inv->B * G.ngrp;
  if (lds_bytes > 160 * 1024 || items > 0x7fffffff) return HSP_EINVAL;
  const int64_t blocks = std::min<int64_t>(items, ds_resident(lds_bytes));
  static hsp_lds_flags flags;
  if (int e = hsp_raise_lds_limit(reinterpret_cast<const void*>(dftseg_pair_kernel), 160 * 1024, flags)) return e;
  hipLaunchKernelGGL(dftseg_pair_kernel, dim3((unsigned)blocks), dim3(512), lds_bytes, static_cast<hipStream_t>(stream), *inv,
                     *fwd, G);
  return (int)hipGetLastError();
}

