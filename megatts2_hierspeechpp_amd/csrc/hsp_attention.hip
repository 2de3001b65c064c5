// Multi-head attention on channel-major (B, H*D, T) tensors, fp32.
//
// Covers timm 0.6.13 Attention as used by DiTConVBlock (modules.py:397,409: no key
// mask, scale D^-0.5), and attentions.MultiHeadAttention (attentions.py:157-188:
// masked_fill(mask == 0, -1e4), optional relative-position window on keys and values).
//
// Whole-row kernels (mha_kernel, mha_mfma_kernel): one workgroup = 16 / 32 queries of one (batch, head), the
// scores for ALL keys are kept in LDS, soft-maxed by rows, then multiplied with V.  They serve every launch
// whose score rows fit the CU's LDS (T <= a few hundred on the 50 Hz path); beyond that the key-streaming
// kernels at the end of this file (online softmax over key blocks) take over, so no launch has a length ceiling.
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
  const int64_t qcs = a.q_cs, kcs = a.k_cs, vcs = a.v_cs, ocs = a.o_cs;  // channel strides (host fills the defaults)
  const float* qh = a.q + (int64_t)b * a.q_bs + (int64_t)h * D * qcs;
  const float* kh = a.k + (int64_t)b * a.k_bs + (int64_t)h * D * kcs;
  const float* vh = a.v + (int64_t)b * a.v_bs + (int64_t)h * D * vcs;
  float* oh = a.o + (int64_t)b * a.o_bs + (int64_t)h * D * ocs;

  // ---- Q tile, pre-scaled (attentions.py:164 divides the query; timm scales the product)
  for (int e = tid; e < D * QT; e += ATT_THREADS) {
    const int i = e % QT, d = e / QT;
    Qs[e] = (i0 + i < Tq) ? qh[(int64_t)d * qcs + i0 + i] * a.qk_scale : 0.0f;
  }
  __syncthreads();

  // ---- scores: wave w owns queries 4w..4w+3, lanes run along keys
  {
    const int iq = wave * 4;
    for (int j = lane; j < Tk; j += 64) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (int d = 0; d < D; ++d) {
        const float kv = kh[(int64_t)d * kcs + j];
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
        if (a.mask_dense && i < Tq && a.mask_dense[(int64_t)b * a.mask_dense_bs + (int64_t)i * Tk + j] == 0.0f) s = -1e4f;
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
        Vs[lane * dpad + dd] = (j0 + lane < Tk) ? vh[(int64_t)(d0 + dd) * vcs + j0 + lane] : 0.0f;
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
        oh[(int64_t)(d0 + d) * ocs + i] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// fp32-MFMA attention (v_mfma_f32_32x32x2_f32, exact fp32) for the no-window case: timm
// Attention of the DiT blocks and the StyleEncoder's self-attention.  One workgroup = 32
// queries of one (batch, head):
//   S^T-free layout: S[32 x Tk] = (scale*Q)^T K  with M = queries, N = keys, k = head dim
//                    (A fragments from the Q tile in LDS, B fragments straight from the
//                    channel-major K plane: 32 consecutive keys per half wave);
//   row softmax in LDS (mask -> -1e4 exactly like attentions.py:175);
//   O^T[D x 32] = V P^T with M = head dim, N = queries, k = keys: V is staged through LDS in
//                    64-key slabs (odd pitch: the column reads of both operands are conflict
//                    free) and the result lands with queries on the lanes -> coalesced stores.
constexpr int MQT = 32;
typedef float mha_f32x16 __attribute__((ext_vector_type(16)));

// NDB = ceil(D / 32).  WHOLE_V: all of V (D x Tk) is staged in LDS once, next to the Q tile, so the kernel
// has three global round trips in sequence (Q + V, the K fragments of this wave's key blocks, nothing else)
// instead of one per 4 head-dim steps and two per 64-key slab: at T <= 256 the launch is pure latency.
template <int NDB, bool WHOLE_V>
__global__ __launch_bounds__(256) void mha_mfma_kernel(const hsp_mha_args a, int n_qt, int sp, int vp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int D = a.D, Tq = a.Tq, Tk = a.Tk;
  constexpr int DP = NDB * 32;
  float* Qs = lds;                 // [DP][32]  scale * q, zero rows beyond D
  float* S = Qs + DP * 32;         // [32][sp]
  float* Vs = S + 32 * sp;         // WHOLE_V: [DP][vp] (vp odd, >= Tk rounded up to 64) ; else [DP][65]
  int bid = blockIdx.x;
  const int qt = bid % n_qt;
  bid /= n_qt;
  const int h = bid % a.H;
  const int b = bid / a.H;
  const int i0 = qt * MQT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, half = lane >> 5;
  const int64_t qcs = a.q_cs, kcs = a.k_cs, vcs = a.v_cs, ocs = a.o_cs;  // channel strides (host fills the defaults)
  const float* qh = a.q + (int64_t)b * a.q_bs + (int64_t)h * D * qcs;
  const float* kh = a.k + (int64_t)b * a.k_bs + (int64_t)h * D * kcs;
  const float* vh = a.v + (int64_t)b * a.v_bs + (int64_t)h * D * vcs;
  float* oh = a.o + (int64_t)b * a.o_bs + (int64_t)h * D * ocs;

  if constexpr (WHOLE_V) {
    // V rows straight into LDS (LDS-DMA, 4 B per lane, nothing waits here); padding is zeroed by hand
    const int tkp = (Tk + 63) & ~63;
    for (int d = wave; d < DP; d += 4)
      for (int j0 = 0; j0 < tkp; j0 += 64) {
        if (d < D && j0 + lane < Tk)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vh + (int64_t)d * vcs + j0 + lane),
                                           (__attribute__((address_space(3))) void*)(Vs + d * vp + j0), 4, 0, 0);
        else
          Vs[d * vp + j0 + lane] = 0.0f;
      }
  }
#pragma unroll
  for (int u = 0; u < DP * 32 / 256; ++u) {
    const int e = tid + 256 * u;
    const int i = e & 31, d = e >> 5;
    Qs[e] = (d < D && i0 + i < Tq) ? qh[(int64_t)d * qcs + i0 + i] * a.qk_scale : 0.0f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- scores: wave w owns key blocks w, w+4, ...; the block's K fragments are fetched in one batch
  // (issuing the first block's fragments together with the Q / V reads saves 1.2 us at T <= 128 but costs
  //  more than that once two workgroups share a CU: measured, not kept)
  const int nkb = (Tk + 31) >> 5;
  for (int jb = wave; jb < nkb; jb += 4) {
    const int j = jb * 32 + l32;
    const bool jok = j < Tk;
    float kf[DP / 2];
#pragma unroll
    for (int kk = 0; kk < DP / 2; ++kk) {
      const int d = 2 * kk + half;
      kf[kk] = (jok && d < D) ? kh[(int64_t)d * kcs + j] : 0.0f;
    }
    mha_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int kk = 0; kk < DP / 2; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[(2 * kk + half) * 32 + l32], kf[kk], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) S[((r & 3) + 8 * (r >> 2) + 4 * half) * sp + j] = acc[r];
  }
  __syncthreads();

  // ---- row softmax over the Tk valid keys (8 rows per wave).  exp on the hardware exp2; the 1 / sum
  // normalisation is applied to the PV output (queries sit on the lanes there), padding columns become 0
  float* inv_s = Qs;   // Qs is dead after the score phase (all waves passed the barrier above): 32 floats reused
  for (int u = 0; u < 8; ++u) {
    const int row_i = wave * 8 + u;
    float* row = S + row_i * sp;
    const int i = i0 + row_i;
    const float mq = (a.mask_q && i < Tq) ? a.mask_q[(int64_t)b * Tq + i] : 1.0f;
    const float* md = (a.mask_dense && i < Tq) ? a.mask_dense + (int64_t)b * a.mask_dense_bs + (int64_t)i * Tk : nullptr;
    float mx = -3.0e38f;
    for (int j = lane; j < Tk; j += 64) {
      float sv = row[j];
      if (a.mask_q && mq * a.mask_k[(int64_t)b * Tk + j] == 0.0f) sv = -1e4f;
      if (md && md[j] == 0.0f) sv = -1e4f;
      row[j] = sv;
      mx = fmaxf(mx, sv);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.0f;
    for (int j = lane; j < nkb * 32 + 32 && j < sp; j += 64) {
      const float e = j < Tk ? __builtin_amdgcn_exp2f((row[j] - mx) * 1.4426950408889634f) : 0.0f;
      row[j] = e;
      sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) inv_s[row_i] = 1.0f / sum;
  }

  // ---- O^T = V P^T: wave w owns head-dim block w (D <= 128)
  mha_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  if constexpr (WHOLE_V) {
    __syncthreads();  // softmax rows complete
    if (wave < NDB) {
      const float* va = Vs + (wave * 32 + l32) * vp + half;
      const float* pb = S + l32 * sp + half;
      const int tk2 = (Tk + 1) & ~1;   // S and Vs are zero beyond Tk
#pragma unroll 8
      for (int jj = 0; jj < tk2; jj += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va[jj], pb[jj], acc, 0, 0, 0);
    }
  } else {
    for (int j0 = 0; j0 < Tk; j0 += 64) {
      __syncthreads();  // softmax complete (first slab) / previous slab consumed
      for (int d = wave; d < DP; d += 4)
        Vs[d * 65 + lane] = (d < D && j0 + lane < Tk) ? vh[(int64_t)d * vcs + j0 + lane] : 0.0f;
      __syncthreads();
      if (wave < NDB) {
        const float* va = Vs + (wave * 32 + l32) * 65 + half;
        const float* pb = S + l32 * sp + j0 + half;
#pragma unroll 8
        for (int jj = 0; jj < 64; jj += 2)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va[jj], pb[jj], acc, 0, 0, 0);
      }
    }
  }
  if (wave < NDB && i0 + l32 < Tq) {
    const float inv = inv_s[l32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (d < D) oh[(int64_t)d * ocs + i0 + l32] = acc[r] * inv;
    }
  }
}

template <int NDB>
int mha_mfma_launch(const hsp_mha_args& a, hipStream_t stream) {
  constexpr int DP = NDB * 32;
  const int sp = ((a.Tk + 31) & ~31) + 33;  // odd pitch, room for one zero slab column block
  const int vp = ((a.Tk + 63) & ~63) + 1;
  const int64_t base = ((int64_t)DP * 32 + 32 * (int64_t)sp) * (int64_t)sizeof(float);
  const int64_t lds_whole = base + (int64_t)DP * vp * (int64_t)sizeof(float);
  const int64_t lds_slab = base + (int64_t)DP * 65 * (int64_t)sizeof(float);
  const int n_qt32 = (a.Tq + MQT - 1) / MQT;
  const unsigned blocks = (unsigned)((int64_t)n_qt32 * a.H * a.B);
  // whole-V needs one round trip less per 64 keys but more LDS: with more workgroups than CUs prefer the
  // footprint that lets two of them share a CU (their latencies then overlap instead of queueing)
  const bool whole = lds_whole <= 160 * 1024 && (blocks <= 256 || lds_whole <= 80 * 1024 || lds_slab > 80 * 1024);
  if (!whole && lds_slab > 160 * 1024) return -1;
  if (whole) {
    static hsp_lds_flags flags;
    if (lds_whole > 32 * 1024)
      if (int e = hsp_raise_lds_limit(reinterpret_cast<const void*>(mha_mfma_kernel<NDB, true>), 160 * 1024, flags)) return e;
    hipLaunchKernelGGL((mha_mfma_kernel<NDB, true>), dim3(blocks), dim3(256), (size_t)lds_whole, stream, a, n_qt32, sp, vp);
  } else {
    static hsp_lds_flags flags;
    if (lds_slab > 32 * 1024)
      if (int e = hsp_raise_lds_limit(reinterpret_cast<const void*>(mha_mfma_kernel<NDB, false>), 160 * 1024, flags)) return e;
    hipLaunchKernelGGL((mha_mfma_kernel<NDB, false>), dim3(blocks), dim3(256), (size_t)lds_slab, stream, a, n_qt32, sp, 65);
  }
  return (int)hipGetLastError();
}


// ---------------------------------------------------------------------------------------
// Key-streaming variants (online softmax): no launch depends on Tk fitting LDS.  Used when the whole-row kernels
// above would need more than the CU's 160 KB (Tk beyond ~1 200 with the MFMA kernel, ~2 200 with the window
// kernel): the denoiser's time conformer on a long prompt (160 frames/s, denoiser/conformer.py:45-60), the
// StyleEncoder / MelEncoder on a minute of mel frames (attentions.py:157-188).  Keys are walked in blocks; a query
// row keeps a running maximum m and sum l, and the PV accumulators are rescaled by exp(m_old - m_new) whenever the
// maximum moves.  masked_fill(-1e4) semantics are unchanged: a masked score is the VALUE -1e4 (a fully masked row
// ends as the uniform average, as in the reference); only keys beyond Tk are excluded (weight 0).
constexpr int SKB = 128;   // keys per block of the MFMA kernel (4 waves x 32)

template <int NDB>
__global__ __launch_bounds__(256) void mha_mfma_stream_kernel(const hsp_mha_args a, int n_qt) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int D = a.D, Tq = a.Tq, Tk = a.Tk;
  constexpr int DP = NDB * 32;
  constexpr int SP = SKB + 1, VP = SKB + 1;
  float* Qs = lds;                 // [DP][32]  scale * q, zero rows beyond D
  float* S = Qs + DP * 32;         // [32][SP]
  float* Vs = S + 32 * SP;         // [DP][VP]
  float* mrow = Vs + DP * VP;      // [32] running maximum
  float* lrow = mrow + 32;         // [32] running sum
  float* arow = lrow + 32;         // [32] rescale factor of the current block
  int bid = blockIdx.x;
  const int qt = bid % n_qt;
  bid /= n_qt;
  const int h = bid % a.H;
  const int b = bid / a.H;
  const int i0 = qt * MQT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, half = lane >> 5;
  const int64_t qcs = a.q_cs, kcs = a.k_cs, vcs = a.v_cs, ocs = a.o_cs;
  const float* qh = a.q + (int64_t)b * a.q_bs + (int64_t)h * D * qcs;
  const float* kh = a.k + (int64_t)b * a.k_bs + (int64_t)h * D * kcs;
  const float* vh = a.v + (int64_t)b * a.v_bs + (int64_t)h * D * vcs;
  float* oh = a.o + (int64_t)b * a.o_bs + (int64_t)h * D * ocs;

#pragma unroll
  for (int u = 0; u < DP * 32 / 256; ++u) {
    const int e = tid + 256 * u;
    const int i = e & 31, d = e >> 5;
    Qs[e] = (d < D && i0 + i < Tq) ? qh[(int64_t)d * qcs + i0 + i] * a.qk_scale : 0.0f;
  }
  if (tid < 32) { mrow[tid] = -3.0e38f; lrow[tid] = 0.0f; }
  mha_f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.0f;

  for (int j0 = 0; j0 < Tk; j0 += SKB) {
    __syncthreads();   // Q / state initialised (first block); S and Vs of the previous block consumed
    // V block -> LDS (zero beyond Tk and beyond D)
    for (int d = wave; d < DP; d += 4)
      for (int jj = lane; jj < SKB; jj += 64)
        Vs[d * VP + jj] = (d < D && j0 + jj < Tk) ? vh[(int64_t)d * vcs + j0 + jj] : 0.0f;
    // scores of this wave's 32 keys
    {
      const int j = j0 + wave * 32 + l32;
      const bool jok = j < Tk;
      float kf[DP / 2];
#pragma unroll
      for (int kk = 0; kk < DP / 2; ++kk) {
        const int d = 2 * kk + half;
        kf[kk] = (jok && d < D) ? kh[(int64_t)d * kcs + j] : 0.0f;
      }
      mha_f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
      for (int kk = 0; kk < DP / 2; ++kk)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[(2 * kk + half) * 32 + l32], kf[kk], acc, 0, 0, 0);
      const float mk = (a.mask_k && jok) ? a.mask_k[(int64_t)b * Tk + j] : 1.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row_i = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int i = i0 + row_i;
        float sv = acc[r];
        if (i < Tq && jok) {
          if (a.mask_q && a.mask_q[(int64_t)b * Tq + i] * mk == 0.0f) sv = -1e4f;
          if (a.mask_dense && a.mask_dense[(int64_t)b * a.mask_dense_bs + (int64_t)i * Tk + j] == 0.0f) sv = -1e4f;
        }
        S[row_i * SP + wave * 32 + l32] = jok ? sv : -3.0e38f;
      }
    }
    __syncthreads();
    // online softmax of rows 8 wave .. 8 wave + 7 over the block's 128 columns (two per lane)
    for (int u = 0; u < 8; ++u) {
      const int row_i = wave * 8 + u;
      float* row = S + row_i * SP;
      const float s0 = row[lane], s1 = row[lane + 64];
      float mx = fmaxf(s0, s1);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      const float mo = mrow[row_i];
      const float mn = fmaxf(mo, mx);
      const float e0 = s0 > -1.0e38f ? __builtin_amdgcn_exp2f((s0 - mn) * 1.4426950408889634f) : 0.0f;
      const float e1 = s1 > -1.0e38f ? __builtin_amdgcn_exp2f((s1 - mn) * 1.4426950408889634f) : 0.0f;
      row[lane] = e0;
      row[lane + 64] = e1;
      float sum = e0 + e1;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
      if (lane == 0) {
        const float al = __builtin_amdgcn_exp2f((mo - mn) * 1.4426950408889634f);   // 0 on the first block
        arow[row_i] = al;
        lrow[row_i] = lrow[row_i] * al + sum;
        mrow[row_i] = mn;
      }
    }
    __syncthreads();
    // O^T += V P^T on the rescaled accumulators: wave w owns head-dim block w, queries sit on the lanes
    if (wave < NDB) {
      const float al = arow[l32];
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[r] *= al;
      const float* va = Vs + (wave * 32 + l32) * VP + half;
      const float* pb = S + l32 * SP + half;
#pragma unroll 8
      for (int jj = 0; jj < SKB; jj += 2) oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(va[jj], pb[jj], oacc, 0, 0, 0);
    }
  }
  if (wave < NDB && i0 + l32 < Tq) {
    const float inv = 1.0f / lrow[l32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (d < D) oh[(int64_t)d * ocs + i0 + l32] = oacc[r] * inv;
    }
  }
}

template <int NDB>
int mha_mfma_stream_launch(const hsp_mha_args& a, hipStream_t stream) {
  constexpr int DP = NDB * 32;
  const size_t lds_bytes = ((size_t)DP * 32 + 32 * (SKB + 1) + (size_t)DP * (SKB + 1) + 96) * sizeof(float);
  const int n_qt = (a.Tq + MQT - 1) / MQT;
  const int64_t blocks = (int64_t)n_qt * a.H * a.B;
  if (blocks <= 0 || blocks > 0x7fffffff) return HSP_EINVAL;
  static hsp_lds_flags flags;
  if (lds_bytes > 32 * 1024)
    if (int e = hsp_raise_lds_limit(reinterpret_cast<const void*>(mha_mfma_stream_kernel<NDB>), (int)lds_bytes, flags)) return e;
  hipLaunchKernelGGL((mha_mfma_stream_kernel<NDB>), dim3((unsigned)blocks), dim3(256), lds_bytes, stream, a, n_qt);
  return (int)hipGetLastError();
}

// Window (relative-position) kernel, key-streaming form.  Same thread roles as mha_kernel: wave w scores
// queries 4w .. 4w+3 with lanes along the keys of a block; thread (d, query group) accumulates P V plus the
// relative-value term of the keys of this block that fall inside a query's window.  D <= 256.
constexpr int WKB = 256;   // keys per block

__global__ __launch_bounds__(ATT_THREADS) void mha_stream_kernel(const hsp_mha_args a, int n_qt, int dpad) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int D = a.D, Tq = a.Tq, Tk = a.Tk;
  constexpr int SPW = WKB + 1;
  float* Qs = lds;                       // [D][QT]
  float* S = Qs + D * QT;                // [QT][SPW]
  float* Vs = S + QT * SPW;              // [64][dpad]
  float* mrow = Vs + 64 * dpad;          // [QT]
  float* lrow = mrow + QT;
  float* arow = lrow + QT;
  int bid = blockIdx.x;
  const int qt = bid % n_qt;
  bid /= n_qt;
  const int h = bid % a.H;
  const int b = bid / a.H;
  const int i0 = qt * QT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t qcs = a.q_cs, kcs = a.k_cs, vcs = a.v_cs, ocs = a.o_cs;
  const float* qh = a.q + (int64_t)b * a.q_bs + (int64_t)h * D * qcs;
  const float* kh = a.k + (int64_t)b * a.k_bs + (int64_t)h * D * kcs;
  const float* vh = a.v + (int64_t)b * a.v_bs + (int64_t)h * D * vcs;
  float* oh = a.o + (int64_t)b * a.o_bs + (int64_t)h * D * ocs;

  for (int e = tid; e < D * QT; e += ATT_THREADS) {
    const int i = e % QT, d = e / QT;
    Qs[e] = (i0 + i < Tq) ? qh[(int64_t)d * qcs + i0 + i] * a.qk_scale : 0.0f;
  }
  if (tid < QT) { mrow[tid] = -3.0e38f; lrow[tid] = 0.0f; }
  const int d = tid & 127, ig = tid >> 7;       // PV role: head dims d and d + 128, queries 8 ig .. 8 ig + 7
  float acc0[8], acc1[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc0[u] = acc1[u] = 0.0f;

  for (int j0 = 0; j0 < Tk; j0 += WKB) {
    const int jn = min(WKB, Tk - j0);
    __syncthreads();   // Q / state ready; previous block's S consumed
    {
      const int iq = wave * 4;
      for (int jj = lane; jj < WKB; jj += 64) {
        const int j = j0 + jj;
        float sv[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        if (jj < jn) {
          float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
          for (int dd = 0; dd < D; ++dd) {
            const float kv = kh[(int64_t)dd * kcs + j];
            const float4 qv = *reinterpret_cast<const float4*>(Qs + dd * QT + iq);
            s0 = fmaf(qv.x, kv, s0);
            s1 = fmaf(qv.y, kv, s1);
            s2 = fmaf(qv.z, kv, s2);
            s3 = fmaf(qv.w, kv, s3);
          }
          sv[0] = s0; sv[1] = s1; sv[2] = s2; sv[3] = s3;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = i0 + iq + u;
            float s = sv[u];
            if (a.rel_k) {
              const int r = j - i;
              if (r >= -a.window && r <= a.window && i < Tq) {
                const float* ek = a.rel_k + (int64_t)(r + a.window) * D;
                float t = 0.0f;
                for (int dd = 0; dd < D; ++dd) t = fmaf(Qs[dd * QT + iq + u], ek[dd], t);
                s += t;
              }
            }
            if (i < Tq) {
              if (a.mask_q && a.mask_q[(int64_t)b * Tq + i] * a.mask_k[(int64_t)b * Tk + j] == 0.0f) s = -1e4f;
              if (a.mask_dense && a.mask_dense[(int64_t)b * a.mask_dense_bs + (int64_t)i * Tk + j] == 0.0f) s = -1e4f;
            }
            sv[u] = s;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) S[(iq + u) * SPW + jj] = sv[u];
      }
    }
    __syncthreads();
    for (int u = 0; u < 4; ++u) {
      const int row_i = wave * 4 + u;
      float* row = S + row_i * SPW;
      float sv[WKB / 64];
      float mx = -3.0e38f;
#pragma unroll
      for (int q = 0; q < WKB / 64; ++q) { sv[q] = row[lane + 64 * q]; mx = fmaxf(mx, sv[q]); }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      const float mo = mrow[row_i];
      const float mn = fmaxf(mo, mx);
      float sum = 0.0f;
#pragma unroll
      for (int q = 0; q < WKB / 64; ++q) {
        const float e = sv[q] > -1.0e38f ? expf(sv[q] - mn) : 0.0f;
        row[lane + 64 * q] = e;
        sum += e;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
      if (lane == 0) {
        const float al = expf(mo - mn);
        arow[row_i] = al;
        lrow[row_i] = lrow[row_i] * al + sum;
        mrow[row_i] = mn;
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) { const float al = arow[ig * 8 + u]; acc0[u] *= al; acc1[u] *= al; }
    for (int d0 = 0; d0 < D; d0 += 128) {
      for (int js = 0; js < jn; js += 64) {
        __syncthreads();
        for (int dd = wave; dd < 128 && d0 + dd < D; dd += 4)
          Vs[lane * dpad + dd] = (js + lane < jn) ? vh[(int64_t)(d0 + dd) * vcs + j0 + js + lane] : 0.0f;
        __syncthreads();
        if (d0 + d < D) {
          const int jm = min(64, jn - js);
          for (int j = 0; j < jm; ++j) {
            const float vv = Vs[j * dpad + d];
            if (d0 == 0) {
#pragma unroll
              for (int u = 0; u < 8; ++u) acc0[u] = fmaf(S[(ig * 8 + u) * SPW + js + j], vv, acc0[u]);
            } else {
#pragma unroll
              for (int u = 0; u < 8; ++u) acc1[u] = fmaf(S[(ig * 8 + u) * SPW + js + j], vv, acc1[u]);
            }
          }
        }
      }
      if (a.rel_v && d0 + d < D) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + ig * 8 + u;
          if (i >= Tq) continue;
          float t = 0.0f;
          for (int r = -a.window; r <= a.window; ++r) {
            const int j = i + r;
            if (j >= j0 && j < j0 + jn) t = fmaf(S[(ig * 8 + u) * SPW + j - j0], a.rel_v[(int64_t)(r + a.window) * D + d0 + d], t);
          }
          if (d0 == 0) acc0[u] += t; else acc1[u] += t;
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int i = i0 + ig * 8 + u;
    if (i >= Tq) continue;
    const float inv = 1.0f / lrow[ig * 8 + u];
    if (d < D) oh[(int64_t)d * ocs + i] = acc0[u] * inv;
    if (d + 128 < D) oh[(int64_t)(d + 128) * ocs + i] = acc1[u] * inv;
  }
}

int mha_stream_launch(const hsp_mha_args& a, hipStream_t stream) {
  if (a.D > 256) return HSP_EINVAL;
  const int n_qt = (a.Tq + QT - 1) / QT;
  const int dpad = (a.D < 128 ? a.D : 128) | 1;
  const size_t lds_bytes = ((size_t)a.D * QT + (size_t)QT * (WKB + 1) + 64 * (size_t)dpad + 3 * QT) * sizeof(float);
  const int64_t blocks = (int64_t)n_qt * a.H * a.B;
  if (blocks <= 0 || blocks > 0x7fffffff) return HSP_EINVAL;
  static hsp_lds_flags flags;
  if (lds_bytes > 32 * 1024)
    if (int e = hsp_raise_lds_limit(reinterpret_cast<const void*>(mha_stream_kernel), 160 * 1024, flags)) return e;   // once per device: the maximum
  hipLaunchKernelGGL(mha_stream_kernel, dim3((unsigned)blocks), dim3(ATT_THREADS), lds_bytes, stream, a, n_qt, dpad);
  return (int)hipGetLastError();
}


// ---------------------------------------------------------------------------------------
// Short-sequence attention without masks or window (the PLM loop's 4 x 69 heads over a growing prefix, timm
// Attention of the DiT blocks: Tk <= 256): latency-oriented.  The whole-row MFMA kernel above stages V through LDS
// (by LDS-DMA or slab by slab) and spends 20 us on a 10-key problem, 56 us at 200 keys (round-3 trace of the PLM
// loop: 4 launches x 200 steps = a quarter of the loop).  Here nothing but the scores touches LDS:
//   * one workgroup (8 waves) = 32 queries of one (batch, head);
//   * S = (scale Q)^T K: wave w owns key blocks w, w + 8, ...; both fragments are coalesced dword loads from the
//     channel-major planes (lanes along queries / keys), the Q fragments are loaded once and kept in registers;
//   * row softmax in LDS, 4 rows per wave;
//   * O^T = V P^T: six (three, ...) waves = head-dim blocks x two key halves.  The V fragment of lane (d, half) is ONE
//     16-B load of V[d][j0 + 4 half .. + 3] feeding four MFMAs (the k-slots of an MFMA may pair any two keys as
//     long as the P fragment pairs the same ones), so V needs no transpose and no staging; the two key halves
//     are added through LDS.
// LDS = 32 x (Tk + pad) scores + the partial-sum buffer: 45 KB at Tk = 256 -> three workgroups per CU.
constexpr int TQT = 32;

template <int NDB, int NW>
__global__ __launch_bounds__(64 * NW) void mha_tok_kernel(const hsp_mha_args a, int n_qt, int sp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KH = NW / NDB >= 2 ? 2 : 1;          // key halves in the PV phase (8 waves: 2, 4 waves: 1)
  const int D = a.D, Tq = a.Tq, Tk = a.Tk;
  float* S = lds;                        // [32][sp]
  float* red = S + 32 * sp;              // [NDB][16][64] partial O of the second key half (KH == 2)
  float* inv_s = red + (KH == 2 ? NDB * 16 * 64 : 0);    // [32]
  int bid = blockIdx.x;
  const int qt = bid % n_qt;
  bid /= n_qt;
  const int h = bid % a.H;
  const int b = bid / a.H;
  const int i0 = qt * TQT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int64_t qcs = a.q_cs, kcs = a.k_cs, vcs = a.v_cs, ocs = a.o_cs;
  const float* qh = a.q + (int64_t)b * a.q_bs + (int64_t)h * D * qcs;
  const float* kh = a.k + (int64_t)b * a.k_bs + (int64_t)h * D * kcs;
  const float* vh = a.v + (int64_t)b * a.v_bs + (int64_t)h * D * vcs;
  float* oh = a.o + (int64_t)b * a.o_bs + (int64_t)h * D * ocs;
  const int nkb = (Tk + 31) >> 5;
#ifdef HSP_TUNING
  unsigned long long* stamps = (a.window == 1003 && blockIdx.x == 0 && tid == 0) ? (unsigned long long*)a.rel_v : nullptr;
#define MT_STAMP(i) do { if (stamps) stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define MT_STAMP(i) do { } while (0)
#endif
  MT_STAMP(0);

  // ---- scores.  Tk <= 256 = eight key blocks = one (two) per wave, so nothing is reused across blocks: both
  // fragments are loaded per group of U k-steps, all of a group's loads in flight together (the loop of
  // hsp_rgemm.hip).  Only ceil(D / 2) k-steps run (35 of the padded 48 at D = 69).
  {
    constexpr int U = NDB == 3 ? 36 : 32;             // k-steps requested together: all 35 of D = 69 (one round trip; in-kernel stamps:
                                                      // two groups of 18 cost 4 us of a 9-us launch), 48 at D = 96 in two
    const int ksteps = (D + 1) >> 1;
    const bool odd_tail = (D & 1) != 0;
    const int iq = min(i0 + l32, Tq - 1);
    for (int jb = wave; jb < nkb; jb += NW) {
      const int j = jb * 32 + l32;
      const float* qp = qh + iq + (int64_t)half * qcs;
      const float* kp = kh + min(j, Tk - 1) + (int64_t)half * kcs;
      const int64_t qst = 2 * qcs, kst = 2 * kcs;
      mha_f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll 1
      for (int kk = 0; kk < ksteps; kk += U) {
        float fq[U], fk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = min(kk + u, ksteps - 1);
          const int kc = (odd_tail && half == 1 && k == ksteps - 1) ? k - 1 : k;   // stay inside the head's rows
          fq[u] = qp[(int64_t)kc * qst];
          fk[u] = kp[(int64_t)kc * kst];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool ok = kk + u < ksteps && !(odd_tail && half == 1 && kk + u == ksteps - 1);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? fq[u] * a.qk_scale : 0.0f, fk[u], acc, 0, 0, 0);
        }
      }
      const bool jok = j < Tk;
#pragma unroll
      for (int r = 0; r < 16; ++r) S[((r & 3) + 8 * (r >> 2) + 4 * half) * sp + j] = jok ? acc[r] : -3.0e38f;
    }
  }
  MT_STAMP(1);
  __syncthreads();
  MT_STAMP(2);
  // ---- V fragments of the first chunk: requested NOW, ahead of the softmax they do not depend on.
  // O^T[d][q] = sum_j V[d][j] P[q][j]: wave = (head-dim block, key part); groups of 8 keys, MAXG groups in flight.
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  constexpr int MAXG = 16;
  const int db = wave % NDB, kh2 = wave / NDB;        // waves >= KH NDB idle in the PV phase
  const int ngrp = (Tk + 7) >> 3;                     // groups of 8 keys
  const int gsplit = KH == 2 ? (ngrp + 1) / 2 : ngrp;
  const int g0 = kh2 == 0 ? 0 : gsplit, g1 = kh2 >= KH ? 0 : (kh2 == 0 ? gsplit : ngrp);
  const int dv = db * 32 + l32;
  const bool dok = dv < D;
  const float* vrow = vh + (int64_t)min(dv, D - 1) * vcs;
  // lane (d, half) wants V[d][jl .. jl + 3], jl = 8 g + 4 half: one 16-B load (4-B aligned: the utterances of a
  // batch sit side by side on the column axis).  A window that would end beyond Tk (last group only) is moved
  // back inside the row -- Tk >= 4 here -- and re-indexed below; what lies beyond Tk is zero.
  f4u v[MAXG];
  auto vload = [&](int gb) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < MAXG; ++u) {
      const int jl = 8 * (gb + u) + 4 * half;
      if (gb + u < g1) v[u] = *reinterpret_cast<const f4u*>(vrow + max(min(jl, Tk - 4), 0));
    }
  };
  vload(g0);
  // ---- row softmax, 32 / NW rows per wave, walked in LOCKSTEP: a wave-wide reduction is six dependent cross-lane
  // steps of ~150 cycles each, and the rows' chains are independent (in-kernel stamps: row after row this phase was
  // 2.9 us of a 9-us launch at 16 keys).  P is left un-normalised, 1 / sum goes to the output.
  const int ncol = nkb * 32;                          // columns written above (the padding holds -3e38 -> 0)
  {
    constexpr int NR = 32 / NW;
    float* rows[NR];
    float mx[NR], sum[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      rows[u] = S + (wave * NR + u) * sp;
      mx[u] = -3.0e38f;
      sum[u] = 0.0f;
    }
    for (int j = lane; j < ncol; j += 64) {
#pragma unroll
      for (int u = 0; u < NR; ++u) mx[u] = fmaxf(mx[u], rows[u][j]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
      for (int u = 0; u < NR; ++u) mx[u] = fmaxf(mx[u], __shfl_xor(mx[u], o, 64));
    }
    for (int j = lane; j < ncol; j += 64) {
#pragma unroll
      for (int u = 0; u < NR; ++u) {
        const float sv = rows[u][j];
        const float e = sv > -1.0e38f ? __builtin_amdgcn_exp2f((sv - mx[u]) * 1.4426950408889634f) : 0.0f;
        rows[u][j] = e;
        sum[u] += e;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
      for (int u = 0; u < NR; ++u) sum[u] += __shfl_xor(sum[u], o, 64);
    }
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < NR; ++u) inv_s[wave * NR + u] = 1.0f / sum[u];
    }
  }
  // zero the columns [ncol, ncol8) the PV groups of eight keys may read
  {
    const int ncol8 = (ncol + 7) & ~7;
    for (int e = tid; e < 32 * (ncol8 - ncol); e += 64 * NW) S[(e / (ncol8 - ncol)) * sp + ncol + e % (ncol8 - ncol)] = 0.0f;
  }
  MT_STAMP(3);
  __syncthreads();
  MT_STAMP(4);
  // ---- O^T += V P^T
  mha_f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.0f;
  if (kh2 < KH) {
    const float* prow = S + l32 * sp + 4 * half;
    const int glast = ngrp - 1;                       // the only group whose window may have been moved back
    for (int gb = g0; gb < g1; gb += MAXG) {
      if (gb > g0) vload(gb);
      if (!dok) {                                     // head-dim rows beyond D (last block only): contribute nothing
#pragma unroll
        for (int u = 0; u < MAXG; ++u) v[u] = f4u{0.0f, 0.0f, 0.0f, 0.0f};
      }
      float pc[4], pn[4];                             // P fragments of the current / next group
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) pc[qq] = prow[8 * gb + qq];
#pragma unroll
      for (int u = 0; u < MAXG; ++u) {
        if (gb + u < g1) {
          if (gb + u + 1 < g1) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) pn[qq] = prow[8 * (gb + u + 1) + qq];
          }
          f4u vv = v[u];
          if (gb + u == glast) {                      // wave-uniform: re-index the moved window, zero what lies beyond Tk
            const int jl = 8 * (gb + u) + 4 * half;
            const int sh = jl - max(min(jl, Tk - 4), 0);
            float t4[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
              const int idx = qq + sh;
              t4[qq] = idx == 0 ? vv[0] : idx == 1 ? vv[1] : idx == 2 ? vv[2] : idx == 3 ? vv[3] : 0.0f;
            }
            vv = f4u{t4[0], t4[1], t4[2], t4[3]};
          }
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[qq], pc[qq], oacc, 0, 0, 0);
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) pc[qq] = pn[qq];
        }
      }
    }
    if (KH == 2 && kh2 == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(db * 16 + r) * 64 + lane] = oacc[r];
    }
  }
  MT_STAMP(5);
  if (KH == 2) __syncthreads();
  MT_STAMP(6);
  if (kh2 == 0 && i0 + l32 < Tq) {
    const float inv = inv_s[l32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = db * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (d < D) oh[(int64_t)d * ocs + i0 + l32] = (oacc[r] + (KH == 2 ? red[(db * 16 + r) * 64 + lane] : 0.0f)) * inv;
    }
  }
  MT_STAMP(7);
#undef MT_STAMP
}

template <int NDB, int NW>
int mha_tok_launch_nw(const hsp_mha_args& a, hipStream_t stream, int64_t blocks, int n_qt) {
  constexpr int KH = NW / NDB >= 2 ? 2 : 1;
  const int nkb = (a.Tk + 31) >> 5;
  const int sp = ((nkb * 32 + 7) & ~7) + 1;
  const size_t lds_bytes = ((size_t)32 * sp + (KH == 2 ? NDB * 16 * 64 : 0) + 32) * sizeof(float);
  static hsp_lds_flags flags;   // raised ONCE per device, so to the kernel's maximum (Tk = 256), not to this launch's size
  constexpr int kMaxLds = (32 * 265 + NDB * 16 * 64 + 32) * (int)sizeof(float);
  if (lds_bytes > 32 * 1024)
    if (int e = hsp_raise_lds_limit(reinterpret_cast<const void*>(mha_tok_kernel<NDB, NW>), kMaxLds, flags)) return e;
  hipLaunchKernelGGL((mha_tok_kernel<NDB, NW>), dim3((unsigned)blocks), dim3(64 * NW), lds_bytes, stream, a, n_qt, sp);
  return (int)hipGetLastError();
}

template <int NDB>
int mha_tok_launch(const hsp_mha_args& a, hipStream_t stream) {
  const int n_qt = (a.Tq + TQT - 1) / TQT;
  const int64_t blocks = (int64_t)n_qt * a.H * a.B;
  if (blocks <= 0 || blocks > 0x7fffffff) return HSP_EINVAL;
  // eight waves: one key block each, two key halves in the PV phase.  (A four-wave form, of which a CU holds more,
  // was measured on 448 / 896 workgroups: 38.7 / 53.5 us against 32.6 / 58.1 -- its per-workgroup chain is twice as
  // long -- and is not built.)
  return mha_tok_launch_nw<NDB, 8>(a, stream, blocks, n_qt);
}

}  // namespace

extern "C" int hsp_mha_f32(const hsp_mha_args* ap, void* stream) {
  if (!ap) return HSP_EINVAL;
  hsp_mha_args a = *ap;
  if (a.q_cs == 0) a.q_cs = a.Tq;
  if (a.k_cs == 0) a.k_cs = a.Tk;
  if (a.v_cs == 0) a.v_cs = a.Tk;
  if (a.o_cs == 0) a.o_cs = a.Tq;
  if (a.q_cs < a.Tq || a.k_cs < a.Tk || a.v_cs < a.Tk || a.o_cs < a.Tq) return HSP_EINVAL;
  if (!a.q || !a.k || !a.v || !a.o || a.B <= 0 || a.H <= 0 || a.D <= 0 || a.Tq <= 0 || a.Tk <= 0) return HSP_EINVAL;
  if ((a.mask_q == nullptr) != (a.mask_k == nullptr)) return HSP_EINVAL;
  // test hook: window = -(w + 1) selects the key-streaming kernels at any Tk, with window w
  const bool force_stream = a.window < 0;
  if (force_stream) a.window = -(a.window + 1);
#ifdef HSP_TUNING
  if (a.window == 1003 && a.rel_v && !a.rel_k) {   // in-kernel phase stamps of workgroup 0 -> rel_v (8 x uint64): tools/mha_stamps.py
    switch ((a.D + 31) / 32) {
      case 1: return mha_tok_launch<1>(a, static_cast<hipStream_t>(stream));
      case 2: return mha_tok_launch<2>(a, static_cast<hipStream_t>(stream));
      default: return mha_tok_launch<3>(a, static_cast<hipStream_t>(stream));
    }
  }
#endif
  if ((a.rel_k || a.rel_v) && (a.window <= 0 || a.Tq != a.Tk)) return HSP_EINVAL;
  if (a.mask_dense && a.mask_dense_bs < (int64_t)a.Tq * a.Tk) return HSP_EINVAL;
  const hipStream_t st = static_cast<hipStream_t>(stream);
  // (the force_stream hook was decoded above, before validation)
  // matrix-core path: no relative-position window, head dim <= 128.  The whole-row kernel while the scores of 32
  // queries fit LDS, the key-streaming one (online softmax) beyond
  if (!a.rel_k && !a.rel_v && a.D <= 128) {
    int e = -1;
    // no masks, at most 256 keys: the latency-oriented kernel (NDB <= 3: two key halves x head-dim blocks = 6 waves)
    if (!force_stream && !a.mask_q && !a.mask_dense && a.Tk >= 4 && a.Tk <= 256 && a.D <= 96) {
      switch ((a.D + 31) / 32) {
        case 1: return mha_tok_launch<1>(a, st);
        case 2: return mha_tok_launch<2>(a, st);
        default: return mha_tok_launch<3>(a, st);
      }
    }
    if (!force_stream) {
      switch ((a.D + 31) / 32) {
        case 1: e = mha_mfma_launch<1>(a, st); break;
        case 2: e = mha_mfma_launch<2>(a, st); break;
        case 3: e = mha_mfma_launch<3>(a, st); break;
        default: e = mha_mfma_launch<4>(a, st); break;
      }
      if (e >= 0) return e;
    }
    switch ((a.D + 31) / 32) {
      case 1: return mha_mfma_stream_launch<1>(a, st);
      case 2: return mha_mfma_stream_launch<2>(a, st);
      case 3: return mha_mfma_stream_launch<3>(a, st);
      default: return mha_mfma_stream_launch<4>(a, st);
    }
  }
  const int n_qt = (a.Tq + QT - 1) / QT;
  const int dpad = (a.D < 128 ? a.D : 128) | 1;
  const int spad = a.Tk + 1;
  const int64_t lds_bytes = ((int64_t)a.D * QT + (int64_t)QT * spad + 64 * dpad) * (int64_t)sizeof(float);
  if (lds_bytes > 160 * 1024 || force_stream) return mha_stream_launch(a, st);
  static hsp_lds_flags flags;
  if (lds_bytes > 32 * 1024)
    if (int e = hsp_raise_lds_limit(reinterpret_cast<const void*>(mha_kernel), 160 * 1024, flags)) return e;
  const int64_t blocks = (int64_t)n_qt * a.H * a.B;
  hipLaunchKernelGGL(mha_kernel, dim3((unsigned)blocks), dim3(ATT_THREADS), (size_t)lds_bytes,
                     static_cast<hipStream_t>(stream), a, n_qt, dpad, spad);
  return (int)hipGetLastError();
}
