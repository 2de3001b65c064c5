// Latency-oriented fp32-MFMA GEMM for the 1x1 ("nn.Linear over tokens") convolutions of the 50 Hz
// part of the path: Y[M x N] = epilogue(W[M x K] X[K x N]) with K of a few hundred and N = a few
// hundred to a few thousand columns (the PLM loop: K = 276 / 1104, N = 16 .. 3200; the DiT / WN /
// attention projections: K = 192 .. 768, N = 200 per utterance).
//
// The implicit-GEMM conv kernel (hsp_conv1d_mfma.hip) stages one 64-channel chunk ahead; at these
// sizes a tile is 5-18 chunks whose global-load round trip (~2 us) is fully exposed each time, so a
// launch costs 20-65 us whatever N is.  Here a 64 x 64 tile keeps THREE 96-channel stages in flight
// (LDS-DMA, 144 KB), the four producer waves only issue DMA and count their own vmcnt, and EIGHT
// consumer waves only read LDS and issue MFMAs: two groups of four split every stage's channels
// (intra-tile split-K -- with fewer tiles than CUs the serial MFMA chain of a tile is what a launch
// waits for) and add their halves through LDS at the end.  Accumulators land with tokens on the
// lanes, so the epilogue stores 128-B rows straight from registers.
//
// Operands: W packed [K][w_ld] (hsp_conv1d_args.w, K == 1 tap), X channel-major with unit column
// stride and 16-B addressable rows; every out-of-range 16-B lane group reads the zero buffer.
#include "hsp_device.h"

// tuning switches exist only in the -DHSP_TUNING build (libhsp_tune.so)
#ifdef HSP_TUNING
#define TG_DBG(a, bit) (((a).debug & (bit)) != 0)
#else
#define TG_DBG(a, bit) false
#endif

namespace {

// tile 64 x 64.  A consumer group eats 48 input channels per stage.
//   SPLIT = false: 4 consumer + 4 producer waves, stage = 48 channels, 3 stages = 72 KB -> two workgroups per CU
//                  (many tiles: a second resident tile hides the first one's load round trip and epilogue)
//   SPLIT = true : 8 consumer + 4 producer waves, stage = 96 channels, 3 stages = 144 KB: two consumer groups split
//                  every stage's channels (intra-tile split-K) and add their halves through LDS at the end
//                  (fewer tiles than CUs: the serial MFMA chain of one tile is what the launch waits for)
constexpr int TG_BM = 64, TG_BN = 64, TG_KH = 48, TG_ST = 3;
typedef float tg_f32x16 __attribute__((ext_vector_type(16)));

#define TG_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define TG_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ void tg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ unsigned tg_lds_addr(const float* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) float*)p;
}

// four k-steps of A and B fragments: rows 2 (4 g + i) of the half-stage, 512 B apart
template <int G>
__device__ __forceinline__ void tg_read4(float (&A)[4], float (&B)[4], unsigned wa, unsigned xa) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(A[i]) : "v"(wa), "n"((4 * G + i) * 2 * TG_BM * 4));
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(B[i]) : "v"(xa), "n"((4 * G + i) * 2 * TG_BN * 4));
  }
}
template <int PENDING>
__device__ __forceinline__ void tg_wait(float (&A)[4], float (&B)[4]) {
  // the fragments become valid here: tie them to the wait so that no MFMA is scheduled above it
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3])
               : "n"(PENDING));
}

// 24 k-steps (48 channels) of one consumer wave: LDS reads of group g+1 are in flight under the MFMAs of g
template <int G = 0>
__device__ __forceinline__ void tg_mma48(tg_f32x16& acc, float (&A0)[4], float (&B0)[4], float (&A1)[4], float (&B1)[4],
                                         unsigned wa, unsigned xa) {
  if constexpr (G < 6) {
    if constexpr (G + 1 < 6) {
      tg_read4<G + 1>(A1, B1, wa, xa);
      tg_wait<8>(A0, B0);
    } else {
      tg_wait<0>(A0, B0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[i], B0[i], acc, 0, 0, 0);
    tg_mma48<G + 1>(acc, A1, B1, A0, B0, wa, xa);
  }
}

template <bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 768 : 512, 1) void tokgemm_kernel(const hsp_conv1d_args a, int n_mt, int n_nt) {
  constexpr int KS = SPLIT ? 2 * TG_KH : TG_KH;          // channels per stage
  constexpr int STAGE = KS * (TG_BM + TG_BN);            // floats per stage
  constexpr int LPW = 2 * (KS / 4) / 4;                  // DMA instructions per producer wave and stage (W + X)
  constexpr int NCW = SPLIT ? 8 : 4;                     // consumer waves
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int bid = blockIdx.x;
  const int mt = bid % n_mt;
  bid /= n_mt;
  const int nt = bid % n_nt;
  const int b = bid / n_nt;
  const int m0 = mt * TG_BM, n0 = nt * TG_BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.Cin;
  const int nstage = (K + KS - 1) / KS;

  if (wave >= NCW) {
    // ------------------------------------------------------------------ producers
    const int pw = wave - NCW;
    const int r_in = lane >> 4, col = (lane & 15) * 4;
    const bool mok = m0 + col < a.M, nok = n0 + col < a.ncols;
    const float* wsrc = a.w + m0 + col;
    const float* xsrc = a.x + (int64_t)b * a.x_bs + n0 + col;
    auto issue = [&](int s) __attribute__((always_inline)) {
      float* Ws = lds + (s % TG_ST) * STAGE;
      float* Xs = Ws + KS * TG_BM;
      const int k0 = s * KS;
#pragma unroll
      for (int q = 0; q < KS / 16; ++q) {               // instruction pw + 4q covers rows 4(pw+4q) .. +3
        const int r0 = 4 * (pw + 4 * q);
        const int k = k0 + r0 + r_in;
        const float* ws = (mok && k < K) ? wsrc + (int64_t)k * a.w_ld : a.zeros;
        __builtin_amdgcn_global_load_lds(TG_GPTR(ws), TG_LPTR(Ws + r0 * TG_BM), 16, 0, 0);
        const float* xs = (nok && k < K) ? xsrc + (int64_t)k * a.x_cs : a.zeros;
        __builtin_amdgcn_global_load_lds(TG_GPTR(xs), TG_LPTR(Xs + r0 * TG_BN), 16, 0, 0);
      }
    };
    const int npre = nstage < TG_ST ? nstage : TG_ST;
    for (int s = 0; s < npre; ++s) issue(s);
    // fused LayerNorm: column sums of (x - pivot) and (x - pivot)^2, pivot = the column's first channel.  Shifting
    // by a sample of the column keeps var = E[d^2] - E[d]^2 free of the cancellation that the raw moments suffer
    // when |mean| >> std (the residual stream of a real checkpoint), at one subtraction per element.
    float s1 = 0.0f, s2 = 0.0f, pivot = 0.0f;
    for (int s = 0; s < nstage; ++s) {
      const int issued = (s + TG_ST < nstage ? s + TG_ST : nstage) - (s + 1);  // stages in flight behind s
      if (issued >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPW) : "memory");
      else if (issued == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tg_barrier();                                     // A_s: stage s is in LDS
      if (a.ln_c1) {
        // this wave's share of the stage's rows (pw, pw+4, ...), lane = column; padded rows are zero
        const float* Xs = lds + (s % TG_ST) * STAGE + KS * TG_BM + lane;
        const int rows = K - s * KS < KS ? K - s * KS : KS;
        if (s == 0) pivot = Xs[0];                      // row 0 of stage 0: the same value in all four waves
        for (int r = pw; r < rows; r += 4) {
          const float v = Xs[r * TG_BN] - pivot;
          s1 += v;
          s2 = fmaf(v, v, s2);
        }
      }
      if (s + TG_ST < nstage) {
        tg_barrier();                                   // B_s: consumers are done with this slot
        if (!TG_DBG(a, 1)) issue(s + TG_ST);
      }
    }
    if (a.ln_c1) {
      float* st = lds + TG_ST * STAGE;                  // [4 producer waves][2][64] + pivot[64]
      st[(pw * 2 + 0) * 64 + lane] = s1;
      st[(pw * 2 + 1) * 64 + lane] = s2;
      if (pw == 0) st[8 * 64 + lane] = pivot;
      tg_barrier();                                     // C: statistics are in LDS
    }
    return;
  }

  // -------------------------------------------------------------------- consumers
  const int grp = SPLIT ? wave >> 2 : 0, w4 = wave & 3;
  const int l32 = lane & 31, half = lane >> 5;
  const int wm = w4 & 1, wn = w4 >> 1;
  tg_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  // epilogue operands of this lane's rows: fetched now, they arrive under the main loop
  constexpr int NR = SPLIT ? 8 : 16;                     // accumulator rows this wave finalises
  const int n = n0 + wn * 32 + l32;
  const int nc = n < a.ncols ? n : a.ncols - 1;
  const int mb = m0 + wm * 32 + 4 * half;
  float bv[NR], rv[NR], yv[NR], cs[NR], c1[NR];
  // second output (hsp_conv1d_args.split_row): whole row tiles at or beyond split_row use the second parameter set
  const bool second = a.split_row > 0 && m0 >= a.split_row;
  const int mo = second ? a.split_row : 0;
  const int mmode = second ? a.mask_mode2 : a.mask_mode;
  const int accum = second ? a.accumulate2 : a.accumulate;
  const int64_t ycs = second ? a.y2_cs : a.y_cs;
  const float mk = mmode != HSP_MASK_NONE ? a.mask[(int64_t)b * a.mask_bs + nc] : 1.0f;
  const float* resb = (a.res && !second) ? a.res + (int64_t)b * a.res_bs + nc : nullptr;
  float* yb = (second ? a.y2 + (int64_t)b * a.y2_bs : a.y + (int64_t)b * a.y_bs) + nc;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int r = NR * grp + i;
    const int m = mb + (r & 3) + 8 * (r >> 2);
    const int mc = m < a.Cout ? m : a.Cout - 1;
    float t = a.bias ? a.bias[mc] : 0.0f;
    if (a.cbias) t += a.cbias[(int64_t)b * a.cbias_bs + mc];
    bv[i] = t;
    cs[i] = a.cscale ? a.cscale[(int64_t)b * a.cscale_bs + mc] : 1.0f;
    rv[i] = resb ? resb[(int64_t)mc * a.res_cs] : 0.0f;
    yv[i] = accum ? yb[(int64_t)(mc - mo) * ycs] : 0.0f;
    c1[i] = a.ln_c1 ? a.ln_c1[mc] : 0.0f;
  }
  float A0[4], B0[4], A1[4], B1[4];
  for (int s = 0; s < nstage; ++s) {
    tg_barrier();                                       // A_s
    const float* Ws = lds + (s % TG_ST) * STAGE + (grp * TG_KH + half) * TG_BM + wm * 32 + l32;
    const float* Xs = lds + (s % TG_ST) * STAGE + KS * TG_BM + (grp * TG_KH + half) * TG_BN + wn * 32 + l32;
    int rows = K - s * KS - grp * TG_KH;
    rows = rows < 0 ? 0 : (rows > TG_KH ? TG_KH : rows);
    if (TG_DBG(a, 2)) rows = 0;
    if (rows == TG_KH) {
      const unsigned wa = tg_lds_addr(Ws), xa = tg_lds_addr(Xs);
      tg_read4<0>(A0, B0, wa, xa);
      tg_mma48<0>(acc, A0, B0, A1, B1, wa, xa);
    } else {
      for (int kk = 0; kk < rows / 2; ++kk)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ws[2 * kk * TG_BM], Xs[2 * kk * TG_BN], acc, 0, 0, 0);
    }
    if (s + TG_ST < nstage) tg_barrier();               // B_s
  }

  // ---- fused LayerNorm: mean / rstd of this lane's column from the producers' partial sums
  float mean = 0.0f, rstd = 1.0f;
  if (a.ln_c1) {
    tg_barrier();                                       // C
    const float* st = lds + TG_ST * STAGE + wn * 32 + l32;
    const float t1 = (st[0 * 64] + st[2 * 64]) + (st[4 * 64] + st[6 * 64]);
    const float t2 = (st[1 * 64] + st[3 * 64]) + (st[5 * 64] + st[7 * 64]);
    const float dm = t1 / (float)K;                     // mean of (x - pivot)
    mean = st[8 * 64] + dm;
    const float var = fmaxf(t2 / (float)K - dm * dm, 0.0f);
    rstd = 1.0f / sqrtf(var + a.ln_eps);
  }

  // ---- SPLIT: combine the two K-groups.  Each finalises 8 of the 16 accumulator rows and hands the other 8
  // over through a stage slot nobody reads any more (slot of stage nstage - 3, or one never issued)
  float fin[NR];
  if constexpr (SPLIT) {
    float* xch = lds + (nstage % TG_ST) * STAGE;         // [dst group][w4][8][64]
    float* dst = xch + (((1 - grp) * 4 + w4) * 8) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i * 64] = grp ? acc[i] : acc[8 + i];
    tg_barrier();
    const float* src = xch + ((grp * 4 + w4) * 8) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 8; ++i) fin[i] = (grp ? acc[8 + i] : acc[i]) + src[i * 64];
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) fin[i] = acc[i];
  }

  // epilogue: fin[i] = Y[m0 + 32 wm + (r&3) + 8 (r>>2) + 4 half][n0 + 32 wn + l32], r = NR grp + i
  if (n >= a.ncols) return;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int r = NR * grp + i;
    const int m = mb + (r & 3) + 8 * (r >> 2);
    if (m >= a.Cout) continue;
    float v = a.ln_c1 ? fmaf(rstd, fmaf(-mean, c1[i], fin[i]), bv[i]) : fin[i] + bv[i];
    v = hsp_apply_act(v, a.act);                        // then the order of hsp_epilogue_store
    if (mmode & HSP_MASK_PRE) v *= mk;
    v *= cs[i];
    v *= a.scale;
    v += rv[i];
    if (mmode & HSP_MASK_POST) v *= mk;
    v += yv[i];
    yb[(int64_t)(m - mo) * ycs] = v * a.post_scale;
  }
}

template <bool SPLIT>
int tg_launch(const hsp_conv1d_args& a, hipStream_t s, int n_mt, int n_nt, int64_t blocks) {
  constexpr int KS = SPLIT ? 2 * TG_KH : TG_KH;
  const size_t lds_bytes = (size_t)TG_ST * KS * (TG_BM + TG_BN) * sizeof(float) + 2304;  // + LayerNorm partials and pivot
  static hsp_lds_flags flags;
  if (int e = hsp_raise_lds_limit(reinterpret_cast<const void*>(tokgemm_kernel<SPLIT>), (int)lds_bytes, flags)) return e;
  hipLaunchKernelGGL(tokgemm_kernel<SPLIT>, dim3((unsigned)blocks), dim3(SPLIT ? 768 : 512), lds_bytes, s, a, n_mt, n_nt);
  return (int)hipGetLastError();
}

}  // namespace

// Host side: eligibility + launch; called by the conv dispatcher (hsp_conv1d_mfma.hip).
// Returns -1 when the shape is not one this kernel takes.
int hsp_tokgemm_try(const hsp_conv1d_args& a, hipStream_t s, int32_t* plan_out) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (a.K != 1 || a.stride != 1 || a.pad != 0 || a.prologue != HSP_PRO_NONE || a.rows != HSP_ROWS_PLAIN) return -1;
  if (a.x_ts != 1 || a.Lin != a.ncols || a.Lout != a.ncols) return -1;
  if ((a.Cin & 3) || (a.ncols & 3) || (a.x_bs & 3) || (a.x_cs & 3) || !al16(a.x) || !al16(a.w) || (a.w_ld & 3)) return -1;
  if (a.Cin < 64 || a.Cin > 4096) return -1;
  if (a.ln_c1 && !(a.ln_eps > 0.0f)) return -1;
  if (a.split_row) {  // second output: whole 64-row tiles, its own 16-B independent destination
    if (a.split_row < 0 || (a.split_row % TG_BM) || a.split_row >= a.Cout || !a.y2 || a.ln_c1) return -1;
    if (a.mask_mode2 != HSP_MASK_NONE && !a.mask) return -1;
  }
  const int n_mt = (a.M + TG_BM - 1) / TG_BM, n_nt = (a.ncols + TG_BN - 1) / TG_BN;
  const int64_t blocks = (int64_t)n_mt * n_nt * a.B;
  if (blocks <= 0 || blocks > 0x7fffffff) return -1;
  // up to one tile per CU: halve each tile's serial chain; beyond: two resident tiles per CU
  const bool split = TG_DBG(a, 256) ? true : (TG_DBG(a, 512) ? false : blocks <= 256);
  if (plan_out) {  // {BM, BN, 0 = "token GEMM" (the conv kernel reports its chunk depth here), LDS bytes}
    plan_out[0] = TG_BM; plan_out[1] = TG_BN; plan_out[2] = 0;
    plan_out[3] = (int32_t)((size_t)TG_ST * (split ? 2 : 1) * TG_KH * (TG_BM + TG_BN) * sizeof(float) + 2304);
    return 0;
  }
  return split ? tg_launch<true>(a, s, n_mt, n_nt, blocks) : tg_launch<false>(a, s, n_mt, n_nt, blocks);
}
