// Throughput-oriented fp32-MFMA GEMM for the LARGER token matrices of the 50 Hz part:
//   Y[M x N] = epilogue(W[M x K] X[K x N]),  from ~96 tiles of 64 x 64 outputs per launch upward
// (the PLM loop beyond ~30 prefix positions: K = 276 / 1104, M = 276 / 828 / 1104, N = 16 n -- reference call sites
// ttv_v1/transformer_mega.py:63-73 (w_q / w_k / w_v, out_proj), :121-126 (ff.0, ff.3); the DiT qkv / proj / fc2 and WN
// res_skip 1x1s of a front group of the vocoder, modules.py:166-174,357-411).
//
// Why a third token-GEMM kernel.  hsp_rgemm.hip (no staging, 32 x 32 / 64 x 32 tiles, K split over the waves) and
// hsp_tokgemm.hip (LDS-DMA, 64 x 64 tiles) are built around the latency of ONE small launch; at 1 600-3 200 columns they
// sat at 35-40 TFLOP/s, and their fused input LayerNorm took its statistics on the producers' critical path (+14 us at
// 3 200 columns).  Two tile shapes here (hsp_bgemm_try picks; tools/gemm_bench.py, profiles/r03_bgemm_bench.txt):
//   64 x 64   three 48-channel stages = 72 KB of LDS -> TWO workgroups per CU: the second resident tile covers the
//             first one's start (1.2 us to the first stage) and its epilogue; the default;
//   128 x 128 one workgroup per CU, half the L2 traffic per flop; from ~850 small tiles (1104 x 3 200).
// (128 x 64 and 64 x 128 exist in the tuning build only: ties.)
//
//   waves 0-3  consumers (2 x 2 over the tile, TM x TN blocks of 32 x 32 each): A / B fragments by plain LDS loads one
//              group of k-steps ahead of the MFMAs that use them (the compiler counts them and places the waits; a
//              scheduling barrier per group keeps the order); with a fused input LayerNorm they also form the column
//              statistics from the B fragments passing through their registers (pivot-shifted sums, as in the other two
//              kernels) -- no extra pass over the staged tile and no work on the producers' critical path;
//   waves 4-7  producers: LDS-DMA only, three (four: 128 x 64) stages in flight; one vector instruction per DMA on
//              interior tiles (lane offset once per tile, the row walked by the scalar unit).
//
// Workgroup order is XCD-aware: the row tiles of one column tile get consecutive logical ids and an XCD takes a
// contiguous range of ids, so a column tile's activations are fetched into ONE L2 and reused there by its row tiles.
//
// Operands as in hsp_tokgemm.hip: W packed [K][w_ld], X channel-major with unit column stride and 16-B addressable
// rows; out-of-range 16-B lane groups read the zero buffer.  Epilogues (bg_epilogue, bg_epilogue_ext): the operation
// order of the other token-GEMM kernels (bias + conditioning bias, LayerNorm correction, pointwise function, masks,
// per-(b, c) scale, scale, residual, accumulate, post_scale), with a compile-time pointwise function.
#include "hsp_device.h"

#ifdef HSP_TUNING
#define BG_DBG(a, bit) (((a).debug & (bit)) != 0)
#else
#define BG_DBG(a, bit) false
#endif

namespace {

typedef float bg_f32x16 __attribute__((ext_vector_type(16)));
constexpr int BG_KS = 48;                                // input channels per stage
// k-steps per fragment group (LDS reads of one group are in flight under the MFMAs of the previous one): a group
// must hold ~256 cycles of MFMAs to cover the LDS round trip -- four 64-cycle MFMAs
constexpr int bg_gs(int tm, int tn) { return tm * tn >= 2 ? 2 : 4; }

#define BG_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define BG_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int TM, int TN>
struct BgCfg {
  static constexpr int BM = 64 * TM, BN = 64 * TN;
  static constexpr int NST = TM * TN == 2 ? 4 : 3;       // stages resident in LDS (64 x 64: 72 KB, two workgroups per CU)
  static constexpr int STAGE = BG_KS * (BM + BN);        // floats per stage
  static constexpr int NIW = BG_KS * BM / 256, NIX = BG_KS * BN / 256;   // DMA instructions (1 KB each) per stage
  static constexpr int LPW = (NIW + NIX) / 4;            // per producer wave
  static constexpr int LDS_BYTES = NST * STAGE * (int)sizeof(float);
  static_assert(NIW % 4 == 0 && NIX % 4 == 0, "four producer waves share a stage evenly");
  static_assert(LPW * (NST - 1) <= 63, "vmcnt is six bits");
};

__device__ __forceinline__ void bg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Fragments of group G (k-steps GS G .. GS G + GS - 1 of the stage): A[step * TM + tm], B[step * TN + tn].
// Plain LDS loads: the compiler counts them (LDS returns in order) and places s_waitcnt lgkmcnt(n) itself.  A first
// version issued them from inline asm with hand-placed waits, as hsp_tokgemm.hip does; there the compiler is free to
// COPY an asm output register before the wait that makes it valid (it did, in the peeled first stage of the
// LayerNorm variants: stale fragments in some waves), so the hand-counted form is not used here.
typedef const __attribute__((address_space(3))) float* bg_lptr;
template <int TM, int TN, int G>
__device__ __forceinline__ void bg_read(float (&A)[bg_gs(TM, TN) * TM], float (&B)[bg_gs(TM, TN) * TN], bg_lptr wa, bg_lptr xa) {
  constexpr int GS = bg_gs(TM, TN);
#pragma unroll
  for (int step = 0; step < GS; ++step) {
    const int kk = GS * G + step;
#pragma unroll
    for (int j = 0; j < TM; ++j) A[step * TM + j] = wa[kk * 2 * 64 * TM + j * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) B[step * TN + j] = xa[kk * 2 * 64 * TN + j * 32];
  }
}

// one stage (24 k-steps) of a consumer wave: the LDS reads of group G + 1 are in flight under the MFMAs of group G
// (a scheduling barrier after every group keeps that order).
// LN: the wave also sums (x - pivot) and (x - pivot)^2 over the channels of ONE of its column blocks, SEL (the two
// waves that share a column range split its two blocks between them when TN == 2: the fp32 MFMAs run on the VALU's
// own lanes, so every statistics instruction is paid in matrix time).  TAIL: the stage holds fewer than 48 channels;
// the rows beyond K are staged as zeros and must not enter the column's sums.
// MOD (round 6, hsp.h ln_scale): the B fragments are multiplied by (1 + scale_b[channel]) on their way into the MFMA --
// `sl` = this stage's 48 factors in LDS (already offset by the lane's half: channel 2 k + half of k-step k); the column
// statistics keep the unscaled values.
template <int TM, int TN, bool LN, int SEL, bool TAIL, bool MOD = false, int G = 0>
__device__ __forceinline__ void bg_stage(bg_f32x16 (&acc)[TM * TN], float (&A0)[bg_gs(TM, TN) * TM],
                                         float (&B0)[bg_gs(TM, TN) * TN], float (&A1)[bg_gs(TM, TN) * TM],
                                         float (&B1)[bg_gs(TM, TN) * TN], bg_lptr wa, bg_lptr xa, float& s1, float& s2,
                                         float pivot, int rows2, bg_lptr sl = nullptr) {
  constexpr int GS = bg_gs(TM, TN), NG = BG_KS / (2 * GS);
  if constexpr (G < NG) {
    if constexpr (G + 1 < NG) bg_read<TM, TN, G + 1>(A1, B1, wa, xa);
    float sv[GS];
    if constexpr (MOD) {
#pragma unroll
      for (int i = 0; i < GS; ++i) sv[i] = sl[2 * (GS * G + i)];
    }
#pragma unroll
    for (int i = 0; i < GS; ++i) {
      if constexpr (LN) {
        float d = B0[i * TN + SEL] - pivot;
        if constexpr (TAIL) d = GS * G + i < rows2 ? d : 0.0f;
        s1 += d;
        s2 = fmaf(d, d, s2);
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const float bv = MOD ? B0[i * TN + tn] * sv[i] : B0[i * TN + tn];
          acc[tm * TN + tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[i * TM + tm], bv, acc[tm * TN + tn], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    bg_stage<TM, TN, LN, SEL, TAIL, MOD, G + 1>(acc, A1, B1, A0, B0, wa, xa, s1, s2, pivot, rows2, sl);
  }
}

// Epilogue of one consumer wave: acc[tm][tn][r] = Y[mw + 32 tm + (r & 3) + 8 (r >> 2) + 4 half][nw + 32 tn + l32].
//
// Two things made a first version of this cost more than the tile's MFMAs (in-kernel stamps, tools/bgemm_stamps.py:
// 45 000 cycles against 42 000): hsp_apply_act's run-time switch inlined 16 TM TN times (330 KB of libm branches to
// fetch), and a 64-bit row-stride multiply + per-element bounds branch in front of every store (quarter-rate VALU
// ops, ~35 instructions per element).  So: ACT is a compile-time pointwise function, a row is addressed as
// wave-uniform base + 32-bit lane offset, and bounds are checked per group of four rows.  Only the plain epilogue
// exists here (bias, LayerNorm correction, NONE / RELU / GELU_TANH, residual): launches that ask for masks, a
// per-channel scale, a running sum or non-unit scales stay with the other token-GEMM kernels (hsp_bgemm_try).
template <int TM, int TN, bool LN, int ACT>
__device__ __forceinline__ void bg_epilogue(const hsp_conv1d_args& a, const bg_f32x16 (&acc)[TM * TN], int b, int mw, int nw,
                                            int l32, int half, float bvl, float c1l, const float (&mean)[TN],
                                            const float (&rstd)[TN]) {
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    // row operands by cross-lane read, with every lane of the wave still active
    float bvr[16], c1r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int src = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;     // the lane that holds this row's operands
      bvr[r] = __shfl(bvl, src, 64);
      c1r[r] = LN ? __shfl(c1l, src, 64) : 0.0f;
    }
    {
      // lane offsets in bytes from the (uniform) address of row mu = mw + 32 tm + (r & 3) + 8 (r >> 2), column nw
      const unsigned yo = 4u * (unsigned)(l32 + 4 * half * a.y_cs), ro = 4u * (unsigned)(l32 + 4 * half * a.res_cs);
      const char* ybase = reinterpret_cast<const char*>(a.y + (int64_t)b * a.y_bs + nw);
      const char* rbase = a.res ? reinterpret_cast<const char*>(a.res + (int64_t)b * a.res_bs + nw) : nullptr;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        if (nw + tn * 32 + l32 < a.ncols) {
          // the block's residual operands in one batch (rows clamped: always a legal address)
          float rv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int mc = min(mw + tm * 32 + (r & 3) + 8 * (r >> 2), a.Cout - 1 - 4 * half);
            rv[r] = rbase ? *reinterpret_cast<const float*>(rbase + (int64_t)mc * a.res_cs * 4 + tn * 128 + ro) : 0.0f;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int mu = mw + tm * 32 + 8 * q;                      // uniform; this lane's rows: mu + 4 half + (0..3)
            if (mu + 4 * half < a.Cout) {                             // Cout % 4 == 0: the four rows stand or fall together
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const int r = 4 * q + i;
                const float fin = acc[tm * TN + tn][r];
                float v = fin + bvr[r];
                if constexpr (LN) v = fmaf(rstd[tn], fmaf(-mean[tn], c1r[r], fin), bvr[r]);
                if constexpr (ACT == HSP_ACT_RELU) v = fmaxf(v, 0.0f);
                else if constexpr (ACT == HSP_ACT_GELU_TANH) v = hsp_apply_act(v, HSP_ACT_GELU_TANH);
                v += rv[r];
                *reinterpret_cast<float*>(const_cast<char*>(ybase) + (int64_t)(mu + i) * a.y_cs * 4 + tn * 128 + yo) = v;
              }
            }
          }
        }
      }
    }
  }
}

// The same with everything hsp_conv1d_args can ask of a plain-row epilogue (the DiT / WN token GEMMs of the vocoder's
// 50 Hz part): mask before / after the residual, per-(b, c) scale, scale, running sum, post_scale and the second output
// of hsp_conv1d_args.split_row (whole 64-row tiles at or beyond split_row write y2 with their own parameter set).
// Same operation order as hsp_epilogue_store; a factor that is not asked for is an exact multiplication by 1.
template <int TM, int TN, int ACT>
__device__ __forceinline__ void bg_epilogue_ext(const hsp_conv1d_args& a, const bg_f32x16 (&acc)[TM * TN], int b, int m0, int mw,
                                                int nw, int l32, int half, float bvl) {
  const bool second = a.split_row > 0 && m0 >= a.split_row;
  const int mo = second ? a.split_row : 0;
  const int mmode = second ? a.mask_mode2 : a.mask_mode;
  const bool accum = (second ? a.accumulate2 : a.accumulate) != 0;
  const int64_t ycs = second ? a.y2_cs : a.y_cs;
  const char* ybase = reinterpret_cast<const char*>((second ? a.y2 + (int64_t)b * a.y2_bs : a.y + (int64_t)b * a.y_bs) + nw);
  const char* rbase = (a.res && !second) ? reinterpret_cast<const char*>(a.res + (int64_t)b * a.res_bs + nw) : nullptr;
  const unsigned yo = 4u * (unsigned)(l32 + 4 * half * (int)ycs), ro = 4u * (unsigned)(l32 + 4 * half * a.res_cs);
  const float csl = a.cscale ? a.cscale[(int64_t)b * a.cscale_bs + min(mw + l32 + 32 * half, a.Cout - 1)] : 1.0f;
  const float sc = a.scale, ps = a.post_scale;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    float bvr[16], csr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int src = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      bvr[r] = __shfl(bvl, src, 64);
      csr[r] = __shfl(csl, src, 64);
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int n = nw + tn * 32 + l32;
      if (n < a.ncols) {
        const float mk = mmode != HSP_MASK_NONE ? a.mask[(int64_t)b * a.mask_bs + n] : 1.0f;
        const float mk_pre = (mmode & HSP_MASK_PRE) ? mk : 1.0f, mk_post = (mmode & HSP_MASK_POST) ? mk : 1.0f;
        float rv[16], yv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int mc = min(mw + tm * 32 + (r & 3) + 8 * (r >> 2), a.Cout - 1 - 4 * half) - mo;
          rv[r] = rbase ? *reinterpret_cast<const float*>(rbase + (int64_t)mc * a.res_cs * 4 + tn * 128 + ro) : 0.0f;
          yv[r] = accum ? *reinterpret_cast<const float*>(ybase + (int64_t)mc * ycs * 4 + tn * 128 + yo) : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int mu = mw + tm * 32 + 8 * q;
          if (mu + 4 * half < a.Cout) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int r = 4 * q + i;
              float v = acc[tm * TN + tn][r] + bvr[r];
              if constexpr (ACT == HSP_ACT_RELU) v = fmaxf(v, 0.0f);
              else if constexpr (ACT == HSP_ACT_GELU_TANH) v = hsp_apply_act(v, HSP_ACT_GELU_TANH);
              v *= mk_pre;
              v *= csr[r];
              v *= sc;
              v += rv[r];
              v *= mk_post;
              v += yv[r];
              *reinterpret_cast<float*>(const_cast<char*>(ybase) + (int64_t)(mu + i - mo) * ycs * 4 + tn * 128 + yo) = v * ps;
            }
          }
        }
      }
    }
  }
}

template <int N>
__device__ __forceinline__ void bg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int TM, int TN, bool LN, bool EXT, bool MOD = false>
__global__ __launch_bounds__(512, 1) void bgemm_kernel(const hsp_conv1d_args a, int n_mt, int n_nt, int per_xcd,
                                                       int total) {
  using C = BgCfg<TM, TN>;
  static_assert(!MOD || (LN && !EXT), "the modulated form is a fused-LayerNorm form");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // logical tile id: XCD x (workgroups x, x + 8, ...) takes ids [x * per_xcd, (x + 1) * per_xcd)
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  int id = xcd * per_xcd + idx;
  if (id >= total) return;                                // whole workgroup, before any barrier
  const int mt = id % n_mt;
  id /= n_mt;
  const int nt = id % n_nt;
  const int b = id / n_nt;
  const int m0 = mt * C::BM, n0 = nt * C::BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.Cin;
  const int nstage = (K + BG_KS - 1) / BG_KS;
#ifdef HSP_TUNING
  // tuning bit 1 << 20: cycle-counter stamps of workgroup 0 (consumer wave 0: 0-3, producer wave 4: 4-6) -> a.filt
  unsigned long long* stamps = ((a.debug & (1 << 20)) && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4))
                                   ? (unsigned long long*)a.filt : nullptr;
#define BG_STAMP(i) do { if (stamps) stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define BG_STAMP(i) do { } while (0)
#endif
  if (wave == 0) BG_STAMP(0);

  if (wave >= 4) {
    // ------------------------------------------------------------------ producers
    const int pw = wave - 4;
    constexpr int LPRW = C::BM / 4, RPIW = 64 / LPRW;      // lanes per staged row, rows per DMA instruction
    constexpr int LPRX = C::BN / 4, RPIX = 64 / LPRX;
    const int rw = lane / LPRW, cw = (lane % LPRW) * 4;
    const int rx = lane / LPRX, cx = (lane % LPRX) * 4;
    const bool mok = m0 + cw < a.M, nok = n0 + cx < a.ncols;
    const float* wsrc = a.w + m0 + cw;
    const float* xsrc = a.x + (int64_t)b * a.x_bs + n0 + cx;
    // Fast staging (the conv kernel's dma_w_fast idea): the fp32 MFMAs of the consumer wave on this SIMD run on the
    // VALU's own lanes, so every vector instruction of the address arithmetic (a 64-bit multiply-add, compares,
    // selects: ~45 cycles per DMA instruction in the general form below, a fifth of a stage's matrix time) is paid in
    // matrix time.  For a tile that lies inside the matrix and a stage that lies inside K -- all but the edge tiles
    // and the last stage -- a lane's byte offset is computed ONCE and each instruction adds it to a wave-uniform
    // base that the scalar unit walks: one vector instruction per DMA.
    const bool wfast = m0 + C::BM <= a.M, xfast = n0 + C::BN <= a.ncols && a.x_cs < (1 << 26);
    // this lane's source inside the tile's first staged rows; hsp_conv1d_mfma_f32's validation bounds the element
    // offsets of the weight matrix and of one utterance to 31 bits
    const float* wlane = a.w + m0 + rw * a.w_ld + cw;
    const float* xlane = a.x + (int64_t)b * a.x_bs + n0 + (xfast ? rx * (int)a.x_cs : 0) + cx;
    const int wld = a.w_ld, xcs = (int)a.x_cs;
    auto issue = [&](int s) __attribute__((always_inline)) {
      float* Ws = lds + (s % C::NST) * C::STAGE;
      float* Xs = Ws + BG_KS * C::BM;
      const int k0 = s * BG_KS;
      const bool kfull = k0 + BG_KS <= K;
      if (wfast && kfull) {
#pragma unroll
        for (int q = 0; q < C::NIW / 4; ++q) {
          const int r0 = (pw + 4 * q) * RPIW;
          const int eo = __builtin_amdgcn_readfirstlane((k0 + r0) * wld);     // scalar unit
          __builtin_amdgcn_global_load_lds(BG_GPTR(wlane + eo), BG_LPTR(Ws + r0 * C::BM), 16, 0, 0);
        }
      } else {
#pragma unroll
        for (int q = 0; q < C::NIW / 4; ++q) {
          const int r0 = (pw + 4 * q) * RPIW;
          const int k = k0 + r0 + rw;
          const float* src = (mok && k < K) ? wsrc + (int64_t)k * a.w_ld : a.zeros;
          __builtin_amdgcn_global_load_lds(BG_GPTR(src), BG_LPTR(Ws + r0 * C::BM), 16, 0, 0);
        }
      }
      if (xfast && kfull) {
#pragma unroll
        for (int q = 0; q < C::NIX / 4; ++q) {
          const int r0 = (pw + 4 * q) * RPIX;
          const int eo = __builtin_amdgcn_readfirstlane((k0 + r0) * xcs);
          __builtin_amdgcn_global_load_lds(BG_GPTR(xlane + eo), BG_LPTR(Xs + r0 * C::BN), 16, 0, 0);
        }
      } else {
#pragma unroll
        for (int q = 0; q < C::NIX / 4; ++q) {
          const int r0 = (pw + 4 * q) * RPIX;
          const int k = k0 + r0 + rx;
          const float* src = (nok && k < K) ? xsrc + (int64_t)k * a.x_cs : a.zeros;
          __builtin_amdgcn_global_load_lds(BG_GPTR(src), BG_LPTR(Xs + r0 * C::BN), 16, 0, 0);
        }
      }
    };
    const int npre = nstage < C::NST ? nstage : C::NST;
    BG_STAMP(4);
    for (int s = 0; s < npre; ++s) issue(s);
    BG_STAMP(5);
    for (int s = 0; s < nstage; ++s) {
      const int behind = (s + C::NST < nstage ? s + C::NST : nstage) - (s + 1);   // stages in flight behind s
      if (behind >= 3) bg_wait_vm<3 * C::LPW>();
      else if (behind == 2) bg_wait_vm<2 * C::LPW>();
      else if (behind == 1) bg_wait_vm<C::LPW>();
      else bg_wait_vm<0>();
      if (s == 0) BG_STAMP(6);
      bg_barrier();                                       // A_s: stage s is in LDS
      if (s + C::NST < nstage) {
        bg_barrier();                                     // B_s: the consumers are done with this slot
        if (!BG_DBG(a, 1)) issue(s + C::NST);
      }
    }
    return;
  }

  // -------------------------------------------------------------------- consumers
  const int l32 = lane & 31, half = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  const int mw = m0 + wm * (32 * TM), nw = n0 + wn * (32 * TN);   // first row / column of this wave's blocks
  bg_f32x16 acc[TM * TN];
#pragma unroll
  for (int t = 0; t < TM * TN; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  // per-row epilogue operands, fetched now (they arrive under the main loop): lane j keeps those of row mw + j
  // (32 TM <= 64 rows); the epilogue picks its rows with a cross-lane read
  float bvl, c1l = 0.0f;
  {
    const int mc = min(mw + lane, a.Cout - 1);
    bvl = a.bias ? a.bias[mc] : 0.0f;
    if (a.cbias) bvl += a.cbias[(int64_t)b * a.cbias_bs + mc];
    if constexpr (LN) c1l = a.ln_c1[(int64_t)b * a.ln_c1_bs + mc];
  }
  // MOD: (1 + scale_b[c]) of this tile's utterance behind the stages (1 beyond K: those rows are staged as zeros);
  // the first stage barrier below publishes it
  float* const scl = lds + C::NST * C::STAGE;
  if constexpr (MOD) {
    const float* sp = a.ln_scale + (int64_t)b * a.ln_scale_bs;
    for (int c = tid; c < nstage * BG_KS; c += 256) scl[c] = c < K ? 1.0f + sp[c] : 1.0f;
  }
  constexpr int GS = bg_gs(TM, TN);
  // LayerNorm statistics: this wave sums ONE of its column blocks (TN == 2: wave wm takes block wm, its partner of
  // the same column range the other one)
  const int sel = TN == 2 ? wm : 0;
  float s1 = 0.0f, s2 = 0.0f, pivot = 0.0f;
  float A0[GS * TM], B0[GS * TN], A1[GS * TM], B1[GS * TN];
  for (int s = 0; s < nstage; ++s) {
    bg_barrier();                                         // A_s
    if (s == 0) BG_STAMP(1);
    const float* Ws = lds + (s % C::NST) * C::STAGE;
    const float* Xs = Ws + BG_KS * C::BM;
    if (LN && s == 0) pivot = Xs[wn * (32 * TN) + sel * 32 + l32];      // channel 0 of the column
    int rows = K - s * BG_KS;
    rows = rows > BG_KS ? BG_KS : rows;
    const bg_lptr wa = (bg_lptr)(Ws + half * C::BM + wm * (32 * TM) + l32);
    const bg_lptr xa = (bg_lptr)(Xs + half * C::BN + wn * (32 * TN) + l32);
    if (!BG_DBG(a, 2)) {
      bg_read<TM, TN, 0>(A0, B0, wa, xa);
      if constexpr (!LN) {
        bg_stage<TM, TN, false, 0, false>(acc, A0, B0, A1, B1, wa, xa, s1, s2, pivot, 0);
      } else if (BG_DBG(a, 1 << 25)) {   // tuning: the LayerNorm variant WITHOUT its statistics (wrong results): what are they worth?
        bg_stage<TM, TN, false, 0, false>(acc, A0, B0, A1, B1, wa, xa, s1, s2, pivot, 0);
        s2 = 1.0f;
      } else if (rows == BG_KS) {
        const bg_lptr sl = (bg_lptr)(scl + s * BG_KS + half);
        if (sel) bg_stage<TM, TN, true, TN - 1, false, MOD>(acc, A0, B0, A1, B1, wa, xa, s1, s2, pivot, 0, sl);
        else bg_stage<TM, TN, true, 0, false, MOD>(acc, A0, B0, A1, B1, wa, xa, s1, s2, pivot, 0, sl);
      } else {
        const bg_lptr sl = (bg_lptr)(scl + s * BG_KS + half);
        if (sel) bg_stage<TM, TN, true, TN - 1, true, MOD>(acc, A0, B0, A1, B1, wa, xa, s1, s2, pivot, rows >> 1, sl);
        else bg_stage<TM, TN, true, 0, true, MOD>(acc, A0, B0, A1, B1, wa, xa, s1, s2, pivot, rows >> 1, sl);
      }
    }
    if (s + C::NST < nstage) bg_barrier();                // B_s
  }

  BG_STAMP(2);
  // ---- fused LayerNorm: mean / rstd of this lane's columns (both halves of a column hold half of its channels)
  float mean[TN], rstd[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    mean[tn] = 0.0f;
    rstd[tn] = 1.0f;
  }
  if constexpr (LN) {
    const float t1 = s1 + __shfl_xor(s1, 32, 64);
    const float t2 = s2 + __shfl_xor(s2, 32, 64);
    const float dm = t1 / (float)K;
    const float mu = pivot + dm;
    const float var = fmaxf(t2 / (float)K - dm * dm, 0.0f);
    float rs = 1.0f / sqrtf(var + a.ln_eps);
    if constexpr (MOD) {                                  // the column mask of the normalised input (hsp.h ln_mask)
      if (a.ln_mask) rs *= a.ln_mask[(int64_t)b * a.ln_mask_bs + min(nw + sel * 32 + l32, a.ncols - 1)];
    }
    if constexpr (TN == 1) {
      mean[0] = mu;
      rstd[0] = rs;
    } else {
      // the partner wave (same wn, other wm) holds the other block's statistics: swap through a stage slot that
      // nobody reads any more (slot of stage nstage - NST, or one never issued); the producers have ended
      float* xch = lds + (nstage % C::NST) * C::STAGE;
      if (half == 0) {
        xch[((wn * 2 + wm) * 2 + 0) * 32 + l32] = mu;
        xch[((wn * 2 + wm) * 2 + 1) * 32 + l32] = rs;
      }
      bg_barrier();
      const float mu2 = xch[((wn * 2 + (1 - wm)) * 2 + 0) * 32 + l32];
      const float rs2 = xch[((wn * 2 + (1 - wm)) * 2 + 1) * 32 + l32];
      mean[0] = wm == 0 ? mu : mu2;
      mean[TN - 1] = wm == 0 ? mu2 : mu;
      rstd[0] = wm == 0 ? rs : rs2;
      rstd[TN - 1] = wm == 0 ? rs2 : rs;
    }
  }

  // ---- epilogue (hsp_bgemm_try admits these three pointwise functions only)
  if constexpr (EXT) {
    static_assert(!LN, "the extended epilogue is not built with the fused LayerNorm");
    if (a.act == HSP_ACT_RELU) bg_epilogue_ext<TM, TN, HSP_ACT_RELU>(a, acc, b, m0, mw, nw, l32, half, bvl);
    else if (a.act == HSP_ACT_GELU_TANH) bg_epilogue_ext<TM, TN, HSP_ACT_GELU_TANH>(a, acc, b, m0, mw, nw, l32, half, bvl);
    else bg_epilogue_ext<TM, TN, HSP_ACT_NONE>(a, acc, b, m0, mw, nw, l32, half, bvl);
  } else {
    if (a.act == HSP_ACT_RELU) bg_epilogue<TM, TN, LN, HSP_ACT_RELU>(a, acc, b, mw, nw, l32, half, bvl, c1l, mean, rstd);
    else if (a.act == HSP_ACT_GELU_TANH) bg_epilogue<TM, TN, LN, HSP_ACT_GELU_TANH>(a, acc, b, mw, nw, l32, half, bvl, c1l, mean, rstd);
    else bg_epilogue<TM, TN, LN, HSP_ACT_NONE>(a, acc, b, mw, nw, l32, half, bvl, c1l, mean, rstd);
  }
  BG_STAMP(3);
#undef BG_STAMP
}

template <int TM, int TN, bool LN, bool EXT, bool MOD = false>
int bg_launch(const hsp_conv1d_args& a, hipStream_t s, int n_mt, int n_nt, int total) {
  using C = BgCfg<TM, TN>;
  static hsp_lds_flags flags;
  // MOD: + the utterance's (1 + scale) vector behind the stages (Cin <= 1024 rounded up to whole stages: <= 4.2 KB)
  const int lds_bytes = C::LDS_BYTES + (MOD ? ((a.Cin + BG_KS - 1) / BG_KS) * BG_KS * (int)sizeof(float) : 0);
  if (int e = hsp_raise_lds_limit(reinterpret_cast<const void*>(bgemm_kernel<TM, TN, LN, EXT, MOD>), C::LDS_BYTES + (MOD ? 4224 : 0), flags)) return e;
  const int per_xcd = (total + 7) / 8;
  hipLaunchKernelGGL((bgemm_kernel<TM, TN, LN, EXT, MOD>), dim3((unsigned)(8 * per_xcd)), dim3(512), lds_bytes, s, a, n_mt, n_nt,
                     per_xcd, total);
  return (int)hipGetLastError();
}

// does the launch need bg_epilogue_ext?
inline bool bg_extended(const hsp_conv1d_args& a) {
  return a.mask_mode != HSP_MASK_NONE || a.cscale || a.accumulate || a.scale != 1.0f || a.post_scale != 1.0f || a.split_row > 0;
}

template <int TM, int TN>
int bg_go(const hsp_conv1d_args& a, hipStream_t s, int32_t* plan_out) {
  using C = BgCfg<TM, TN>;
  const int n_mt = (a.M + C::BM - 1) / C::BM, n_nt = (a.ncols + C::BN - 1) / C::BN;
  const int64_t total = (int64_t)n_mt * n_nt * a.B;
  if (total <= 0 || total > 0x3fffffff) return -1;
  if (plan_out) {  // {BM, BN, -2 = "block token GEMM", LDS bytes}
    plan_out[0] = C::BM; plan_out[1] = C::BN; plan_out[2] = -2; plan_out[3] = C::LDS_BYTES;
    return 0;
  }
  if (bg_extended(a)) return bg_launch<TM, TN, false, true>(a, s, n_mt, n_nt, (int)total);   // never with ln_c1 (hsp_bgemm_try)
  if constexpr (TM == 1 && TN == 1) {
    if (a.ln_scale) return bg_launch<1, 1, true, false, true>(a, s, n_mt, n_nt, (int)total);   // modulated LayerNorm: 64 x 64 only
  }
  return a.ln_c1 ? bg_launch<TM, TN, true, false>(a, s, n_mt, n_nt, (int)total)
                 : bg_launch<TM, TN, false, false>(a, s, n_mt, n_nt, (int)total);
}

}  // namespace

// Host side: eligibility + tile choice + launch; called by the conv dispatcher ahead of the two latency-oriented
// token GEMMs.  Returns -1 when the shape is not one this kernel takes.
int hsp_bgemm_try(const hsp_conv1d_args& a, hipStream_t s, int32_t* plan_out) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (BG_DBG(a, 1 << 22)) return -1;
  if (a.K != 1 || a.stride != 1 || a.pad != 0 || a.prologue != HSP_PRO_NONE || a.rows != HSP_ROWS_PLAIN) return -1;
  if (a.x_ts != 1 || a.Lin != a.ncols || a.Lout != a.ncols) return -1;
  if ((a.Cin & 3) || (a.ncols & 3) || (a.x_bs & 3) || (a.x_cs & 3) || !al16(a.x) || !al16(a.w) || (a.w_ld & 3)) return -1;
  if (a.Cin < 96 || a.Cin > 8192) return -1;
  if (a.ln_c1 && !(a.ln_eps > 0.0f)) return -1;
  if (a.ln_scale && (!a.ln_c1 || a.bias || !a.cbias || a.Cin > 1024 || a.act != HSP_ACT_NONE || a.res)) return -1;   // hsp.h ln_scale
  if (!a.ln_scale && (a.ln_mask || a.ln_c1_bs)) return -1;
  if (a.mask_mode != HSP_MASK_NONE && !a.mask) return -1;
  if (a.act != HSP_ACT_NONE && a.act != HSP_ACT_RELU && a.act != HSP_ACT_GELU_TANH) return -1;
  // the epilogues address a row as uniform base + 32-bit lane offset and test bounds per group of four rows
  if (a.y_cs >= (1 << 24) || a.res_cs >= (1 << 24) || (a.Cout & 3)) return -1;
  const bool ext = bg_extended(a);
  if (ext && a.ln_c1) return -1;                          // the extended epilogue is not built with the fused LayerNorm
  if (a.split_row) {                                      // second output: whole 64-row tiles (64 x 64 shape only)
    if (a.split_row < 0 || (a.split_row % 64) || a.split_row >= a.Cout || !a.y2 || a.y2_cs >= (1 << 24)) return -1;
    if (a.mask_mode2 != HSP_MASK_NONE && !a.mask) return -1;
  }
#ifdef HSP_TUNING
  if (!a.split_row) {
    if (a.debug & (1 << 18)) return bg_go<2, 2>(a, s, plan_out);
    if (a.debug & (1 << 19)) return bg_go<2, 1>(a, s, plan_out);
    if (a.debug & (1 << 21)) return bg_go<1, 2>(a, s, plan_out);
  }
  if (a.debug & (1 << 23)) return bg_go<1, 1>(a, s, plan_out);
#endif
  // Tile choice (tools/gemm_bench.py, profiles/r03_bgemm_bench.txt).  64 x 64 tiles at two workgroups per CU are the
  // default: a second resident tile covers the first one's 1.8-us start and its epilogue, which a lone 128 x 128
  // tile per CU leaves exposed, and up to 3 200 columns that outweighs the doubled L2 traffic.  Below ~96 tiles the
  // latency-oriented kernels win (7-11 us floors against 9.5-11.5 here); from ~850 tiles (1104 x 3 200) the
  // 128 x 128 shape is ahead (33 against 35 us).
  auto tiles = [&](int bm, int bn) { return (int64_t)((a.M + bm - 1) / bm) * ((a.ncols + bn - 1) / bn) * a.B; };
  auto waste_ok = [&](int bm) { return 4 * (int64_t)(((a.M + bm - 1) / bm) * bm - a.M) <= a.M; };
  const int64_t t64 = tiles(64, 64);
  if (a.ln_scale) return bg_go<1, 1>(a, s, plan_out);     // the modulated LayerNorm: this kernel's 64 x 64 shape or nothing
  if (t64 < 96 && !a.split_row) return -1;   // (a second-output launch has no other kernel: better one launch here than two)
  // (round 4, tools/gemm_sweep.py over B in {1..64} x T in {50, 200, 1000}: profiles/r04_gemm_dispatch_table.txt)  The
  // 128 x 128 shape runs one workgroup per CU, so what it costs is ROUNDS of 256 tiles: it wins only for many rows
  // (M >= 768: the PLM's ff.0 / q-k-v) and when its last round is nearly full -- 4 000 columns of ff.0 are 288 tiles
  // = two rounds for the work of 1.1 (60.5 against 44.9 us on 64 x 64), 3 200 columns are 225 tiles = one (32.4
  // against 35.9).  Round 3's rule (from 850 small tiles upward) was up to 1.48 x off the better kernel.
  // very large launches of a short K (the 192-channel 1x1s over >= 5 000 small tiles): the conv kernel's 128 x 128
  // tiles at two per CU are ahead (WN res_skip at 64 x 1 000 frames: 95 against 117 us)
  if (t64 >= 5000 && a.Cin <= 256 && !a.ln_c1 && !a.split_row && !bg_extended(a)) return -1;
  if (t64 >= 700 && a.M >= 768 && waste_ok(128) && a.ncols >= 512 && !a.split_row) {
    const int64_t t128 = tiles(128, 128);
    const int64_t rounds = (t128 + 255) / 256;
    if (100 * rounds * 256 <= 115 * t128) return bg_go<2, 2>(a, s, plan_out);
  }
  return bg_go<1, 1>(a, s, plan_out);
}
