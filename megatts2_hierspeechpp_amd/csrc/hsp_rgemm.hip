// Register-path fp32-MFMA GEMM for the token matrices of the 50 Hz part (the PLM loop above all):
//   Y[M x N] = epilogue(W[M x K] X[K x N]),  K = 192 .. 1104, N = 16 .. a few thousand columns.
//
// Why a second token-GEMM kernel.  hsp_tokgemm.hip stages both operands through LDS with LDS-DMA; measured
// (round 3, tools/conv_sweep.py on the PLM shapes, libhsp_tune.so): a launch with the DMA and the MFMAs switched
// off still takes 12.5 us, because every tile first pulls three stages (147 KB) through a path that sustains
// ~20 GB/s per CU -- 7 us before the first MFMA -- and a K = 1104 tile needs 565 KB = 28 us however few tiles the
// launch has.  LDS-DMA is the right tool where a staged byte is reused 64-128 times (the conv kernel needs ~10 GB/s
// per CU); a 64 x 64 GEMM tile reuses it 32 times and starves.
//
// Here nothing is staged.  The packed weight is [K][M] (rows fastest) and the activations are channel-major
// [K][N], so BOTH MFMA fragments of v_mfma_f32_32x32x2_f32 are plain coalesced dword loads: lane (l32, half) reads
// W[2 kk + half][m0 + l32] and X[2 kk + half][n0 + l32] -- 128 contiguous bytes per half wave, straight from L2
// (weights) / L2-MALL (activations) into the registers the MFMA reads.  A workgroup is 8 waves on ONE small output
// tile (32 x 32 or 64 x 32): the waves split the tile's 32 x 32 blocks AND its K range, each wave runs a
// register-prefetched chain of K / (2 ksplit) MFMAs, and the partial sums meet in LDS.  Many small workgroups
// instead of a few pipelined ones: 4 of them share a CU, so one wave's load latency is another's MFMA time, and a
// launch is one L2 round trip + a short MFMA chain + one LDS exchange.
//
// Epilogue = hsp_tokgemm.hip's (bias, conditioning bias, fused input LayerNorm, pointwise function, masks,
// per-(b, c) scale, residual, accumulate), same operation order.  The LayerNorm statistics come from the B
// fragments the waves load anyway (pivot-shifted sums, as in the LDS kernel).
#include "hsp_device.h"

#ifdef HSP_TUNING
#define RG_DBG(a, bit) (((a).debug & (bit)) != 0)
#else
#define RG_DBG(a, bit) false
#endif

namespace {

typedef float rg_f32x16 __attribute__((ext_vector_type(16)));
constexpr int RG_WAVES = 8;
// RG_U = k-steps per register group: 2 RG_U loads requested together, then their RG_U MFMAs.  A launch at these sizes
// is a chain of L2 round trips (1-2 us each) around 0.5 us of MFMAs, so the group covers a wave's WHOLE share of K where
// registers allow: 18 k-steps at K = 276 with the K range split eight ways (one round trip; groups of 8 with the next
// group prefetched made it three -- the prefetch hides 0.2 us of MFMAs, not the trip), 36 for K > 512 (K = 1104: two).

// cfg: nbm x nbn blocks of 32 x 32 per workgroup (1 x 1 or 2 x 1), ksplit = 8 / (nbm * nbn) waves per block
template <int RG_U>
__global__ __launch_bounds__(64 * RG_WAVES) void rgemm_kernel(const hsp_conv1d_args a, int n_mt, int n_nt, int nbm,
                                                              int nbn) {
  __shared__ float part[RG_WAVES][16][64];      // partial accumulators (32 KB)
  __shared__ float stat[RG_WAVES][2][32];       // LayerNorm partial sums per wave
  const int nblk = nbm * nbn, ksplit = RG_WAVES / nblk;
  int bid = blockIdx.x;
  const int mt = bid % n_mt;
  bid /= n_mt;
  const int nt = bid % n_nt;
  const int b = bid / n_nt;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, l32 = lane & 31, half = lane >> 5;
  const int blk = wave % nblk, ks = wave / nblk;
  const int bm = blk % nbm, bn = blk / nbm;
  const int mb = mt * (32 * nbm) + bm * 32;     // first row / column of this wave's block
  const int nb = nt * (32 * nbn) + bn * 32;
  const int K = a.Cin;
  const int ksteps = (K + 1) >> 1;
  const int per = (ksteps + ksplit - 1) / ksplit;
  const int kk0 = ks * per, kk1 = min(kk0 + per, ksteps);

#ifdef HSP_TUNING
  // tuning bit 1 << 20: cycle-counter stamps of workgroup (gridDim.x / 2), wave 0, lane 0 -> a.filt (6 x uint64)
  unsigned long long* stamps = ((a.debug & (1 << 20)) && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0)
                                   ? (unsigned long long*)a.filt : nullptr;
#define RG_STAMP(i) do { if (stamps) stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define RG_STAMP(i) do { } while (0)
#endif
  RG_STAMP(0);
  // clamped (always legal) fragment addresses: rows >= M / columns >= ncols compute garbage nobody stores
  const int m = min(mb + l32, a.M - 1);
  const int n = min(nb + l32, a.ncols - 1);
  const float* wp = a.w + m + (int64_t)half * a.w_ld;
  const float* xp = a.x + (int64_t)b * a.x_bs + (int64_t)n * a.x_ts + (int64_t)half * a.x_cs;
  const int64_t wst = 2 * (int64_t)a.w_ld, xst = 2 * a.x_cs;
  const bool odd_tail = (K & 1) != 0;           // the last k-step's second channel does not exist

  // ---- epilogue operands of the accumulator rows this wave finalises (r = ks, ks + ksplit, ...): fetched now
  const int nr = 16 / ksplit;                   // 2 or 4 (ksplit 8 / 4)
  const int ncol = nb + l32;
  const int nc = min(ncol, a.ncols - 1);
  const float mk = a.mask_mode != HSP_MASK_NONE ? a.mask[(int64_t)b * a.mask_bs + nc] : 1.0f;
  const float* resb = a.res ? a.res + (int64_t)b * a.res_bs + (int64_t)nc * (a.res_ts > 1 ? a.res_ts : 1) : nullptr;
  float* yb = a.y + (int64_t)b * a.y_bs + nc;
  float bv[4], rv[4], yv[4], cs[4], c1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bv[i] = rv[i] = yv[i] = c1[i] = 0.0f;
    cs[i] = 1.0f;
    if (i < nr) {
      const int r = ks + i * ksplit;
      const int mr = min(mb + (r & 3) + 8 * (r >> 2) + 4 * half, a.Cout - 1);
      float t = a.bias ? a.bias[mr] : 0.0f;
      if (a.cbias) t += a.cbias[(int64_t)b * a.cbias_bs + mr];
      bv[i] = t;
      if (a.cscale) cs[i] = a.cscale[(int64_t)b * a.cscale_bs + mr];
      if (resb) rv[i] = resb[(int64_t)mr * a.res_cs];
      if (a.accumulate) yv[i] = yb[(int64_t)mr * a.y_cs];
      if (a.ln_c1) c1[i] = a.ln_c1[mr];
    }
  }
  const float pivot = a.ln_c1 ? a.x[(int64_t)b * a.x_bs + (int64_t)n * a.x_ts] : 0.0f;   // channel 0 of this lane's column

  // ---- main loop: groups of RG_U k-steps, the next group's fragments in flight under this group's MFMAs
  rg_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float s1 = 0.0f, s2 = 0.0f;
  float fa[RG_U], fb[RG_U];
  auto load = [&](int kk, float (&A)[RG_U], float (&B)[RG_U]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < RG_U; ++u) {
      int k = min(kk + u, kk1 - 1);             // clamped: the surplus steps of the last group are zeroed below
      // odd K: the second channel of the last k-step does not exist -- W row K / x channel K would be one row past the
      // packed weight and the activations.  Read the previous step's (legal) address instead; the value is zeroed below.
      if (odd_tail && half == 1 && k == ksteps - 1) k -= 1;
      A[u] = wp[(int64_t)k * wst];
      B[u] = xp[(int64_t)k * xst];
    }
  };
  if (!RG_DBG(a, 2)) {
#pragma unroll 1
    for (int kk = kk0; kk < kk1; kk += RG_U) {
      load(kk, fa, fb);
#pragma unroll
      for (int u = 0; u < RG_U; ++u) {
        const bool ok = kk + u < kk1 && !(odd_tail && half == 1 && kk + u == ksteps - 1);
        const float av = ok ? fa[u] : 0.0f;
        const float bvv = ok ? fb[u] : 0.0f;
        if (a.ln_c1 && ok) {
          const float d = bvv - pivot;
          s1 += d;
          s2 = fmaf(d, d, s2);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bvv, acc, 0, 0, 0);
      }
    }
  }

  RG_STAMP(1);
  // ---- K-split exchange through LDS
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][r][lane] = acc[r];
  if (a.ln_c1) {
    s1 += __shfl_xor(s1, 32, 64);               // both halves of a column
    s2 += __shfl_xor(s2, 32, 64);
    if (half == 0) { stat[wave][0][l32] = s1; stat[wave][1][l32] = s2; }
  }
  RG_STAMP(2);
  __syncthreads();
  RG_STAMP(3);
  float mean = 0.0f, rstd = 1.0f;
  if (a.ln_c1) {
    float t1 = 0.0f, t2 = 0.0f;
    for (int q = 0; q < ksplit; ++q) {          // the waves of this block, in a fixed order
      t1 += stat[blk + q * nblk][0][l32];
      t2 += stat[blk + q * nblk][1][l32];
    }
    const float dm = t1 / (float)K;
    mean = pivot + dm;
    const float var = fmaxf(t2 / (float)K - dm * dm, 0.0f);
    rstd = 1.0f / sqrtf(var + a.ln_eps);
  }
  if (ncol >= a.ncols) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i >= nr) break;
    const int r = ks + i * ksplit;
    const int mr = mb + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (mr >= a.Cout) continue;
    float fin = 0.0f;
    for (int q = 0; q < ksplit; ++q) fin += part[blk + q * nblk][r][lane];   // fixed order: deterministic
    float v = a.ln_c1 ? fmaf(rstd, fmaf(-mean, c1[i], fin), bv[i]) : fin + bv[i];
    v = hsp_apply_act(v, a.act);                // then the order of hsp_epilogue_store
    if (a.mask_mode & HSP_MASK_PRE) v *= mk;
    v *= cs[i];
    v *= a.scale;
    v += rv[i];
    if (a.mask_mode & HSP_MASK_POST) v *= mk;
    v += yv[i];
    yb[(int64_t)mr * a.y_cs] = v * a.post_scale;
  }
  RG_STAMP(4);
#undef RG_STAMP
}

}  // namespace

// Host side: eligibility + launch; called by the conv dispatcher ahead of the LDS-DMA token GEMM.
// Returns -1 when the shape is not one this kernel takes.
int hsp_rgemm_try(const hsp_conv1d_args& a, hipStream_t s, int32_t* plan_out) {
  if (a.K != 1 || a.stride != 1 || a.pad != 0 || a.prologue != HSP_PRO_NONE || a.rows != HSP_ROWS_PLAIN) return -1;
  if (a.x_ts < 1 || a.Lin != a.ncols || a.Lout != a.ncols || a.split_row) return -1;   // any column stride of x
  if (a.Cin < 32 || a.Cin > 8192) return -1;
  if (a.ln_c1 && !(a.ln_eps > 0.0f)) return -1;
  if (a.mask_mode != HSP_MASK_NONE && !a.mask) return -1;
  // Where it wins (tools/gemm_bench.py, hipGraph replay, profiles/r03_gemm_bench.txt): up to ~1.4 M outputs per
  // launch -- 7-8 us against 12-15 us for the LDS-DMA kernel at a few hundred columns (its floor is the three-stage
  // prologue), 14 against 29 us at K = 1104 -- and a tie around 1 600 x 828; beyond that the LDS kernel's operand
  // reuse wins (3 200 columns: 34-49 us against 43-52), so larger launches stay there.
  auto tiles = [&](int bm, int bn) { return (int64_t)((a.M + bm - 1) / bm) * ((a.ncols + bn - 1) / bn) * a.B; };
  const bool forced = RG_DBG(a, 8 | 16) || (a.res && a.res_ts > 1);   // a strided residual has no other kernel
  const int64_t outs = (int64_t)a.M * a.ncols * a.B;
  if (!forced && a.x_ts == 1 && (outs > 1400000 || outs * a.Cin > 500000000)) return -1;
  // many short utterances ([B, C, T <= 64]: every utterance is two ragged 32-column tiles): beyond ~400 such tiles the
  // conv kernel's tiles win by 1.4-1.6 x, below the register path does by up to 1.8 x (profiles/r04_gemm_dispatch_table.txt)
  // (a fused input LayerNorm that the block GEMM did not take has no other kernel: e.g. the ONE new column per utterance of
  // the PLM loop's cached layer 0 -- B utterances x 1 column, 26 x 16 tiles -- round 6)
  if (!forced && !a.ln_c1 && a.ncols <= 64 && tiles(32, 32) > 400) return -1;
  // tile: 64 x 32 where that still fills the chip, else 32 x 32 (a 64 x 64 / K-split-2 form was measured and lost)
  int nbm = 1, nbn = 1;
  if (tiles(64, 32) >= 200) { nbm = 2; nbn = 1; }
#ifdef HSP_TUNING
  if (a.debug & 8) { nbm = 2; nbn = 1; }
  if (a.debug & 16) { nbm = 1; nbn = 1; }
#endif
  const int n_mt = (a.M + 32 * nbm - 1) / (32 * nbm), n_nt = (a.ncols + 32 * nbn - 1) / (32 * nbn);
  const int64_t blocks = (int64_t)n_mt * n_nt * a.B;
  if (blocks <= 0 || blocks > 0x7fffffff) return -1;
  if (plan_out) {  // {BM, BN, -1 = "register-path token GEMM" (0 = the LDS-DMA one, > 0 = the conv kernel's chunk depth), LDS bytes}
    plan_out[0] = 32 * nbm; plan_out[1] = 32 * nbn; plan_out[2] = -1; plan_out[3] = (int32_t)(sizeof(float) * (RG_WAVES * 16 * 64 + RG_WAVES * 64));
    return 0;
  }
  if (a.Cin > 512) hipLaunchKernelGGL(rgemm_kernel<36>, dim3((unsigned)blocks), dim3(64 * RG_WAVES), 0, s, a, n_mt, n_nt, nbm, nbn);
  else hipLaunchKernelGGL(rgemm_kernel<18>, dim3((unsigned)blocks), dim3(64 * RG_WAVES), 0, s, a, n_mt, n_nt, nbm, nbn);
  return (int)hipGetLastError();
}
