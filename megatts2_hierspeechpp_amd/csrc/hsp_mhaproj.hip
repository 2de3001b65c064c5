// hsp_mha_proj_f32: self-attention over ALL heads of one utterance's query tile + the output projection + its
// epilogue (bias, mask, per-(b, c) scale, residual) in ONE launch (round 4).
//
// Replaces two launches per layer of the Mega-TTS2 PLM loop -- scaled_dot_product_attention + out_proj + residual,
// ttv_v1/transformer_mega.py:63-87,121-123 -- and of a DiT block -- timm Attention's softmax(q k^T) v + proj, then
// `x + gate_msa * attn(...)`, modules.py:397,409.  A workgroup owns 16 queries of one utterance for every head, so the
// H * D input rows of the projection are local: no grid-wide step between the two products.
//
// Everything runs on v_mfma_f32_16x16x4_f32 (exact fp32; A[l & 15][k = l >> 4], B[k = l >> 4][l & 15], C/D: column
// l & 15, rows 4 (l >> 4) + reg).  Lane l = (x = l & 15, g = l >> 4).  Three products, chained WITHOUT a transpose:
//
//   1. S^T[key][q] = sum_d K[d][key] * (scale Q[d][q]).   A = K^T, B = Q.  A lane's A values for FOUR MFMAs come from
//      one 16-B load of K[d = 4 ks + g][key0 .. key0 + 3], key0 = 64 kg + 4 x: the four score tiles j = 0..3 of a
//      64-key group hold the keys {64 kg + 4 m + j}.  Accumulator (j, reg i) of lane (q, g) = key 64 kg + 16 g + 4 i + j.
//   2. softmax over keys in registers: lane-local over (group, j, i), then across g with two xor-shuffles.
//   3. O[d][q] = sum_key V[d][key] P[key][q].   B = the score accumulator (j, i) as it stands: its k-slot g holds
//      key 64 kg + 16 g + 4 i + j, so the A lane (d, g) needs V[d][64 kg + 16 g + 4 i + j]: component j of ONE
//      16-B load at key 64 kg + 16 g + 4 i (the trick of mha_tok_kernel: an MFMA's k-slots may pair any keys as long
//      as both operands pair the same ones).
//   4. Y[m][q] = sum_c Wt[m][c] O[c][q] with O in LDS as [c][16 q] and Wt = the nn.Linear weight as stored
//      ([out][in] row-major): A lane (m, g) loads Wt[m][16 cb + 4 g .. + 3] (16 B) for four MFMAs, whose B values are
//      O[16 cb + 4 g + i][q].
//
// Eight waves = heads x key splits (PLM: 4 x 2, DiT: 2 x 4); a wave's partial (max, sum, O) over its keys is merged
// through LDS.  In the projection the eight waves split the row blocks.  What bounds the launch is not MFMA time but
// what ONE workgroup can pull from L2: K and V of every head + the whole projection matrix, ~600 KB at 100 keys
// (tools/micro/cu_fetch_bw.hip: 55-64 GB/s per workgroup of eight waves) -- so the first projection fragments are
// requested before the attention phases, and lanes whose keys lie beyond Tk read one shared address.
#include <type_traits>
#include "hsp_device.h"

namespace {
typedef float mp_f32x4 __attribute__((ext_vector_type(4)));
typedef float mp_f4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int MP_QT = 16;  // queries per workgroup
constexpr int MP_NW = 8;   // waves
constexpr float MP_LOG2E = 1.4426950408889634f;
constexpr float MP_NEG = -3.0e38f;

template <int H_, int D_>
struct MpCfg {
  static constexpr int H = H_, D = D_, C = H_ * D_;
  static constexpr int KS = MP_NW / H_;             // waves (key splits) per head
  static constexpr int NK = (D_ + 3) / 4;           // k-steps of the score product
  static constexpr int NDB = (D_ + 15) / 16;        // head-dim blocks
  static constexpr int NCB = (C + 15) / 16;         // 16-channel blocks of the projection's K axis
  static constexpr int NRBT = (C + 15) / 16;        // its row blocks (M == C)
  static constexpr int NRB = (NRBT + MP_NW - 1) / MP_NW;
  static constexpr int OROWS = NCB * 16;
  static constexpr int PF = 8;                      // projection fragments requested ahead (k blocks)
  // VE head-dim blocks of a key group's V are requested one group ahead (the first group's: together with Q and K, one
  // round trip), the other NDB - VE follow behind the group's score MFMAs: K + V + Q fragments of a lane must leave
  // room for the accumulators and addresses under the 256-register budget (-Rpass-analysis=kernel-resource-usage)
  static constexpr int VE_ = (170 - 5 * NK) / 16;
  static constexpr int VE = VE_ < 0 ? 0 : (VE_ > NDB ? NDB : VE_);
  // LDS floats: O [OROWS][16] | partial O [NW][NDB * 4][64] | partial max / sum [NW][16] each
  static constexpr int LDS_O = OROWS * MP_QT;
  static constexpr int LDS_P = MP_NW * NDB * 4 * 64;
  static constexpr int LDS_TOTAL = LDS_O + LDS_P + 2 * MP_NW * MP_QT;
  static_assert(MP_NW % H_ == 0, "waves = heads x key splits");
  static_assert(C % 4 == 0, "16-B projection fragments never straddle the end of a weight row");
};

__device__ __forceinline__ mp_f32x4 mp_zero4() { return mp_f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }

// out[j] = v[j + sh] while j + sh <= 3, else 0: a 16-B window that was moved back by sh columns to stay inside the row.
// A two-stage barrel shifter of selects (written as a chain of `idx == k ? v[k] : ...` hipcc turns it into a cascade of
// divergent branches: 14 000 lines of ISA for this kernel).
__device__ __forceinline__ mp_f32x4 mp_shift(mp_f32x4 v, int sh) {
  const bool s1 = (sh & 1) != 0, s2 = (sh & 2) != 0, s4 = sh >= 4;
  const float a0 = s1 ? v[1] : v[0], a1 = s1 ? v[2] : v[1], a2 = s1 ? v[3] : v[2], a3 = s1 ? 0.0f : v[3];
  mp_f32x4 r;
  r[0] = s4 ? 0.0f : (s2 ? a2 : a0);
  r[1] = s4 ? 0.0f : (s2 ? a3 : a1);
  r[2] = (s4 || s2) ? 0.0f : a2;
  r[3] = (s4 || s2) ? 0.0f : a3;
  return r;
}

// workgroup barrier that does NOT drain this wave's outstanding global loads (__syncthreads() waits vmcnt(0): here that
// would be the projection fragments and epilogue operands requested just before -- a whole L2 round trip, 6 us measured)
__device__ __forceinline__ void mp_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <class Cfg>
__global__ __launch_bounds__(64 * MP_NW) void mha_proj_kernel(const hsp_mha_proj_args a, int n_qt) {
  constexpr int D = Cfg::D, C = Cfg::C, KS = Cfg::KS, NK = Cfg::NK, NDB = Cfg::NDB, VE = Cfg::VE;
  constexpr int NCB = Cfg::NCB, NRBT = Cfg::NRBT, NRB = Cfg::NRB, PF = Cfg::PF;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const Oall = lds;                            // [OROWS][16]
  float* const Opart = lds + Cfg::LDS_O;              // [NW][NDB * 4][64]
  float* const Mpart = Opart + Cfg::LDS_P;            // [NW][16]
  float* const Lpart = Mpart + MP_NW * MP_QT;         // [NW][16]

  // utterance fastest: with B a multiple of 8 the query tiles of one utterance share an XCD (and its L2 copy of K, V)
  const int b = blockIdx.x % a.B;
  const int qt = blockIdx.x / a.B;
  const int i0 = qt * MP_QT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, g = lane >> 4;
  // wave -> (head, key split): waves w and w + 4 share a SIMD, and with few keys only the first splits have work --
  // heads vary fastest so that every SIMD hosts a first split (split-major order left SIMDs 1 and 3 idle through the
  // attention phases of the PLM's first 64 steps while 0 and 2 multiplied two heads each)
  const int h = wave % Cfg::H, ksp = wave / Cfg::H;
  const int Tq = a.Tq, Tk = a.Tk;
  const int qi = min(i0 + x, Tq - 1);                 // this lane's query (clamped: surplus columns compute garbage nobody stores)
#ifdef HSP_TUNING
  // tuning build, debug bit 1: cycle-counter stamps of the middle workgroup's wave 0 -> a.cscale (10 x uint64; cscale unused)
  unsigned long long* stamps = ((a.debug & 1) && blockIdx.x == gridDim.x / 2 && tid == 0) ? (unsigned long long*)a.cscale : nullptr;
  const float* const cscale = (a.debug & 1) ? nullptr : a.cscale;
#define MP_STAMP(i) do { if (stamps) stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
  const float* const cscale = a.cscale;
#define MP_STAMP(i) do { } while (0)
#endif
  MP_STAMP(0);

  // Every global load of this kernel is UNCONDITIONAL (clamped addresses, results selected afterwards): a load inside a
  // branch -- divergent or wave-uniform -- is waited for at once (s_waitcnt vmcnt(0) at the join), and the launch is a
  // chain of L2 round trips as it is.  Loads are issued in program order where they should leave; compiler fences
  // (mp_fence) keep memory operations from drifting across the phase boundaries.
  auto mp_fence = []() __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);                // MFMAs too: a fragment register is re-requested only after its last use was issued
  };

  const int ngt = (Tk + 63) >> 6;                     // 64-key groups in all; wave (h, ksp) owns ksp, ksp + KS, ...
  const float* qh = a.q + (int64_t)b * a.q_bs + (int64_t)h * D * a.q_cs + qi;
  const float* kh = a.k + (int64_t)b * a.k_bs + (int64_t)h * D * a.k_cs;
  const float* vh = a.v + (int64_t)b * a.v_bs + (int64_t)h * D * a.v_cs;
  const bool ragged = (Tk & 3) != 0;                  // some 16-B window straddles the end of the row
  // offsets inside one utterance's planes are 32-bit (the entry point checks C * stride < 2^31): 64-bit multiplies
  // in the address arithmetic cost VALU time and -- worse -- hipcc parks the high half of a 64-bit product in the
  // destination register of a load in flight and then waits vmcnt(0) for it
  const int qcs = (int)a.q_cs, kcs = (int)a.k_cs, vcs = (int)a.v_cs;

  // K fragments of key group kg: lane (x, g) reads K[4 ks + g][64 kg + 4 x .. + 3] (window moved back inside the row)
  auto kload = [&](int kg, mp_f32x4 (&kf)[NK]) __attribute__((always_inline)) {
    const int a0 = max(min(64 * kg + 4 * x, Tk - 4), 0);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks)
      kf[ks] = *reinterpret_cast<const mp_f4u*>(kh + (min(4 * ks + g, D - 1) * kcs + a0));
  };
  // V fragments of (key group kg, head-dim block db): lane (x, g) reads V[16 db + x][64 kg + 16 g + 4 i .. + 3]
  auto vload = [&](int kg, int db, mp_f32x4 (&vf)[4]) __attribute__((always_inline)) {
    const float* vrow = vh + min(16 * db + x, D - 1) * vcs;
#pragma unroll
    for (int i = 0; i < 4; ++i) vf[i] = *reinterpret_cast<const mp_f4u*>(vrow + max(min(64 * kg + 16 * g + 4 * i, Tk - 4), 0));
  };

  // ---- requests of the first round trip: Q, K and (as far as registers allow) V of this wave's first key group
  float fq[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) fq[ks] = qh[min(4 * ks + g, D - 1) * qcs];
  mp_f32x4 kf[NK];
  kload(ksp, kf);
  mp_f32x4 vf[NDB][4];
#pragma unroll
  for (int db = 0; db < VE; ++db) vload(ksp, db, vf[db]);
  mp_fence();
  // rows [C, OROWS) of the projection's B operand are zero
  for (int e = tid; e < (Cfg::OROWS - C) * MP_QT; e += 64 * MP_NW) Oall[C * MP_QT + e] = 0.0f;
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) fq[ks] = 4 * ks + g < D ? fq[ks] * a.qk_scale : 0.0f;
  MP_STAMP(1);

  // ---- 1.-3. this wave's key groups kg = ksp, ksp + KS, ... one after the other with a running (max, sum, O) -- the
  // online softmax: any number of keys, a fixed register budget.  The first group's K and V travel in the first round
  // trip; a later group requests its K behind the previous group's last MFMAs and its V behind its own score MFMAs
  // (K and V fragments of two groups do not fit the register file side by side), so every group after the first
  // exposes a round trip: Tk <= 64 KS (128 keys in the PLM shape, 256 in the DiT shape) is the fast case.
  float m_run = MP_NEG, l_run = 0.0f;                 // l_run: this LANE's share of the sum (reduced over g at the end)
  mp_f32x4 oacc[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db) oacc[db] = mp_zero4();
  const int ngw = ngt > ksp ? (ngt - ksp + KS - 1) / KS : 0;   // wave-uniform
  auto group = [&](int kg, auto first_tag) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const bool tail = ragged && 64 * kg + 64 > Tk;    // wave-uniform: the group that holds the end of a ragged row
    if (tail) {
      const int key0 = 64 * kg + 4 * x;
      const int sh = key0 - max(min(key0, Tk - 4), 0);
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) kf[ks] = mp_shift(kf[ks], sh);
    }
    mp_f32x4 sacc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sacc[j] = mp_zero4();
#pragma unroll
    for (int ks = 0; ks < NK; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j) sacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[ks][j], fq[ks], sacc[j], 0, 0, 0);
    mp_fence();
#pragma unroll
    for (int db = FIRST ? VE : 0; db < NDB; ++db) vload(kg, db, vf[db]);   // what did not travel ahead
    mp_fence();
    // softmax step: new running maximum (uniform over the four lanes g of a query: the PV MFMAs sum over g)
    float mg = MP_NEG;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float sv = 64 * kg + 16 * g + 4 * i + j < Tk ? sacc[j][i] : MP_NEG;
        sacc[j][i] = sv;
        mg = fmaxf(mg, sv);
      }
    mg = fmaxf(mg, __shfl_xor(mg, 16, 64));
    mg = fmaxf(mg, __shfl_xor(mg, 32, 64));
    const float m_new = fmaxf(m_run, mg);             // finite: every group holds at least one real key
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * MP_LOG2E);   // first group: 2^(-huge) = 0
    m_run = m_new;
    float ls = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float sv = sacc[j][i];
        const float e = sv > -1.0e38f ? __builtin_amdgcn_exp2f((sv - m_new) * MP_LOG2E) : 0.0f;
        sacc[j][i] = e;
        ls += e;
      }
    l_run = fmaf(l_run, alpha, ls);
    if (!FIRST) {
#pragma unroll
      for (int db = 0; db < NDB; ++db) oacc[db] *= alpha;
    }
    // O^T += V P^T
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
      if (tail) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int key0 = 64 * kg + 16 * g + 4 * i;
          vf[db][i] = mp_shift(vf[db][i], key0 - max(min(key0, Tk - 4), 0));
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (64 * kg + 4 * i < Tk) {                   // wave-uniform: some k-slot of these MFMAs holds a real key
#pragma unroll
          for (int j = 0; j < 4; ++j) oacc[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[db][i][j], sacc[j][i], oacc[db], 0, 0, 0);
        }
      }
    }
    mp_fence();
    kload(kg + KS, kf);                               // the next group's K (an address inside the row when there is none)
    mp_fence();
  };
  if (ngw > 0) group(ksp, std::true_type{});
#pragma unroll 1
  for (int gi = 1; gi < ngw; ++gi) group(ksp + KS * gi, std::false_type{});
  float mx = m_run, sum = l_run;
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  MP_STAMP(4);
  // ---- requests of the projection phase: the first PF k blocks of this wave's weight rows and every epilogue operand
  const float* wrow[NRB];
#pragma unroll
  for (int r = 0; r < NRB; ++r) wrow[r] = a.wt + min(16 * (wave + MP_NW * r) + x, C - 1) * a.wt_ld;
  mp_f32x4 wf[PF][NRB];
  auto wload = [&](int cb, mp_f32x4 (&dst)[NRB]) __attribute__((always_inline)) {
    const int c0 = min(16 * cb + 4 * g, C - 4);       // beyond C (last block only): any legal address, the O rows there are zero
#pragma unroll
    for (int r = 0; r < NRB; ++r) dst[r] = *reinterpret_cast<const mp_f4u*>(wrow[r] + c0);
  };
#pragma unroll
  for (int p = 0; p < PF; ++p) wload(p, wf[p]);
  // epilogue operands (unconditional, clamped; NULL operands read the weight matrix and are ignored): bias and cscale
  // of a row block's four rows 4 g .. 4 g + 3 are one 16-B load each, the residual is four rows apart
  mp_f32x4 e_bias[NRB], e_cs[NRB];
  float e_res[NRB][4], e_mk;
  {
    const float* mp = a.mask ? a.mask + (int64_t)b * a.mask_bs + qi : a.wt;
    e_mk = *mp;
    const float* rb = (a.res ? a.res : a.wt) + (a.res ? (int64_t)b * a.res_bs + (int64_t)qi * a.res_ts : 0);
    const int rcs = a.res ? (int)a.res_cs : 0;
    const float* cs = cscale ? cscale + (int64_t)b * a.cscale_bs : a.wt;
    const float* bs = a.bias ? a.bias : a.wt;
#pragma unroll
    for (int r = 0; r < NRB; ++r) {
      const int m0 = min(16 * (wave + MP_NW * r) + 4 * g, C - 4);      // C % 4 == 0: a row block's 4-groups are whole or absent
      e_bias[r] = *reinterpret_cast<const mp_f4u*>(bs + (a.bias ? m0 : 0));
      e_cs[r] = *reinterpret_cast<const mp_f4u*>(cs + (cscale ? m0 : 0));
#pragma unroll
      for (int i = 0; i < 4; ++i) e_res[r][i] = rb[(m0 + i) * rcs];
    }
  }
  mp_fence();
  // partials -> LDS.  Keys beyond Tk carry P = 0, so whatever their V window held contributes 0 * finite = 0.
  {
    float* op = Opart + wave * (NDB * 4 * 64) + lane;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int i = 0; i < 4; ++i) op[(db * 4 + i) * 64] = oacc[db][i];
    if (g == 0) {
      Mpart[wave * MP_QT + x] = mx;
      Lpart[wave * MP_QT + x] = sum;
    }
  }
  mp_barrier();
  MP_STAMP(5);
  // ---- merge the KS partials of every head: wave (h, ksp) takes the head-dim blocks ksp, ksp + KS, ...
  {
    float mt = MP_NEG;
#pragma unroll
    for (int s = 0; s < KS; ++s) mt = fmaxf(mt, Mpart[(h + Cfg::H * s) * MP_QT + x]);
    float f[KS], lt = 0.0f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      f[s] = __builtin_amdgcn_exp2f((Mpart[(h + Cfg::H * s) * MP_QT + x] - mt) * MP_LOG2E);   // a wave without keys: 2^(-huge) = 0
      lt = fmaf(Lpart[(h + Cfg::H * s) * MP_QT + x], f[s], lt);
    }
    const float inv = 1.0f / lt;
#pragma unroll
    for (int db = ksp; db < NDB; db += KS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float o = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; ++s) o = fmaf(Opart[(h + Cfg::H * s) * (NDB * 4 * 64) + (db * 4 + i) * 64 + lane], f[s], o);
        const int d = 16 * db + 4 * g + i;
        if (d < D) Oall[(h * D + d) * MP_QT + x] = o * inv;
      }
    }
  }
  mp_barrier();
  MP_STAMP(6);

  // ---- 4. projection: wave w owns the row blocks w, w + 8, ...
  mp_f32x4 yacc[NRB];
#pragma unroll
  for (int r = 0; r < NRB; ++r) yacc[r] = mp_zero4();
  const float* ob = Oall + (4 * g) * MP_QT + x;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    float of[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) of[i] = ob[(16 * cb + i) * MP_QT];
#pragma unroll
    for (int r = 0; r < NRB; ++r) {
      if (wave + MP_NW * r < NRBT) {                  // wave-uniform, MFMAs only
#pragma unroll
        for (int i = 0; i < 4; ++i) yacc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[cb % PF][r][i], of[i], yacc[r], 0, 0, 0);
      }
    }
    if (cb + PF < NCB) {
      mp_fence();
      wload(cb + PF, wf[cb % PF]);
      mp_fence();
    }
  }
  MP_STAMP(7);
  // ---- epilogue: y = ((W o + bias) [* mask]) [* cscale] [+ res]   (operands in registers since before the merge)
  if (i0 + x < Tq) {
    float* yb = a.y + (int64_t)b * a.y_bs + (int64_t)qi * a.y_ts;
    const int ycs = (int)a.y_cs;
#pragma unroll
    for (int r = 0; r < NRB; ++r) {
      if (wave + MP_NW * r < NRBT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = 16 * (wave + MP_NW * r) + 4 * g + i;
          float v = yacc[r][i] + (a.bias ? e_bias[r][i] : 0.0f);
          if (a.mask) v *= e_mk;
          if (cscale) v *= e_cs[r][i];
          if (a.res) v += e_res[r][i];
          if (m < C) yb[m * ycs] = v;
        }
      }
    }
  }
  MP_STAMP(8);
#undef MP_STAMP
}

template <class Cfg>
int mp_launch(const hsp_mha_proj_args& a, hipStream_t s) {
  const int n_qt = (a.Tq + MP_QT - 1) / MP_QT;
  const int64_t blocks = (int64_t)n_qt * a.B;
  if (blocks <= 0 || blocks > 0x7fffffff) return HSP_EINVAL;
  constexpr int lds_bytes = Cfg::LDS_TOTAL * (int)sizeof(float);
  static_assert(lds_bytes <= 160 * 1024, "LDS");
  static hsp_lds_flags flags;
  if (lds_bytes > 32 * 1024)
    if (int e = hsp_raise_lds_limit(reinterpret_cast<const void*>(mha_proj_kernel<Cfg>), lds_bytes, flags)) return e;
  hipLaunchKernelGGL((mha_proj_kernel<Cfg>), dim3((unsigned)blocks), dim3(64 * MP_NW), lds_bytes, s, a, n_qt);
  return (int)hipGetLastError();
}
}  // namespace

extern "C" int hsp_mha_proj_supported(int32_t H, int32_t D, int32_t M, int32_t Tk) {
  return ((H == 4 && D == 69) || (H == 2 && D == 96)) && M == H * D && Tk >= 4 && Tk <= (1 << 20);
}

extern "C" int hsp_mha_proj_f32(const hsp_mha_proj_args* ap, void* stream) {
  if (!ap) return HSP_EINVAL;
  const hsp_mha_proj_args& a = *ap;
  if (!a.q || !a.k || !a.v || !a.wt || !a.y || a.B <= 0 || a.Tq <= 0) return HSP_EINVAL;
#ifndef HSP_TUNING
  if (a.debug != 0) return HSP_EINVAL;   // the tuning switches exist only in libhsp_tune.so
#endif
  if (!hsp_mha_proj_supported(a.H, a.D, a.M, a.Tk)) return HSP_EINVAL;
  if (a.wt_ld < a.M || (a.wt_ld & 3) || a.q_cs < 1 || a.k_cs < a.Tk || a.v_cs < a.Tk) return HSP_EINVAL;
  if (a.y_ts < 1 || (a.res && a.res_ts < 1)) return HSP_EINVAL;
  {  // 32-bit offsets inside one utterance's planes
    const int64_t lim = (int64_t)1 << 31, Cc = (int64_t)a.H * a.D;
    if (Cc * a.q_cs >= lim || Cc * a.k_cs >= lim || Cc * a.v_cs >= lim || (int64_t)a.M * a.wt_ld >= lim ||
        (int64_t)a.M * a.y_cs >= lim || (a.res && (int64_t)a.M * a.res_cs >= lim)) return HSP_EINVAL;
  }
  const hipStream_t s = static_cast<hipStream_t>(stream);
  if (a.H == 4) return mp_launch<MpCfg<4, 69>>(a, s);
  return mp_launch<MpCfg<2, 96>>(a, s);
}
