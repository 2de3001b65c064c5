// Two chained GEMMs in ONE launch for the 50 Hz part of the path (T = a few hundred frames per utterance):
//
//   WN layer   (modules.py:148-176, commons.py:107-114):  x_in = Conv_k(x) (+ g_l);  acts = tanh(a) * sigmoid(b);
//              rs = Conv1x1(acts);  x' = (x + rs[:H]) * mask;  out += rs[H:]
//   DiT FFN    (modules.py:382-388):  y = Conv_k(h) -> GELU(tanh) -> Conv1x1 -> * mask * gate + x
//
// Round 1 ran these as 2-3 launches per layer, each filling / draining a short K loop on ragged tiles
// (35-60 TFLOP/s, 40 % of the 13 ms this region took per step).  Here one workgroup owns 32 columns of one
// utterance and ALL rows:
//
//   phase 1   acc1[384 x 32] += W1[k-rows] x window(x)     implicit GEMM over (channel pair, tap) steps
//   gate      pre-activations -> LDS (U), bias + conditioning bias, tanh*sigmoid or GELU in place: the
//             activations never exist in HBM
//   phase 2   acc2[M2 x 32] += W2[rows of this part] x U   (a second part of 384 phase-1 rows, if any, repeats
//             phase 1 / gate / phase 2 on the same acc2: the FFN's 768 hidden rows are two parts)
//   epilogue  the residual / accumulate / mask / per-channel-gate tail of the 1x1 layers' own argument structs
//
// The input window of all channels is staged once (LDS-DMA, 16-B lanes); weights stream through a 3-slot LDS ring
// of 24 KB chunks (8 MFMA k-steps of phase 1 = 16 k-rows x 384 rows, or 16 / 32 channels of W2), issued two chunks
// ahead by three producer waves that do nothing else; one workgroup barrier per chunk.  Eight consumer waves =
// two groups that split every chunk's k-steps (a tile's serial MFMA chain is what the launch waits for) and add
// their halves through LDS.  fp32 MFMA (v_mfma_f32_32x32x2_f32): exact fp32.
#include <type_traits>
#include "hsp_device.h"

// tuning switches exist only in the -DHSP_TUNING build (libhsp_tune.so): bit 0 = producers stage only the first
// two chunks, bit 1 = consumers skip their MFMAs (results are then wrong; the time is what is measured)
#ifdef HSP_TUNING
#define G2_DBG(a, bit) (((a).debug & (bit)) != 0)
#else
#define G2_DBG(a, bit) false
#endif

// In-kernel stamps (tuning build, debug bit 256): wave 0 (consumer) and wave 8 (producer) of every workgroup write
// s_memtime at the phase boundaries to the buffer in.filt points to (unused by this kernel otherwise):
// [block][role 0/1][8] 64-bit ticks.  Diagnostic only; nothing reads the buffer on the device.
#ifdef HSP_TUNING
#define G2_STAMP(k)                                                                                         \
  do {                                                                                                      \
    if ((in.debug & 256) && lane == 0 && (wave == 0 || wave == G2_NCW))                                     \
      reinterpret_cast<unsigned long long*>(const_cast<float*>(in.filt))[(blockIdx.x * 2 + (wave ? 1 : 0)) * 8 + (k)] = \
          __builtin_amdgcn_s_memtime();                                                                     \
  } while (0)
#else
#define G2_STAMP(k) do {} while (0)
#endif

namespace {

typedef float g2_f32x16 __attribute__((ext_vector_type(16)));
// C/D map of the 32x32 MFMA forms: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#define HSP_ACC_ROW_G2(r, half) (((r) & 3) + 8 * ((r) >> 2) + 4 * (half))

constexpr int G2_BN = 32;            // columns per workgroup
constexpr int G2_R1 = 384;           // phase-1 rows per part (12 blocks of 32: 3 per consumer wave)
constexpr int G2_SLOT = 6144;        // floats per ring slot (24 KB)
constexpr int G2_NSLOT = 3;
constexpr int G2_NPW = 3;            // producer waves: wave i issues instruction i of every 3-instruction k-step
constexpr int G2_NCW = 8;            // consumer waves: 2 K-groups x 4
constexpr int G2_THREADS = 64 * (G2_NCW + G2_NPW);
constexpr int G2_IPC = 8;            // DMA instructions per producer wave and chunk (24 KB / 1 KB / 3)

#define G2_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define G2_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ void g2_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned g2_lds_addr(const float* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) float*)p;
}
// v = lds[addr + OFF bytes]: the offset rides in the instruction, so a group of fragment reads shares ONE address
// register (a register per read made the compiler run out of VGPRs and spill)
// The destination is written BY REFERENCE: the read is asynchronous, and a value returned from a helper (or routed
// through a ?:) may be copied to another register before the data has landed -- only the g2_wait* below make it valid.
template <int OFF>
__device__ __forceinline__ void g2_ds_read(float& dst, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// all but the N youngest LDS reads have landed.  The fragment registers are tied to the wait as in/out operands:
// without that nothing stops the compiler from scheduling an MFMA that reads them above the wait.
template <int N>
__device__ __forceinline__ void g2_wait4(float& b, float& a0, float& a1, float& a2) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(b), "+v"(a0), "+v"(a1), "+v"(a2) : "n"(N));
}
__device__ __forceinline__ void g2_wait_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <int N>
__device__ __forceinline__ void g2_wait3(float& b, float& a0, float& a1) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(b), "+v"(a0), "+v"(a1) : "n"(N));
}

struct G2Plan {
  int a0;        // aligned halo: the window starts at t0 - a0 (a0 = pad rounded up to 4)
  int xp;        // window pitch = 32 + 2 a0 floats (a multiple of 8)
  int xwin_sz;   // C1 * xp
  int u_off, ring_off, tab_off, total;   // floats
};
constexpr int G2_MAXPARTS = 4;
// tables: bias1[parts x 384] (phase-1 bias + conditioning bias per packed row), bias2[384], cscale2[384]
constexpr int G2_TAB = G2_MAXPARTS * G2_R1 + 2 * 384;
__host__ __device__ inline G2Plan g2_plan(int C1, int pad) {
  G2Plan p;
  p.a0 = (pad + 3) & ~3;
  p.xp = G2_BN + 2 * p.a0;
  p.xwin_sz = C1 * p.xp;
  p.u_off = (p.xwin_sz + 63) & ~63;
  p.ring_off = p.u_off + G2_R1 * G2_BN;
  p.tab_off = p.ring_off + G2_NSLOT * G2_SLOT;
  p.total = p.tab_off + G2_TAB;
  return p;
}

// The epilogue fields of one of the two 1x1 argument structs, selected FIELD BY FIELD: a reference to a by-value
// kernel argument chosen at run time (`second ? o2 : o1`) makes the compiler copy all three structs to scratch.
struct G2Out {
  const float* res; int64_t res_bs, res_cs;
  float* y; int64_t y_bs, y_cs;
  const float* mask; int64_t mask_bs;
  const float* bias; const float* cbias; int64_t cbias_bs;
  const float* cscale; int64_t cscale_bs;
  int mask_mode, accumulate, Cout, act;
  float scale, post_scale;
};
#define G2_PICK(f) (second ? o2.f : o1.f)
__device__ __forceinline__ G2Out g2_pick(const hsp_conv1d_args& o1, const hsp_conv1d_args& o2, bool second) {
  return G2Out{G2_PICK(res), G2_PICK(res_bs), G2_PICK(res_cs), G2_PICK(y), G2_PICK(y_bs), G2_PICK(y_cs), G2_PICK(mask),
               G2_PICK(mask_bs), G2_PICK(bias), G2_PICK(cbias), G2_PICK(cbias_bs), G2_PICK(cscale), G2_PICK(cscale_bs),
               G2_PICK(mask_mode), G2_PICK(accumulate), G2_PICK(Cout), G2_PICK(act), G2_PICK(scale), G2_PICK(post_scale)};
}

// NB2PW: 32-row blocks of the second GEMM per consumer wave (M2 <= 128 NB2PW).  GATE: phase-1 rows are WN's gated
// packing (32-row blocks alternate tanh-half / sigmoid-half) and a part yields 192 activation channels; otherwise a
// part yields 384 channels through the pointwise function in.act.
// MULTI: more than one part (the second GEMM's accumulators then live through phase 1 of the later parts; with a
// single part they are born after the gate, which keeps the register allocation of phase 1 small).
//
// Control flow: producers and consumers run SEPARATE loops with the same barrier sequence
//   per part: NP1 x [chunk barrier]   3 x [combine / gate barrier]   NP2 x [chunk barrier]      then 2 at the end.
// (A shared loop with `if (producer) ... else ...` inside makes the accumulators live across the join: the compiler
// then copies all 48 of them once per chunk -- VALU work, i.e. fp32 MFMA time.)
template <int NB2PW, bool GATE, bool MULTI>
__global__ __launch_bounds__(G2_THREADS, 3) void gemm2_kernel(const hsp_conv1d_args in, const hsp_conv1d_args o1,
                                                              const hsp_conv1d_args o2, const int split,
                                                              const int nparts, const int n_nt, const int xvec) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CP = GATE ? G2_R1 / 2 : G2_R1;   // activation channels per part = K of phase 2 per part
  constexpr int M2 = NB2PW == 3 ? 384 : 192;     // rows of the second GEMM (= o1.Cout + o2.Cout): 12 or 6 blocks of
                                                 // 32 over four waves -> NB2PW = 3 or 2 (the fourth wave idles at 192)
  static_assert(NB2PW == 2 || NB2PW == 3, "M2 is 192 or 384");
  constexpr int KC2 = G2_SLOT / M2;              // channels per phase-2 chunk (16 for M2 = 384, 32 for 192)
  constexpr int NP2 = CP / KC2;
  const G2Plan P = g2_plan(in.Cin, in.pad);
  float* const xwin = lds;
  float* const U = lds + P.u_off;
  float* const ring = lds + P.ring_off;
  float* const tab1 = lds + P.tab_off;                 // [nparts][384]: bias + conditioning bias of phase-1 row m
  float* const tab2b = tab1 + G2_MAXPARTS * G2_R1;      // [M2]: bias (+ conditioning bias) of output row m
  float* const tab2s = tab2b + 384;                     // [M2]: per-channel gate x scale of output row m

  const int b = blockIdx.x / n_nt, nt = blockIdx.x % n_nt;
  const int t0 = nt * G2_BN;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;

  const int C1 = in.Cin, K1 = in.K;
  const int NP1 = ((C1 >> 1) * K1) >> 3;         // phase-1 chunks of 8 k-steps (the host requires a whole number)
  const int NCH = nparts * (NP1 + NP2);          // chunks of the whole schedule

  G2_STAMP(0);
  // ------------------------------------------------------------------ per-row constants -> LDS (everyone; they are
  // read after the first workgroup barrier at the earliest).  Loads issue back to back: one round trip, instead of
  // one per accumulator element in the combine step and the epilogue.
  {
    for (int idx = threadIdx.x; idx < nparts * G2_R1; idx += G2_THREADS) {
      const int part = idx / G2_R1, m = idx - part * G2_R1;
      int ch;
      if (GATE) ch = ((m >> 5) & 1) * (CP * nparts) + part * CP + (m >> 6) * 32 + (m & 31);
      else ch = part * CP + m;
      float v = in.bias ? in.bias[ch] : 0.0f;
      if (in.cbias) v += in.cbias[(int64_t)b * in.cbias_bs + ch];
      tab1[idx] = v;
    }
    for (int m = threadIdx.x; m < M2; m += G2_THREADS) {
      const bool second = split > 0 && m >= split;
      const G2Out o = g2_pick(o1, o2, second);
      const int co = m - (second ? split : 0);
      float v = o.bias ? o.bias[co] : 0.0f;
      if (o.cbias) v += o.cbias[(int64_t)b * o.cbias_bs + co];
      tab2b[m] = v;
      tab2s[m] = (o.cscale ? o.cscale[(int64_t)b * o.cscale_bs + co] : 1.0f) * o.scale;
    }
  }
  G2_STAMP(1);

  if (wave >= G2_NCW) {
    // ================================================================================================ producers
    // chunk n -> ring slot n % 3; a producer wave owns instruction `pw` of every 3-instruction row pair / row group
    const int pw = wave - G2_NCW;
    const int f16 = 64 * pw + lane;              // 16-B lane index inside the 192 lanes of a k-step (2 x 384 floats)
    const unsigned off1 = 4u * (unsigned)((f16 / 96) * in.w_ld + (f16 % 96) * 4);
    // phase 2: a chunk is KC2 rows of M2 floats = 24 instructions; the lane pattern repeats every 3 instructions
    // (192 lanes = 768 floats = 2 rows of 384 or 4 rows of 192)
    const unsigned off2 = 4u * (unsigned)((f16 / (M2 >> 2)) * o1.w_ld + (f16 % (M2 >> 2)) * 4);
    // running schedule position of the NEXT chunk to issue (no division by run-time values inside the loop)
    int is_part = 0, is_r = 0, is_cp = 0, is_j = 0;
    auto issue = [&](int n) __attribute__((always_inline)) {
      float* const slot = ring + (n % G2_NSLOT) * G2_SLOT;
      if (is_r < NP1) {
        // 8 k-steps of phase 1: step -> (channel pair cp, tap j); its two k-rows are w1[(j C1 + 2 cp)(+1)][part rows]
        int cp = is_cp, j = is_j;
#pragma unroll
        for (int i = 0; i < G2_IPC; ++i) {
          const char* base = reinterpret_cast<const char*>(in.w + (size_t)(j * C1 + 2 * cp) * in.w_ld + is_part * G2_R1);
          __builtin_amdgcn_global_load_lds(G2_GPTR(base + off1), G2_LPTR(slot + i * 768 + pw * 256), 16, 0, 0);
          if (++j == K1) { j = 0; ++cp; }
        }
        is_cp = cp;
        is_j = j;
      } else {
        const int k0 = is_part * CP + (is_r - NP1) * KC2;   // first W2 row of the chunk
        constexpr int rows_per3 = 768 / M2;                 // 2 or 4 rows per 3 instructions
#pragma unroll
        for (int i = 0; i < G2_IPC; ++i) {
          const char* base = reinterpret_cast<const char*>(o1.w + (size_t)(k0 + i * rows_per3) * o1.w_ld);
          __builtin_amdgcn_global_load_lds(G2_GPTR(base + off2), G2_LPTR(slot + i * 768 + pw * 256), 16, 0, 0);
        }
      }
      if (++is_r == NP1 + NP2) { is_r = 0; ++is_part; is_cp = 0; is_j = 0; }
    };
    // input window of all channels
    const float* xb = in.x + (int64_t)b * in.x_bs;
    if (xvec) {
      const int l4 = P.xp >> 2;                    // 16-B lanes per window row
      const int nl = C1 * l4;
      for (int f = 64 * pw + lane; f < ((nl + 63) & ~63); f += 64 * G2_NPW) {
        const int row = f / l4, c = f - row * l4;
        const int t = t0 - P.a0 + 4 * c;
        if (f < nl) {
          const float* src = (t >= 0 && t < in.Lin) ? xb + (int64_t)row * in.x_cs + t : in.zeros;
          __builtin_amdgcn_global_load_lds(G2_GPTR(src), G2_LPTR(xwin + 4 * (f - lane)), 16, 0, 0);
        }
      }
    } else {
      // rows that are not 16-B addressable (T % 4 != 0): one float per lane
      const int nl = C1 * P.xp;
      for (int f = 64 * pw + lane; f < ((nl + 63) & ~63); f += 64 * G2_NPW) {
        const int row = f / P.xp, c = f - row * P.xp;
        const int t = t0 - P.a0 + c;
        if (f < nl) {
          const float* src = (t >= 0 && t < in.Lin) ? xb + (int64_t)row * in.x_cs + t : in.zeros;
          __builtin_amdgcn_global_load_lds(G2_GPTR(src), G2_LPTR(xwin + (f - lane)), 4, 0, 0);
        }
      }
    }
    issue(0);
    if (NCH > 1) issue(1);
    G2_STAMP(2);
    int n = 0;
    auto chunk_rounds = [&](int count) __attribute__((always_inline)) {
      for (int c = 0; c < count; ++c, ++n) {
        if (n + 1 < NCH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G2_IPC) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        g2_barrier();                             // chunk n is in LDS; every consumer is done with chunk n - 1
        if (n + 2 < NCH && !G2_DBG(in, 1)) issue(n + 2);
      }
    };
    for (int part = 0; part < nparts; ++part) {
      chunk_rounds(NP1);
      G2_STAMP(3);
      g2_barrier();
      g2_barrier();
      g2_barrier();
      G2_STAMP(4);
      chunk_rounds(NP2);
    }
    G2_STAMP(6);
    g2_barrier();
    g2_barrier();
    return;
  }

  // ================================================================================================== consumers
  const int grp = wave >> 2, w4 = wave & 3;
  const int l32 = lane & 31, half = lane >> 5;
  g2_f32x16 acc2[NB2PW];
  auto zero_acc2 = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NB2PW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[i][r] = 0.0f;
  };
  if constexpr (MULTI) zero_acc2();
  G2_STAMP(2);

  int n = 0;   // chunk sequence number
  // phase 2 of one part: acc2 += W2[rows of the part] x activations (U).
  // Software pipeline over "sub-chunks" of 4 k-steps (one per chunk at M2 = 384, two at M2 = 192): the fragments of
  // sub-chunk k + 1 are requested in the MIDDLE of sub-chunk k's MFMAs (two register sets), right behind the barrier
  // that says its chunk has landed, so neither the LDS round trip nor the barrier sits between two MFMA groups.  The
  // barrier's own lgkmcnt(0) retires the current set's remaining reads (issued a whole sub-chunk earlier).
  auto phase2 = [&]() __attribute__((always_inline)) {
    // a wave whose blocks lie past M2 (the fourth wave when M2 = 192) works on the last NB2PW blocks instead -- reads in
    // range, results never stored -- so every wave runs the same instruction stream
    const int blk0 = (M2 >> 5) - w4 * NB2PW > 0 ? w4 * NB2PW : (M2 >> 5) - NB2PW;
    constexpr int nst = KC2 >> 2;                 // steps per group and chunk: KC2 / 2 steps split in two (4 or 8)
    constexpr int SPC = nst / 4;                  // sub-chunks per chunk (1 or 2)
    constexpr int NSUB = NP2 * SPC;
    static_assert(NSUB % 2 == 0, "the pipeline is unrolled by two");
    float A0[4][NB2PW], B0[4], A1[4][NB2PW], B1[4];
    auto rd2 = [&](float (&A)[4][NB2PW], float (&B)[4], int k) __attribute__((always_inline)) {
      // sub-chunk k: chunk c = k / SPC (slot of sequence number n0 + c), steps ls0 .. ls0 + 3 inside the chunk (ls0 a
      // multiple of 4): activation channels ka0 + 2 s + half with ka0 a multiple of 8 -> constant offsets between steps
      const int c = k / SPC;
      const float* const slot = ring + ((n + c) % G2_NSLOT) * G2_SLOT;
      const int ls0 = grp * nst + (k % SPC) * 4;
      const int ka0 = c * KC2 + 2 * ls0 + half;
      const int urow0 = GATE ? ka0 + ((ka0 >> 5) << 5) : ka0;
      const unsigned ua = g2_lds_addr(U + urow0 * G2_BN + l32);
      const unsigned wa = g2_lds_addr(slot + (2 * ls0 + half) * M2 + blk0 * 32 + l32);
      auto one = [&](auto ss) __attribute__((always_inline)) {
        constexpr int s = decltype(ss)::value;
        g2_ds_read<2 * s * G2_BN * 4>(B[s], ua);
        g2_ds_read<(2 * s * M2) * 4>(A[s][0], wa);
        if constexpr (NB2PW > 1) g2_ds_read<(2 * s * M2 + 32) * 4>(A[s][1], wa);
        if constexpr (NB2PW > 2) g2_ds_read<(2 * s * M2 + 64) * 4>(A[s][2], wa);
      };
      one(std::integral_constant<int, 0>{});
      one(std::integral_constant<int, 1>{});
      one(std::integral_constant<int, 2>{});
      one(std::integral_constant<int, 3>{});
    };
    // MFMAs of steps S0, S0 + 1; WAIT: the set was requested last (counted waits), else it has been retired already
    auto mm2 = [&](float (&A)[4][NB2PW], float (&B)[4], auto s0, auto waitflag) __attribute__((always_inline)) {
      constexpr int S0 = decltype(s0)::value;
      constexpr bool WAIT = decltype(waitflag)::value;
#pragma unroll
      for (int s = S0; s < S0 + 2; ++s) {
        if constexpr (WAIT) {
          if constexpr (NB2PW == 3) { if (s == 0) g2_wait4<12>(B[0], A[0][0], A[0][1], A[0][2]); else g2_wait4<8>(B[1], A[1][0], A[1][1], A[1][2]); }
          else { if (s == 0) g2_wait3<9>(B[0], A[0][0], A[0][1]); else g2_wait3<6>(B[1], A[1][0], A[1][1]); }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NB2PW; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s][i], B[s], acc2[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    using T0 = std::integral_constant<int, 0>;
    using T2 = std::integral_constant<int, 2>;
    g2_barrier();                                 // chunk n landed (and the activations are complete in U)
    if (!G2_DBG(in, 2)) rd2(A0, B0, 0);
    for (int k = 0; k < NSUB; k += 2) {
      if (!G2_DBG(in, 2)) mm2(A0, B0, T0{}, std::true_type{});
      if ((k + 1) % SPC == 0) g2_barrier(); else g2_wait_all();
      if (!G2_DBG(in, 2)) { rd2(A1, B1, k + 1); mm2(A0, B0, T2{}, std::false_type{}); mm2(A1, B1, T0{}, std::true_type{}); }
      if (k + 2 < NSUB) {
        if ((k + 2) % SPC == 0) g2_barrier(); else g2_wait_all();
        if (!G2_DBG(in, 2)) rd2(A0, B0, k + 2);
      }
      if (!G2_DBG(in, 2)) mm2(A1, B1, T2{}, std::false_type{});
    }
    n += NP2;
  };

  for (int part = 0;; ++part) {
    // ================================================================ phase 1
    {
      g2_f32x16 acc1[3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[i][r] = 0.0f;
      // this group's first k-step of the part is 4 grp: (channel pair, tap) -> window offset 2 cp xp + j dil, kept as
      // SCALARS and advanced step by step (a division or a vector select here would be VALU work inside the loop)
      int c_j = __builtin_amdgcn_readfirstlane((4 * grp) % K1);
      int c_off = __builtin_amdgcn_readfirstlane(2 * ((4 * grp) / K1) * P.xp + c_j * in.dil);
      const int wrap = 2 * P.xp - (K1 - 1) * in.dil;   // offset step from (cp, K1 - 1) to (cp + 1, 0)
      auto next_step = [&]() __attribute__((always_inline)) {
        const bool w = c_j + 1 == K1;
        c_off += w ? wrap : in.dil;
        c_j = w ? 0 : c_j + 1;
      };
      const unsigned xa = g2_lds_addr(xwin + half * P.xp + l32 + (P.a0 - in.pad));
      // same software pipeline as phase 2: chunk c + 1's fragments are requested in the middle of chunk c's MFMAs
      float A0[4][3], B0[4], A1[4][3], B1[4];
      auto rd1 = [&](float (&A)[4][3], float (&B)[4], int c) __attribute__((always_inline)) {
        const float* const slot = ring + ((n + c) % G2_NSLOT) * G2_SLOT;
        const unsigned sa = g2_lds_addr(slot + (8 * grp + half) * G2_R1 + w4 * 96 + l32);
        auto one = [&](auto ss) __attribute__((always_inline)) {
          constexpr int s = decltype(ss)::value;
          g2_ds_read<0>(B[s], xa + 4u * (unsigned)c_off);
          g2_ds_read<(2 * s * G2_R1) * 4>(A[s][0], sa);
          g2_ds_read<(2 * s * G2_R1 + 32) * 4>(A[s][1], sa);
          g2_ds_read<(2 * s * G2_R1 + 64) * 4>(A[s][2], sa);
          next_step();
        };
        one(std::integral_constant<int, 0>{});
        one(std::integral_constant<int, 1>{});
        one(std::integral_constant<int, 2>{});
        one(std::integral_constant<int, 3>{});
        // the other group's 4 steps lie between this chunk's and the next one's
        next_step();
        next_step();
        next_step();
        next_step();
      };
      auto mm1 = [&](float (&A)[4][3], float (&B)[4], auto s0, auto waitflag) __attribute__((always_inline)) {
        constexpr int S0 = decltype(s0)::value;
        constexpr bool WAIT = decltype(waitflag)::value;
#pragma unroll
        for (int s = S0; s < S0 + 2; ++s) {
          if constexpr (WAIT) { if (s == 0) g2_wait4<12>(B[0], A[0][0], A[0][1], A[0][2]); else g2_wait4<8>(B[1], A[1][0], A[1][1], A[1][2]); }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 3; ++i) acc1[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s][i], B[s], acc1[i], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      using T0 = std::integral_constant<int, 0>;
      using T2 = std::integral_constant<int, 2>;
      g2_barrier();                               // chunk n landed
      if (!G2_DBG(in, 2)) rd1(A0, B0, 0);
      for (int c = 0; c < NP1; c += 2) {          // NP1 is even (host check)
        if (!G2_DBG(in, 2)) mm1(A0, B0, T0{}, std::true_type{});
        g2_barrier();                             // chunk n + c + 1 landed; A0 / B0 fully in registers (lgkmcnt(0))
        if (!G2_DBG(in, 2)) { rd1(A1, B1, c + 1); mm1(A0, B0, T2{}, std::false_type{}); mm1(A1, B1, T0{}, std::true_type{}); }
        if (c + 2 < NP1) {
          g2_barrier();
          if (!G2_DBG(in, 2)) rd1(A0, B0, c + 2);
        }
        if (!G2_DBG(in, 2)) mm1(A1, B1, T2{}, std::false_type{});
      }
      n += NP1;
      G2_STAMP(3);
      // ============================================================== combine the K-groups, bias, gate -> U
      // U through a pointer the compiler cannot see through, re-made every part: otherwise it hoists the ~200 loop-
      // invariant LDS addresses of this section out of the part loop and keeps them in scratch.
      unsigned ub = g2_lds_addr(U), tb = g2_lds_addr(tab1);
      asm volatile("" : "+v"(ub), "+v"(tb));
      typedef __attribute__((address_space(3))) float lds_f32;
      lds_f32* const Ul = reinterpret_cast<lds_f32*>(static_cast<uintptr_t>(ub));
      const lds_f32* const T1l = reinterpret_cast<const lds_f32*>(static_cast<uintptr_t>(tb));
      g2_barrier();                               // every consumer is done with phase 1
      if (grp == 1) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) Ul[(w4 * 96 + i * 32 + HSP_ACC_ROW_G2(r, half)) * G2_BN + l32] = acc1[i][r];
      }
      g2_barrier();
      if (grp == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = w4 * 96 + i * 32 + HSP_ACC_ROW_G2(r, half);    // packed phase-1 row inside the part
            Ul[m * G2_BN + l32] = acc1[i][r] + Ul[m * G2_BN + l32] + T1l[part * G2_R1 + m];
          }
        }
      }
      g2_barrier();
      {
        const int tid = threadIdx.x;              // 0 .. 511
        if (GATE) {
          for (int idx = tid; idx < CP * G2_BN; idx += 64 * G2_NCW) {
            const int c = idx >> 5, nn = idx & 31;
            const int ra = ((c >> 5) << 6) + (c & 31);                   // a-row of channel c; its b-row is 32 below
            const float va = Ul[ra * G2_BN + nn], vb = Ul[(ra + 32) * G2_BN + nn];
            Ul[ra * G2_BN + nn] = hsp_tanh(va) * hsp_sigmoid(vb);
          }
        } else {
          for (int idx = tid; idx < CP * G2_BN; idx += 64 * G2_NCW) Ul[idx] = hsp_apply_act(Ul[idx], in.act);
        }
      }
      G2_STAMP(4);
    }
    // ================================================================ phase 2 (the last part's runs after the loop)
    if (part == nparts - 1) break;
    phase2();
  }

  // epilogue operands, requested now (acc1's registers are free), they arrive under the last phase 2: the residual OR
  // the running sum of an output element (the host refuses to fuse a layer that has both on the same rows), and this
  // lane's mask value.  Group g owns accumulator registers [8 g, 8 g + 8) of its blocks.
  float eop[NB2PW][8], emk[NB2PW];
  {
    int tcl = t0 + l32 < in.Lin ? t0 + l32 : in.Lin - 1;
#pragma unroll
    for (int i = 0; i < NB2PW; ++i) {
      const int mblk = (w4 * NB2PW + i) * 32;
      const bool second = split > 0 && mblk >= split;
      const G2Out o = g2_pick(o1, o2, second);
      const int mo = second ? split : 0;
      emk[i] = 1.0f;
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) eop[i][r8] = 0.0f;
      if (mblk < M2) {                             // whole passes under wave-uniform branches: the loads of a
        int cof[8];                                // pass issue back to back
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
          const int co = mblk + HSP_ACC_ROW_G2(8 * grp + r8, half) - mo;
          cof[r8] = co < o.Cout ? co : o.Cout - 1;
        }
        if (o.res) {
          const float* const rb = o.res + (int64_t)b * o.res_bs + tcl;
#pragma unroll
          for (int r8 = 0; r8 < 8; ++r8) eop[i][r8] = rb[(int64_t)cof[r8] * o.res_cs];
        } else if (o.accumulate) {
          const float* const yb = o.y + (int64_t)b * o.y_bs + tcl;
#pragma unroll
          for (int r8 = 0; r8 < 8; ++r8) eop[i][r8] = yb[(int64_t)cof[r8] * o.y_cs];
        }
        if (o.mask_mode != HSP_MASK_NONE) emk[i] = o.mask[(int64_t)b * o.mask_bs + tcl];
      }
    }
  }
  G2_STAMP(5);
  if constexpr (!MULTI) zero_acc2();
  phase2();
  G2_STAMP(6);

  // ==================================================================== exchange halves, epilogue
  g2_barrier();                                   // every consumer is done with U and the ring
  // group g finalises accumulator registers [8 g, 8 g + 8) of its blocks and hands the other eight to its partner
  float* const xch = U;                           // [dst group][w4][NB2PW][8][64]
  {
    float* dst = xch + ((((1 - grp) * 4 + w4) * NB2PW) * 8) * 64 + lane;
#pragma unroll
    for (int i = 0; i < NB2PW; ++i)
#pragma unroll
      for (int r = 0; r < 8; ++r) dst[(i * 8 + r) * 64] = grp ? acc2[i][r] : acc2[i][8 + r];
  }
  g2_barrier();
  const float* src = xch + (((grp * 4 + w4) * NB2PW) * 8) * 64 + lane;
  const int t = t0 + l32;
  if (t >= in.Lin) return;
#pragma unroll
  for (int i = 0; i < NB2PW; ++i) {
    const int mblk = (w4 * NB2PW + i) * 32;
    if (mblk >= M2) continue;   // wave-uniform
    const bool second = split > 0 && mblk >= split;
    const G2Out o = g2_pick(o1, o2, second);
    const int mo = second ? split : 0;
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) {
      const int r = 8 * grp + r8;
      const int m = mblk + HSP_ACC_ROW_G2(r, half);
      const int co = m - mo;
      if (co >= o.Cout) continue;
      // the order of hsp_epilogue_store (hsp_device.h), operands from LDS tables / registers
      float v = (grp ? acc2[i][8 + r8] : acc2[i][r8]) + src[(i * 8 + r8) * 64] + tab2b[m];
      v = hsp_apply_act(v, o.act);
      if (o.mask_mode & HSP_MASK_PRE) v *= emk[i];
      v *= tab2s[m];
      if (o.res) v += eop[i][r8];
      if (o.mask_mode & HSP_MASK_POST) v *= emk[i];
      if (!o.res) v += eop[i][r8];                  // running sum (0 when the layer does not accumulate)
      o.y[(int64_t)b * o.y_bs + (int64_t)co * o.y_cs + t] = v * o.post_scale;
    }
  }
  G2_STAMP(7);
}

bool g2_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// the 1x1 layer `o` reads the activations the conv `in` would have written
bool g2_consumes(const hsp_conv1d_args& in, const hsp_conv1d_args& o, int acts_channels) {
  return o.K == 1 && o.stride == 1 && o.pad == 0 && o.dil >= 1 && o.rows == HSP_ROWS_PLAIN && o.prologue == HSP_PRO_NONE &&
         o.x == in.y && o.x_bs == in.y_bs && o.x_cs == in.y_cs && o.x_ts == 1 && o.B == in.B && o.Cin == acts_channels &&
         o.Lin == in.Lout && o.Lout == in.Lout && o.ncols == in.Lout && !o.ln_c1 && !o.split_row && o.M == o.Cout &&
         (o.Cout & 31) == 0 && (o.w_ld & 3) == 0 && g2_al16(o.w) && o.y && o.y != in.x &&
         (o.mask_mode == HSP_MASK_NONE || o.mask) && !(o.res && o.accumulate) && !(o.res && o.res_ts > 1);
}

template <int NB2PW, bool GATE, bool MULTI>
int g2_launch(const hsp_conv1d_args& in, const hsp_conv1d_args& o1, const hsp_conv1d_args& o2, int split, int nparts,
              hipStream_t s) {
  const int n_nt = (in.Lin + G2_BN - 1) / G2_BN;
  const int64_t blocks = (int64_t)in.B * n_nt;
  if (blocks <= 0 || blocks > 0x7fffffff) return HSP_EINVAL;
  const int lds_bytes = g2_plan(in.Cin, in.pad).total * (int)sizeof(float);
  static hsp_lds_flags flags;
  if (int e = hsp_raise_lds_limit(reinterpret_cast<const void*>(gemm2_kernel<NB2PW, GATE, MULTI>), 160 * 1024, flags)) return e;
  const int xvec = g2_al16(in.x) && (in.x_cs & 3) == 0 && (in.x_bs & 3) == 0 && (in.Lin & 3) == 0;
  hipLaunchKernelGGL((gemm2_kernel<NB2PW, GATE, MULTI>), dim3((unsigned)blocks), dim3(G2_THREADS), (size_t)lds_bytes, s, in, o1, o2,
                     split, nparts, n_nt, xvec);
  return (int)hipGetLastError();
}

}  // namespace

// Fused launch of `in` (conv) -> gate / pointwise -> `o1` (+ `o2`: further rows of the same 1x1 matrix).
// Returns -1 when the argument structs do not describe a pattern the kernel covers (the caller then launches
// the layers one by one), 0 / hipError_t otherwise; dry_run: only the check (0 = would fuse).  in.y is only a name here: the activations are never written.
int hsp_gemm2_try(const hsp_conv1d_args& in, const hsp_conv1d_args& o1, const hsp_conv1d_args* o2, hipStream_t s,
                  bool dry_run) {
  const bool gate = in.rows == HSP_ROWS_GATE_WN;
  if (!gate && in.rows != HSP_ROWS_PLAIN) return -1;
#ifndef HSP_TUNING
  if (in.debug || o1.debug) return -1;
#endif
  if (!in.x || !in.w || !in.y || !in.zeros || !g2_al16(in.zeros)) return -1;
  if (in.stride != 1 || in.x_ts != 1 || in.prologue != HSP_PRO_NONE || in.K < 1 || (in.K & 1) == 0 || in.dil < 1) return -1;
  if (in.pad * 2 != in.dil * (in.K - 1) || in.Lin != in.Lout || in.ncols != in.Lin) return -1;
  if (!g2_al16(in.w) || (in.w_ld & 3) || (in.Cin & 1) || in.Cin < 2) return -1;
  if (in.mask_mode != HSP_MASK_NONE || in.cscale || in.res || in.accumulate || in.scale != 1.0f || in.post_scale != 1.0f ||
      in.ln_c1 || in.split_row)
    return -1;
  int nparts, acts;
  if (gate) {
    if (in.gate_half <= 0 || in.gate_half % (G2_R1 / 2) || in.M != 2 * in.gate_half || in.act != HSP_ACT_NONE) return -1;
    nparts = in.gate_half / (G2_R1 / 2);
    acts = in.gate_half;
  } else {
    if (in.M != in.Cout || in.M % G2_R1) return -1;
    nparts = in.M / G2_R1;
    acts = in.M;
  }
  if (in.w_ld < nparts * G2_R1 || nparts > G2_MAXPARTS) return -1;
  if (((in.Cin >> 1) * in.K) & 15) return -1;     // phase 1 runs in whole chunks of 8 k-steps, an even number of them
  if (!g2_consumes(in, o1, acts)) return -1;
  int split = 0, M2 = o1.Cout;
  if (o2) {
    if (!g2_consumes(in, *o2, acts) || o2->w != o1.w + o1.Cout || o2->w_ld != o1.w_ld) return -1;
    split = o1.Cout;
    M2 += o2->Cout;
  }
  if (M2 != 192 && M2 != 384) return -1;          // chunk = 6144 floats = 32 or 16 rows of W2
  if (o1.w_ld < M2) return -1;
  // 32-bit element offsets inside one utterance / matrix
  if ((int64_t)in.Cin * in.x_cs + in.Lin >= (1ll << 31) || (int64_t)in.K * in.Cin * in.w_ld >= (1ll << 31) ||
      (int64_t)acts * o1.w_ld >= (1ll << 31))
    return -1;
  if (g2_plan(in.Cin, in.pad).total * (int)sizeof(float) > 160 * 1024) return -1;
  if (dry_run) return 0;
  const hsp_conv1d_args& o2r = o2 ? *o2 : o1;
  // instantiations: WN layers (gated, one part: H = 192) with 384 or 192 output rows; FFN-style pairs (pointwise,
  // any number of parts) with 192 or 384 output rows; gated with several parts (H = 384, 576, 768)
  if (gate && nparts == 1)
    return M2 == 384 ? g2_launch<3, true, false>(in, o1, o2r, split, nparts, s) : g2_launch<2, true, false>(in, o1, o2r, split, nparts, s);
  if (gate)
    return M2 == 384 ? g2_launch<3, true, true>(in, o1, o2r, split, nparts, s) : g2_launch<2, true, true>(in, o1, o2r, split, nparts, s);
  return M2 == 384 ? g2_launch<3, false, true>(in, o1, o2r, split, nparts, s) : g2_launch<2, false, true>(in, o1, o2r, split, nparts, s);
}
