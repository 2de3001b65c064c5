// hsp_conv1d_mfma_f32: argument validation and the choice of kernel, tile shape and epilogue kind.
// The kernel itself is the template in hsp_conv1d_mfma_kernel.h, instantiated per tile shape by
// hsp_conv1d_tile.hip; 1x1 convs over short column axes go to the token GEMM (hsp_tokgemm.hip).
//
// Reference call sites: see include/hsp.h (hsp_conv1d_args).
#include "hsp_conv1d_mfma_kernel.h"

int hsp_rgemm_try(const hsp_conv1d_args& a, hipStream_t s, int32_t* plan_out);    // hsp_rgemm.hip; likewise
int hsp_bgemm_try(const hsp_conv1d_args& a, hipStream_t s, int32_t* plan_out);    // hsp_bgemm.hip; likewise

namespace {
using namespace hspconv;

int validate(const hsp_conv1d_args& a) {
  if (!a.x || !a.w || !a.y) return HSP_EINVAL;
#ifndef HSP_TUNING
  if (a.debug != 0) return HSP_EINVAL;  // tuning switches exist only in libhsp_tune.so
#endif
  if (a.B <= 0 || a.Cin <= 0 || a.Lin <= 0 || a.K <= 0 || a.M <= 0 || a.Cout <= 0 || a.Lout <= 0 || a.ncols <= 0)
    return HSP_EINVAL;
  if (a.stride != 1 || (a.M & 3) || (a.w_ld & 3) || a.w_ld < a.M || a.dil < 1) return HSP_EINVAL;
  if ((reinterpret_cast<uintptr_t>(a.w) & 15) != 0 || a.w_bs < 0 || (a.w_bs & 3)) return HSP_EINVAL;
  if (a.w_bs && (a.ln_c1 || a.split_row || (a.res && a.res_ts > 1))) return HSP_EINVAL;   // per-batch weights: conv tiles only
  // the kernel indexes one utterance / the weight matrix with 32-bit element offsets
  if ((int64_t)a.K * a.Cin * a.w_ld >= (1ll << 31)) return HSP_EINVAL;
  if ((int64_t)a.Cin * a.x_cs + (int64_t)a.Lin * a.x_ts >= (1ll << 31) || a.x_cs < 0 || a.x_ts < 0) return HSP_EINVAL;
  if (a.y_cs < 0 || (int64_t)a.Cout * a.y_cs + a.Lout >= (1ll << 31)) return HSP_EINVAL;
  if (a.res && (a.res_cs < 0 || (int64_t)a.Cout * a.res_cs + a.Lout >= (1ll << 31))) return HSP_EINVAL;
  if (a.res_ts < 0 || (a.res && a.res_ts > 1 && (int64_t)a.Cout * a.res_cs + (int64_t)a.Lout * a.res_ts >= (1ll << 31))) return HSP_EINVAL;
  if (a.prologue == HSP_PRO_ACT1D && (!a.alpha_exp || !a.beta_inv || !a.filt || a.x_ts != 1)) return HSP_EINVAL;
  if (!a.zeros || (reinterpret_cast<uintptr_t>(a.zeros) & 15) != 0) return HSP_EINVAL;
  if (a.prologue != HSP_PRO_NONE && a.prologue != HSP_PRO_LRELU && a.prologue != HSP_PRO_ACT1D) return HSP_EINVAL;
  if (a.mask_mode != HSP_MASK_NONE && !a.mask) return HSP_EINVAL;
  // modulated input LayerNorm (hsp.h ln_scale): with ln_c1 only; its per-utterance bias arrives as cbias
  if ((a.ln_scale || a.ln_mask || a.ln_c1_bs) && (!a.ln_c1 || !a.ln_scale || a.bias || !a.cbias || a.Cin > 1024 || a.ln_c1_bs < 0)) return HSP_EINVAL;
  if (a.rows == HSP_ROWS_GATE_WN || a.rows == HSP_ROWS_GATE_GLU) {
    if (a.gate_half <= 0 || (a.gate_half & 31) || a.M != 2 * a.gate_half || a.Cout != a.gate_half) return HSP_EINVAL;
    if (a.prologue == HSP_PRO_ACT1D) return HSP_EINVAL;
  } else if (a.rows == HSP_ROWS_SHUFFLE) {
    if (a.up <= 0 || a.up > 16 || a.M != a.Cout * a.up || a.M > (1 << 16)) return HSP_EINVAL;
  } else if (a.rows != HSP_ROWS_PLAIN) {
    return HSP_EINVAL;
  }
  return 0;
}

int dispatch(const hsp_conv1d_args& a, hipStream_t s, int32_t* plan_out) {
  const bool gated = a.rows == HSP_ROWS_GATE_WN || a.rows == HSP_ROWS_GATE_GLU;
  const bool act = a.prologue == HSP_PRO_ACT1D;
  // "short": 128 x 128 tiles would leave at least half of the 256 CUs idle; prefer 64 x 64 tiles with deep
  // chunks so that more CUs get work.  A lone 64 x 64 tile takes half as long as a lone 128 x 128 one (measured:
  // 36 us against 67 us at C = 128, k = 7) for a quarter of the work, so the small shape wins up to 64 big tiles,
  // ties up to 128 and loses 2x beyond (tools/conv_sweep.py, profiles/r02_tile_threshold.txt).
  const bool short_seq = (int64_t)((a.M + 127) / 128) * ((a.ncols + 127) / 128) * a.B <= 128;
  // a residual read at a column stride: the register-path token GEMM or nothing
  if (a.res && a.res_ts > 1) {
    const int e = hsp_rgemm_try(a, s, plan_out);
    return e >= 0 ? e : HSP_EINVAL;
  }
  // 1x1 GEMMs with enough outputs for about one 128 x 128 / 128 x 64 tile per CU (the PLM loop beyond ~60 prefix
  // positions): the throughput-oriented token GEMM (hsp_bgemm.hip)
  if (a.K == 1 && !a.w_bs && !HSP_DBG(a, 128)) {
    const int e = hsp_bgemm_try(a, s, plan_out);
    if (e >= 0) return e;
  }
  if (a.ln_scale) return HSP_EINVAL;              // the modulated LayerNorm exists on the block token GEMM only
  // (tuning bit 1 << 24: try the register-path token GEMM whatever the launch size -- tools/gemm_sweep.py)
  if ((short_seq || a.ln_c1 || a.split_row || HSP_DBG(a, 1 << 24)) && !a.w_bs && !HSP_DBG(a, 128)) {
    // 1x1 GEMMs over a few thousand token columns: the register-path kernel (hsp_rgemm.hip).  (The LDS-DMA token GEMM
    // of rounds 1-3, hsp_tokgemm.hip, was retired in round 4: tools/gemm_sweep.py found no (shape, B, T) cell of
    // profiles/r04_gemm_dispatch_table.txt where it beat the block GEMM or the register path by more than 2 %.)
    const int e = hsp_rgemm_try(a, s, plan_out);
    if (e >= 0) return e;
  }
  if (a.ln_c1 || a.split_row) return HSP_EINVAL;  // fused input LayerNorm / second output: token-GEMM path only
  int epi = select_epilogue(a);
  if (act && epi != HSP_EPI_INIT) epi = HSP_EPI_GEN;  // the activation shapes carry INIT and GEN only
  // the window pitch is a compile-time constant of the shape (tile columns + 64): halos beyond 61 columns
  // (the 64-tap blocks of the wav2vec2 positional conv; WN in-layers with dilation_rate > 1) go to the two
  // shapes with a wider pitch (plain / gated rows; up to 125 columns, beyond: HSP_EINVAL from pick_lkc)
  if ((a.K - 1) * a.dil + 3 > 64)
    return gated ? hsp_conv_tile_S64GW(a, epi, act, s, plan_out) : hsp_conv_tile_S64W(a, epi, act, s, plan_out);
#ifdef HSP_TUNING
  // force a tile shape (results stay right)
  if (!gated && a.debug & 256) return hsp_conv_tile_M128(a, epi, act, s, plan_out);
  if (!gated && a.debug & 1024) return hsp_conv_tile_M64(a, epi, act, s, plan_out);
  if (!gated && a.debug & 4096) return hsp_conv_tile_S64(a, epi, act, s, plan_out);
  if (!gated && a.debug & 8192) return hsp_conv_tile_M64P(a, epi, act, s, plan_out);
#endif
  if (short_seq) {
    if (gated) {
      // (round 5) a launch of a few 64 x 128 tiles is bound by ONE tile's MFMA chain per SIMD: 64 x 64 tiles whose
      // chunk's K range is split between two wave pairs put twice the CUs on half the chain each (one WN layer of H = 192
      // at 8 x 200 frames: 57.9 -> 40.3 us, at 1 x 200: 56.3 -> 39.0).  Only up to 48 such tiles (a request of up to four
      // utterances): in the 32-utterance step the four batch groups of the front part run on four streams and fill the idle
      // CUs with each other's launches -- there the wider footprint LOSES (same box: 57.5-57.6 -> 57.8 ms per step with
      // the split tile on the 96-tile in-layers, profiles/r05_ab_s64g2*.json).  Tuning bits: 1 << 27 never, 1 << 28 up to
      // 512 tiles.
      const int64_t t128 = (int64_t)((a.M + 63) / 64) * ((a.ncols + 127) / 128) * a.B;
      if ((t128 <= 48 || (HSP_DBG(a, 1 << 28) && t128 <= 512)) && !HSP_DBG(a, 1 << 27)) {
        const int e = hsp_conv_tile_S64G2(a, epi, act, s, plan_out);
        if (e != HSP_EINVAL) return e;
      }
      return hsp_conv_tile_S64G(a, epi, act, s, plan_out);
    }
    if (a.M > 32) return hsp_conv_tile_S64(a, epi, act, s, plan_out);
    return hsp_conv_tile_S32(a, epi, act, s, plan_out);
  }
  if (gated) return hsp_conv_tile_M128(a, epi, act, s, plan_out);
  // M > 128: 128 x 128 tiles at two workgroups per CU.  (A 256 x 128 one-per-CU shape existed in round 1; in
  // units of one 128 x 128 tile's MFMA time a CU spends 2 ceil(n256 / 256) against ceil(n128 / 256) <= that, and
  // it lost or tied on every layer of the path, so it is gone.)
  if (a.M > 64) return hsp_conv_tile_M128(a, epi, act, s, plan_out);
  if (a.M > 32) return act ? hsp_conv_tile_M64(a, epi, act, s, plan_out) : hsp_conv_tile_M64P(a, epi, act, s, plan_out);
  return act ? hsp_conv_tile_M32(a, epi, act, s, plan_out) : hsp_conv_tile_M32P(a, epi, act, s, plan_out);
}

}  // namespace

extern "C" int hsp_conv1d_mfma_f32(const hsp_conv1d_args* a, void* stream) {
  if (!a) return HSP_EINVAL;
  if (int e = validate(*a)) return e;
  return dispatch(*a, static_cast<hipStream_t>(stream), nullptr);
}

extern "C" int hsp_conv1d_mfma_plan(const hsp_conv1d_args* a, int32_t out4[4]) {
  if (!a || !out4) return HSP_EINVAL;
  if (int e = validate(*a)) return e;
  return dispatch(*a, nullptr, out4);
}
