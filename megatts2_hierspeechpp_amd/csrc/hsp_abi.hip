// Dispatching entry points under the names SURVEY.md §8(b) lists as the minimum C ABI of the path.  Host code
// only: each one forwards to the kernels' own entry points (hsp_conv1d_mfma_f32 / hsp_conv1d_direct_f32 /
// hsp_layernorm_mod_f32) with the selection rule the Python mirror (hip_layers.Conv1d.forward) applies.
#include "hsp_device.h"

namespace {
// shapes the MFMA kernels do not take (stride, degenerate channel / length counts, SiLU prologue) go to the VALU kernel
bool wants_direct(const hsp_conv1d_args& a) {
  const bool gated = a.rows == HSP_ROWS_GATE_WN || a.rows == HSP_ROWS_GATE_GLU;
  return !gated && a.rows == HSP_ROWS_PLAIN && a.prologue != HSP_PRO_ACT1D && !a.ln_c1 && !a.w_bs &&
         (a.stride != 1 || a.Cin < 8 || a.Cout < 8 || a.Lout < 8 || a.prologue == HSP_PRO_SILU);
}
}  // namespace

extern "C" int hsp_conv1d_f32(const hsp_conv1d_args* a, void* stream) {
  if (!a || a->rows == HSP_ROWS_SHUFFLE) return HSP_EINVAL;
  return wants_direct(*a) ? hsp_conv1d_direct_f32(a, stream) : hsp_conv1d_mfma_f32(a, stream);
}

extern "C" int hsp_convtr1d_f32(const hsp_conv1d_args* a, void* stream) {
  if (!a || a->rows != HSP_ROWS_SHUFFLE) return HSP_EINVAL;
  return hsp_conv1d_mfma_f32(a, stream);
}

// One WN layer (modules.py:148-176): gated in-conv, then the res / skip 1x1s.  Round 2 ran this as ONE launch where it
// could (gemm2_kernel: the gated activations of a 32-column tile kept in LDS); round 4 retired that kernel with numbers
// -- tools/stage_split.py, same box, prior encoder + flows of the 32 x 4 s step: 9.84 / 9.64 / 9.77 ms with it against
// 9.06 / 8.47 / 7.85 ms layer by layer at one / two / four batch groups (profiles/r04_stage_split_policies.txt): every
// 32-column tile streamed the whole weight set through its CU (16 flop per L2 byte).  The entry point stays: it is the
// name SURVEY.md 8(b) gives the operation, and it issues the layer's launches.
extern "C" int hsp_wn_layer_f32(const hsp_conv1d_args* in_layer, const hsp_conv1d_args* res, const hsp_conv1d_args* skip,
                                void* stream) {
  if (!in_layer || in_layer->rows != HSP_ROWS_GATE_WN || (!res && !skip)) return HSP_EINVAL;
  if ((res && res->rows != HSP_ROWS_PLAIN) || (skip && skip->rows != HSP_ROWS_PLAIN)) return HSP_EINVAL;
  if (int e = hsp_conv1d_mfma_f32(in_layer, stream)) return e;
  if (res)
    if (int e = hsp_conv1d_f32(res, stream)) return e;
  return skip ? hsp_conv1d_f32(skip, stream) : 0;
}

// The DiT block's FFN_Conv (modules.py:382-388): conv k = 5 -> GELU -> 1x1 (+ gate, residual, mask in the second
// launch's epilogue).  Same history as hsp_wn_layer_f32.
extern "C" int hsp_ffn_conv_f32(const hsp_conv1d_args* fc1, const hsp_conv1d_args* fc2, void* stream) {
  if (!fc1 || !fc2 || fc1->rows != HSP_ROWS_PLAIN || fc2->rows != HSP_ROWS_PLAIN) return HSP_EINVAL;
  if (int e1 = hsp_conv1d_f32(fc1, stream)) return e1;
  return hsp_conv1d_f32(fc2, stream);
}

extern "C" int hsp_layernorm_modulate_f32(const float* x, float* y, int32_t B, int32_t C, int32_t T, float eps,
                                          const float* mask, const float* shift, const float* scale, int64_t mod_bs,
                                          void* stream) {
  return hsp_layernorm_mod_f32(x, y, B, C, T, eps, mask, shift, scale, mod_bs, nullptr, nullptr, stream);
}
