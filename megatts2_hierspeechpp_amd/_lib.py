"""ctypes binding of libhsp.so (the C ABI declared in include/hsp.h).

The product path has no CPU or eager fallback: if the shared library is missing, or a
tensor is not an fp32 tensor on a ROCm device, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HSP_LIB", os.path.join(_HERE, "libhsp.so"))  # HSP_LIB: kernel A/B builds

# enums of include/hsp.h
PRO_NONE, PRO_LRELU, PRO_ACT1D, PRO_SILU = 0, 1, 2, 3
ACT_NONE, ACT_TANH, ACT_GELU_TANH, ACT_RELU, ACT_MISH, ACT_SILU, ACT_SOFTPLUS, ACT_GELU_ERF = 0, 1, 2, 3, 4, 5, 6, 7
ROWS_PLAIN, ROWS_GATE_WN, ROWS_GATE_GLU, ROWS_SHUFFLE = 0, 1, 2, 3
MASK_NONE, MASK_PRE, MASK_POST, MASK_BOTH = 0, 1, 2, 3
EINVAL = -1

_fp = C.c_void_p


class Conv1dArgs(C.Structure):
    """Mirror of ``hsp_conv1d_args`` (include/hsp.h) -- field order is the ABI."""
    _fields_ = [
        ("x", _fp), ("x_bs", C.c_int64), ("x_cs", C.c_int64), ("x_ts", C.c_int64),
        ("B", C.c_int32), ("Cin", C.c_int32), ("Lin", C.c_int32),
        ("w", _fp), ("K", C.c_int32), ("M", C.c_int32), ("dil", C.c_int32), ("pad", C.c_int32), ("stride", C.c_int32),
        ("zeros", _fp), ("w_ld", C.c_int32),
        ("y", _fp), ("y_bs", C.c_int64), ("y_cs", C.c_int64),
        ("Cout", C.c_int32), ("Lout", C.c_int32), ("ncols", C.c_int32),
        ("prologue", C.c_int32), ("slope", C.c_float),
        ("alpha_exp", _fp), ("beta_inv", _fp), ("filt", _fp),
        ("rows", C.c_int32), ("gate_half", C.c_int32), ("up", C.c_int32), ("shuf_pad", C.c_int32),
        ("bias", _fp), ("cbias", _fp), ("cbias_bs", C.c_int64),
        ("act", C.c_int32),
        ("mask", _fp), ("mask_bs", C.c_int64), ("mask_mode", C.c_int32),
        ("cscale", _fp), ("cscale_bs", C.c_int64), ("scale", C.c_float),
        ("res", _fp), ("res_bs", C.c_int64), ("res_cs", C.c_int64),
        ("accumulate", C.c_int32), ("post_scale", C.c_float), ("debug", C.c_int32),
        ("ln_c1", _fp), ("ln_eps", C.c_float),
        ("split_row", C.c_int32), ("accumulate2", C.c_int32), ("mask_mode2", C.c_int32), ("y2", _fp),
        ("y2_bs", C.c_int64), ("y2_cs", C.c_int64), ("res_ts", C.c_int64), ("w_bs", C.c_int64),
        ("ln_scale", _fp), ("ln_scale_bs", C.c_int64), ("ln_c1_bs", C.c_int64), ("ln_mask", _fp), ("ln_mask_bs", C.c_int64),
    ]


class MhaArgs(C.Structure):
    """Mirror of ``hsp_mha_args``."""
    _fields_ = [
        ("q", _fp), ("k", _fp), ("v", _fp), ("o", _fp),
        ("q_bs", C.c_int64), ("k_bs", C.c_int64), ("v_bs", C.c_int64), ("o_bs", C.c_int64),
        ("B", C.c_int32), ("H", C.c_int32), ("D", C.c_int32), ("Tq", C.c_int32), ("Tk", C.c_int32),
        ("qk_scale", C.c_float),
        ("mask_q", _fp), ("mask_k", _fp), ("rel_k", _fp), ("rel_v", _fp), ("window", C.c_int32),
        ("q_cs", C.c_int64), ("k_cs", C.c_int64), ("v_cs", C.c_int64), ("o_cs", C.c_int64),
        ("mask_dense", _fp), ("mask_dense_bs", C.c_int64),
    ]


class MhaProjArgs(C.Structure):
    """Mirror of ``hsp_mha_proj_args``."""
    _fields_ = [
        ("q", _fp), ("k", _fp), ("v", _fp),
        ("q_bs", C.c_int64), ("q_cs", C.c_int64), ("k_bs", C.c_int64), ("k_cs", C.c_int64), ("v_bs", C.c_int64),
        ("v_cs", C.c_int64),
        ("B", C.c_int32), ("H", C.c_int32), ("D", C.c_int32), ("Tq", C.c_int32), ("Tk", C.c_int32),
        ("qk_scale", C.c_float),
        ("wt", _fp), ("M", C.c_int32), ("wt_ld", C.c_int32),
        ("bias", _fp), ("mask", _fp), ("mask_bs", C.c_int64), ("cscale", _fp), ("cscale_bs", C.c_int64),
        ("res", _fp), ("res_bs", C.c_int64), ("res_cs", C.c_int64), ("res_ts", C.c_int64),
        ("y", _fp), ("y_bs", C.c_int64), ("y_cs", C.c_int64), ("y_ts", C.c_int64), ("debug", C.c_int32),
    ]


class DftSegArgs(C.Structure):
    """Mirror of ``hsp_dftseg_args``."""
    _fields_ = [
        ("x", _fp), ("x_bs", C.c_int64), ("x_cs", C.c_int64),
        ("y", _fp), ("y_bs", C.c_int64), ("y_cs", C.c_int64),
        ("B", C.c_int32), ("C", C.c_int32), ("L", C.c_int32),
        ("k", C.c_int32), ("dil", C.c_int32), ("pad", C.c_int32), ("nseg", C.c_int32), ("Np", C.c_int32),
        ("xf", _fp), ("xf_bs", C.c_int64), ("dft", _fp),
        ("bias", _fp), ("res", _fp), ("res_bs", C.c_int64), ("res_cs", C.c_int64),
        ("accumulate", C.c_int32), ("post_scale", C.c_float),
        ("act_alpha_exp", _fp), ("act_beta_inv", _fp), ("act_filt", _fp), ("prod3", C.c_int32),
    ]


class Cprod3Args(C.Structure):
    """Mirror of ``hsp_cprod3_args``."""
    _fields_ = [
        ("xf", _fp), ("yf", _fp), ("w", _fp), ("zeros", _fp), ("xf_bs", C.c_int64), ("yf_bs", C.c_int64),
        ("bins", C.c_int32), ("C", C.c_int32), ("Np", C.c_int32), ("debug", C.c_int32),
    ]


WSPEC_BLOCK, WSPEC_THREE = 0, 1


# symbol -> (restype, argtypes); every symbol include/hsp.h declares
SIGNATURES = {
    "hsp_version": (C.c_int, []),
    "hsp_arch": (C.c_char_p, []),
    "hsp_conv1d_mfma_f32": (C.c_int, [C.POINTER(Conv1dArgs), _fp]),
    "hsp_conv1d_direct_f32": (C.c_int, [C.POINTER(Conv1dArgs), _fp]),
    "hsp_conv1d_mfma_plan": (C.c_int, [C.POINTER(Conv1dArgs), C.POINTER(C.c_int32 * 4)]),
    "hsp_act1d_snakebeta_f32": (C.c_int, [_fp, _fp, C.c_int32, C.c_int32, C.c_int32, _fp, _fp, _fp, _fp]),
    "hsp_snake_consts_f32": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int32, _fp]),
    "hsp_fold_weight_norm_f32": (C.c_int, [_fp, _fp, _fp, C.c_int32, C.c_int32, _fp]),
    "hsp_gather_f32": (C.c_int, [_fp, _fp, _fp, C.c_int64, _fp]),
    "hsp_sequence_mask_f32": (C.c_int, [_fp, _fp, C.c_int32, C.c_int32, _fp]),
    "hsp_flip_channels_f32": (C.c_int, [_fp, _fp, C.c_int32, C.c_int32, C.c_int32, _fp]),
    "hsp_sample_prior_f32": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_float, _fp]),
    "hsp_layernorm_mod_f32": (C.c_int, [_fp, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_float, _fp, _fp, _fp,
                                        C.c_int64, _fp, _fp, _fp]),
    "hsp_mha_f32": (C.c_int, [C.POINTER(MhaArgs), _fp]),
    "hsp_mha_proj_f32": (C.c_int, [C.POINTER(MhaProjArgs), _fp]),
    "hsp_dftseg_fwd_f32": (C.c_int, [C.POINTER(DftSegArgs), _fp]),
    "hsp_dftseg_inv_f32": (C.c_int, [C.POINTER(DftSegArgs), _fp]),
    "hsp_dftseg_tables_f32": (C.c_int, [_fp, _fp]),
    "hsp_dftseg_pair_supported": (C.c_int, [C.POINTER(DftSegArgs), C.POINTER(DftSegArgs)]),
    "hsp_dftseg_pair_f32": (C.c_int, [C.POINTER(DftSegArgs), C.POINTER(DftSegArgs), _fp]),
    "hsp_dftseg_supported": (C.c_int, [C.POINTER(DftSegArgs)]),
    "hsp_cprod3_f32": (C.c_int, [C.POINTER(Cprod3Args), _fp]),
    "hsp_cprod3_supported": (C.c_int, [C.POINTER(Cprod3Args)]),
    "hsp_dftseg_weight_spectrum_f32": (C.c_int, [_fp, C.c_int32, C.c_int32, C.c_int32, _fp, _fp, C.c_int32, _fp]),
    "hsp_mha_proj_supported": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "hsp_masked_mean_f32": (C.c_int, [_fp, _fp, _fp, C.c_int32, C.c_int32, C.c_int32, _fp]),
    "hsp_mask_mul_f32": (C.c_int, [_fp, _fp, _fp, C.c_int32, C.c_int32, C.c_int32, _fp]),
    "hsp_linear_interp_f32": (C.c_int, [_fp, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _fp]),
    "hsp_axpby_f32": (C.c_int, [_fp, _fp, _fp, C.c_float, C.c_float, C.c_int64, _fp]),
    "hsp_plm_embed_f32": (C.c_int, [_fp, C.c_int64, C.c_int64, C.c_int32, _fp, C.c_int64, _fp, C.c_int32, C.c_int32,
                                    _fp, C.c_int32, _fp, _fp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _fp]),
    "hsp_plm_embed_step_f32": (C.c_int, [_fp, C.c_int64, C.c_int64, C.c_int32, _fp, C.c_int64, _fp, C.c_int32, C.c_int32,
                                         _fp, C.c_int32, _fp, _fp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _fp, C.c_int64,
                                         C.c_int64, C.c_int32, _fp]),
    "hsp_argmax_f32": (C.c_int, [_fp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _fp, C.c_int64, _fp]),
    "hsp_embedding_sum_f32": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_float, _fp,
                                        C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _fp]),
    "hsp_lstm_bidir_f32": (C.c_int, [_fp, C.c_int64, _fp, _fp, _fp, _fp, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                     C.c_int32, _fp]),
    "hsp_duration_f32": (C.c_int, [_fp, C.c_int64, _fp, C.c_float, _fp, C.c_int64, _fp, C.c_int32, C.c_int32, _fp]),
    "hsp_gaussian_upsample_f32": (C.c_int, [_fp, C.c_int64, C.c_int64, _fp, C.c_int64, _fp, C.c_int64, _fp, _fp, _fp,
                                            C.c_int32, C.c_int32, C.c_int32, C.c_int32, _fp]),
    "hsp_act_f32": (C.c_int, [_fp, _fp, C.c_int64, C.c_int32, _fp]),
    "hsp_reflect_pad_f32": (C.c_int, [_fp, C.c_int64, _fp, C.c_int32, C.c_int32, C.c_int32, _fp]),
    "hsp_f0_convert_f32": (C.c_int, [_fp, C.c_int32, _fp, C.c_int32, _fp, _fp]),
    "hsp_sum_sq_f32": (C.c_int, [_fp, C.c_int64, _fp, _fp]),
    "hsp_mag_pha_f32": (C.c_int, [_fp, C.c_int64, _fp, _fp, C.c_int32, C.c_int32, C.c_float, _fp]),
    "hsp_instnorm_prelu_f32": (C.c_int, [_fp, C.c_int64, C.c_int32, C.c_int64, _fp, _fp, _fp, C.c_float, _fp]),
    "hsp_dwconv_bn_silu_f32": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_float, _fp, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, _fp]),
    "hsp_lsigmoid_mul_f32": (C.c_int, [_fp, _fp, C.c_float, _fp, _fp, C.c_int32, C.c_int32, _fp]),
    "hsp_atan2_f32": (C.c_int, [_fp, _fp, _fp, C.c_int64, _fp]),
    "hsp_polar_f32": (C.c_int, [_fp, _fp, C.c_float, _fp, C.c_int64, _fp, C.c_int64, C.c_int32, C.c_int32, _fp]),
    "hsp_istft_ola_f32": (C.c_int, [_fp, C.c_int64, _fp, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_float, _fp]),
    "hsp_maxpool1d_f32": (C.c_int, [_fp, C.c_int64, C.c_int64, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _fp]),
    "hsp_vq_nearest_f32": (C.c_int, [_fp, C.c_int64, C.c_int64, _fp, _fp, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32, _fp]),
    "hsp_add_cbias_f32": (C.c_int, [_fp, C.c_int64, C.c_int64, _fp, C.c_int64, _fp, C.c_int32, C.c_int32, C.c_int32, _fp]),
    "hsp_zero_below_f32": (C.c_int, [_fp, C.c_float, _fp, C.c_int64, _fp]),
    "hsp_peak_int16": (C.c_int, [_fp, C.c_int64, _fp, C.c_float, _fp, C.c_int64, C.c_int32, C.c_int64, _fp]),
    "hsp_copy_strided_f32": (C.c_int, [_fp, C.c_int64, C.c_int64, C.c_int64, _fp, C.c_int32, C.c_int32, C.c_int32,
                                       _fp]),
    "hsp_conv1d_f32": (C.c_int, [C.POINTER(Conv1dArgs), _fp]),
    "hsp_convtr1d_f32": (C.c_int, [C.POINTER(Conv1dArgs), _fp]),
    "hsp_wn_layer_f32": (C.c_int, [C.POINTER(Conv1dArgs), C.POINTER(Conv1dArgs), C.POINTER(Conv1dArgs), _fp]),
    "hsp_ffn_conv_f32": (C.c_int, [C.POINTER(Conv1dArgs), C.POINTER(Conv1dArgs), _fp]),
    "hsp_layernorm_modulate_f32": (C.c_int, [_fp, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_float, _fp, _fp, _fp,
                                             C.c_int64, _fp]),
    "hsp_stft_frames_f32": (C.c_int, [_fp, _fp, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      _fp]),
    "hsp_power_mel_log_f32": (C.c_int, [_fp, C.c_int64, C.c_int32, _fp, _fp, _fp, _fp, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_float, _fp]),
}

_lib: Optional[C.CDLL] = None


class HspError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libhsp.so once.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HspError(
                f"{LIB_PATH} not found: build the HIP kernels first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C megatts2_hierspeechpp_amd/csrc). "
                "There is no CPU fallback for the product path.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        kind = "invalid arguments" if code == EINVAL else f"hipError_t {code}"
        raise HspError(f"{what} failed: {kind}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of an fp32 (or int) ROCm tensor; None passes through as NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HspError("libhsp kernels run on the GPU only (got a CPU tensor); there is no CPU fallback")
    return t.data_ptr()


def fptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is not None and t.dtype != torch.float32:
        raise HspError(f"expected float32, got {t.dtype}")
    return ptr(t)


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream
