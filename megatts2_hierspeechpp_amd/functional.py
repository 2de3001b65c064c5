"""Thin Python wrappers (torch tensors in/out) over the pointwise / reduction entry
points of libhsp.so.  No torch arithmetic happens here: every op is one HIP launch."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib as L


def _new_like(x):
    return torch.empty(x.shape, dtype=torch.float32, device=x.device)


def _c(x: torch.Tensor) -> torch.Tensor:
    return x if x.is_contiguous() else x.contiguous()


def sequence_mask(length: torch.Tensor, max_length: int) -> torch.Tensor:
    """commons.sequence_mask (reference commons.py:128-132) as a float mask [B, 1, T]."""
    length = _c(length.to(torch.int64))
    B = length.shape[0]
    mask = torch.empty(B, 1, max_length, dtype=torch.float32, device=length.device)
    L.check(L.lib().hsp_sequence_mask_f32(L.ptr(length), L.fptr(mask), B, max_length, L.stream_ptr()),
            "hsp_sequence_mask_f32")
    return mask


ACT_HOOK = None  # measurement hook (bench.py): hook(algorithmic_bytes, ev_start, ev_end) around every stand-alone activation


def act1d(x, ea, binv, filt, out=None):
    x = _c(x)
    B, Cc, T = x.shape
    out = _new_like(x) if out is None else out
    hook = ACT_HOOK
    if hook is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(L.lib().hsp_act1d_snakebeta_f32(L.fptr(x), L.fptr(out), B, Cc, T, L.fptr(ea), L.fptr(binv), L.fptr(filt),
                                            L.stream_ptr()), "hsp_act1d_snakebeta_f32")
    if hook is not None:
        e1.record()
        hook(8 * B * Cc * T, e0, e1)   # one fp32 read + one fp32 write per element
    return out


def flip_channels(x):
    x = _c(x)
    B, Cc, T = x.shape
    y = _new_like(x)
    L.check(L.lib().hsp_flip_channels_f32(L.fptr(x), L.fptr(y), B, Cc, T, L.stream_ptr()), "hsp_flip_channels_f32")
    return y


def sample_prior(stats, noise, mask, noise_scale: float):
    """z = (m + noise * exp(logs) * noise_scale) * mask; stats = [B, 2C, T]."""
    stats, noise, mask = _c(stats), _c(noise), _c(mask)
    B, C2, T = stats.shape
    z = torch.empty(B, C2 // 2, T, dtype=torch.float32, device=stats.device)
    assert noise.shape == z.shape, (noise.shape, z.shape)
    L.check(L.lib().hsp_sample_prior_f32(L.fptr(stats), L.fptr(noise), L.fptr(mask), L.fptr(z), B, C2 // 2, T,
                                         float(noise_scale), L.stream_ptr()), "hsp_sample_prior_f32")
    return z


def layernorm_mod(x, eps: float, mask=None, shift=None, scale=None, gamma=None, beta=None):
    """LayerNorm over C of [B, C, T] (+ optional affine), * mask, then * (1 + scale) + shift."""
    x = _c(x)
    B, Cc, T = x.shape
    y = _new_like(x)
    mod_bs = 0
    if shift is not None:
        assert shift.stride(1) == 1 and scale.stride(1) == 1 and shift.stride(0) == scale.stride(0)
        mod_bs = shift.stride(0)
    from . import hip_layers
    if hip_layers.SURVEY_ABI and gamma is None and beta is None:   # SURVEY.md §8(b) name of the same launch
        L.check(L.lib().hsp_layernorm_modulate_f32(L.fptr(x), L.fptr(y), B, Cc, T, float(eps),
                                                   L.fptr(_c(mask)) if mask is not None else None,
                                                   L.fptr(shift), L.fptr(scale), mod_bs, L.stream_ptr()),
                "hsp_layernorm_modulate_f32")
        return y
    L.check(L.lib().hsp_layernorm_mod_f32(L.fptr(x), L.fptr(y), B, Cc, T, float(eps),
                                          L.fptr(_c(mask)) if mask is not None else None,
                                          L.fptr(shift), L.fptr(scale), mod_bs, L.fptr(gamma), L.fptr(beta),
                                          L.stream_ptr()), "hsp_layernorm_mod_f32")
    return y


def mha(q, k, v, n_heads: int, qk_scale: float, mask_q=None, mask_k=None, rel_k=None, rel_v=None, window=0,
        out=None, mask_dense=None, force_stream=False):
    """q [B, H*D, Tq], k/v [B, H*D, Tk] (any batch / channel strides, unit time stride) -> [B, H*D, Tq].
    Strided views let a batch live side by side on the column axis of one [C, B*T] matrix
    (``x.view(C, B, T).permute(1, 0, 2)``), the layout of the PLM loop."""
    B, HD, Tq = q.shape
    Tk = k.shape[2]
    o = torch.empty(B, HD, Tq, dtype=torch.float32, device=q.device) if out is None else out
    for t_, T_ in ((q, Tq), (k, Tk), (v, Tk), (o, Tq)):
        assert (t_.stride(2) == 1 or T_ == 1) and t_.stride(1) >= T_, "attention operands need unit time stride"
    a = L.MhaArgs()
    a.q, a.k, a.v, a.o = L.fptr(q), L.fptr(k), L.fptr(v), L.fptr(o)
    a.q_bs, a.k_bs, a.v_bs, a.o_bs = q.stride(0), k.stride(0), v.stride(0), o.stride(0)
    a.q_cs, a.k_cs, a.v_cs, a.o_cs = q.stride(1), k.stride(1), v.stride(1), o.stride(1)
    a.B, a.H, a.D, a.Tq, a.Tk = B, n_heads, HD // n_heads, Tq, Tk
    a.qk_scale = float(qk_scale)
    if mask_q is not None:
        a.mask_q, a.mask_k = L.fptr(_c(mask_q)), L.fptr(_c(mask_k))
    if rel_k is not None:
        a.rel_k, a.rel_v, a.window = L.fptr(_c(rel_k)), L.fptr(_c(rel_v)), window
    if mask_dense is not None:
        # the reference's general attn_mask [B, 1, Tq, Tk] / [B, Tq, Tk] (attentions.py:147-155): 0 -> -1e4
        md = mask_dense.reshape(B, Tq, Tk)
        if md.dtype != torch.float32:
            md = md.to(torch.float32)            # a dtype cast of the caller's bool mask, no arithmetic
        md = _c(md)
        a.mask_dense, a.mask_dense_bs = L.fptr(md), md.stride(0)
    if force_stream:                              # tests: the key-streaming kernels at any length (hsp.h)
        a.window = -(a.window + 1)
    L.check(L.lib().hsp_mha_f32(C.byref(a), L.stream_ptr()), "hsp_mha_f32")
    return o


# HSP_FUSE_MHA_PROJ=0: attention and its output projection as two launches again (same-box A/B, tests of both paths)
FUSE_MHA_PROJ = os.environ.get("HSP_FUSE_MHA_PROJ", "1") == "1"


def mha_proj_supported(n_heads: int, head_dim: int, m: int, tk: int) -> bool:
    return FUSE_MHA_PROJ and bool(L.lib().hsp_mha_proj_supported(n_heads, head_dim, m, tk))


def mha_proj(q, k, v, n_heads: int, qk_scale: float, wt, bias=None, mask=None, cscale=None, res=None, out=None):
    """Attention over all heads + output projection + epilogue in one launch (hsp_mha_proj_f32):
    y = ((wt @ attention(q, k, v) + bias) * mask) * cscale + res.  q [B, H*D, Tq], k / v [B, H*D, Tk] with unit time
    stride (strided views as for ``mha``); ``wt`` [M, H*D] row-major (the nn.Linear weight as stored); ``res`` / ``out``
    [B, M, Tq] with ANY strides; ``mask`` [B, 1, Tq] or [B, Tq]; ``cscale`` [B, M]."""
    B, HD, Tq = q.shape
    Tk = k.shape[2]
    M = wt.shape[0]
    y = torch.empty(B, M, Tq, dtype=torch.float32, device=q.device) if out is None else out
    assert y.shape == (B, M, Tq) and wt.shape == (M, HD) and wt.stride(1) == 1
    for t_, T_ in ((q, Tq), (k, Tk), (v, Tk)):
        assert (t_.stride(2) == 1 or T_ == 1), "attention operands need unit time stride"
    a = L.MhaProjArgs()
    a.q, a.k, a.v = L.fptr(q), L.fptr(k), L.fptr(v)
    a.q_bs, a.q_cs, a.k_bs, a.k_cs, a.v_bs, a.v_cs = q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1)
    a.B, a.H, a.D, a.Tq, a.Tk = B, n_heads, HD // n_heads, Tq, Tk
    a.qk_scale = float(qk_scale)
    a.wt, a.M, a.wt_ld = L.fptr(wt), M, wt.stride(0)
    if bias is not None:
        a.bias = L.fptr(bias)
    if mask is not None:
        mk = mask.reshape(B, -1)
        assert mk.shape[1] == Tq and (mk.stride(1) == 1 or Tq == 1)
        a.mask, a.mask_bs = L.fptr(mk), mk.stride(0)
    if cscale is not None:
        assert cscale.shape[:2] == (B, M) and cscale.stride(1) == 1
        a.cscale, a.cscale_bs = L.fptr(cscale), cscale.stride(0)
    if res is not None:
        assert res.shape == y.shape
        a.res, a.res_bs, a.res_cs, a.res_ts = L.fptr(res), res.stride(0), res.stride(1), max(res.stride(2), 1)
    a.y, a.y_bs, a.y_cs, a.y_ts = L.fptr(y), y.stride(0), y.stride(1), max(y.stride(2), 1)
    L.check(L.lib().hsp_mha_proj_f32(C.byref(a), L.stream_ptr()), "hsp_mha_proj_f32")
    return y


def masked_mean(x, mask):
    x, mask = _c(x), _c(mask)
    B, Cc, T = x.shape
    out = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
    L.check(L.lib().hsp_masked_mean_f32(L.fptr(x), L.fptr(mask), L.fptr(out), B, Cc, T, L.stream_ptr()),
            "hsp_masked_mean_f32")
    return out


def mask_mul(x, mask):
    x, mask = _c(x), _c(mask)
    B, Cc, T = x.shape
    y = _new_like(x)
    L.check(L.lib().hsp_mask_mul_f32(L.fptr(x), L.fptr(mask), L.fptr(y), B, Cc, T, L.stream_ptr()), "hsp_mask_mul_f32")
    return y


def axpby(x, z, a: float, b: float):
    x, z = _c(x), _c(z)
    assert x.shape == z.shape
    y = _new_like(x)
    L.check(L.lib().hsp_axpby_f32(L.fptr(x), L.fptr(z), L.fptr(y), float(a), float(b), x.numel(), L.stream_ptr()),
            "hsp_axpby_f32")
    return y


def linear_interp(x, out_len: int):
    """F.interpolate(x, out_len, mode='linear') along the last axis of [B, C, L]."""
    x = _c(x)
    B, Cc, Lin = x.shape
    y = torch.empty(B, Cc, out_len, dtype=torch.float32, device=x.device)
    L.check(L.lib().hsp_linear_interp_f32(L.fptr(x), L.fptr(y), B, Cc, Lin, out_len, L.stream_ptr()),
            "hsp_linear_interp_f32")
    return y


def copy_strided(x):
    """Contiguous copy of a strided [B, C, T] view (one launch, no torch arithmetic)."""
    B, Cc, T = x.shape
    y = torch.empty(B, Cc, T, dtype=torch.float32, device=x.device)
    L.check(L.lib().hsp_copy_strided_f32(L.fptr(x), x.stride(0), x.stride(1), x.stride(2), L.fptr(y), B, Cc, T,
                                         L.stream_ptr()), "hsp_copy_strided_f32")
    return y


def add_cbias(x, cb):
    """x [B, C, T] (strided ok) + cb [B, C(, 1)] broadcast over T -> contiguous [B, C, T]."""
    B, Cc, T = x.shape
    assert x.stride(2) == 1 and cb.stride(1) == 1 and cb.shape[:2] == (B, Cc)
    y = torch.empty(B, Cc, T, dtype=torch.float32, device=x.device)
    L.check(L.lib().hsp_add_cbias_f32(L.fptr(x), x.stride(0), x.stride(1), L.fptr(cb), cb.stride(0), L.fptr(y), B, Cc, T,
                                      L.stream_ptr()), "hsp_add_cbias_f32")
    return y


def embedding_sum(ids, tables, n_rows, scale: float, channels: int, out=None):
    """Channel-major sum of up to three embedding lookups: ids = list of int64 [B, T], tables = list of
    flat fp32 tables [rows * channels] -> [B, channels, T]."""
    B, T = ids[0].shape
    if out is None:
        out = torch.empty(B, channels, T, dtype=torch.float32, device=ids[0].device)
    ids = [_c(i.to(torch.int64)) for i in ids] + [None] * (3 - len(ids))
    tables = list(tables) + [None] * (3 - len(tables))
    n_rows = list(n_rows) + [0] * (3 - len(n_rows))
    L.check(L.lib().hsp_embedding_sum_f32(L.ptr(ids[0]), L.ptr(ids[1]), L.ptr(ids[2]), L.fptr(tables[0]),
                                          L.fptr(tables[1]), L.fptr(tables[2]), n_rows[0], n_rows[1], n_rows[2],
                                          float(scale), L.fptr(out), out.stride(0), out.stride(1), B, channels, T,
                                          L.stream_ptr()), "hsp_embedding_sum_f32")
    return out


def act(x, kind: int):
    """y = act(x) elementwise (HSP_ACT_*)."""
    x = _c(x)
    y = torch.empty_like(x)
    L.check(L.lib().hsp_act_f32(L.fptr(x), L.fptr(y), x.numel(), kind, L.stream_ptr()), "hsp_act_f32")
    return y


def reflect_pad(x, pad: int):
    """F.pad(x, (pad, pad), "reflect") on the last axis of [B, L] / [B, 1, L] audio."""
    shp = x.shape
    x2 = x.reshape(-1, shp[-1])
    assert x2.stride(1) == 1
    y = torch.empty(x2.shape[0], shp[-1] + 2 * pad, dtype=torch.float32, device=x.device)
    L.check(L.lib().hsp_reflect_pad_f32(L.fptr(x2), x2.stride(0), L.fptr(y), x2.shape[0], shp[-1], pad, L.stream_ptr()),
            "hsp_reflect_pad_f32")
    return y.reshape(*shp[:-1], shp[-1] + 2 * pad)


def f0_convert(f0_src, f0_trg):
    """inference_vc.py:80-81,104-105 for one utterance: -> log(f0' + 1) with the source's voiced frames moved to the
    target speaker's voiced mean / std."""
    s, t = _c(f0_src.reshape(-1)), _c(f0_trg.reshape(-1))
    out = torch.empty_like(s)
    L.check(L.lib().hsp_f0_convert_f32(L.fptr(s), s.numel(), L.fptr(t), t.numel(), L.fptr(out), L.stream_ptr()),
            "hsp_f0_convert_f32")
    return out.reshape(f0_src.shape)
