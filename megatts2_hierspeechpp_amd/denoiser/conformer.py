"""denoiser/conformer.py with the reference's class names and state-dict keys.  Tensors are channel-major
``[A, C, N]`` where the reference holds ``[A, N, C]``: the feed-forward and convolution modules work along N, and --
because the reference builds ``nn.MultiheadAttention`` without ``batch_first`` (conformer.py:48) -- the attention
runs along A with N as its batch.  That behaviour is kept as it is."""
from __future__ import annotations

import torch
from torch import nn

from .. import _lib as L
from .. import functional as Fh
from ..hip_layers import Conv1d, HipLayer, LinearCT, _SubArena
from ..ttv_v1.transformer_mega import LayerNorm
from .utils import Vec


class FeedForwardModule(nn.Module):
    """conformer.py:9-22: LayerNorm, Linear, SiLU, Dropout, Linear, Dropout (keys ffm.0 / ffm.1 / ffm.4)."""

    def __init__(self, dim, mult=4, dropout=0):
        super().__init__()
        self.ffm = nn.ModuleDict({"0": LayerNorm(dim), "1": LinearCT(dim, dim * mult), "4": LinearCT(dim * mult, dim)})

    def forward(self, x, scale=1.0):
        """x + scale * ffm(x)."""
        h = self.ffm["1"](self.ffm["0"](x), act=L.ACT_SILU)
        return self.ffm["4"](h, scale=scale, res=x)


class _DepthwiseConv1d(Vec):
    """nn.Conv1d(C, C, k, padding = k // 2, groups = C): weight [C, 1, k], bias [C]."""

    NAMES = ("weight", "bias")

    def __init__(self, channels, k):
        super().__init__()
        self.channels, self.k = channels, k
        self.weight = nn.Parameter(torch.zeros(channels, 1, k), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(channels), requires_grad=False)


class _BatchNorm1d(Vec):
    """nn.BatchNorm1d in eval mode (running statistics)."""

    NAMES = ("weight", "bias", "running_mean", "running_var")

    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(channels), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(channels), requires_grad=False)
        self.register_buffer("running_mean", torch.zeros(channels))
        self.register_buffer("running_var", torch.ones(channels))
        self.register_buffer("num_batches_tracked", torch.zeros((), dtype=torch.int64))


class ConformerConvModule(nn.Module):
    """conformer.py:25-42 (keys ccm.0 LayerNorm, ccm.2 pointwise conv + GLU, ccm.4 depthwise conv, ccm.5 BatchNorm,
    ccm.7 pointwise conv): the GLU is the conv's gated epilogue, depthwise conv + BatchNorm + SiLU are one launch."""

    def __init__(self, dim, expansion_factor=2, kernel_size=31, dropout=0.):
        super().__init__()
        inner = dim * expansion_factor
        self.inner = inner
        self.ccm = nn.ModuleDict({"0": LayerNorm(dim), "2": Conv1d(dim, inner * 2, 1, rows=L.ROWS_GATE_GLU),
                                  "4": _DepthwiseConv1d(inner, kernel_size), "5": _BatchNorm1d(inner),
                                  "7": Conv1d(inner, dim, 1)})

    def forward(self, x):
        """x + ccm(x)."""
        h = self.ccm["2"](self.ccm["0"](x))                    # [A, inner, N]
        dw, bn = self.ccm["4"], self.ccm["5"]
        y = torch.empty_like(h)
        A, Cc, N = h.shape
        L.check(L.lib().hsp_dwconv_bn_silu_f32(L.fptr(h), L.fptr(dw.dev("weight")), L.fptr(dw.dev("bias")),
                                               L.fptr(bn.dev("weight")), L.fptr(bn.dev("bias")),
                                               L.fptr(bn.dev("running_mean")), L.fptr(bn.dev("running_var")),
                                               float(bn.eps), L.fptr(y), A, Cc, N, dw.k, L.stream_ptr()),
                "hsp_dwconv_bn_silu_f32")
        return self.ccm["7"](y, res=x)


class _MultiheadAttention(HipLayer):
    """torch.nn.MultiheadAttention(dim, n_head) parameters (``in_proj_weight`` / ``in_proj_bias`` / ``out_proj``);
    the stacked q/k/v projection is one GEMM."""

    def __init__(self, dim, n_head):
        super().__init__()
        self.dim, self.n_head = dim, n_head
        self.in_proj_weight = nn.Parameter(torch.zeros(3 * dim, dim), requires_grad=False)
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * dim), requires_grad=False)
        self.out_proj = LinearCT(dim, dim)
        self.__dict__["_qkv"] = LinearCT(dim, 3 * dim)      # not registered: its rows are the in_proj parameters

    def hsp_requests(self):
        return [("qkv." + n, numel) for n, numel in self._qkv.hsp_requests()]

    def hsp_fill(self, arena, materialize):
        if materialize:
            self._qkv.weight.data = self.in_proj_weight.data
            self._qkv.bias.data = self.in_proj_bias.data
        self._qkv.hsp_fill(_SubArena(arena, self, "qkv."), materialize)


class AttentionModule(nn.Module):
    """conformer.py:45-56."""

    def __init__(self, dim, n_head=8, dropout=0.):
        super().__init__()
        self.attn = _MultiheadAttention(dim, n_head)
        self.layernorm = LayerNorm(dim)

    def forward(self, x):
        """x + attn(layernorm(x)), attention along dim 0 of [A, C, N] for every n."""
        m = self.attn
        qkv = m._qkv(self.layernorm(x))                                   # [A, 3C, N]
        qkv_t = Fh.copy_strided(qkv.permute(2, 1, 0))                     # [N, 3C, A]: A is the sequence axis
        C_ = m.dim
        o = Fh.mha(qkv_t[:, :C_], qkv_t[:, C_:2 * C_], qkv_t[:, 2 * C_:], m.n_head, (C_ // m.n_head) ** -0.5)
        return m.out_proj(Fh.copy_strided(o.permute(2, 1, 0)), res=x)     # back to [A, C, N]


class ConformerBlock(nn.Module):
    """conformer.py:59-76."""

    def __init__(self, dim, n_head=8, ffm_mult=4, ccm_expansion_factor=2, ccm_kernel_size=31, ffm_dropout=0.,
                 attn_dropout=0., ccm_dropout=0.):
        super().__init__()
        self.ffm1 = FeedForwardModule(dim, ffm_mult, dropout=ffm_dropout)
        self.attn = AttentionModule(dim, n_head, dropout=attn_dropout)
        self.ccm = ConformerConvModule(dim, ccm_expansion_factor, ccm_kernel_size, dropout=ccm_dropout)
        self.ffm2 = FeedForwardModule(dim, ffm_mult, dropout=ffm_dropout)
        self.post_norm = LayerNorm(dim)

    def forward(self, x):
        x = self.ffm1(x, scale=0.5)
        x = self.attn(x)
        x = self.ccm(x)
        x = self.ffm2(x, scale=0.5)
        return self.post_norm(x)
