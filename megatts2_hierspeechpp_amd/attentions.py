"""attentions.MultiHeadAttention with the reference's parameter names (reference:
attentions.py:109-188).  Used by StyleEncoder (no relative window) and, for the
text front-end, with window_size=4."""
from __future__ import annotations

import math

import torch
from torch import nn

from . import _lib as L
from . import functional as Fh
from .hip_layers import Conv1d, HipLayer
from .modules import LayerNorm


class Encoder(nn.Module):
    """attentions.Encoder (attentions.py:13-50): post-LN transformer with windowed
    relative-position self-attention and a conv FFN (text / w2v / mel encoders of the front-end)."""

    def __init__(self, hidden_channels, filter_channels, n_heads, n_layers, kernel_size=1, p_dropout=0., window_size=4,
                 **kwargs):
        super().__init__()
        self.hidden_channels, self.filter_channels = hidden_channels, filter_channels
        self.n_heads, self.n_layers, self.kernel_size, self.window_size = n_heads, n_layers, kernel_size, window_size
        self.attn_layers = nn.ModuleList()
        self.norm_layers_1 = nn.ModuleList()
        self.ffn_layers = nn.ModuleList()
        self.norm_layers_2 = nn.ModuleList()
        for _ in range(n_layers):
            self.attn_layers.append(MultiHeadAttention(hidden_channels, hidden_channels, n_heads, p_dropout=p_dropout,
                                                       window_size=window_size))
            self.norm_layers_1.append(LayerNorm(hidden_channels))
            self.ffn_layers.append(FFN(hidden_channels, hidden_channels, filter_channels, kernel_size, p_dropout=p_dropout))
            self.norm_layers_2.append(LayerNorm(hidden_channels))

    def forward(self, x, x_mask):
        x = Fh.mask_mul(x, x_mask)
        for i in range(self.n_layers):
            x = self.norm_layers_1[i](self.attn_layers[i](x, x, mask_q=x_mask, mask_k=x_mask, res=x))
            x = self.norm_layers_2[i](self.ffn_layers[i](x, x_mask, res=x))
        return Fh.mask_mul(x, x_mask)


class FFN(nn.Module):
    """attentions.FFN (attentions.py:266-313), non-causal 'same' padding, ReLU."""

    def __init__(self, in_channels, out_channels, filter_channels, kernel_size, p_dropout=0., activation=None,
                 causal=False):
        super().__init__()
        if causal or activation == "gelu":
            raise NotImplementedError("only the non-causal ReLU FFN is instantiated by the reference's encoders")
        if kernel_size % 2 != 1:
            raise NotImplementedError("even FFN kernels need asymmetric padding")
        self.kernel_size = kernel_size
        self.conv_1 = Conv1d(in_channels, filter_channels, kernel_size, padding=(kernel_size - 1) // 2)
        self.conv_2 = Conv1d(filter_channels, out_channels, kernel_size, padding=(kernel_size - 1) // 2)

    def forward(self, x, x_mask, res=None):
        # conv_1(pad(x * mask)) -> relu -> conv_2(pad(h * mask)) * mask  [+ residual for the caller's x + y]
        h = self.conv_1(Fh.mask_mul(x, x_mask), act=L.ACT_RELU, mask=x_mask, mask_mode=L.MASK_PRE)
        return self.conv_2(h, mask=x_mask, mask_mode=L.MASK_PRE, res=res)


class MultiHeadAttention(HipLayer):
    def __init__(self, channels, out_channels, n_heads, p_dropout=0., window_size=None, heads_share=True,
                 block_length=None, proximal_bias=False, proximal_init=False):
        super().__init__()
        assert channels % n_heads == 0
        if block_length is not None or proximal_bias:
            raise NotImplementedError("block_length / proximal_bias are never enabled by the reference's models")
        self.channels, self.out_channels, self.n_heads = channels, out_channels, n_heads
        self.window_size = window_size
        self.k_channels = channels // n_heads
        self.conv_q = Conv1d(channels, channels, 1)
        self.conv_k = Conv1d(channels, channels, 1)
        self.conv_v = Conv1d(channels, channels, 1)
        self.conv_o = Conv1d(channels, out_channels, 1)
        if window_size is not None:
            if not heads_share:
                raise NotImplementedError("heads_share=False is never used by the reference")
            self.emb_rel_k = nn.Parameter(torch.zeros(1, window_size * 2 + 1, self.k_channels), requires_grad=False)
            self.emb_rel_v = nn.Parameter(torch.zeros(1, window_size * 2 + 1, self.k_channels), requires_grad=False)
        self._rk = self._rv = None

    def hsp_requests(self):
        n = (2 * self.window_size + 1) * self.k_channels if self.window_size is not None else 0
        return [("rk", n), ("rv", n)] if n else []

    def hsp_fill(self, arena, materialize):
        if self.window_size is None:
            return
        self._rk, self._rv = arena.view(self, "rk"), arena.view(self, "rv")
        if materialize:
            self._rk.copy_(self.emb_rel_k.data.reshape(-1))
            self._rv.copy_(self.emb_rel_v.data.reshape(-1))

    def forward(self, x, c, attn_mask=None, *, mask_q=None, mask_k=None, res=None, cbias=None, out=None):
        """The reference's signature ``forward(x, c, attn_mask=None)`` (attentions.py:147-155): ``attn_mask``
        [B, 1, Tq, Tk] (bool or float, 0 = masked, shared by the heads) is applied element by element inside the
        attention kernel (hsp_mha_args.mask_dense).  The mirrors' own call sites pass the two [B, 1, T] factors
        ``mask_q`` / ``mask_k`` instead -- the reference's masks are always that outer product (attentions.py:39,
        styleencoder.py:71) -- which saves building and reading the [Tq, Tk] matrix."""
        q, k, v = self.conv_q(x), self.conv_k(c), self.conv_v(c)
        rel_k = self._rk if self.window_size is not None else None
        rel_v = self._rv if self.window_size is not None else None
        o = Fh.mha(q, k, v, self.n_heads, 1.0 / math.sqrt(self.k_channels), mask_q=mask_q, mask_k=mask_k,
                   rel_k=rel_k, rel_v=rel_v, window=self.window_size or 0, mask_dense=attn_mask)
        return self.conv_o(o, res=res, cbias=cbias, out=out)
