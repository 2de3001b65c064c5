"""attentions.MultiHeadAttention with the reference's parameter names (reference:
attentions.py:109-188).  Used by StyleEncoder (no relative window) and, for the
text front-end, with window_size=4."""
from __future__ import annotations

import math

import torch
from torch import nn

from . import functional as Fh
from .hip_layers import Conv1d


class MultiHeadAttention(nn.Module):
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

    def forward(self, x, c, attn_mask=None, *, mask_q=None, mask_k=None, res=None):
        """``attn_mask`` of the reference is always mask_k[b, j] * mask_q[b, i]
        (attentions.py:39, styleencoder.py:71); pass the two [B, 1, T] factors."""
        if attn_mask is not None:
            raise NotImplementedError("pass mask_q / mask_k ([B,1,T]) instead of the outer-product attn_mask")
        q, k, v = self.conv_q(x), self.conv_k(c), self.conv_v(c)
        rel_k = self.emb_rel_k[0] if self.window_size is not None else None
        rel_v = self.emb_rel_v[0] if self.window_size is not None else None
        o = Fh.mha(q, k, v, self.n_heads, 1.0 / math.sqrt(self.k_channels), mask_q=mask_q, mask_k=mask_k,
                   rel_k=rel_k, rel_v=rel_v, window=self.window_size or 0)
        return self.conv_o(o, res=res)
