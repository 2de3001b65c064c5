"""HIP mirror of reference ttv_v1/vits_models.py (DurationPredictor only; the rest is training code)."""
from __future__ import annotations

from torch import nn

from .. import _lib as L
from .. import functional as Fh
from ..hip_layers import Conv1d
from ..modules import LayerNorm
from .lstm import LSTM


class DurationPredictor(nn.Module):
    """vits_models.DurationPredictor (:94-142): cond -> 2-layer BiLSTM -> LayerNorm -> ReLU -> 1x1 proj
    -> softplus.  ``lengths`` replaces the reference's ``x * x_mask`` before the (unpacked) LSTM: each
    utterance runs over its own phones, which is what the reference computes at B = 1."""

    def __init__(self, in_channels, filter_channels, kernel_size, p_dropout, gin_channels=0):
        super().__init__()
        self.in_channels, self.filter_channels, self.gin_channels = in_channels, filter_channels, gin_channels
        self.lstms = LSTM(in_channels, filter_channels, num_layers=2, bidirectional=True, batch_first=True)
        self.norm_2 = LayerNorm(filter_channels * 2)
        self.proj = Conv1d(filter_channels * 2, 1, 1)
        if gin_channels != 0:
            self.cond = Conv1d(gin_channels, in_channels, 1)

    def forward(self, x, x_mask, g=None, lengths=None):
        if g is not None:
            x = Fh.add_cbias(x, self.cond(g, force_direct=True))
        h = self.norm_2(self.lstms(x, lengths))
        # relu -> proj(x * mask) -> softplus -> * mask   (:131-136); padded columns are zero via the POST mask
        return self.proj(h, lrelu=0.0, act=L.ACT_SOFTPLUS, mask=x_mask, mask_mode=L.MASK_POST)
