"""HIP mirror of reference ttv_v1/Gaussian.py: RangePredictor and GaussianUpsampling."""
from __future__ import annotations

import torch
from torch import nn

from .. import _lib as L
from ..hip_layers import LinearCT
from .lstm import LSTM


class LinearNorm(nn.Module):
    """Gaussian.LinearNorm (:7-22): a named wrapper around nn.Linear (key ``linear_layer``)."""

    def __init__(self, in_dim, out_dim, bias=True, w_init_gain="linear"):
        super().__init__()
        self.linear_layer = LinearCT(in_dim, out_dim, bias=bias)

    def forward(self, x, **kw):
        return self.linear_layer(x, **kw)


class RangePredictor(nn.Module):
    """Gaussian.RangePredictor (:76-117): BiLSTM over cat(encoder outputs, durations) -> Linear -> softplus."""

    def __init__(self, in_channel, out_channel):
        super().__init__()
        self.in_channel, self.out_channel = in_channel, out_channel
        self.lstm = LSTM(in_channel, out_channel, 1, batch_first=True, bidirectional=True)
        self.proj = LinearNorm(out_channel * 2, 1)

    def forward(self, xd, input_lengths):
        """xd [B, 257, N] = cat(x, durations) on the channel axis -> ranges [B, 1, N] (softplus output;
        the clamp against the durations is applied inside the Gaussian upsampling kernel)."""
        return self.proj(self.lstm(xd, input_lengths), act=L.ACT_SOFTPLUS)


class GaussianUpsampling(nn.Module):
    """Gaussian.GaussianUpsampling (:24-69); T (the longest utterance's frame count) is passed by the
    caller, who already fetched the frame counts from the device (the reference's ``.item()``)."""

    def forward(self, x, dur, rng, input_lengths, frames, T):
        B, C, N = x.shape
        assert x.stride(2) == 1 and dur.stride(-1) == 1 and rng.is_contiguous()
        out = torch.empty(B, C, T, dtype=torch.float32, device=x.device)
        L.check(L.lib().hsp_gaussian_upsample_f32(L.fptr(x), x.stride(0), x.stride(1), L.fptr(dur), dur.stride(0),
                                                  L.fptr(rng), rng.stride(0), L.ptr(input_lengths), L.fptr(frames),
                                                  L.fptr(out), B, C, N, T, L.stream_ptr()),
                "hsp_gaussian_upsample_f32")
        return out
