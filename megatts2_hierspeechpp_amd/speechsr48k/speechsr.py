"""SpeechSR (16 kHz -> 48 kHz, or 24 kHz with ``upsample_rates=[1.5]``-style factors) with the
reference's call surface (reference: speechsr48k/speechsr.py, speechsr24k/speechsr.py):
``SynthesizerTrn(spec_channels, segment_size, resblock, ...)``, ``forward(x)``, ``infer(x, max_len)``
and the ``dec.*`` checkpoint keys of the shipped ``G_100000.pth``.  SURVEY.md §8a row A15."""
from __future__ import annotations

import torch
from torch import nn

from .. import _lib as L
from .. import activations
from .. import functional as Fh
from ..alias_free_torch import Activation1d
from ..hierspeechpp_speechsynthesizer import AMPBlock1, _amp_stage
from ..hip_layers import Conv1d, entry as _entry, finalize as _finalize

AMPBlock0 = AMPBlock1  # identical forward (speechsr48k/speechsr.py:53-62 vs hierspeechpp_speechsynthesizer.py:377-386)


class Generator(nn.Module):
    """speechsr48k/speechsr.py:67-109.  conv_pre (1 -> C) -> linear x3 interpolation -> mean of
    three AMP blocks -> anti-aliased SnakeBeta -> conv_post -> tanh."""

    def __init__(self, initial_channel, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, gin_channels=0):
        super().__init__()
        self.num_kernels, self.num_upsamples = len(resblock_kernel_sizes), len(upsample_rates)
        self.upsample_rates = upsample_rates
        ch = upsample_initial_channel
        self.conv_pre = Conv1d(initial_channel, ch, 7, padding=3, weight_norm=True)
        self.resblocks = nn.ModuleList([AMPBlock0(ch, k, d, activation="snakebeta")
                                        for k, d in zip(resblock_kernel_sizes, resblock_dilation_sizes)])
        self.activation_post = Activation1d(activation=activations.SnakeBeta(ch, alpha_logscale=True))
        self.conv_post = Conv1d(ch, 1, 7, padding=3, bias=False)
        if gin_channels != 0:
            self.cond = Conv1d(gin_channels, ch, 1)

    def forward(self, x, g=None):
        x = self.conv_pre(x, cbias=self.cond(g) if g is not None else None)
        for i in range(self.num_upsamples):
            # the reference hard-codes `int(x.shape[-1] * 3)` (48k) / `* 1.5` (24k): speechsr.py:96
            x = Fh.linear_interp(x, int(x.shape[-1] * self.upsample_rates[i]))
            x = _amp_stage(self.resblocks, i * self.num_kernels, self.num_kernels, x)
        x = self.activation_post(x)
        return self.conv_post(x, act=L.ACT_TANH)


class SynthesizerTrn(nn.Module):
    """speechsr48k/speechsr.py:214-252 (generator only; the discriminators are training code)."""

    def __init__(self, spec_channels, segment_size, resblock, resblock_kernel_sizes, resblock_dilation_sizes,
                 upsample_rates, upsample_initial_channel, upsample_kernel_sizes, **kwargs):
        super().__init__()
        self.spec_channels, self.segment_size = spec_channels, segment_size
        self.dec = Generator(1, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                             upsample_initial_channel, upsample_kernel_sizes)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        if "model" in state_dict and not any(k.startswith("dec.") for k in state_dict):
            state_dict = state_dict["model"]
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def finalize(self, device, materialize: bool = True):
        return _finalize(self, device, materialize)

    @_entry
    @torch.no_grad()
    def forward(self, x):
        return self.dec(x)

    @_entry
    @torch.no_grad()
    def infer(self, x, max_len=None):
        return self.dec(x[:, :, :max_len])
