"""SpeechSR 16 kHz -> 24 kHz with the reference's call surface (reference: speechsr24k/speechsr.py).  The 24 kHz
model is the 48 kHz one with ONE difference: ``Generator.forward`` interpolates to ``int(L * 1.5)`` samples
(speechsr24k/speechsr.py:96) -- hard-coded there, the shipped config.json still says ``upsample_rates: [3]`` -- so
this module only pins that factor; everything else (AMP block, checkpoint keys of ``G_340000.pth``) is shared."""
from __future__ import annotations

from ..speechsr48k import speechsr as _sr48

AMPBlock0 = _sr48.AMPBlock0
INTERP_FACTOR = 1.5   # speechsr24k/speechsr.py:96


class Generator(_sr48.Generator):
    def __init__(self, initial_channel, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, gin_channels=0):
        super().__init__(initial_channel, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                         upsample_initial_channel, upsample_kernel_sizes, gin_channels=gin_channels)
        self.upsample_rates = [INTERP_FACTOR] * len(upsample_rates)   # the forward ignores the configured rate


class SynthesizerTrn(_sr48.SynthesizerTrn):
    """speechsr24k/speechsr.py:215-253 (generator only)."""

    def __init__(self, spec_channels, segment_size, resblock, resblock_kernel_sizes, resblock_dilation_sizes,
                 upsample_rates, upsample_initial_channel, upsample_kernel_sizes, **kwargs):
        super().__init__(spec_channels, segment_size, resblock, resblock_kernel_sizes, resblock_dilation_sizes,
                         upsample_rates, upsample_initial_channel, upsample_kernel_sizes, **kwargs)
        self.dec = Generator(1, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                             upsample_initial_channel, upsample_kernel_sizes)
