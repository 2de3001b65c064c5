"""Anti-aliased activation with the reference's module tree (reference:
alias_free_torch/act.py, resample.py, filter.py) so that the ``upsample.filter`` and
``downsample.lowpass.filter`` buffers of a checkpoint load.  ``Activation1d`` is never
run as three ops here: it is either the prologue of the following conv
(hsp_conv1d_mfma_f32, HSP_PRO_ACT1D) or one fused stand-alone launch."""
import torch
from torch import nn

from . import _lib as L
from . import functional as Fh
from .hip_layers import HipLayer
from .synth import kaiser_sinc_filter12


class _Filter(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("filter", torch.from_numpy(kaiser_sinc_filter12()).view(1, 1, 12).clone())


class UpSample1d(_Filter):
    def __init__(self, ratio=2, kernel_size=12):
        assert ratio == 2 and kernel_size == 12
        super().__init__()


class LowPassFilter1d(_Filter):
    pass


class DownSample1d(nn.Module):
    def __init__(self, ratio=2, kernel_size=12):
        assert ratio == 2 and kernel_size == 12
        super().__init__()
        self.lowpass = LowPassFilter1d()


class Activation1d(HipLayer):
    def __init__(self, activation, up_ratio: int = 2, down_ratio: int = 2, up_kernel_size: int = 12,
                 down_kernel_size: int = 12):
        super().__init__()
        self.act = activation
        self.upsample = UpSample1d(up_ratio, up_kernel_size)
        self.downsample = DownSample1d(down_ratio, down_kernel_size)
        self.channels = activation.in_features
        self._ea = self._binv = self._filt = None

    def hsp_requests(self):
        return [("ea", self.channels), ("binv", self.channels), ("filt", 24)]

    def hsp_fill(self, arena, materialize):
        self._ea, self._binv, self._filt = arena.view(self, "ea"), arena.view(self, "binv"), arena.view(self, "filt")
        if materialize:
            L.check(L.lib().hsp_snake_consts_f32(L.fptr(self.act.alpha.data), L.fptr(self.act.beta.data),
                                                 L.fptr(self._ea), L.fptr(self._binv), self.channels, L.stream_ptr()),
                    "hsp_snake_consts_f32")
            self._filt[:12].copy_(self.upsample.filter.reshape(12))
            self._filt[12:].copy_(self.downsample.lowpass.filter.reshape(12))

    def forward(self, x):
        if self._ea is None:
            raise L.HspError("Activation1d used before finalize()")
        return Fh.act1d(x, self._ea, self._binv, self._filt)
