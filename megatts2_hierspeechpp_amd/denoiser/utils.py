"""denoiser/utils.py: padding helpers and the learnable sigmoid (parameters only; the arithmetic is in libhsp)."""
from __future__ import annotations

import torch
from torch import nn

from ..hip_layers import HipLayer


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)


def get_padding_2d(kernel_size, dilation=(1, 1)):
    return (int((kernel_size[0] * dilation[0] - dilation[0]) / 2), int((kernel_size[1] * dilation[1] - dilation[1]) / 2))


class Vec(HipLayer):
    """Device copies of a module's small 1-D parameters (arena views named after the parameters)."""

    NAMES = ()

    def hsp_requests(self):
        return [(n, getattr(self, n).numel()) for n in self.NAMES]

    def hsp_fill(self, arena, materialize):
        for n in self.NAMES:
            v = arena.view(self, n)
            self.__dict__["_" + n] = v
            if materialize:
                v.copy_(getattr(self, n).data.reshape(-1).float())

    def dev(self, n):
        v = self.__dict__.get("_" + n)
        if v is None:
            from .. import _lib as L
            raise L.HspError(f"{type(self).__name__} used before finalize()")
        return v


class LearnableSigmoid_2d(Vec):
    """denoiser/utils.py:44-53: beta * sigmoid(slope * x), slope [in_features, 1]."""

    NAMES = ("slope",)

    def __init__(self, in_features, beta=1):
        super().__init__()
        self.beta = beta
        self.slope = nn.Parameter(torch.ones(in_features, 1), requires_grad=False)
