"""HIP mirror of reference ttv_v1/modules.py members used at inference (WN and LayerNorm are the
same code as the top-level modules.py there; ResBlock1 is HiFi-GAN's)."""
from __future__ import annotations

from torch import nn

from ..hip_layers import Conv1d
from ..modules import LRELU_SLOPE, WN, LayerNorm  # noqa: F401  (re-exported like the reference module)


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)


class ResBlock1(nn.Module):
    """ttv_v1/modules.ResBlock1 (:187-223), mask-free form used by PitchPredictor: three
    [lrelu -> dilated conv -> lrelu -> conv -> + x] stages; both leaky-ReLUs run as conv prologues."""

    def __init__(self, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.convs1 = nn.ModuleList([Conv1d(channels, channels, kernel_size, dilation=d,
                                            padding=get_padding(kernel_size, d), weight_norm=True) for d in dilation])
        self.convs2 = nn.ModuleList([Conv1d(channels, channels, kernel_size, padding=get_padding(kernel_size, 1),
                                            weight_norm=True) for _ in dilation])

    def forward(self, x, x_mask=None, out=None, accumulate=False, post_scale=1.0):
        """``out`` / ``accumulate`` / ``post_scale`` fuse the caller's ``xs += block(x)`` and ``xs / n``.
        ``x_mask`` (x must arrive masked): zero the padding after every conv, which equals the
        reference's ``xt * x_mask`` before each conv plus the final ``x * x_mask`` (:213-222)."""
        from .. import _lib as L
        mk = dict(mask=x_mask, mask_mode=L.MASK_POST) if x_mask is not None else {}
        n = len(self.convs1)
        for i, (c1, c2) in enumerate(zip(self.convs1, self.convs2)):
            xt = c1(x, lrelu=LRELU_SLOPE, **mk)
            if i == n - 1 and out is not None:
                return c2(xt, lrelu=LRELU_SLOPE, res=x, out=out, accumulate=accumulate, post_scale=post_scale, **mk)
            x = c2(xt, lrelu=LRELU_SLOPE, res=x, **mk)
        return x
