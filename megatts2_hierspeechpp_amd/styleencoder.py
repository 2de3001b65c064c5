"""StyleEncoder with the reference's module tree (reference: styleencoder.py)."""
from __future__ import annotations

from torch import nn

from . import _lib as L
from . import attentions
from . import functional as Fh
from .hip_layers import Conv1d


class Conv1dGLU(nn.Module):
    """styleencoder.py:13-32: conv -> split -> x1 * sigmoid(x2) -> + residual (one launch)."""

    def __init__(self, in_channels, out_channels, kernel_size, dropout):
        super().__init__()
        assert in_channels == out_channels and out_channels % 32 == 0
        self.out_channels = out_channels
        self.conv1 = Conv1d(in_channels, 2 * out_channels, kernel_size, padding=2, rows=L.ROWS_GATE_GLU)

    def forward(self, x, mask=None):
        return self.conv1(x, res=x, mask=mask, mask_mode=L.MASK_POST if mask is not None else L.MASK_NONE)


class StyleEncoder(nn.Module):
    def __init__(self, in_dim=513, hidden_dim=128, out_dim=256):
        super().__init__()
        self.in_dim, self.hidden_dim, self.out_dim = in_dim, hidden_dim, out_dim
        self.kernel_size, self.n_head, self.dropout = 5, 2, 0.1
        # nn.Sequential(Conv, Mish, Dropout, Conv, Mish, Dropout): parameters at 0 and 3
        self.spectral = nn.ModuleList([Conv1d(in_dim, hidden_dim, 1), nn.Identity(), nn.Identity(),
                                       Conv1d(hidden_dim, hidden_dim, 1), nn.Identity(), nn.Identity()])
        self.temporal = nn.ModuleList([Conv1dGLU(hidden_dim, hidden_dim, 5, 0.1), Conv1dGLU(hidden_dim, hidden_dim, 5, 0.1)])
        self.slf_attn = attentions.MultiHeadAttention(hidden_dim, hidden_dim, self.n_head, p_dropout=0.1,
                                                      proximal_bias=False, proximal_init=True)
        self.fc = Conv1d(hidden_dim, out_dim, 1)

    def forward(self, x, mask=None):
        """styleencoder.py:63-81: x [B, in_dim, T], mask [B, 1, T] -> [B, out_dim]."""
        assert mask is not None
        x = self.spectral[0](x, act=L.ACT_MISH)
        x = self.spectral[3](x, act=L.ACT_MISH, mask=mask, mask_mode=L.MASK_PRE)
        x = self.temporal[0](x)
        x = self.temporal[1](x, mask=mask)
        x = self.slf_attn(x, x, mask_q=mask, mask_k=mask, res=x)
        x = self.fc(x)
        return Fh.masked_mean(x, mask)
