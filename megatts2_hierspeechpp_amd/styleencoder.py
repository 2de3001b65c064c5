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

    def forward(self, x, mask=None, per_utterance=False):
        """styleencoder.py:63-81: x [B, in_dim, T], mask [B, 1, T] -> [B, out_dim].

        The reference does not mask between its two Conv1dGLU layers and its average pool sums the
        padded frames too, so in a ragged batch the padding of a short row leaks into its style vector;
        that is reproduced by default (the vocoder path takes batches).  ``per_utterance=True`` masks there too, making every row equal to the B = 1 result on
        that utterance alone -- the semantics of the B = 1-only text front-end."""
        assert mask is not None
        x = self.spectral[0](x, act=L.ACT_MISH)
        x = self.spectral[3](x, act=L.ACT_MISH, mask=mask, mask_mode=L.MASK_PRE)
        x = self.temporal[0](x, mask=mask if per_utterance else None)
        x = self.temporal[1](x, mask=mask)
        x = self.slf_attn(x, x, mask_q=mask, mask_k=mask, res=x)
        # temporal_avg_pool (:83-91) sums over ALL frames, padding included, and divides by the length
        x = self.fc(x, **(dict(mask=mask, mask_mode=L.MASK_POST) if per_utterance else {}))
        return Fh.masked_mean(x, mask)
