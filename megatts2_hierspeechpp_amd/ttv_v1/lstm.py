"""torch.nn.LSTM (bidirectional, batch_first) on channel-major tensors, HIP path.

Parameter names and shapes are nn.LSTM's (``weight_ih_l0``, ``weight_hh_l0_reverse`` ...), so the
reference's checkpoints load unchanged.  Per layer: the input projection of both directions is one
1x1 GEMM on the conv kernel (``x W_ih^T + b_ih``), the recurrence is ``hsp_lstm_bidir_f32``.
Every utterance runs over its own length (what ``pack_padded_sequence`` / a B = 1 call does)."""
from __future__ import annotations

import torch
from torch import nn

from .. import _lib as L
from ..hip_layers import Conv1d, HipLayer


class _InputProjection(Conv1d):
    """Rows = [W_ih ; W_ih_reverse] of one layer; owns no parameters."""

    def __init__(self, lstm, layer, cin, hidden):
        super().__init__(cin, 8 * hidden, 1, bias=True, weight_2d=True)
        del self._parameters["weight"], self._parameters["bias"]
        self.__dict__["_src"] = (lstm, layer)

    def _folded(self):
        lstm, l = self._src
        return torch.cat([getattr(lstm, f"weight_ih_l{l}").data, getattr(lstm, f"weight_ih_l{l}_reverse").data], 0).contiguous()

    def _bias_src(self):
        lstm, l = self._src
        return torch.cat([getattr(lstm, f"bias_ih_l{l}").data, getattr(lstm, f"bias_ih_l{l}_reverse").data], 0)


class LSTM(HipLayer):
    def __init__(self, input_size, hidden_size, num_layers=1, batch_first=True, bidirectional=True):
        super().__init__()
        if not (batch_first and bidirectional):
            raise NotImplementedError("the front-end only builds bidirectional batch_first LSTMs")
        if hidden_size > 256 or hidden_size % 4:
            raise NotImplementedError("hsp_lstm_bidir_f32 holds one gate row per thread: hidden_size <= 256")
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        H = hidden_size
        for l in range(num_layers):
            cin = input_size if l == 0 else 2 * H
            for suf in ("", "_reverse"):
                self.register_parameter(f"weight_ih_l{l}{suf}", nn.Parameter(torch.zeros(4 * H, cin), requires_grad=False))
                self.register_parameter(f"weight_hh_l{l}{suf}", nn.Parameter(torch.zeros(4 * H, H), requires_grad=False))
                self.register_parameter(f"bias_ih_l{l}{suf}", nn.Parameter(torch.zeros(4 * H), requires_grad=False))
                self.register_parameter(f"bias_hh_l{l}{suf}", nn.Parameter(torch.zeros(4 * H), requires_grad=False))
        self.proj = nn.ModuleList([_InputProjection(self, l, input_size if l == 0 else 2 * H, H) for l in range(num_layers)])
        self._whh = self._bhh = None

    def hsp_requests(self):
        H = self.hidden_size
        return [("whh", self.num_layers * 2 * H * 4 * H), ("bhh", self.num_layers * 2 * 4 * H)]

    def hsp_fill(self, arena, materialize):
        H = self.hidden_size
        self._whh = arena.view(self, "whh").view(self.num_layers, 2, H, 4 * H)
        self._bhh = arena.view(self, "bhh").view(self.num_layers, 2, 4 * H)
        if materialize:
            for l in range(self.num_layers):
                for d, suf in enumerate(("", "_reverse")):
                    self._whh[l, d].copy_(getattr(self, f"weight_hh_l{l}{suf}").data.t())
                    self._bhh[l, d].copy_(getattr(self, f"bias_hh_l{l}{suf}").data)

    def forward(self, x, lengths):
        """x [B, In, N] (any batch / channel stride), lengths int64 [B] -> [B, 2H, N]."""
        if self._whh is None:
            raise L.HspError("LSTM used before finalize()")
        B, _, N = x.shape
        H = self.hidden_size
        lengths = lengths.to(torch.int64).contiguous()
        for l in range(self.num_layers):
            xp = self.proj[l](x)                                            # [B, 8H, N]
            y = torch.empty(B, 2 * H, N, dtype=torch.float32, device=x.device)
            L.check(L.lib().hsp_lstm_bidir_f32(L.fptr(xp), xp.stride(0), L.fptr(self._whh[l]), L.fptr(self._bhh[l]),
                                               L.ptr(lengths), L.fptr(y), y.stride(0), y.stride(1), B, H, N,
                                               L.stream_ptr()), "hsp_lstm_bidir_f32")
            x = y
        return x
