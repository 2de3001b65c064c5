"""denoiser/generator.py (MP-SENet ``MPNet``) with the reference's class names, constructor arguments and state-dict
keys; every tensor stays ``[1, C, T, F]`` (the reference's layout) and all arithmetic is in libhsp.so.

How the 2-D pieces map onto the 1-D kernels (one utterance, as denoiser/infer.py feeds it):
  * ``Conv2d(k = (kh, kw), dilation = (d, 1))``: the T rows are the batch of a Conv1d along F; tap row i reads the
    input rows shifted by (i - kh // 2) d -- kh launches on row-shifted views, accumulating into the output (rows the
    shift pushes outside [0, T) are the conv's zero padding: those launches simply cover fewer rows);
  * the dense concatenation ``cat([x, skip])`` is one ``[1, 5 * 64, T, F]`` buffer filled from the back, so a layer's
    input is a channel slice of it and nothing is copied;
  * ``Conv2d((1, 3), stride (1, 2))`` / ``ConvTranspose2d((1, 3), stride (1, 2))``: the polyphase strided conv /
    transposed conv of the vocoder along F;
  * ``InstanceNorm2d + PReLU``: one launch per layer (``hsp_instnorm_prelu_f32``), in place."""
from __future__ import annotations

import torch
from torch import nn

from .. import _lib as L
from .. import functional as Fh
from ..hip_layers import Conv1d, ConvTranspose1d, HipLayer, PolyphaseConv1d, _SubArena, entry as _entry, finalize as _finalize
from .conformer import ConformerBlock
from .utils import LearnableSigmoid_2d, Vec, get_padding_2d


class _Conv2d(HipLayer):
    """nn.Conv2d(cin, cout, (kh, kw), stride (1, sw), dilation (dh, 1), 'same' padding along T, ``pw`` along F);
    parameters keep nn.Conv2d's names and shapes."""

    def __init__(self, cin, cout, kernel_size, stride=(1, 1), dilation=(1, 1), padding=(0, 0)):
        super().__init__()
        kh, kw = kernel_size
        assert stride[0] == 1 and dilation[1] == 1 and padding[0] * 2 == dilation[0] * (kh - 1)
        self.cin, self.cout, self.kh, self.kw, self.sw, self.dh, self.pw = cin, cout, kh, kw, stride[1], dilation[0], padding[1]
        self.weight = nn.Parameter(torch.zeros(cout, cin, kh, kw), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)
        mid = kh // 2
        if self.sw == 1:
            rows = [Conv1d(cin, cout, kw, padding=self.pw, bias=(i == mid)) for i in range(kh)]
        else:
            assert kh == 1 and self.pw == 0
            rows = [PolyphaseConv1d(cin, cout, kw, self.sw)]
        self.__dict__["_rows"] = rows     # not registered: no duplicate state_dict keys

    def hsp_requests(self):
        return [(f"r{i}.{n}", numel) for i, r in enumerate(self._rows) for n, numel in r.hsp_requests()]

    def hsp_fill(self, arena, materialize):
        for i, r in enumerate(self._rows):
            if materialize:
                r.weight.data = self.weight.data[:, :, i, :].contiguous()
                if r.bias is not None:
                    r.bias.data = self.bias.data
            r.hsp_fill(_SubArena(arena, self, f"r{i}."), materialize)

    def forward(self, x, out=None):
        """x [1, cin, T, F] (any channel stride) -> out [1, cout, T, F'] (written in place when given)."""
        _, _, T, F_ = x.shape
        rows_of = lambda z, t0, t1: z[0, :, t0:t1].permute(1, 0, 2)      # [T', C, F]: rows as the batch
        if self.sw != 1:
            assert out is None
            y = self._rows[0](rows_of(x, 0, T))                           # [T, cout, F']
            return Fh.copy_strided(y.permute(1, 0, 2)).unsqueeze(0)       # channel planes contiguous again
        Fo = F_ + 2 * self.pw - (self.kw - 1)
        if out is None:
            out = torch.empty(1, self.cout, T, Fo, dtype=torch.float32, device=x.device)
        mid = self.kh // 2
        for n, i in enumerate([mid] + [j for j in range(self.kh) if j != mid]):
            sh = (i - mid) * self.dh
            t0, t1 = max(0, -sh), min(T, T - sh)
            if t1 <= t0:
                continue
            self._rows[i](rows_of(x, t0 + sh, t1 + sh), out=rows_of(out, t0, t1), accumulate=n > 0)
        return out


class _InstanceNorm2d(Vec):
    NAMES = ("weight", "bias")

    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(channels), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(channels), requires_grad=False)


class _PReLU(Vec):
    NAMES = ("weight",)

    def __init__(self, channels):
        super().__init__()
        self.weight = nn.Parameter(torch.full((channels,), 0.25), requires_grad=False)


def _norm_act(norm: _InstanceNorm2d, act: _PReLU, x):
    """In place on x [1, C, T, F] (each channel plane contiguous)."""
    _, Cc, T, F_ = x.shape
    assert x.stride(3) == 1 and x.stride(2) == F_
    L.check(L.lib().hsp_instnorm_prelu_f32(L.fptr(x), x.stride(1), Cc, T * F_, L.fptr(norm.dev("weight")),
                                           L.fptr(norm.dev("bias")), L.fptr(act.dev("weight")), float(norm.eps),
                                           L.stream_ptr()), "hsp_instnorm_prelu_f32")
    return x


def _seq(*mods):
    """nn.Sequential's key layout ('0', '1', ...) without its forward."""
    return nn.ModuleDict({str(i): m for i, m in enumerate(mods) if m is not None})


class DenseBlock(nn.Module):
    """generator.py:11-32."""

    def __init__(self, h, kernel_size=(3, 3), depth=4):
        super().__init__()
        self.h, self.depth = h, depth
        self.dense_block = nn.ModuleList([])
        for i in range(depth):
            dil = 2 ** i
            self.dense_block.append(_seq(
                _Conv2d(h.dense_channel * (i + 1), h.dense_channel, kernel_size, dilation=(dil, 1),
                        padding=get_padding_2d(kernel_size, (dil, 1))),
                _InstanceNorm2d(h.dense_channel), _PReLU(h.dense_channel)))

    def forward(self, x):
        """x [1, C, T, F] -> [1, C, T, F].  ``skip = cat([x_i, skip])`` lives in one buffer: slot depth holds the
        input, slot depth - 1 - i the output of layer i, and layer i reads slots depth - i .. depth."""
        _, Cc, T, F_ = x.shape
        assert x.stride(3) == 1 and x.stride(2) == F_
        buf = torch.empty(1, Cc * (self.depth + 1), T, F_, dtype=torch.float32, device=x.device)
        L.check(L.lib().hsp_copy_strided_f32(L.fptr(x), x.stride(0), x.stride(1), 1, L.fptr(buf[:, self.depth * Cc:]), 1, Cc,
                                             T * F_, L.stream_ptr()), "hsp_copy_strided_f32")
        for i in range(self.depth):
            blk = self.dense_block[i]
            o = buf[:, (self.depth - 1 - i) * Cc:(self.depth - i) * Cc]
            blk["0"](buf[:, (self.depth - i) * Cc:], out=o)
            _norm_act(blk["1"], blk["2"], o)
        return buf[:, :Cc]


class DenseEncoder(nn.Module):
    """generator.py:35-56."""

    def __init__(self, h, in_channel):
        super().__init__()
        self.h = h
        self.dense_conv_1 = _seq(_Conv2d(in_channel, h.dense_channel, (1, 1)), _InstanceNorm2d(h.dense_channel),
                                 _PReLU(h.dense_channel))
        self.dense_block = DenseBlock(h, depth=4)
        self.dense_conv_2 = _seq(_Conv2d(h.dense_channel, h.dense_channel, (1, 3), (1, 2)),
                                 _InstanceNorm2d(h.dense_channel), _PReLU(h.dense_channel))

    def forward(self, x):
        c1, c2 = self.dense_conv_1, self.dense_conv_2
        x = _norm_act(c1["1"], c1["2"], c1["0"](x))
        x = self.dense_block(x)
        return _norm_act(c2["1"], c2["2"], c2["0"](x))


class _ConvTranspose2d(HipLayer):
    """nn.ConvTranspose2d(cin, cout, (1, kw), (1, sw)): weight [cin, cout, 1, kw]; a ConvTranspose1d along F."""

    def __init__(self, cin, cout, kernel_size, stride):
        super().__init__()
        assert kernel_size[0] == 1 and stride[0] == 1
        self.cin, self.cout = cin, cout
        self.weight = nn.Parameter(torch.zeros(cin, cout, 1, kernel_size[1]), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)
        self.__dict__["_tr"] = ConvTranspose1d(cin, cout, kernel_size[1], stride[1])

    def hsp_requests(self):
        return [("t." + n, numel) for n, numel in self._tr.hsp_requests()]

    def hsp_fill(self, arena, materialize):
        if materialize:
            self._tr.weight.data = self.weight.data[:, :, 0, :].contiguous()
            self._tr.bias.data = self.bias.data
        self._tr.hsp_fill(_SubArena(arena, self, "t."), materialize)

    def forward(self, x):
        """x [1, cin, T, F] -> [1, cout, T, (F - 1) sw + kw]."""
        _, _, T, F_ = x.shape
        Fo = (F_ - 1) * self._tr.up + self._tr.k
        out = torch.empty(1, self.cout, T, Fo, dtype=torch.float32, device=x.device)
        self._tr(x[0].permute(1, 0, 2), out=out[0].permute(1, 0, 2))
        return out


class MaskDecoder(nn.Module):
    """generator.py:59-79."""

    def __init__(self, h, out_channel=1):
        super().__init__()
        self.dense_block = DenseBlock(h, depth=4)
        self.mask_conv = _seq(_ConvTranspose2d(h.dense_channel, h.dense_channel, (1, 3), (1, 2)),
                              _Conv2d(h.dense_channel, out_channel, (1, 1)), _InstanceNorm2d(out_channel),
                              _PReLU(out_channel), _Conv2d(out_channel, out_channel, (1, 1)))
        self.lsigmoid = LearnableSigmoid_2d(h.n_fft // 2 + 1, beta=h.beta)

    def forward(self, x, mag_tf):
        """x [1, C, T, F'], mag_tf [T, F] -> mag_tf * mask [T, F] (the product of generator.py:140 included)."""
        mc = self.mask_conv
        m = mc["1"](mc["0"](self.dense_block(x)))
        m = mc["4"](_norm_act(mc["2"], mc["3"], m))                  # [1, 1, T, F]
        T, F_ = mag_tf.shape
        out = torch.empty_like(mag_tf)
        L.check(L.lib().hsp_lsigmoid_mul_f32(L.fptr(m), L.fptr(self.lsigmoid.dev("slope")), float(self.lsigmoid.beta),
                                             L.fptr(mag_tf), L.fptr(out), T, F_, L.stream_ptr()), "hsp_lsigmoid_mul_f32")
        return out


class PhaseDecoder(nn.Module):
    """generator.py:82-97."""

    def __init__(self, h, out_channel=1):
        super().__init__()
        self.dense_block = DenseBlock(h, depth=4)
        self.phase_conv = _seq(_ConvTranspose2d(h.dense_channel, h.dense_channel, (1, 3), (1, 2)),
                               _InstanceNorm2d(h.dense_channel), _PReLU(h.dense_channel))
        self.phase_conv_r = _Conv2d(h.dense_channel, out_channel, (1, 1))
        self.phase_conv_i = _Conv2d(h.dense_channel, out_channel, (1, 1))

    def forward(self, x):
        """x [1, C, T, F'] -> atan2(x_i, x_r) [T, F]."""
        pc = self.phase_conv
        p = _norm_act(pc["1"], pc["2"], pc["0"](self.dense_block(x)))
        x_r, x_i = self.phase_conv_r(p), self.phase_conv_i(p)
        out = torch.empty(x_r.shape[2], x_r.shape[3], dtype=torch.float32, device=x.device)
        L.check(L.lib().hsp_atan2_f32(L.fptr(x_i), L.fptr(x_r), L.fptr(out), out.numel(), L.stream_ptr()), "hsp_atan2_f32")
        return out


class TSConformerBlock(nn.Module):
    """generator.py:100-115: a conformer over (f, t) rows then one over (t, f) rows; see conformer.py for which axis
    each of their modules works along."""

    def __init__(self, h):
        super().__init__()
        self.h = h
        self.time_conformer = ConformerBlock(dim=h.dense_channel, n_head=4, ccm_kernel_size=31, ffm_dropout=0.2,
                                             attn_dropout=0.2)
        self.freq_conformer = ConformerBlock(dim=h.dense_channel, n_head=4, ccm_kernel_size=31, ffm_dropout=0.2,
                                             attn_dropout=0.2)

    def forward(self, x):
        """x [1, C, T, F] -> [1, C, T, F]."""
        xt = Fh.copy_strided(x[0].permute(2, 0, 1))                   # [F, C, T]   (reference: view(b f, t, c))
        xt = Fh.axpby(self.time_conformer(xt), xt, 1.0, 1.0)
        xf = Fh.copy_strided(xt.permute(2, 1, 0))                     # [T, C, F]   (reference: view(b t, f, c))
        xf = Fh.axpby(self.freq_conformer(xf), xf, 1.0, 1.0)
        return Fh.copy_strided(xf.permute(1, 0, 2)).unsqueeze(0)      # [1, C, T, F]


class MPNet(nn.Module):
    """generator.py:118-147.  ``forward(noisy_mag [1, F, T], noisy_pha [1, F, T])`` ->
    ``(denoised_mag [1, F, T], denoised_pha [1, F, T], denoised_com [1, F, T, 2])`` as the reference; one utterance
    per call (B = 1 is what denoiser/infer.py passes)."""

    def __init__(self, h, num_tscblocks=4):
        super().__init__()
        self.h = h
        self.num_tscblocks = num_tscblocks
        self.dense_encoder = DenseEncoder(h, in_channel=2)
        self.TSConformer = nn.ModuleList([TSConformerBlock(h) for _ in range(num_tscblocks)])
        self.mask_decoder = MaskDecoder(h, out_channel=1)
        self.phase_decoder = PhaseDecoder(h, out_channel=1)

    def finalize(self, device, materialize: bool = True):
        self.arena = _finalize(self, device, materialize)
        return self

    @_entry
    @torch.no_grad()
    def forward(self, noisy_mag, noisy_pha):
        if noisy_mag.dim() != 3 or noisy_mag.shape[0] != 1 or noisy_mag.shape != noisy_pha.shape:
            raise L.HspError("MPNet takes one utterance: noisy_mag / noisy_pha [1, F, T]")
        _, F_, T = noisy_mag.shape
        x = torch.empty(1, 2, T, F_, dtype=torch.float32, device=noisy_mag.device)      # cat((mag, pha), 1) as [1, 2, T, F]
        for i, src in enumerate((noisy_mag, noisy_pha)):
            v = src.permute(0, 2, 1)                                                    # [1, T, F] view of [1, F, T]
            L.check(L.lib().hsp_copy_strided_f32(L.fptr(v), v.stride(0), v.stride(1), v.stride(2), L.fptr(x[:, i]), 1, T, F_,
                                                 L.stream_ptr()), "hsp_copy_strided_f32")
        h = self.dense_encoder(x)
        for blk in self.TSConformer:
            h = blk(h)
        mag_tf = self.mask_decoder(h, x[0, 0])                                           # [T, F]
        pha_tf = self.phase_decoder(h)                                                   # [T, F]
        d_mag = Fh.copy_strided(mag_tf.t().unsqueeze(0))                                 # [1, F, T]
        d_pha = Fh.copy_strided(pha_tf.t().unsqueeze(0))
        re = torch.empty(F_, T, dtype=torch.float32, device=x.device)
        im = torch.empty(F_, T, dtype=torch.float32, device=x.device)
        L.check(L.lib().hsp_polar_f32(L.fptr(d_mag), L.fptr(d_pha), 1.0, L.fptr(re), T, L.fptr(im), T, F_, T, L.stream_ptr()),
                "hsp_polar_f32")
        return d_mag, d_pha, torch.stack((re, im), dim=-1).unsqueeze(0)       # a copy, no arithmetic
