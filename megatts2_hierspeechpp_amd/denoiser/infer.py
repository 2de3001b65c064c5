"""denoiser/infer.py: ``denoise(noisy_wav, model, hps)`` with the reference's signature.  torch.stft / torch.istft
(third party) are restated as the framing kernel of the prompt mel front-end, a DFT product on the MFMA GEMM against a
host-built (float64 -> fp32) basis, and an overlap-add kernel; magnitude compression, phase and the polar
re-composition are pointwise launches.  One host synchronisation: the norm factor ``sqrt(len / sum(x^2))`` is read
back once (prompt pre-processing, outside any timed path)."""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import nn

from .. import _lib as L
from .. import functional as Fh
from ..hip_layers import Conv1d, finalize as _finalize


class _Stft(nn.Module):
    """Forward and inverse DFT bases of a (n_fft, hop) pair as packed 1 x 1 conv weights, plus the Hann window."""

    def __init__(self, n_fft, hop_size, win_size):
        super().__init__()
        if win_size != n_fft or n_fft % 2:
            raise L.HspError("denoiser STFT: only win_size == n_fft (even) is built")
        self.n_fft, self.hop, self.n_freqs = n_fft, hop_size, n_fft // 2 + 1
        nf = self.n_freqs
        n = np.arange(n_fft, dtype=np.float64)
        f = np.arange(nf, dtype=np.float64)
        ang = 2.0 * np.pi * ((f[:, None] * n[None, :]) % n_fft) / n_fft
        self.dft = Conv1d(n_fft, 2 * nf, 1, bias=False)           # rows: cos | -sin   (torch.stft, onesided)
        self.idft = Conv1d(2 * nf, n_fft, 1, bias=False)          # irfft: DC / Nyquist once, the others twice; their
        wgt = np.full(nf, 2.0)                                    # imaginary parts are ignored as a C2R FFT does
        wgt[0] = wgt[-1] = 1.0
        inv = np.concatenate([np.cos(ang) * wgt[:, None], -np.sin(ang) * wgt[:, None]], 0).T / n_fft   # [n_fft, 2 nf]
        inv[:, nf] = 0.0
        inv[:, 2 * nf - 1] = 0.0
        with torch.no_grad():
            self.dft.weight.copy_(torch.from_numpy(np.concatenate([np.cos(ang), -np.sin(ang)], 0).astype(np.float32))
                                  .reshape(self.dft.weight.shape))
            self.idft.weight.copy_(torch.from_numpy(inv.astype(np.float32)).reshape(self.idft.weight.shape))
        self._window = None

    def finalize(self, device):
        _finalize(self, device)
        self._window = torch.hann_window(self.n_fft, periodic=True, dtype=torch.float32).to(device)
        return self


_STFT = {}


def _stft_for(device, n_fft, hop_size, win_size):
    key = (device.type, device.index, n_fft, hop_size, win_size)
    if key not in _STFT:
        _STFT[key] = _Stft(n_fft, hop_size, win_size).finalize(device)
    return _STFT[key]


def mag_pha_stft(y, n_fft, hop_size, win_size, compress_factor=1.0, center=True):
    """infer.py:12-24: y [1, L] -> (mag [1, F, T], pha [1, F, T], com [1, F, T, 2])."""
    if not center or y.dim() != 2 or y.shape[0] != 1:
        raise L.HspError("mag_pha_stft: one utterance [1, L], center=True")
    if not y.is_cuda or y.dtype != torch.float32:
        raise L.HspError("the denoiser runs on the GPU in float32 only; there is no CPU fallback")
    st = _stft_for(y.device, n_fft, hop_size, win_size)
    y = y.contiguous()
    Ls = y.shape[1]
    if Ls <= n_fft // 2:
        raise L.HspError(f"mag_pha_stft: reflect padding needs more than {n_fft // 2} samples, got {Ls}")
    T = 1 + Ls // hop_size
    f_ld = (T + 3) & ~3
    frames = torch.empty(1, n_fft, f_ld, dtype=torch.float32, device=y.device)
    L.check(L.lib().hsp_stft_frames_f32(L.fptr(y), L.fptr(st._window), L.fptr(frames), 1, Ls, n_fft, hop_size, T, f_ld,
                                        L.stream_ptr()), "hsp_stft_frames_f32")
    spec = st.dft(frames)                                          # [1, 2 F, f_ld]: real | imaginary rows
    nf = st.n_freqs
    mag = torch.empty(1, nf, T, dtype=torch.float32, device=y.device)
    pha = torch.empty(1, nf, T, dtype=torch.float32, device=y.device)
    L.check(L.lib().hsp_mag_pha_f32(L.fptr(spec), spec.stride(1), L.fptr(mag), L.fptr(pha), nf, T, float(compress_factor),
                                    L.stream_ptr()), "hsp_mag_pha_f32")
    return mag, pha, _com(mag, pha)


def _com(mag, pha):
    _, nf, T = mag.shape
    re = torch.empty(nf, T, dtype=torch.float32, device=mag.device)
    im = torch.empty(nf, T, dtype=torch.float32, device=mag.device)
    L.check(L.lib().hsp_polar_f32(L.fptr(mag), L.fptr(pha), 1.0, L.fptr(re), T, L.fptr(im), T, nf, T, L.stream_ptr()),
            "hsp_polar_f32")
    return torch.stack((re, im), dim=-1).unsqueeze(0)


def mag_pha_istft(mag, pha, n_fft, hop_size, win_size, compress_factor=1.0, center=True, scale=1.0):
    """infer.py:26-32: mag, pha [1, F, T] -> wav [1, hop (T - 1)] (times ``scale``)."""
    if not center or mag.dim() != 3 or mag.shape[0] != 1 or mag.shape != pha.shape:
        raise L.HspError("mag_pha_istft: one utterance [1, F, T], center=True")
    st = _stft_for(mag.device, n_fft, hop_size, win_size)
    mag, pha = mag.contiguous(), pha.contiguous()
    _, nf, T = mag.shape
    if nf != st.n_freqs or T < 2:
        raise L.HspError(f"mag_pha_istft: expected {st.n_freqs} bins and at least two frames")
    t_ld = (T + 3) & ~3
    spec = torch.zeros(1, 2 * nf, t_ld, dtype=torch.float32, device=mag.device)
    L.check(L.lib().hsp_polar_f32(L.fptr(mag), L.fptr(pha), 1.0 / float(compress_factor), L.fptr(spec), t_ld,
                                  L.fptr(spec[0, nf:]), t_ld, nf, T, L.stream_ptr()), "hsp_polar_f32")
    frames = st.idft(spec)                                         # [1, n_fft, t_ld]
    wav = torch.empty(1, hop_size * (T - 1), dtype=torch.float32, device=mag.device)
    L.check(L.lib().hsp_istft_ola_f32(L.fptr(frames), frames.stride(1), L.fptr(st._window), L.fptr(wav), n_fft, hop_size, T,
                                      float(scale), L.stream_ptr()), "hsp_istft_ola_f32")
    return wav


@torch.no_grad()
def denoise(noisy_wav, model, hps):
    """infer.py:3-10: noisy_wav [L] on the GPU -> denoised [1, hop (T - 1)]."""
    if noisy_wav.dim() != 1:
        raise L.HspError("denoise takes one 1-D waveform, as the reference")
    if not noisy_wav.is_cuda or noisy_wav.dtype != torch.float32:
        raise L.HspError("the denoiser runs on the GPU in float32 only; there is no CPU fallback")
    x = noisy_wav.contiguous()
    ss = torch.empty(1, dtype=torch.float32, device=x.device)
    L.check(L.lib().hsp_sum_sq_f32(L.fptr(x), x.numel(), L.fptr(ss), L.stream_ptr()), "hsp_sum_sq_f32")
    ssv = float(ss.item())
    if not ssv > 0.0:
        raise L.HspError("denoise(): the prompt is silent (sum of squares 0); the reference's norm factor is inf there "
                         "and its output NaN")
    norm = math.sqrt(x.numel() / ssv)
    y = Fh.axpby(x, x, norm, 0.0).unsqueeze(0)
    amp, pha, _ = mag_pha_stft(y, hps.n_fft, hps.hop_size, hps.win_size, hps.compress_factor)
    amp_g, pha_g, _ = model(amp, pha)
    return mag_pha_istft(amp_g, pha_g, hps.n_fft, hps.hop_size, hps.win_size, hps.compress_factor, scale=1.0 / norm)
