"""Mirror of the reference's ``Mels_preprocess.py``: ``MelSpectrogramFixed`` turns a
16 kHz prompt waveform into the log-mel the style / prompt encoders read (inference_plm.py:134,150).

The reference delegates to ``torchaudio.transforms.MelSpectrogram`` (third-party, not in this image); the
algorithm restated here is torchaudio 0.13.1's: centre-padded (reflect) STFT with a periodic Hann window, power
spectrogram, HTK triangular mel filters without normalisation, then the wrapper's ``log(x + 0.001)[..., :-1]``.

HIP path, three launches: ``hsp_stft_frames_f32`` (reflect index + window) -> the DFT as an fp32-MFMA GEMM through
``hsp_conv1d_mfma_f32`` (K = 1, Cin = n_fft, rows = cos | -sin bases built in float64) -> ``hsp_power_mel_log_f32``.
An FFT would save arithmetic (1282 x 1280 MACs per frame here), but a prompt is ~150-500 frames: the whole
transform is 0.2-0.6 GFLOP per utterance, microseconds on the matrix cores, and the GEMM keeps fp32 rounding
comparable to an fp32 FFT (measured against the oracle in tests/test_gpu_parity.py)."""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import nn

from . import _lib as L
from .hip_layers import Conv1d, entry as _entry, finalize as _finalize


def melscale_fbanks_htk(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> np.ndarray:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk") in float32 steps like torchaudio
    (linspace / pow in fp32): [n_freqs, n_mels]."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    fb = torch.clamp(torch.min(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]), min=0.0)
    return fb.numpy()


class MelSpectrogramFixed(nn.Module):
    """Mels_preprocess.py:8-18.  Same constructor keywords as the reference's call (inference_plm.py:204-213):
    sample_rate, n_fft, win_length, hop_length, f_min, f_max, n_mels, window_fn (only torch.hann_window)."""

    def __init__(self, sample_rate=16000, n_fft=400, win_length=None, hop_length=None, f_min=0.0, f_max=None,
                 n_mels=128, window_fn=torch.hann_window, **unsupported):
        super().__init__()
        if unsupported:
            raise L.HspError(f"MelSpectrogramFixed: unsupported torchaudio options {sorted(unsupported)}")
        win_length = n_fft if win_length is None else win_length
        hop_length = win_length // 2 if hop_length is None else hop_length
        if win_length != n_fft or window_fn is not torch.hann_window or n_fft % 2:
            raise L.HspError("MelSpectrogramFixed: only win_length == n_fft (even) with torch.hann_window is built")
        self.sample_rate, self.n_fft, self.hop_length, self.n_mels = sample_rate, n_fft, hop_length, n_mels
        self.f_min = float(f_min)
        self.f_max = float(f_max if f_max is not None else sample_rate // 2)
        self.n_freqs = n_fft // 2 + 1
        # DFT bases: row f = cos(2 pi f n / N), row n_freqs + f = -sin(2 pi f n / N); float64 -> fp32
        self.dft = Conv1d(n_fft, 2 * self.n_freqs, 1, bias=False)
        n = np.arange(n_fft, dtype=np.float64)
        f = np.arange(self.n_freqs, dtype=np.float64)
        ang = 2.0 * np.pi * ((f[:, None] * n[None, :]) % n_fft) / n_fft
        basis = np.concatenate([np.cos(ang), -np.sin(ang)], 0).astype(np.float32)
        with torch.no_grad():
            self.dft.weight.copy_(torch.from_numpy(basis).reshape(self.dft.weight.shape))
        self._window = self._fb = self._lo = self._hi = None

    def finalize(self, device):
        """Pack the DFT matrix and move the window / filter bank to ``device`` (call once)."""
        device = torch.device(device)
        _finalize(self, device)
        fb = melscale_fbanks_htk(self.n_freqs, self.f_min, self.f_max, self.n_mels, self.sample_rate)
        nz = fb > 0
        lo = np.where(nz.any(0), nz.argmax(0), 0).astype(np.int32)
        hi = np.where(nz.any(0), self.n_freqs - nz[::-1].argmax(0), 0).astype(np.int32)
        self._window = torch.hann_window(self.n_fft, periodic=True, dtype=torch.float32).to(device)
        self._fb = torch.from_numpy(np.ascontiguousarray(fb)).to(device)
        self._lo, self._hi = torch.from_numpy(lo).to(device), torch.from_numpy(hi).to(device)
        return self

    @_entry
    @torch.no_grad()
    def forward(self, x):
        """x [..., L] fp32 on the GPU (L > n_fft / 2) -> log-mel [..., n_mels, L // hop_length]."""
        if self._window is None:
            raise L.HspError("MelSpectrogramFixed used before finalize(device)")
        if not x.is_cuda or x.dtype != torch.float32:
            raise L.HspError("MelSpectrogramFixed runs on the GPU in float32 only; there is no CPU fallback")
        lead = x.shape[:-1]
        xs = x.reshape(-1, x.shape[-1]).contiguous()
        B, Ls = xs.shape
        if Ls <= self.n_fft // 2:
            raise L.HspError(f"MelSpectrogramFixed: reflect padding needs more than {self.n_fft // 2} samples, got {Ls}")
        T = 1 + Ls // self.hop_length
        f_ld = (T + 3) & ~3
        frames = torch.empty(B, self.n_fft, f_ld, dtype=torch.float32, device=xs.device)  # pitch columns zeroed by the kernel
        L.check(L.lib().hsp_stft_frames_f32(L.fptr(xs), L.fptr(self._window), L.fptr(frames), B, Ls, self.n_fft,
                                            self.hop_length, T, f_ld, L.stream_ptr()), "hsp_stft_frames_f32")
        spec = self.dft(frames)                                   # [B, 2 n_freqs, f_ld]: real | imaginary rows
        T_out = T - 1                                             # the wrapper drops the last frame
        out = torch.empty(B, self.n_mels, max(T_out, 0), dtype=torch.float32, device=xs.device)
        if T_out > 0:
            L.check(L.lib().hsp_power_mel_log_f32(L.fptr(spec), spec.stride(0), spec.stride(1), L.fptr(self._fb),
                                                  L.ptr(self._lo), L.ptr(self._hi), L.fptr(out), B, self.n_freqs,
                                                  self.n_mels, T_out, 0.001, L.stream_ptr()), "hsp_power_mel_log_f32")
        return out.reshape(*lead, self.n_mels, max(T_out, 0))
