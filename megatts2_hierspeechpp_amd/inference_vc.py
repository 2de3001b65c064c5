"""The tensor core of the reference's voice-conversion harness (reference: inference_vc.py:70-160): source waveform
-> reflect pad 40 -> wav2vec2 hidden state 7 -> (with the source's and the prompt's F0 tracks) F0 conversion ->
prompt mels -> ``voice_conversion_noise_control`` -> peak-normalised int16.

Outside this module, as in inference_plm: audio file decoding / resampling (torchaudio plumbing) and the YAAPT pitch
tracker (third-party ``amfm_decompy``, CPU numpy code, absent from this image): the F0 tracks are inputs here, at the
tracker's rate of 200 Hz (4 per w2v frame), zeros where unvoiced."""
from __future__ import annotations

import torch
from torch import nn

from . import functional as Fh
from .inference_plm import peak_int16


class VcModels(nn.Module):
    """The models the harness loads (inference_vc.py:model_load): the vocoder and the wav2vec2 producer."""

    def __init__(self, voc_cfg, spec_channels=641, segment_frames=61440 // 320, w2v_layer=7):
        super().__init__()
        from .extract_w2v import Wav2vec2
        from .hierspeechpp_speechsynthesizer import SynthesizerTrn
        self.voc = SynthesizerTrn(spec_channels, segment_frames, **voc_cfg)
        self.w2v = Wav2vec2(layer=w2v_layer)

    def finalize(self, device, materialize: bool = True):
        from .hip_layers import finalize
        self.arena = finalize(self, device, materialize)
        return self


def pad_source(audio, hop: int = 1280):
    """inference_vc.py:74-75: zero-pad the source to the next multiple of 1280 samples (always at least one sample)."""
    n = audio.shape[-1]
    p = (n // hop + 1) * hop - n
    out = torch.zeros(*audio.shape[:-1], n + p, dtype=torch.float32, device=audio.device)
    out[..., :n] = audio
    return out


@torch.no_grad()
def vc(models: VcModels, mel_fn, source_audio, f0_src, target_audio, f0_trg, noise_scale_vc=0.333, denoise_ratio=0.0,
       denoised_audio=None, noise=None, return_float=False):
    """source_audio [1, Ls] (16 kHz, already padded by pad_source), f0_src [1, Ls / 80] (YAAPT, 0 = unvoiced),
    target_audio [1, Lt], f0_trg [1, Lt / 80] -> int16 waveform [320 T] (and the float audio with return_float).
    ``denoised_audio``: the denoiser's output for the prompt (inference_vc.py:118-121); None = the prompt itself, which is
    what the reference does at denoise_ratio == 0."""
    x_w2v = models.w2v(Fh.reflect_pad(source_audio, 40))                       # :85-86
    T = x_w2v.shape[2]
    x_length = torch.tensor([T], dtype=torch.int64, device=x_w2v.device)
    lf0 = Fh.f0_convert(f0_src, f0_trg)                                        # :80-81,104-105
    second = target_audio if denoised_audio is None else denoised_audio
    both = torch.empty(2, target_audio.shape[-1], dtype=torch.float32, device=target_audio.device)
    both[0].copy_(target_audio.reshape(-1))
    both[1].copy_(second.reshape(-1)[:target_audio.shape[-1]])
    trg_mel = mel_fn(both)                                                     # [2, 80, Tm]  (:113-126)
    trg_len = torch.tensor([trg_mel.shape[2]] * 2, dtype=torch.int64, device=x_w2v.device)
    audio = models.voc.voice_conversion_noise_control(x_w2v, x_length, trg_mel, trg_len, lf0.reshape(1, -1)[:, :4 * T],
                                                      noise_scale=noise_scale_vc, denoise_ratio=denoise_ratio, noise=noise)
    wav = peak_int16(audio.reshape(1, -1), torch.tensor([audio.shape[-1]], device=audio.device))
    return (wav.reshape(-1), audio) if return_float else wav.reshape(-1)
