"""HIP mirror of the tensor core of the reference's inference_plm.py (its ``tts()`` between mel
extraction and wav writing, :156-190, and the model bundle of ``model_load`` :203-263).

Out of scope here (host plumbing, not the hot path): text cleaning / phonemisation
(``get_text``), audio file IO and resampling -- callers hand over phone ids and the prompt waveform (or its mels),
exactly the tensors ``tts()`` feeds its models.  The prompt mel transform is ``Mels_preprocess.MelSpectrogramFixed``,
the optional prompt denoiser ``denoiser.generator.MPNet`` + ``denoiser.infer.denoise`` (:142-147)."""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from . import _lib as L
from .hierspeechpp_speechsynthesizer import SynthesizerTrn
from .hip_layers import finalize as _finalize
from .ttv_v1.t2w2v_transformer import Megatts2PLM1
from .ttv_v1.t2w2v_transformer import SynthesizerTrn as Text2W2V

# text/symbols_lmdh.py: len(symbols), len(tone_symbols), len(language_symbols)
N_VOCAB, N_TONE, N_LANGUAGE = 126, 11, 4


class TtsModels(nn.Module):
    """The ``hierspeech`` tuple of inference_plm.py:tts as one module: ``voc`` = net_g (the hierarchical
    synthesizer), ``ttv`` = text2w2v, ``plm`` = the prosody LM, optional ``sr`` = SpeechSR.  One weight
    arena for all of them -> one RCCL broadcast in a multi-GPU job."""

    def __init__(self, voc_cfg: dict, ttv_cfg: dict, speechsr: Optional[nn.Module] = None):
        super().__init__()
        self.voc = SynthesizerTrn(641, 61440 // 320, **voc_cfg)
        self.ttv = Text2W2V(N_VOCAB, N_TONE, N_LANGUAGE, 641, 320, 16000, 60, **ttv_cfg)
        self.plm = Megatts2PLM1()
        if speechsr is not None:
            self.sr = speechsr

    def finalize(self, device, materialize: bool = True):
        self.arena = _finalize(self, device, materialize)
        return self


def zero_below(x, thr: float):
    x = x.contiguous()
    y = torch.empty_like(x)
    L.check(L.lib().hsp_zero_below_f32(L.fptr(x), float(thr), L.fptr(y), x.numel(), L.stream_ptr()), "hsp_zero_below_f32")
    return y


def peak_int16(audio, lengths=None, gain: float = 0.999):
    """[B, 1, n] / [B, n] fp32 -> int16 [B, n], every row scaled by its own peak (over ``lengths[b]`` samples)."""
    a = audio.reshape(audio.shape[0], -1)
    assert a.stride(1) == 1
    out = torch.empty(a.shape, dtype=torch.int16, device=a.device)
    L.check(L.lib().hsp_peak_int16(L.fptr(a), a.stride(0), L.ptr(lengths.to(torch.int64).contiguous()) if lengths is not None else None,
                                   float(gain), L.ptr(out), out.stride(0), a.shape[0], a.shape[1], L.stream_ptr()),
            "hsp_peak_int16")
    return out


def prompt_mels(mel_fn, audio, denoiser=None, hps_denoiser=None):
    """The two prompt mels of inference_plm.py:130-150: ``src_mel_ttv`` from the prompt zero-padded to the next
    multiple of 1600 samples (always at least one sample of padding, :131-134), and ``src_mel`` [2, 80, T] from the
    un-padded prompt stacked with itself (denoise_ratio = 0, :142-143) or with its denoised version cut to the same
    length (``denoiser`` = a finalized denoiser.generator.MPNet, ``hps_denoiser`` its config: :144-150; the denoiser
    sees the PADDED prompt, as in the reference).  ``audio`` [1, n] fp32 on the GPU; ``mel_fn`` a finalized
    Mels_preprocess.MelSpectrogramFixed."""
    n = audio.shape[-1]
    padded = torch.zeros(audio.shape[0], (n // 1600 + 1) * 1600, dtype=audio.dtype, device=audio.device)
    padded[:, :n].copy_(audio)
    src_mel_ttv = mel_fn(padded)
    if denoiser is None:
        src_mel = mel_fn(audio)
        src_mel = src_mel.repeat(2, 1, 1) if src_mel.shape[0] == 1 else torch.cat([src_mel, src_mel], 0)
        return src_mel_ttv, src_mel
    if audio.shape[0] != 1:
        raise L.HspError("the prompt denoiser takes one prompt per call, as the reference")
    from .denoiser.infer import denoise
    den = denoise(padded[0], denoiser, hps_denoiser)                  # [1, len(padded)] (1600 is a multiple of the hop)
    both = torch.cat([padded, den[:, :padded.shape[-1]]], 0)[:, :n]   # :147,150 (copies, no arithmetic)
    return src_mel_ttv, mel_fn(both.contiguous())


def write_wav(path, sample_rate: int, pcm):
    """scipy.io.wavfile.write(path, rate, int16 array) of inference_plm.py:195-200: 16-bit mono PCM RIFF."""
    import wave
    import numpy as np
    a = pcm.detach().cpu().numpy() if isinstance(pcm, torch.Tensor) else np.asarray(pcm)
    if a.dtype != np.int16 or a.ndim != 1:
        raise ValueError("write_wav takes a 1-D int16 array (one utterance)")
    with wave.open(str(path), "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(int(sample_rate))
        f.writeframes(a.astype("<i2").tobytes())


@torch.no_grad()
def tts(models: TtsModels, text, text_length, tone, language, src_mel_ttv, src_mel_ttv_length, src_mel, src_length2,
        noise_scale_vc: float = 0.333, denoise_ratio: float = 0.0, output_sr: int = 16000, dur=None, noise=None,
        return_float: bool = False):
    """inference_plm.py:tts :156-190 on tensors.

    text / tone / language int64 [B, N], text_length [B]; src_mel_ttv [B, 80, Tm'] (prompt mel for the
    front-end) with lengths; src_mel [2B, 80, Tm] = the B prompt mels followed by the B denoised prompt
    mels (the reference's ``torch.cat([audio, denoised])`` at B = 1) with ``src_length2`` [2B].
    Returns int16 audio [B, n] (n = 320 * frames, x3 / x1.5 with SpeechSR), rows peak-normalised over
    their own length.  B > 1 runs the utterances side by side; rows are independent up to the
    vocoder, whose convolutions see a shorter row's zero padding exactly as the reference's own batched
    ``infer`` does (equal-length batches are exact)."""
    B = text.shape[0]
    x_frame, g, x_lengths, x_mask = models.ttv.inf_extract_tc_latent(text, text_length, src_mel_ttv, src_mel_ttv_length,
                                                                     tone, language, dur=dur)
    codes = models.plm.infer(x_frame)
    w2v_x, pitch = models.ttv.inf_plm_gen(x_frame, g, codes.unsqueeze(1) if B == 1 else codes, x_lengths, x_mask)
    pitch = zero_below(pitch, math.log(55.0))                                  # :166 pitch clipping
    T2 = w2v_x.shape[2]
    if B == 1:
        src_length = torch.full((1,), T2, dtype=torch.int64, device=w2v_x.device)   # :163 the whole padded length
        audio = models.voc.voice_conversion_noise_control(w2v_x, src_length, src_mel, src_length2, pitch,
                                                          noise_scale=noise_scale_vc, denoise_ratio=denoise_ratio,
                                                          noise=noise)
        n_valid = None
    else:
        frames = torch.ceil(x_lengths).to(torch.int64)
        audio = models.voc.voice_conversion_noise_control(w2v_x, frames, src_mel, src_length2, pitch.unsqueeze(1),
                                                          noise_scale=noise_scale_vc, denoise_ratio=denoise_ratio,
                                                          noise=noise)
        n_valid = frames * 320
    if output_sr in (24000, 48000):
        audio = models.sr(audio)
        if n_valid is not None:
            n_valid = n_valid * output_sr // 16000
    wav = peak_int16(audio, n_valid)
    return (wav, audio) if return_float else wav


@torch.no_grad()
def tts_from_prompt(models: TtsModels, mel_fn, text, tone, language, prompt_audio, output_path=None,
                    noise_scale_vc: float = 0.333, output_sr: int = 16000, dur=None, noise=None,
                    denoise_ratio: float = 0.0, denoiser=None, hps_denoiser=None):
    """inference_plm.py:tts :126-201 from the prompt WAVEFORM on: prompt mels (:130-150, `prompt_mels`; with
    ``denoise_ratio`` > 0 the second prompt mel comes from the denoised prompt and the style vectors are mixed by
    voice_conversion_noise_control), text -> w2v / f0 -> waveform (`tts`), optional 16-bit WAV (:195-200).
    text / tone / language int64 [1, N] on the GPU; prompt_audio fp32 [1, n] at 16 kHz on the GPU;
    ``mel_fn`` a finalized Mels_preprocess.MelSpectrogramFixed.  Returns int16 [n_out]."""
    if denoise_ratio != 0 and denoiser is None:
        raise L.HspError("denoise_ratio > 0 needs the denoiser model (denoiser.generator.MPNet), as inference_plm.py:144-147")
    src_mel_ttv, src_mel = prompt_mels(mel_fn, prompt_audio, denoiser if denoise_ratio != 0 else None, hps_denoiser)
    dev = prompt_audio.device
    B = text.shape[0]
    assert B == 1 and prompt_audio.shape[0] == 1, "the reference harness synthesises one utterance per call"
    text_length = torch.full((B,), text.shape[1], dtype=torch.int64, device=dev)
    ttv_len = torch.full((B,), src_mel_ttv.shape[2], dtype=torch.int64, device=dev)
    src_length2 = torch.full((2 * B,), src_mel.shape[2], dtype=torch.int64, device=dev)
    wav = tts(models, text, text_length, tone, language, src_mel_ttv, ttv_len, src_mel, src_length2,
              noise_scale_vc=noise_scale_vc, denoise_ratio=float(denoise_ratio), output_sr=output_sr, dur=dur, noise=noise)[0]
    if output_path is not None:
        write_wav(output_path, output_sr if output_sr in (24000, 48000) else 16000, wav)
    return wav
