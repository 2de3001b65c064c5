"""Synthetic weights and inputs (there are no checkpoints or datasets in this project).

The recipe follows SURVEY.md §8(c)/(d): every tensor is drawn from a generator seeded by
``(seed, crc32(key))`` so any side (golden-vector script, tests, bench, each rank of a
multi-GPU run) can regenerate any tensor independently of iteration order.

* weight-normed layers: ``weight_v ~ N(0,1)``, ``weight_g ~ U(0.3, 0.7)`` per element (round 6: a constant gain
  would hide a gain applied along the wrong axis -- ConvTranspose1d carries one per INPUT channel, the wav2vec2
  positional conv one per TAP; mean 0.5 keeps SURVEY §8(c)'s conditioning)
* biases ``0.1·N``; SnakeBeta ``alpha, beta ~ 0.5·N``
* plain conv / linear weights ``N(0,1)/sqrt(fan_in)`` (this also re-randomises the layers
  the reference zero-initialises -- ``post`` and ``adaLN_modulation`` -- without which
  the flow parity would be vacuous)
* nn.LSTM matrices ``N(0,1)/sqrt(fan_in)``, nn.LayerNorm gains ``1 + 0.1·N``
* resample filters: the closed-form 12-tap kaiser-sinc
* denoiser: PReLU slopes ``0.25 + 0.1·N``, BatchNorm statistics with positive variance, ``in_proj`` like a Linear
"""
from __future__ import annotations

import math
import re
import zlib
from typing import Dict, Iterable, Tuple

import numpy as np

_PRELU_KEY = re.compile(r"(dense_conv_\d|dense_block\.\d+|phase_conv)\.2\.weight$|mask_conv\.3\.weight$")


def kaiser_sinc_filter12() -> np.ndarray:
    """12-tap kaiser-windowed sinc, cutoff 0.25, half-width 0.3 (reference:
    alias_free_torch/filter.py:28-57 with the arguments of resample.py:17-19).
    Computed in float64, returned as float32."""
    ksize, cutoff, half_width = 12, 0.25, 0.3
    half = ksize // 2
    amp = 2.285 * (half - 1) * math.pi * (4 * half_width) + 7.95
    beta = 0.1102 * (amp - 8.7)
    window = np.kaiser(ksize, beta)
    time = np.arange(-half, half) + 0.5
    filt = 2 * cutoff * window * np.sinc(2 * cutoff * time)
    return (filt / filt.sum()).astype(np.float32)


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(key.encode())])


def synth_tensor(key: str, shape: Tuple[int, ...], seed: int = 0) -> np.ndarray:
    r = _rng(seed, key)
    leaf = key.rsplit(".", 1)[-1]
    n = lambda: r.standard_normal(shape).astype(np.float32)
    if leaf == "weight_g":                             # a gain per channel / tap: U(0.3, 0.7), mean 0.5 (SURVEY §8c conditioning)
        return r.uniform(0.3, 0.7, shape).astype(np.float32)
    if leaf == "weight_v":
        return n()
    if leaf == "bias":
        return 0.1 * n()
    if leaf in ("alpha", "beta") and ".act." in key:
        return 0.5 * n()
    if leaf == "filter":
        return kaiser_sinc_filter12().reshape(shape)
    if leaf == "gamma":
        return 1.0 + 0.1 * n()
    if leaf == "beta":
        return 0.1 * n()
    if leaf == "weight" and len(shape) == 1 and _PRELU_KEY.search(key):   # nn.PReLU slope (MP-SENet denoiser)
        return 0.25 + 0.1 * n()
    if leaf == "slope":                                # LearnableSigmoid_2d
        return 1.0 + 0.3 * n()
    if leaf == "running_var":                          # nn.BatchNorm1d statistics
        return 0.5 + np.abs(n())
    if leaf == "running_mean":
        return 0.1 * n()
    if leaf == "num_batches_tracked":
        return np.zeros(shape, np.int64)
    if leaf == "in_proj_weight":                       # nn.MultiheadAttention
        return n() / math.sqrt(shape[1])
    if leaf == "in_proj_bias":
        return 0.1 * n()
    if leaf in ("emb_rel_k", "emb_rel_v"):
        return n() * (shape[-1] ** -0.5)
    if leaf.startswith(("weight_ih", "weight_hh")):   # nn.LSTM
        return n() / math.sqrt(shape[1])
    if leaf.startswith(("bias_ih", "bias_hh")):
        return 0.1 * n()
    if leaf == "weight" and len(shape) == 1:           # nn.LayerNorm gain
        return 1.0 + 0.1 * n()
    if leaf == "weight":
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        return n() / math.sqrt(max(fan_in, 1))
    return n()


def synth_state_dict(shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0) -> Dict[str, np.ndarray]:
    return {k: synth_tensor(k, tuple(s), seed) for k, s in shapes}


def synth_inputs(batch: int, frames: int, seed: int = 20240, mel_frames: int | None = None):
    """Synthetic (mel, w2v, length, f0, noise) of SURVEY.md §8(d): w2v ~ N(0,1)
    [B,1024,T]; f0 [B,1,4T] = log(U(80,400)+1) with unvoiced runs (>=10 frames, ~30 %)
    set to 0; mel ~ N(-4,2) clipped to [-6.9, 3] [B,80,T]; noise ~ N(0,1) [B,192,T]."""
    r = np.random.default_rng(seed)
    tm = frames if mel_frames is None else mel_frames
    w2v = r.standard_normal((batch, 1024, frames)).astype(np.float32)
    f0 = np.log(r.uniform(80.0, 400.0, (batch, 1, 4 * frames)) + 1.0).astype(np.float32)
    for b in range(batch):
        pos = 0
        while pos < 4 * frames:
            run = int(r.integers(10, 40))
            if r.uniform() < 0.3:
                f0[b, 0, pos:pos + run] = 0.0
            pos += run
    mel = np.clip(r.normal(-4.0, 2.0, (batch, 80, tm)), -6.9, 3.0).astype(np.float32)
    noise = r.standard_normal((batch, 192, frames)).astype(np.float32)
    length = np.full((batch,), frames, np.int64)
    return dict(mel=mel, w2v=w2v, length=length, f0=f0, noise=noise)
