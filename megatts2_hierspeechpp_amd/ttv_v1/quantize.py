"""HIP mirror of reference ttv_v1/quantize.py + core_vq.py, decode side only (n_q = 1)."""
from __future__ import annotations

import torch
from torch import nn

from .. import functional as Fh
from ..hip_layers import HipLayer


class EuclideanCodebook(HipLayer):
    """core_vq.EuclideanCodebook: only the ``embed`` table is read at inference (:188-190)."""

    def __init__(self, dim, codebook_size):
        super().__init__()
        self.dim, self.codebook_size = dim, codebook_size
        self.register_buffer("embed", torch.zeros(codebook_size, dim))
        self._w = None

    def hsp_requests(self):
        return [("w", self.codebook_size * self.dim)]

    def hsp_fill(self, arena, materialize):
        self._w = arena.view(self, "w")
        if materialize:
            self._w.copy_(self.embed.reshape(-1))


class VectorQuantization(nn.Module):
    def __init__(self, dim, codebook_size):
        super().__init__()
        self._codebook = EuclideanCodebook(dim, codebook_size)


class ResidualVectorQuantization(nn.Module):
    def __init__(self, num_quantizers, dim, codebook_size):
        super().__init__()
        self.layers = nn.ModuleList([VectorQuantization(dim, codebook_size) for _ in range(num_quantizers)])


class ResidualVectorQuantizer(nn.Module):
    """quantize.ResidualVectorQuantizer (:112-120 decode).  Key layout ``vq.layers.0._codebook.embed``."""

    def __init__(self, dimension=256, n_q=8, bins=1024, **kwargs):
        super().__init__()
        if n_q != 1:
            raise NotImplementedError("the front-end builds ResidualVectorQuantizer(dimension=20, n_q=1, bins=1024)")
        self.dimension, self.n_q, self.bins = dimension, n_q, bins
        self.vq = ResidualVectorQuantization(n_q, dimension, bins)

    def decode(self, codes: torch.Tensor, st: int = 0) -> torch.Tensor:
        """codes int64 [B, T] (one utterance per row; the reference passes [n_q = 1, B = 1, T]) ->
        quantized [B, dimension, T] (core_vq.py:380-386 -> :298-305 -> :188-190)."""
        if codes.dim() == 3:
            codes = codes[:, 0] if codes.shape[1] == 1 else codes[0]
        cb = self.vq.layers[0]._codebook
        return Fh.embedding_sum([codes], [cb._w], [self.bins], 1.0, self.dimension)
