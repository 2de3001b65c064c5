"""HIP mirror of reference ttv_v1/t2w2v_transformer.py (inference members only)."""
from __future__ import annotations

import math

import torch
from torch import nn

from .. import _lib as L
from ..hip_layers import HipLayer, LinearCT, finalize as _finalize
from .transformer_mega import TransformerEncoder, TransformerEncoderLayer


class SinePositionalEmbedding(HipLayer):
    """t2w2v_transformer.SinePositionalEmbedding (:466-514).  The 4000-row table the reference
    precomputes on the host in its constructor (:482) is computed the same way (torch CPU fp32 ops)
    and kept transposed ``[dim][4000]`` in the weight arena; adding it is fused into
    ``hsp_plm_embed_f32``."""

    N_POS = 4000

    def __init__(self, dim_model: int, dropout: float = 0.0, scale: bool = False, alpha: bool = False):
        super().__init__()
        if scale:
            raise NotImplementedError("the PLM builds SinePositionalEmbedding with scale=False")
        self.dim_model = dim_model
        self.alpha = nn.Parameter(torch.ones(1), requires_grad=False)
        self._pe_t = self._alpha = None

    def table(self) -> torch.Tensor:
        pe = torch.zeros(self.N_POS, self.dim_model)
        position = torch.arange(0, self.N_POS, dtype=torch.float32).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, self.dim_model, 2, dtype=torch.float32)
                             * -(math.log(10000.0) / self.dim_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        return pe

    def hsp_requests(self):
        return [("pe_t", self.dim_model * self.N_POS), ("alpha", 1)]

    def hsp_fill(self, arena, materialize):
        self._pe_t, self._alpha = arena.view(self, "pe_t"), arena.view(self, "alpha")
        if materialize:
            self._pe_t.copy_(self.table().t().contiguous().reshape(-1))
            self._alpha.copy_(self.alpha.data)


class Embedding(HipLayer):
    """torch.nn.Embedding table kept in the weight arena (parameter name ``weight``)."""

    def __init__(self, num_embeddings, embedding_dim):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.weight = nn.Parameter(torch.zeros(num_embeddings, embedding_dim), requires_grad=False)
        self._w = None

    def hsp_requests(self):
        return [("w", self.num_embeddings * self.embedding_dim)]

    def hsp_fill(self, arena, materialize):
        self._w = arena.view(self, "w")
        if materialize:
            self._w.copy_(self.weight.data.reshape(-1))


class Megatts2PLM1(nn.Module):
    """t2w2v_transformer.Megatts2PLM1 (:627-718): greedy prosody-code generation.

    ``infer`` follows the reference loop exactly -- at step t the whole prefix of t+1 positions is
    re-encoded bidirectionally (no KV cache is possible) and the last position's logits pick the next
    code -- but it takes a batch: the reference's go token is ``[1, 1]`` so it runs one utterance per
    call; utterances are independent, so B rows run the same loop side by side (rows shorter than
    the longest simply produce codes past their length that the caller drops).  The code buffer
    lives in device memory and every step is a fixed chain of launches, so the loop has no host
    synchronisation and can be captured in a hipGraph."""

    GO_ID = 1024

    def __init__(self, n_layers: int = 4, n_heads: int = 4, vq_dim: int = 20, tc_latent_dim: int = 256,
                 vq_bins: int = 1024, kernel_size: int = 9, dropout: float = 0.1):
        super().__init__()
        d_model = vq_dim + tc_latent_dim
        self.d_model, self.vq_dim, self.tc_latent_dim, self.vq_bins = d_model, vq_dim, tc_latent_dim, vq_bins
        self.plm = TransformerEncoder(
            TransformerEncoderLayer(dim=d_model, ff_dim=d_model * 4, n_heads=n_heads, dropout=dropout, conv_ff=False),
            num_layers=n_layers)
        self.predict_layer = LinearCT(d_model, vq_bins, bias=False)
        self.pos_emb = SinePositionalEmbedding(d_model)
        self.pc_embedding = Embedding(vq_bins + 2, vq_dim)

    def finalize(self, device, materialize: bool = True):
        self.arena = _finalize(self, device, materialize)
        return self

    def _embed(self, tc, codes, n):
        """[1, d_model, B*n]: utterance b occupies columns b*n .. b*n+n-1."""
        B = tc.shape[0]
        x = torch.empty(1, self.d_model, B * n, dtype=torch.float32, device=tc.device)
        L.check(L.lib().hsp_plm_embed_f32(L.fptr(tc), tc.stride(0), tc.stride(1), self.tc_latent_dim, L.ptr(codes),
                                          codes.stride(0), L.fptr(self.pc_embedding._w), self.vq_dim,
                                          self.pc_embedding.num_embeddings, L.fptr(self.pos_emb._pe_t),
                                          self.pos_emb.N_POS, L.fptr(self.pos_emb._alpha), L.fptr(x), n, B * n, B, n,
                                          L.stream_ptr()), "hsp_plm_embed_f32")
        return x

    def step_logits(self, tc_latent, codes, n, out=None):
        """Logits of position n-1 given the first n columns of ``tc_latent`` [B, 256, T] and of
        ``codes`` [B, >= n] (go token first): one pass of the loop body (:710-716) -> [1, vq_bins, B]."""
        B = tc_latent.shape[0]
        x = self.plm(self._embed(tc_latent, codes, n), batch=(B, n), last_only=True)
        return self.predict_layer(x, out=out)

    @torch.no_grad()
    def infer(self, tc_latent: torch.Tensor, return_logits: bool = False):
        """tc_latent (B, D, T) -> int64 codes (B, T)  [+ fp32 logits (B, T, vq_bins)]."""
        if self.pos_emb._pe_t is None:
            raise L.HspError("Megatts2PLM1 used before finalize()")
        B, D, T = tc_latent.shape
        assert D == self.tc_latent_dim and tc_latent.stride(2) == 1 and T <= self.pos_emb.N_POS
        codes = torch.empty(B, T + 1, dtype=torch.int64, device=tc_latent.device)
        codes[:, 0] = self.GO_ID
        all_logits = torch.empty(T, self.vq_bins, B, dtype=torch.float32, device=tc_latent.device) if return_logits \
            else None
        for t in range(T):
            lg = self.step_logits(tc_latent, codes, t + 1, out=all_logits[t:t + 1] if return_logits else None)
            L.check(L.lib().hsp_argmax_f32(L.fptr(lg), 1, B, B, self.vq_bins, L.ptr(codes[:, t + 1:]),
                                           codes.stride(0), L.stream_ptr()), "hsp_argmax_f32")
        return (codes[:, 1:], all_logits.permute(2, 0, 1)) if return_logits else codes[:, 1:]
