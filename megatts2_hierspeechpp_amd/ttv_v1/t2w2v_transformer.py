"""HIP mirror of reference ttv_v1/t2w2v_transformer.py (inference members only)."""
from __future__ import annotations

import math
import os

import torch
from torch import nn

from .. import _lib as L
from .. import attentions
from .. import functional as Fh
from ..hip_layers import Conv1d, ConvTranspose1d, HipLayer, LinearCT, entry as _entry, finalize as _finalize
from ..styleencoder import StyleEncoder
from . import modules
from .Gaussian import GaussianUpsampling, RangePredictor
from .modules import WN
from .quantize import ResidualVectorQuantizer
from .transformer_mega import TransformerEncoder, TransformerEncoderLayer
from .vits_models import DurationPredictor


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


# (round 6)  Layer 0 of the greedy loop keeps its q / k / v of old positions: its input is the embedding
# matrix, whose columns are final once their code is chosen, so every step embeds and projects ONE new column per utterance
# instead of the whole prefix (transformer_mega.MultiHeadAttention.forward_cached).  HSP_PLM_CACHE_L0=0: the full re-projection.
PLM_CACHE_L0 = os.environ.get("HSP_PLM_CACHE_L0", "1") == "1"


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

    def _embed(self, tc, codes, n, prev_logits=None):
        """[1, d_model, Np]: utterance b occupies columns b*n .. b*n+n-1; Np = B*n rounded up to a multiple
        of 4 (16-B rows for the token GEMM's DMA), padding columns are zero.  ``prev_logits`` [1, vq_bins, B]: the
        scores of step n-1, whose argmax this launch takes (and stores to ``codes[:, n-1]``) before it embeds."""
        B = tc.shape[0]
        x = torch.empty(1, self.d_model, (B * n + 3) & ~3, dtype=torch.float32, device=tc.device)
        head = (L.fptr(tc), tc.stride(0), tc.stride(1), self.tc_latent_dim, L.ptr(codes), codes.stride(0),
                L.fptr(self.pc_embedding._w), self.vq_dim, self.pc_embedding.num_embeddings, L.fptr(self.pos_emb._pe_t),
                self.pos_emb.N_POS, L.fptr(self.pos_emb._alpha), L.fptr(x), n, x.shape[2], B, n)
        if prev_logits is None:
            L.check(L.lib().hsp_plm_embed_f32(*head, L.stream_ptr()), "hsp_plm_embed_f32")
        else:
            lg = prev_logits
            assert lg.shape == (1, self.vq_bins, B) and lg.stride(2) == 1
            L.check(L.lib().hsp_plm_embed_step_f32(*head, L.fptr(lg), 1, lg.stride(1), self.vq_bins, L.stream_ptr()),
                    "hsp_plm_embed_step_f32")
        return x

    def _embed_one(self, tc, codes, t, cache, prev_logits):
        """Position t of every utterance into ``cache.emb[:, :, t]`` ([D, B, Tp]: fixed pitch) -- the same launch as _embed
        with every per-position operand shifted to column t and n = 1 (the kernel then takes codes[:, t] = argmax of
        ``prev_logits`` first, as in the full form)."""
        B = tc.shape[0]
        emb = cache.emb
        x_t = emb[:, :, t]                                           # [D, B] view: element (c, b) at c * B Tp + b Tp
        tc_t, codes_t = tc[:, :, t], codes[:, t]
        pe_t = self.pos_emb._pe_t[t:]                                # flat [D][N_POS] table: row c, position t at c * N_POS + t
        head = (L.fptr(tc_t), tc.stride(0), tc.stride(1), self.tc_latent_dim, L.ptr(codes_t), codes.stride(0),
                L.fptr(self.pc_embedding._w), self.vq_dim, self.pc_embedding.num_embeddings, L.fptr(pe_t),
                self.pos_emb.N_POS, L.fptr(self.pos_emb._alpha), L.fptr(x_t), emb.stride(1), emb.stride(0), B, 1)
        if prev_logits is None:
            L.check(L.lib().hsp_plm_embed_f32(*head, L.stream_ptr()), "hsp_plm_embed_f32")
        else:
            lg = prev_logits
            assert lg.shape == (1, self.vq_bins, B) and lg.stride(2) == 1
            L.check(L.lib().hsp_plm_embed_step_f32(*head, L.fptr(lg), 1, lg.stride(1), self.vq_bins, L.stream_ptr()),
                    "hsp_plm_embed_step_f32")

    @_entry
    def step_logits(self, tc_latent, codes, n, out=None, prev_logits=None, cache=None):
        """Logits of position n-1 given the first n columns of ``tc_latent`` [B, 256, T] and of
        ``codes`` [B, >= n] (go token first): one pass of the loop body (:710-716) -> [1, vq_bins, B].
        With ``prev_logits`` (the result of the call for n-1) ``codes[:, n-1]`` is not read but first set to their
        argmax, inside the embedding launch."""
        B = tc_latent.shape[0]
        # the last layer only produces the last position of every utterance when those B columns form a
        # 16-B addressable matrix for the fused-LayerNorm GEMM; otherwise it runs in full
        last_only = B % 4 == 0
        att = self.plm.layers[0].attn
        if cache is not None:
            # layer 0 incrementally (PLM_CACHE_L0): only position n - 1 is embedded and projected, the older columns are kept
            self._embed_one(tc_latent, codes, n - 1, cache, prev_logits)
        if cache is not None and Fh.mha_proj_supported(att.n_heads, att.head_dim, self.d_model, n):
            x = self.plm(None, batch=(B, n), last_only=last_only, cache=cache)
        elif cache is not None:
            # the first three steps (fewer than four keys: no fused attention kernel): the new column still enters the cache,
            # the step itself runs in the full form (its embedding launch takes the same argmax again: idempotent)
            att.qkv(cache.emb[:, :, n - 1:n].permute(1, 0, 2), out=cache.qkv[:, :, n - 1:n].permute(1, 0, 2))
            x = self.plm(self._embed(tc_latent, codes, n, prev_logits), batch=(B, n), last_only=last_only)
        else:
            x = self.plm(self._embed(tc_latent, codes, n, prev_logits), batch=(B, n), last_only=last_only)
        if not last_only:
            x = Fh.copy_strided(x[0][:, :B * n].reshape(self.d_model, B, n)[:, :, n - 1].unsqueeze(0))
        return self.predict_layer(x, out=out)

    @_entry
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
        lg = None
        cache = None
        if PLM_CACHE_L0 and self.plm.num_layers > 1 and \
                Fh.mha_proj_supported(self.plm.layers[0].attn.n_heads, self.plm.layers[0].attn.head_dim, self.d_model, T):
            import types
            Tp = (T + 3) & ~3
            cache = types.SimpleNamespace(
                emb=torch.empty(self.d_model, B, Tp, dtype=torch.float32, device=tc_latent.device),
                qkv=torch.empty(3 * self.d_model, B, Tp, dtype=torch.float32, device=tc_latent.device))
        for t in range(T):
            # the greedy choice of step t-1 is taken inside step t's embedding launch; only the last step's needs its own
            lg = self.step_logits(tc_latent, codes, t + 1, out=all_logits[t:t + 1] if return_logits else None,
                                  prev_logits=lg, cache=cache)
        L.check(L.lib().hsp_argmax_f32(L.fptr(lg), 1, B, B, self.vq_bins, L.ptr(codes[:, T:]), codes.stride(0),
                                       L.stream_ptr()), "hsp_argmax_f32")
        return (codes[:, 1:], all_logits.permute(2, 0, 1)) if return_logits else codes[:, 1:]


# ======================================================================= front-end
class TextEncoder(nn.Module):
    """t2w2v_transformer.TextEncoder (:82-143).  ``cond`` / ``proj`` of the reference are dead code on
    every path (their uses are commented out) and are not mirrored."""

    def __init__(self, n_vocab, n_tone, n_language, out_channels, hidden_channels, filter_channels, n_heads, n_layers,
                 kernel_size, p_dropout):
        super().__init__()
        self.n_vocab, self.n_tone, self.n_language, self.hidden_channels = n_vocab, n_tone, n_language, hidden_channels
        self.emb = Embedding(n_vocab, hidden_channels)
        self.emb_tone = Embedding(n_tone, hidden_channels)
        self.emb_language = Embedding(n_language, hidden_channels)
        self.encoder = attentions.Encoder(hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout)
        self.encoder2 = attentions.Encoder(hidden_channels, filter_channels, n_heads, 1, kernel_size, p_dropout)

    def forward(self, x, x_lengths, g, tone, language, out=None):
        h = Fh.embedding_sum([x, tone, language], [self.emb._w, self.emb_tone._w, self.emb_language._w],
                             [self.n_vocab, self.n_tone, self.n_language], math.sqrt(self.hidden_channels),
                             self.hidden_channels)
        x_mask = Fh.sequence_mask(x_lengths, h.shape[2])
        h = self.encoder(h, x_mask)
        h = self.encoder2(h, x_mask)
        return h, x_mask


class MelEncoder(nn.Module):
    """t2w2v_transformer.MelEncoder (:145-179)."""

    def __init__(self, out_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout):
        super().__init__()
        self.encoder = attentions.Encoder(hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout)
        self.proj = Conv1d(hidden_channels, out_channels, 1)

    def forward(self, x, x_lengths):
        x_mask = Fh.sequence_mask(x_lengths, x.shape[2])
        return self.proj(self.encoder(x, x_mask), mask=x_mask, mask_mode=L.MASK_POST), x_mask


class W2VEncoder(nn.Module):
    """t2w2v_transformer.W2VEncoder (:182-226); ``project`` / ``proj`` are dead code there."""

    def __init__(self, out_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout):
        super().__init__()
        self.cond = Conv1d(256, hidden_channels, 1)
        self.encoder = attentions.Encoder(hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout)
        self.encoder2 = attentions.Encoder(hidden_channels, filter_channels, n_heads, 1, kernel_size, p_dropout)

    def forward(self, x, x_lengths, g, x_mask=None, cond_added=False):
        """``cond_added``: the caller already fused ``+ cond(g)`` into the conv that produced x."""
        if x_mask is None:
            x_mask = Fh.sequence_mask(x_lengths, x.shape[2])
        if not cond_added:
            x = Fh.add_cbias(x, self.cond(g, force_direct=True))
        return self.encoder2(self.encoder(x, x_mask), x_mask), x_mask


class W2VDecoder(nn.Module):
    """t2w2v_transformer.W2VDecoder (:377-405)."""

    def __init__(self, in_channels, hidden_channels, kernel_size, dilation_rate, n_layers, output_size=1024,
                 gin_channels=0, p_dropout=0):
        super().__init__()
        self.pre = Conv1d(in_channels, hidden_channels, 1)
        self.enc = WN(hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=gin_channels, p_dropout=p_dropout)
        self.proj = Conv1d(hidden_channels, output_size, 1)

    def forward(self, x, x_mask, g=None):
        # pre(x * mask) * mask: x arrives masked from the encoder
        x = self.pre(x, mask=x_mask, mask_mode=L.MASK_POST)
        x = self.enc(x, x_mask, g=g)
        return self.proj(x, mask=x_mask, mask_mode=L.MASK_POST)


class PitchPredictor(nn.Module):
    """t2w2v_transformer.PitchPredictor (:408-463): HiFi-GAN style x4 upsampler, w2v -> log-f0."""

    def __init__(self):
        super().__init__()
        ks, ups, ch0 = [3, 5, 7], [(2, 4), (2, 4)], 256
        self.num_kernels, self.num_upsamples = len(ks), len(ups)
        self.conv_pre = Conv1d(1024, ch0, 7, padding=3)
        self.ups = nn.ModuleList([ConvTranspose1d(ch0 // 2 ** i, ch0 // 2 ** (i + 1), k, u, padding=(k - u) // 2,
                                                  weight_norm=True) for i, (u, k) in enumerate(ups)])
        self.resblocks = nn.ModuleList([modules.ResBlock1(ch0 // 2 ** (i + 1), k, (1, 3, 5))
                                        for i in range(len(ups)) for k in ks])
        self.conv_post = Conv1d(ch0 // 2 ** len(ups), 1, 7, padding=3, bias=False)
        self.cond = Conv1d(256, ch0, 1)

    def forward(self, x, g, lengths=None):
        """``lengths`` (int64 [B], 50 Hz frames): zero every intermediate past an utterance's end, so a
        short row of a batch sees the same zero padding as when it runs alone (the reference has no mask
        here and runs B = 1)."""
        B, _, T = x.shape
        mk = (lambda r: Fh.sequence_mask(lengths * r, T * r)) if lengths is not None else (lambda r: None)
        mm = lambda m: dict(mask=m, mask_mode=L.MASK_POST) if m is not None else {}
        x = self.conv_pre(x, cbias=self.cond(g, force_direct=True), **mm(mk(1)))
        for i in range(self.num_upsamples):
            m = mk(2 ** (i + 1))
            x = self.ups[i](x, lrelu=modules.LRELU_SLOPE)
            if m is not None:
                x = Fh.mask_mul(x, m)
            xs = torch.empty_like(x)
            for j in range(self.num_kernels):
                self.resblocks[i * self.num_kernels + j](x, x_mask=m, out=xs, accumulate=j > 0,
                                                         post_scale=1.0 / self.num_kernels if j == self.num_kernels - 1 else 1.0)
            x = xs
        return self.conv_post(x, lrelu=0.01)   # F.leaky_relu default slope (:458)


class PLMConv(nn.Module):
    """t2w2v_transformer.PLMConv (:517-528): x -> conv2(conv1(x * m) * m) * m on the 20-channel prosody track."""

    def __init__(self, hidden_channels=80):
        super().__init__()
        self.conv1 = Conv1d(hidden_channels, hidden_channels, 5, padding=2)
        self.conv2 = Conv1d(hidden_channels, hidden_channels, 5, padding=2)

    def forward(self, x, mask, premasked=False):
        if not premasked:
            x = Fh.mask_mul(x, mask)
        h = self.conv1(x, mask=mask, mask_mode=L.MASK_POST, force_direct=True)
        return self.conv2(h, mask=mask, mask_mode=L.MASK_POST, force_direct=True)


class SynthesizerTrn(nn.Module):
    """t2w2v_transformer.SynthesizerTrn (:721-1077), inference members only
    (``inf_extract_tc_latent``, ``inf_plm_gen`` and the older non-PLM ``infer``).  Same constructor signature;
    the ``lr`` sub-module, which only the training ``forward`` touches, is not built, so reference
    checkpoints load with ``strict=False``.

    The reference front-end is B = 1 only (RangePredictor's ``.squeeze()`` and the [B,1,N] x [B,N]
    broadcast at :961 break for B > 1).  Here a batch is B independent utterances: every stage honours
    the per-utterance lengths so that row b equals the reference run on utterance b alone, and frames
    past an utterance's length are zero."""

    def __init__(self, n_vocab, n_tone, n_language, spec_channels, hop_length, sampling_rate, segment_size,
                 inter_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout, resblock,
                 resblock_kernel_sizes, resblock_dilation_sizes, gin_channels=256, prosody_size=20, cfg=False,
                 freeze_quantizer=None, **kwargs):
        super().__init__()
        self.inter_channels, self.hidden_channels, self.stride = inter_channels, hidden_channels, 8
        ic = inter_channels
        self.enc_p = TextEncoder(n_vocab, n_tone, n_language, out_channels=ic, hidden_channels=ic, filter_channels=ic * 4,
                                 n_heads=4, n_layers=3, kernel_size=9, p_dropout=0.2)
        self.mel_encoder = MelEncoder(out_channels=256, hidden_channels=80, filter_channels=80 * 4, n_heads=4, n_layers=2,
                                      kernel_size=9, p_dropout=0.2)
        self.mha = attentions.MultiHeadAttention(ic, ic, n_heads=4, p_dropout=0.2)
        self.cond_g = Conv1d(256, ic, 1)
        self.w2v_encoder = W2VEncoder(out_channels=ic, hidden_channels=ic, filter_channels=ic * 4, n_heads=4, n_layers=3,
                                      kernel_size=9, p_dropout=0.2)
        self.w2v_decoder = W2VDecoder(ic, ic * 2, 5, 1, 8, output_size=1024, p_dropout=0.1, gin_channels=256)
        self.emb_g = StyleEncoder(in_dim=80, hidden_dim=256, out_dim=256)
        self.duration_predictor = DurationPredictor(hidden_channels, 256, 3, 0.5, gin_channels=gin_channels)
        self.RangePredictor = RangePredictor(257, 256)
        self.gaussian = GaussianUpsampling()
        self.dur_downsample = Conv1d(hidden_channels, hidden_channels, 1)   # stride 2: run on an x[..., ::2] view
        self.pp = PitchPredictor()
        self.quantizer = ResidualVectorQuantizer(dimension=20, n_q=1, bins=1024)
        self.ssl_proj = Conv1d(20, ic, 1)
        # legacy (non-PLM) prosody path of infer(): the prompt mel's first 20 bins -> PLMConv -> max-pool 8 -> PLMConv
        # -> nearest code (:794-797); nn.MaxPool1d has no parameters
        self.plm_conv1 = PLMConv(hidden_channels=20)
        self.plm_conv2 = PLMConv(hidden_channels=20)

    # keys of the reference checkpoint that no inference path reads
    UNUSED = ("lr.", "enc_p.cond.", "enc_p.proj.", "w2v_encoder.project.", "w2v_encoder.proj.")
    UNUSED_LEAVES = ("_codebook.inited", "_codebook.cluster_size", "_codebook.embed_avg")

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """Accepts the reference's checkpoints: a bare state dict or utils.load_checkpoint's {'model': ...}
        (inference_plm.py:233); training-only / dead keys are skipped."""
        if "model" in state_dict and not any(k.startswith("enc_p.") for k in state_dict):
            state_dict = state_dict["model"]
        sd = {k: v for k, v in state_dict.items() if not k.startswith(self.UNUSED) and not k.endswith(self.UNUSED_LEAVES)}
        return super().load_state_dict(sd, strict=strict, **kw)

    def finalize(self, device, materialize: bool = True):
        self.arena = _finalize(self, device, materialize)
        return self

    @_entry
    @torch.no_grad()
    def inf_extract_tc_latent(self, x, x_lengths, y_mel, y_length, tone, language, mrte_mel=None, mrte_mel_lengths=None,
                              length_scale=1, dur=None):
        """(:937-982) ids/tone/language int64 [B, N], x_lengths [B], y_mel [B, 80, Tm], y_length [B] ->
        x_frame [B, 256, T2], g [B, 256, 1], x_lengths (float, frames / 2) [B], x_mask BOOL [B, 1, T2] (the
        reference's dtypes, :979-982; ``inf_plm_gen`` takes that mask or a float one).
        ``dur`` [B, N] (optional) overrides the predicted durations (BASELINE config 3 pins them)."""
        B, N = x.shape
        C = self.inter_channels
        x_lengths = x_lengths.to(torch.int64)
        ref_mel, ref_len = (mrte_mel, mrte_mel_lengths) if mrte_mel is not None else (y_mel, y_length)
        g = self.emb_g(ref_mel, Fh.sequence_mask(ref_len, ref_mel.shape[2]), per_utterance=True).unsqueeze(-1)
        h, x_mask = self.enc_p(x, x_lengths, g, tone, language)
        mel_out, h_mask = self.mel_encoder(ref_mel, ref_len)
        # x = x + mha(x, mel) + cond_g(g), written into the first 256 channels of the RangePredictor input
        xd = torch.empty(B, C + 1, N, dtype=torch.float32, device=x.device)
        xs = xd[:, :C]
        self.mha(h, mel_out, mask_q=x_mask, mask_k=h_mask, res=h, cbias=self.cond_g(g, force_direct=True), out=xs)
        frames = torch.empty(B, dtype=torch.float32, device=x.device)
        dcol = xd[:, C]
        if dur is None:
            logw = self.duration_predictor(xs, x_mask, g=g, lengths=x_lengths)
            lw = (L.fptr(logw), logw.stride(0))
        else:
            dcol.copy_(dur.to(torch.float32))
            lw = (None, 0)
        L.check(L.lib().hsp_duration_f32(lw[0], lw[1], L.ptr(x_lengths), float(length_scale), L.fptr(dcol), dcol.stride(0),
                                         L.fptr(frames), B, N, L.stream_ptr()), "hsp_duration_f32")
        rng = self.RangePredictor(xd, x_lengths)
        frames_h = frames.cpu()                       # the reference's `.item()` (Gaussian.py:53): T is data dependent
        T = int(frames_h.max().item())
        x_frame = self.gaussian(xs, dcol, rng, x_lengths, frames, T)
        frame_lengths = frames_h / 2                   # (:976) float, may end in .5
        T2 = (T - 1) // 2 + 1
        len2 = torch.ceil(frame_lengths).to(torch.int64).to(x.device)
        mask2 = Fh.sequence_mask(len2, T2)
        x_frame = self.dur_downsample(x_frame[:, :, ::2], mask=mask2, mask_mode=L.MASK_POST)
        return x_frame, g, frame_lengths.to(x.device), mask2.to(torch.bool)   # a dtype cast, as the reference's (:982)

    @_entry
    @torch.no_grad()
    def inf_plm_gen(self, x_frame, g, codes, x_lengths, x_mask):
        """(:984-994) codes int64 [B, T2] (or the reference's [1, 1, T2]) -> w2v [B, 1024, T2], lf0 [B, 4 T2].
        ``x_mask`` (bool as returned by inf_extract_tc_latent, or float) is accepted for the reference's signature; the
        masks used here are rebuilt from ``x_lengths`` by the mask kernel (same values: :975-982)."""
        T2 = x_frame.shape[2]
        q = self.quantizer.decode(codes)
        # x_frame + ssl_proj(quantized) [+ w2v_encoder.cond(g), fused]
        x = self.ssl_proj(q, res=x_frame, cbias=self.w2v_encoder.cond(g, force_direct=True))
        len2 = torch.ceil(x_lengths.to(torch.float32)).to(torch.int64)
        mask = Fh.sequence_mask(len2, T2)
        x2v_enc, _ = self.w2v_encoder(x, x_lengths, g, x_mask=mask, cond_added=True)
        w2v_pred = self.w2v_decoder(x2v_enc, mask, g=g)
        lf0 = self.pp(w2v_pred, g, lengths=len2)
        mask4 = Fh.sequence_mask(4 * len2, 4 * T2)
        return w2v_pred, Fh.mask_mul(lf0, mask4).squeeze(1)

    @_entry
    @torch.no_grad()
    def infer(self, x, x_lengths, mel_spk, mel_spk_lengths, tone, language, dur=None, mrte_mel=None, mrte_mel_lengths=None,
              noise_scale=1, noise_scale_w=1, length_scale=1, denoise_ratio=0):
        """The older non-PLM text -> (w2v, lf0) path (:996-1077; call site inference.py:158): the prosody codes come
        from the PROMPT mel (first 20 bins -> PLMConv -> max-pool 8 -> PLMConv -> nearest code of the 1024-entry
        codebook, every code held for 8 frames) instead of from the prosody LM.

        The reference adds the projected codes, one per prompt-mel frame, to the text-derived frames (:1057), so it
        only runs when the caller passes ``dur`` [B, N] with sum(dur) / 2 == mel_spk.shape[2] (its RangePredictor call
        also needs ``dur`` 2-D) and the mel length is a multiple of 8 (:999); the same holds here, anything else raises.
        ``x_lengths`` of the encoder come from the PREDICTED durations (:1031-1035), as in the reference.  ``mrte_mel``
        is accepted and -- as in the reference, which overwrites g at :1009 -- does not change the style vector.
        B = 1 (the reference's RangePredictor .squeeze()); a batch here is B independent utterances."""
        if dur is None:
            raise L.HspError("infer(): the reference path needs explicit durations (dur [B, N]); see the docstring")
        B, N = x.shape
        C = self.inter_channels
        Tm = mel_spk.shape[2]
        if Tm % self.stride:
            raise L.HspError("infer(): the prompt mel length must be a multiple of 8 (sequence_mask(len / 8, T / 8))")
        x_lengths = x_lengths.to(torch.int64)
        mel_mask = Fh.sequence_mask(mel_spk_lengths, Tm)
        g = self.emb_g(mel_spk, mel_mask, per_utterance=True).unsqueeze(-1)
        h, x_mask = self.enc_p(x, x_lengths, g, tone, language)
        ref_mel, ref_len = (mrte_mel, mrte_mel_lengths) if mrte_mel is not None else (mel_spk, mel_spk_lengths)
        mel_out, h_mask = self.mel_encoder(ref_mel, ref_len)
        xd = torch.empty(B, C + 1, N, dtype=torch.float32, device=x.device)
        xs = xd[:, :C]
        self.mha(h, mel_out, mask_q=x_mask, mask_k=h_mask, res=h, cbias=self.cond_g(g, force_direct=True), out=xs)
        # predicted durations: only their sum is used (the encoder's lengths); the frames follow the caller's `dur`
        logw = self.duration_predictor(xs, x_mask, g=g, lengths=x_lengths)
        dpred = torch.empty(B, N, dtype=torch.float32, device=x.device)
        frames_pred = torch.empty(B, dtype=torch.float32, device=x.device)
        L.check(L.lib().hsp_duration_f32(L.fptr(logw), logw.stride(0), L.ptr(x_lengths), 1.0, L.fptr(dpred), dpred.stride(0),
                                         L.fptr(frames_pred), B, N, L.stream_ptr()), "hsp_duration_f32")
        dcol = xd[:, C]
        dcol.copy_(dur.reshape(B, N).to(torch.float32))
        frames = torch.empty(B, dtype=torch.float32, device=x.device)
        L.check(L.lib().hsp_duration_f32(None, 0, L.ptr(x_lengths), 1.0, L.fptr(dcol), dcol.stride(0), L.fptr(frames), B, N,
                                         L.stream_ptr()), "hsp_duration_f32")
        rng = self.RangePredictor(xd, x_lengths)
        frames_h = frames.cpu()
        T = int(frames_h.max().item())
        T2 = (T - 1) // 2 + 1
        if T2 != Tm:
            raise L.HspError(f"infer(): sum(dur) / 2 = {T2} frames but the prompt mel has {Tm}: the reference adds the "
                             "two tensors frame by frame (t2w2v_transformer.py:1057)")
        x_frame = self.gaussian(xs, dcol, rng, x_lengths, frames, T)
        x_frame = self.dur_downsample(x_frame[:, :, ::2])
        # prosody codes of the prompt
        q_in = self.plm_conv1(mel_spk[:, :20], mel_mask)
        pooled = torch.empty(B, 20, Tm // self.stride, dtype=torch.float32, device=x.device)
        L.check(L.lib().hsp_maxpool1d_f32(L.fptr(q_in), q_in.stride(0), q_in.stride(1), L.fptr(pooled), B, 20, Tm, self.stride,
                                          L.stream_ptr()), "hsp_maxpool1d_f32")
        pool_len = torch.div(mel_spk_lengths.to(torch.int64) + self.stride - 1, self.stride, rounding_mode="floor")
        pool_mask = Fh.sequence_mask(pool_len, Tm // self.stride)   # t < len / 8 (float compare) == t < ceil(len / 8)
        q_in = self.plm_conv2(pooled, pool_mask)
        codes = torch.empty(B, Tm, dtype=torch.int64, device=x.device)
        cb = self.quantizer.vq.layers[0]._codebook
        L.check(L.lib().hsp_vq_nearest_f32(L.fptr(q_in), q_in.stride(0), q_in.stride(1), L.fptr(cb._w), L.ptr(codes),
                                           codes.stride(0), B, 20, Tm // self.stride, self.quantizer.bins, self.stride, Tm,
                                           L.stream_ptr()), "hsp_vq_nearest_f32")
        q = self.quantizer.decode(codes)
        # x_frame + ssl_proj(q * mel_mask) * mel_mask [+ w2v_encoder.cond(g), fused]
        xq = self.ssl_proj(Fh.mask_mul(q, mel_mask), mask=mel_mask, mask_mode=L.MASK_PRE, res=x_frame,
                           cbias=None)
        xq = Fh.add_cbias(xq, self.w2v_encoder.cond(g, force_direct=True))
        enc_len = frames_pred / 2                                     # (:1031-1035) float, from the predicted durations
        len2 = torch.clamp(torch.ceil(enc_len), max=T2).to(torch.int64)
        y_mask = Fh.sequence_mask(len2, T2)
        x2v_enc, _ = self.w2v_encoder(xq, enc_len, g, x_mask=y_mask, cond_added=True)
        w2v_pred = self.w2v_decoder(x2v_enc, y_mask, g=g)
        lf0 = self.pp(w2v_pred, g)
        return w2v_pred, lf0.squeeze(1)
