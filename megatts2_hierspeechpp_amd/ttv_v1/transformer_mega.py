"""Pre-LN transformer encoder of the Mega-TTS2 PLM (reference ttv_v1/transformer_mega.py).

The reference works time-major ``[B, T, D]``; here a batch is ONE channel-major matrix
``[1, D, B*T]`` with the utterances side by side on the column axis (``batch=(B, T)`` tells the
attention where they are), so that every ``nn.Linear`` is a single GEMM over all B*T tokens on the
MFMA conv kernel; the three q/k/v projections run as one stacked GEMM, attention is ``hsp_mha_f32``
on strided views.  Inference only: dropout is the identity
(transformer_mega.py:78 passes ``dropout_p = 0`` outside training)."""
from __future__ import annotations

import math

import torch
from torch import nn

from .. import _lib as L
from .. import functional as Fh
from ..hip_layers import HipLayer, LinearCT, StackedLinearCT


class LayerNorm(HipLayer):
    """torch.nn.LayerNorm(dim) (parameter names ``weight`` / ``bias``) over the channel axis of [B, D, T]."""

    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(dim), requires_grad=False)
        self._g = self._b = None

    def hsp_requests(self):
        return [("g", self.dim), ("b", self.dim)]

    def hsp_fill(self, arena, materialize):
        self._g, self._b = arena.view(self, "g"), arena.view(self, "b")
        if materialize:
            self._g.copy_(self.weight.data)
            self._b.copy_(self.bias.data)

    def forward(self, x):
        if self._g is None:
            raise L.HspError("LayerNorm used before finalize()")
        return Fh.layernorm_mod(x, self.eps, gamma=self._g, beta=self._b)


class MultiHeadAttention(nn.Module):
    """transformer_mega.MultiHeadAttention (:44-87), self-attention without a mask (the only form
    Megatts2PLM1.infer uses: ``self.plm(x_pos)`` passes no lengths, t2w2v_transformer.py:715)."""

    def __init__(self, qkv_dim, n_heads=8, dropout=0., pre_norm=None):
        """``pre_norm``: the LayerNorm the encoder layer applies to this module's input; it is folded into
        the stacked q/k/v GEMM (hsp_conv1d_args.ln_c1) and ``forward`` then takes the un-normalised x."""
        super().__init__()
        assert qkv_dim % n_heads == 0
        self.n_heads, self.head_dim, self.qkv_dim = n_heads, qkv_dim // n_heads, qkv_dim
        self.w_q = LinearCT(qkv_dim, qkv_dim, packed=False)
        self.w_k = LinearCT(qkv_dim, qkv_dim, packed=False)
        self.w_v = LinearCT(qkv_dim, qkv_dim, packed=False)
        self.qkv = StackedLinearCT([self.w_q, self.w_k, self.w_v])
        if pre_norm is not None:
            self.qkv.fuse_input_layernorm(pre_norm)
        self.out_proj = nn.ModuleList([LinearCT(qkv_dim, qkv_dim)])  # nn.Sequential(Linear, Dropout): key "out_proj.0"
        self.out_proj[0].keep_rowmajor_weight()   # A operand of the projection inside hsp_mha_proj_f32

    def forward_cached(self, cache, n, out_cols):
        """Layer 0 of the greedy loop with its q / k / v of OLD positions kept (round 6, VERDICT r05 item 3's small win): the
        layer's input is the embedding matrix, whose columns do not change once their code is chosen
        (t2w2v_transformer.py:708-716: position j of step t is position j of every later step), and q / k / v are
        per-column functions of it, so step t computes the ONE new column per utterance -- a [3D x B] LayerNorm-GEMM
        instead of [3D x B n] -- into ``cache.qkv`` [3D, B, Tp] beside the ``cache.emb`` [D, B, Tp] it was computed from.
        Attention reads the first n columns of every utterance through strided views; the output (+ residual = the
        embeddings) goes to the side-by-side layout [1, D, Np] the rest of the step works in."""
        D, B, Tp = self.qkv_dim, cache.emb.shape[1], cache.emb.shape[2]
        t = n - 1
        xin = cache.emb[:, :, t:t + 1].permute(1, 0, 2)            # [B, D, 1]: utterances as the batch, ONE column each
        self.qkv(xin, out=cache.qkv[:, :, t:t + 1].permute(1, 0, 2))
        per = lambda m: m[:, :, :n].permute(1, 0, 2)               # [B, C, n] view of a [C, B, Tp] matrix
        q, k, v = (per(cache.qkv[i * D:(i + 1) * D]) for i in range(3))
        Np = (B * n + 3) & ~3
        y = torch.empty(1, D, Np, dtype=torch.float32, device=cache.emb.device) if Np == B * n else \
            torch.zeros(1, D, Np, dtype=torch.float32, device=cache.emb.device)
        lin = self.out_proj[0]
        Fh.mha_proj(q, k, v, self.n_heads, 1.0 / math.sqrt(self.head_dim), lin._wt, bias=lin._b, res=per(cache.emb),
                    out=out_cols(y))
        return y

    def forward(self, x, kv=None, mask=None, res=None, batch=None, last_only=False):
        """x [1, D, B*T] -> out_proj(attention) [+ res]; ``last_only`` -> [1, D, B]: only the last
        position of every utterance (columns T-1, 2T-1, ...), ``res`` then being [1, D, B] already."""
        if kv is not None or mask is not None:
            raise NotImplementedError("cross-attention / masks are not on the PLM inference path")
        D = self.qkv_dim
        B, T = batch if batch is not None else (x.shape[0], x.shape[2])
        qkv = self.qkv(x)
        if batch is not None and Fh.mha_proj_supported(self.n_heads, self.head_dim, D, T):
            # ONE launch: attention over all heads of a 16-query tile + out_proj + bias + residual (round 4)
            per = lambda m: m[:, :B * T].reshape(-1, B, T).permute(1, 0, 2)       # [B, C, T] view of a [C, Np] matrix
            q, k, v = (per(qkv[0, i * D:(i + 1) * D]) for i in range(3))
            lin = self.out_proj[0]
            if last_only:
                # only the last position of every utterance: y [1, D, B], res [1, D, B] (a strided view of x)
                y = torch.empty(1, D, B, dtype=torch.float32, device=x.device)
                as_b = lambda m: m[0].permute(1, 0).unsqueeze(2)                     # [B, D, 1] view of [1, D, B]
                Fh.mha_proj(q[:, :, T - 1:], k, v, self.n_heads, 1.0 / math.sqrt(self.head_dim), lin._wt, bias=lin._b,
                            res=None if res is None else as_b(res), out=as_b(y))
                return y
            y = torch.empty_like(x) if x.shape[2] == B * T else torch.zeros_like(x)
            Fh.mha_proj(q, k, v, self.n_heads, 1.0 / math.sqrt(self.head_dim), lin._wt, bias=lin._b,
                        res=None if res is None else per(res[0]), out=per(y[0]))
            return y
        # padding columns (odd batch sizes only) must stay finite: they flow through the following GEMMs
        o = torch.empty_like(x) if batch is None or x.shape[2] == B * T else torch.zeros_like(x)
        # [B, C, T] view of a [C, Np] matrix (Np >= B*T: rows may be padded to a multiple of 4 columns)
        per_utt = lambda m: m[:, :B * T].reshape(-1, B, T).permute(1, 0, 2) if batch is not None else m
        q, k, v = (per_utt(qkv[0, i * D:(i + 1) * D]) if batch is not None else qkv[:, i * D:(i + 1) * D]
                   for i in range(3))
        Fh.mha(q, k, v, self.n_heads, 1.0 / math.sqrt(self.head_dim), out=per_utt(o[0]) if batch is not None else o)
        if last_only:
            o = o[0][:, :B * T].reshape(D, B, T)[:, :, T - 1].unsqueeze(0)  # [1, D, B] view, time stride T
        return self.out_proj[0](o, res=res)


class TransformerEncoderLayer(nn.Module):
    """transformer_mega.TransformerEncoderLayer (:89-132), ``conv_ff=False``."""

    def __init__(self, dim, ff_dim, conv_ff=False, n_heads=8, dropout=0.):
        super().__init__()
        if conv_ff:
            raise NotImplementedError("Megatts2PLM1 builds its layers with conv_ff=False")
        self.dim, self.ff_dim, self.conv_ff, self.n_heads = dim, ff_dim, conv_ff, n_heads
        self.norm1 = LayerNorm(dim)
        self.norm2 = LayerNorm(dim)
        # both LayerNorms run inside the GEMM that consumes them (statistics from the staged input tile)
        self.attn = MultiHeadAttention(dim, n_heads=n_heads, dropout=dropout, pre_norm=self.norm1)
        # nn.Sequential(Linear, ReLU, Dropout, Linear): keys "ff.0" and "ff.3"
        self.ff = nn.ModuleDict({"0": LinearCT(dim, ff_dim), "3": LinearCT(ff_dim, dim)})
        self.ff["0"].fuse_input_layernorm(self.norm2)

    def forward(self, x, mask=None, batch=None, last_only=False, cache=None):
        """``last_only`` returns just the last position of every utterance ``[1, D, B]`` (all the
        greedy loop reads from the final layer); attention still sees the whole prefix.  ``cache`` (layer 0 of the greedy
        loop): x is None, the layer input lives in ``cache.emb`` (MultiHeadAttention.forward_cached)."""
        if cache is not None:
            B, T = batch
            x = self.attn.forward_cached(cache, T, lambda y: y[0][:, :B * T].reshape(-1, B, T).permute(1, 0, 2))
            h = self.ff["0"](x, act=L.ACT_RELU)
            return self.ff["3"](h, res=x)
        res = x
        if last_only:
            B, T = batch
            # the last position of every utterance, read in place at column stride T (hsp_conv1d_args.res_ts)
            res = x[0][:, :B * T].reshape(-1, B, T)[:, :, T - 1].unsqueeze(0)
        x = self.attn(x, mask=mask, res=res, batch=batch, last_only=last_only)   # norm1 fused into q/k/v
        h = self.ff["0"](x, act=L.ACT_RELU)                                       # norm2 fused into ff.0
        return self.ff["3"](h, res=x)


class TransformerEncoder(nn.Module):
    """transformer_mega.TransformerEncoder (:135-163)."""

    def __init__(self, encoder_layer: TransformerEncoderLayer, num_layers: int, norm=None):
        super().__init__()
        e = encoder_layer
        self.layers = nn.ModuleList([e] + [TransformerEncoderLayer(e.dim, e.ff_dim, e.conv_ff, e.n_heads)
                                           for _ in range(num_layers - 1)])
        self.num_layers, self.norm = num_layers, norm

    def forward(self, x, x_lens=None, causal=False, batch=None, last_only=False, cache=None):
        if x_lens is not None or causal:
            raise NotImplementedError("length / causal masks belong to the training forward, not to infer()")
        for i, layer in enumerate(self.layers):
            x = layer(x, batch=batch, last_only=last_only and i == self.num_layers - 1,
                      cache=cache if i == 0 and self.num_layers > 1 else None)
        if self.norm is not None:
            x = self.norm(x)
        return x
