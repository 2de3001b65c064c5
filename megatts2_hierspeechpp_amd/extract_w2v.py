"""wav2vec2 feature producer of the voice-conversion harness with the reference's call surface (reference:
extract_w2v.py:16-46, used at inference_vc.py:85-86): ``Wav2vec2(layer=7)``, ``forward(x [B, 1, t]) -> [B, 1024, T]`` =
``hidden_states[7]`` of HF ``Wav2Vec2ForPreTraining`` (MMS-300M: third-party ``transformers``, pinned 4.34.0 upstream;
its modules are restated here on libhsp kernels, class by class, with HF's parameter names so that the reference's
checkpoint keys ``wav2vec2.wav2vec2.*`` load unchanged):

  feature encoder   7 x [Conv1d (k 10/3/3/3/3/2/2, stride 5/2/2/2/2/2/2, bias) -> LayerNorm over channels -> GELU]
                    (Wav2Vec2LayerNormConvLayer); strided convs run as polyphase unit-stride MFMA launches
  projection        LayerNorm(512) -> Linear(512 -> 1024)                         (Wav2Vec2FeatureProjection)
  positions         x + GELU(weight-normed grouped Conv1d k 128, 16 groups)[..., :-1]   (Wav2Vec2PositionalConvEmbedding)
  encoder           pre-LN layers 0 .. layer-1: x += out_proj(MHA(LN x));  x += W2 GELU(W1 LN x)
                    (Wav2Vec2EncoderLayerStableLayerNorm); hidden_states[layer] is the stream BEFORE layer `layer`,
                    so no final LayerNorm and no layer >= `layer` is ever run -- their checkpoint keys (and the
                    quantizer / projection heads of the pre-training model) are skipped on load.

No attention mask: the harness feeds one utterance (B utterances of equal length work the same way)."""
from __future__ import annotations

import torch
from torch import nn

from . import _lib as L
from . import functional as Fh
from .hip_layers import Conv1d, GroupedPosConv1d, LinearCT, PolyphaseConv1d, StackedLinearCT, entry as _entry, finalize as _finalize
from .ttv_v1.transformer_mega import LayerNorm

CONV_KERNEL = (10, 3, 3, 3, 3, 2, 2)
CONV_STRIDE = (5, 2, 2, 2, 2, 2, 2)


class _ConvLayer(nn.Module):
    """HF Wav2Vec2LayerNormConvLayer."""

    def __init__(self, cin, cout, k, stride):
        super().__init__()
        self.conv = Conv1d(cin, cout, k, stride=stride) if cin < 8 else PolyphaseConv1d(cin, cout, k, stride)
        self.layer_norm = LayerNorm(cout, eps=1e-5)

    def forward(self, x):
        return Fh.act(self.layer_norm(self.conv(x)), L.ACT_GELU_ERF)


class _FeatureEncoder(nn.Module):
    def __init__(self, dim=512):
        super().__init__()
        self.conv_layers = nn.ModuleList([_ConvLayer(1 if i == 0 else dim, dim, k, s)
                                          for i, (k, s) in enumerate(zip(CONV_KERNEL, CONV_STRIDE))])

    def forward(self, x):
        for layer in self.conv_layers:
            x = layer(x)
        return x


class _FeatureProjection(nn.Module):
    def __init__(self, cin=512, hidden=1024):
        super().__init__()
        self.layer_norm = LayerNorm(cin, eps=1e-5)
        self.projection = LinearCT(cin, hidden)
        self.projection.fuse_input_layernorm(self.layer_norm)   # statistics taken inside the GEMM's staged tile

    def forward(self, x):
        return self.projection(x)


class _PosConvEmbed(nn.Module):
    def __init__(self, hidden=1024, k=128, groups=16):
        super().__init__()
        self.conv = GroupedPosConv1d(hidden, k, groups)


class _Attention(nn.Module):
    """HF Wav2Vec2Attention (eager): 16 heads x 64, scores scaled by 64^-0.5, no mask."""

    def __init__(self, hidden=1024, heads=16):
        super().__init__()
        self.heads, self.scale = heads, (hidden // heads) ** -0.5
        self.q_proj = LinearCT(hidden, hidden, packed=False)
        self.k_proj = LinearCT(hidden, hidden, packed=False)
        self.v_proj = LinearCT(hidden, hidden, packed=False)
        self.out_proj = LinearCT(hidden, hidden)
        self.qkv = StackedLinearCT([self.q_proj, self.k_proj, self.v_proj])


class _FeedForward(nn.Module):
    def __init__(self, hidden=1024, inter=4096):
        super().__init__()
        self.intermediate_dense = LinearCT(hidden, inter)
        self.output_dense = LinearCT(inter, hidden)


class _EncoderLayer(nn.Module):
    """HF Wav2Vec2EncoderLayerStableLayerNorm."""

    def __init__(self, hidden=1024, heads=16, inter=4096):
        super().__init__()
        self.attention = _Attention(hidden, heads)
        self.layer_norm = LayerNorm(hidden, eps=1e-5)
        self.feed_forward = _FeedForward(hidden, inter)
        self.final_layer_norm = LayerNorm(hidden, eps=1e-5)
        self.attention.qkv.fuse_input_layernorm(self.layer_norm)
        self.feed_forward.intermediate_dense.fuse_input_layernorm(self.final_layer_norm)

    def forward(self, x):
        C = x.shape[1]
        qkv = self.attention.qkv(x)                       # LayerNorm fused into the stacked q / k / v GEMM
        o = Fh.mha(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], self.attention.heads, self.attention.scale)
        x = self.attention.out_proj(o, res=x)
        h = self.feed_forward.intermediate_dense(x, act=L.ACT_GELU_ERF)
        return self.feed_forward.output_dense(h, res=x)


class _Encoder(nn.Module):
    def __init__(self, n_layers, hidden=1024):
        super().__init__()
        self.pos_conv_embed = _PosConvEmbed(hidden)
        self.layers = nn.ModuleList([_EncoderLayer(hidden) for _ in range(n_layers)])


class _Model(nn.Module):
    """HF Wav2Vec2Model, the members hidden_states[layer] depends on."""

    def __init__(self, n_layers):
        super().__init__()
        self.feature_extractor = _FeatureEncoder()
        self.feature_projection = _FeatureProjection()
        self.encoder = _Encoder(n_layers)


class _ForPreTraining(nn.Module):
    def __init__(self, n_layers):
        super().__init__()
        self.wav2vec2 = _Model(n_layers)


class Wav2vec2(nn.Module):
    """extract_w2v.Wav2vec2 (:16-46): ``forward(x [B, 1, t] or [B, t]) -> [B, 1024, T]``."""

    def __init__(self, layer: int = 7, w2v: str = "mms"):
        super().__init__()
        self.feature_layer = layer
        self.wav2vec2 = _ForPreTraining(layer)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """Takes the reference module's state dict (HF key names).  Keys of parts hidden_states[layer] does not depend
        on (encoder layers >= layer, the final encoder LayerNorm, masking / quantizer / projection heads) are skipped;
        torch >= 2.1 weight-norm parametrization names are mapped to weight_g / weight_v."""
        keep = {}
        mine = set(super().state_dict().keys())
        for k, v in state_dict.items():
            k2 = k.replace("conv.parametrizations.weight.original0", "conv.weight_g") \
                  .replace("conv.parametrizations.weight.original1", "conv.weight_v")
            if k2 in mine:
                keep[k2] = v
        return super().load_state_dict(keep, strict=strict, **kw)

    def finalize(self, device, materialize: bool = True):
        self.arena = _finalize(self, device, materialize)
        return self

    @_entry
    @torch.no_grad()
    def forward(self, x):
        m = self.wav2vec2.wav2vec2
        if x.dim() == 2:
            x = x.unsqueeze(1)
        h = m.feature_projection(m.feature_extractor(x))
        h = m.encoder.pos_conv_embed.conv(h)              # h + GELU(pos_conv(h))
        for layer in m.encoder.layers:
            h = layer(h)
        return h
