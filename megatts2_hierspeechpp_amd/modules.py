"""WN, DiT-style coupling layers and Flip with the reference's class names, constructor
arguments and state-dict keys (reference: modules.py).  Every tensor stays channel-major
(B, C, T); the reference's transposes to (B, T, C) around the DiT blocks are layout
changes only.  All arithmetic is in libhsp.so."""
from __future__ import annotations

import contextlib
import ctypes as C
import os

import torch
from torch import nn

from . import _lib as L
from . import functional as Fh
from . import hip_layers
from .hip_layers import Conv1d, HipLayer, Linear

LRELU_SLOPE = 0.1  # modules.py:17
# (round 6, VERDICT r05 item 1b: shorter dependent launch chains in the 50 Hz part)
FOLD_MASK = os.environ.get("HSP_FOLD_MASK", "1") == "1"   # WN: the final mask multiply inside the skip launches
FOLD_FLIP = os.environ.get("HSP_FOLD_FLIP", "1") == "1"   # coupling blocks: Flip inside the next layer's packed pre / post
# DiT blocks: norm1 + modulate inside the qkv GEMM (hsp_conv1d_args.ln_scale) where the caller supplies the per-utterance
# c1 / bias vectors (`modq`: extra rows of the stacked adaLN GEMM, hip_layers.ModulatedNormRows) -- 96 LayerNorm launches
# per 32-utterance step fewer.  MEASURED AND NOT KEPT AS THE DEFAULT (round 6): same-box 59.07 ms per step
# with the LayerNorm launched on its own against 59.17 with the fold (profiles/r06_ab_fold_ln.json) -- the launches it
# removes were worth at most 0.38 ms (tools/front_probe.py), and the GEMM pays for the factor it multiplies into every
# staged fragment and the adaLN GEMM for twice the rows.  Read when a model is BUILT (the extra rows are stacked then) and
# at every call; HSP_FOLD_LN=1 switches it on, tests/test_gpu_parity.py::test_modulated_layernorm_inside_the_qkv_gemm runs it.
FOLD_LN = os.environ.get("HSP_FOLD_LN", "0") == "1"

def _fuse(x) -> bool:
    """Route a WN layer / DiT FFN through the entry points SURVEY.md 8(b) names for them (hsp_wn_layer_f32,
    hsp_ffn_conv_f32) -- only under HSP_SURVEY_ABI: they issue the same launches the default path issues itself.
    (Rounds 2-3 had a one-launch kernel behind them, csrc/hsp_gemm2.hip, chosen by a tile-count policy; it lost to the
    separate launches at every batch grouping of the step and was retired in round 4:
    profiles/r04_stage_split_policies.txt.)"""
    return hip_layers.SURVEY_ABI


class LayerNorm(HipLayer):
    """modules.LayerNorm (modules.py:19-31): affine LayerNorm over the channel axis of [B, C, T]."""

    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.channels, self.eps = channels, eps
        self.gamma = nn.Parameter(torch.ones(channels), requires_grad=False)
        self.beta = nn.Parameter(torch.zeros(channels), requires_grad=False)
        self._g = self._b = None

    def hsp_requests(self):
        return [("g", self.channels), ("b", self.channels)]

    def hsp_fill(self, arena, materialize):
        self._g, self._b = arena.view(self, "g"), arena.view(self, "b")
        if materialize:
            self._g.copy_(self.gamma.data)
            self._b.copy_(self.beta.data)

    def forward(self, x):
        if self._g is None:
            raise L.HspError("LayerNorm used before finalize()")
        return Fh.layernorm_mod(x, self.eps, gamma=self._g, beta=self._b)


class WN(nn.Module):
    """modules.WN (modules.py:111-176).  Two launches per layer: the gated in-conv (conv k + conditioning bias +
    tanh * sigmoid in the epilogue) and one token-GEMM launch writing both row halves of res_skip (residual update of
    x, accumulation of the skip output; hsp_conv1d_args.split_row)."""

    def __init__(self, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0, p_dropout=0):
        super().__init__()
        assert kernel_size % 2 == 1
        assert hidden_channels % 32 == 0, "gated MFMA epilogue needs hidden_channels % 32 == 0"
        self.hidden_channels, self.kernel_size = hidden_channels, kernel_size
        self.dilation_rate, self.n_layers, self.gin_channels = dilation_rate, n_layers, gin_channels
        self.in_layers = nn.ModuleList()
        self.res_skip_layers = nn.ModuleList()
        if gin_channels != 0:
            self.cond_layer = Conv1d(gin_channels, 2 * hidden_channels * n_layers, 1, weight_norm=True)
        for i in range(n_layers):
            d = dilation_rate ** i
            if (kernel_size - 1) * d + 3 > 125:   # include/hsp.h: the wide-pitch gated tile (S64GW) holds a 125-column halo
                raise L.HspError(f"WN layer {i}: halo (k - 1) * dilation + 3 = {(kernel_size - 1) * d + 3} columns exceeds "
                                 "the widest window of the gated conv kernel (125)")
            self.in_layers.append(Conv1d(hidden_channels, 2 * hidden_channels, kernel_size, dilation=d,
                                         padding=int((kernel_size * d - d) / 2), weight_norm=True,
                                         rows=L.ROWS_GATE_WN))
            rs = 2 * hidden_channels if i < n_layers - 1 else hidden_channels
            self.res_skip_layers.append(Conv1d(hidden_channels, rs, 1, weight_norm=True))

    def forward(self, x, x_mask, g=None, **kwargs):
        """`acts` (the gated activations) is a workspace between a layer's two launches.  x is never updated in place."""
        H = self.hidden_channels
        gc = self.cond_layer(g) if g is not None else None  # [B, 2H*n, 1]
        out = None
        acts = torch.empty(x.shape[0], H, x.shape[2], dtype=torch.float32, device=x.device)
        fuse = _fuse(x)
        # The final `output * x_mask` (modules.py:176) is applied where each skip contribution is produced: a 0/1 mask
        # times every term of the sum IS the mask times the sum, bit for bit (mask(s_1 + ... + s_n): the same additions on
        # the kept columns, exact zeros on the others) -- one launch per WN less (round 6; HSP_FOLD_MASK=0: the old form)
        skip_mask = L.MASK_PRE if (FOLD_MASK and x_mask is not None) else L.MASK_NONE
        for i in range(self.n_layers):
            cb = gc[:, 2 * H * i: 2 * H * (i + 1)] if gc is not None else None
            last = i == self.n_layers - 1
            if not fuse and not last and H % 64 == 0:
                # layer by layer: gated conv, then BOTH halves of res_skip in one token-GEMM launch
                # (hsp_conv1d_args.split_row; None = the library has no such kernel for this shape)
                self.in_layers[i](x, cbias=cb, out=acts)
                both = self.res_skip_layers[i](acts, res=x, mask=x_mask, mask_mode=L.MASK_POST,
                                               split_out=(H, out, out is not None), mask_mode2=skip_mask)
                if both is not None:
                    x, out = both
                    continue
                x_new = self.res_skip_layers[i](acts, row_range=(0, H), res=x, mask=x_mask, mask_mode=L.MASK_POST)
                out = self.res_skip_layers[i](acts, row_range=(H, 2 * H), out=out, accumulate=out is not None,
                                              mask=x_mask, mask_mode=skip_mask)
                x = x_new
                continue
            with (hip_layers.deferred() if fuse else contextlib.nullcontext()) as args:
                self.in_layers[i](x, cbias=cb, out=acts)
                if not last:
                    x_new = self.res_skip_layers[i](acts, row_range=(0, H), res=x, mask=x_mask, mask_mode=L.MASK_POST)
                    out = self.res_skip_layers[i](acts, row_range=(H, 2 * H), out=out, accumulate=out is not None,
                                                  mask=x_mask, mask_mode=skip_mask)
                else:
                    # last layer: output = output + rs (modules.py:175-176)
                    out = self.res_skip_layers[i](acts, out=out, accumulate=out is not None, mask=x_mask,
                                                  mask_mode=skip_mask)
            if fuse:
                hip_layers.launch_group("hsp_wn_layer_f32", L.lib().hsp_wn_layer_f32,
                                        [args[0], args[1] if not last else None, args[-1]])
            if not last:
                x = x_new
        return out if skip_mask != L.MASK_NONE else Fh.mask_mul(out, x_mask)


class Flip(nn.Module):
    """modules.Flip (modules.py:270-277)."""

    def forward(self, x, *args, reverse=False, **kwargs):
        y = Fh.flip_channels(x)
        if not reverse:
            raise NotImplementedError("training direction (logdet) is out of scope")
        return y


class Attention(nn.Module):
    """timm==0.6.13 vision_transformer.Attention (third party; imported at modules.py:13)
    on channel-major tensors: qkv / proj keep nn.Linear's 2-D weights."""

    def __init__(self, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = Conv1d(dim, dim * 3, 1, bias=qkv_bias, weight_2d=True)
        self.proj = Conv1d(dim, dim, 1, weight_2d=True)
        self.proj.keep_rowmajor_weight()   # A operand of the projection inside hsp_mha_proj_f32


class FFN_Conv(nn.Module):
    """modules.FFN_Conv (modules.py:357-388)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, kernel=5, p_dropout=0.1, **_):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = Conv1d(in_features, hidden_features, kernel, padding=(kernel - 1) // 2)
        self.fc2 = Conv1d(hidden_features, out_features, 1)


class DiTConVBlock(nn.Module):
    """modules.DiTConVBlock (modules.py:390-411).  x [B, C, T], c [B, C], x_mask [B, 1, T]
    (the reference passes x as [B, T, C] and x_mask as [B, T, 1]; same values)."""

    def __init__(self, hidden_size, num_heads, mlp_ratio=4.0, kernel=9, p_dropout=0.1, **block_kwargs):
        super().__init__()
        self.hidden_size = hidden_size
        self.attn = Attention(hidden_size, num_heads=num_heads, qkv_bias=True)
        self.mlp = FFN_Conv(hidden_size, int(hidden_size * mlp_ratio), kernel=kernel, p_dropout=p_dropout)
        # nn.Sequential(SiLU, Linear) in the reference: index 1 carries the parameters
        self.adaLN_modulation = nn.ModuleList([nn.Identity(), Linear(hidden_size, 6 * hidden_size)])

    def forward(self, x, c, x_mask, c_silu=None, mod=None, premasked=False, modq=None):
        """``c_silu`` = SiLU(c) precomputed by the caller (the same for every block of a flow); ``mod`` =
        this block's adaLN_modulation output [B, 6C, 1] when the caller ran all blocks' Linears as one GEMM;
        ``premasked``: x is already zero outside the mask (the output of a masked launch), so the leading
        ``x * x_mask`` of modules.py:407 is the identity and is not launched."""
        C = self.hidden_size
        if not premasked:
            x = Fh.mask_mul(x, x_mask)
        if mod is not None:
            pass
        elif c_silu is not None:
            mod = self.adaLN_modulation[1](c_silu)           # [B, 6C, 1]
        else:
            mod = self.adaLN_modulation[1](c, silu_in=True)  # [B, 6C, 1]
        sh_a, sc_a, g_a, sh_m, sc_m, g_m = (mod[:, i * C:(i + 1) * C, 0] for i in range(6))
        qkv = None
        if modq is not None and FOLD_LN:
            # ``modq`` [B, 6C, 1]: (c1_b | bias_b) of the qkv layer for this block (ModulatedNormRows): LayerNorm, mask and
            # modulate run inside the qkv GEMM on the un-normalised x; None = no kernel for this shape
            qkv = self.attn.qkv(x, ln_mod=(sc_a, modq[:, :3 * C, 0], modq[:, 3 * C:6 * C, 0], x_mask, 1e-6))
        if qkv is None:
            h = Fh.layernorm_mod(x, 1e-6, mask=x_mask, shift=sh_a, scale=sc_a)
            qkv = self.attn.qkv(h)
        if Fh.mha_proj_supported(self.attn.num_heads, C // self.attn.num_heads, C, x.shape[2]):
            # attention + proj + `x + gate_msa * (.) * mask` in ONE launch (round 4, csrc/hsp_mhaproj.hip)
            x = Fh.mha_proj(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], self.attn.num_heads, self.attn.scale,
                            self.attn.proj._wt, bias=self.attn.proj._b, mask=x_mask, cscale=g_a, res=x)
        else:
            o = Fh.mha(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], self.attn.num_heads, self.attn.scale)
            x = self.attn.proj(o, mask=x_mask, mask_mode=L.MASK_PRE, cscale=g_a, res=x)
        h = Fh.layernorm_mod(x, 1e-6, shift=sh_m, scale=sc_m)
        # fc1 -> GELU -> fc2; fc2(y * mask) * mask == (W y + b) * mask for a 1x1 conv and a 0/1 mask
        fuse = _fuse(x)
        with (hip_layers.deferred() if fuse else contextlib.nullcontext()) as args:
            y = self.mlp.fc1(h, act=L.ACT_GELU_TANH)
            out = self.mlp.fc2(y, mask=x_mask, mask_mode=L.MASK_PRE, cscale=g_m, res=x)
        if fuse:
            hip_layers.launch_group("hsp_ffn_conv_f32", L.lib().hsp_ffn_conv_f32, args)
        return out


class ResidualCouplingLayer_Transformer_simple(nn.Module):
    """modules.ResidualCouplingLayer_Transformer_simple (modules.py:413-488), reverse
    direction, mean_only."""

    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=0.1,
                 mean_only=False):
        super().__init__()
        assert channels % 2 == 0, "channels should be divisible by 2"
        if not mean_only:
            raise NotImplementedError("only mean_only=True couplings exist on the hot path")
        self.channels, self.hidden_channels, self.half_channels = channels, hidden_channels, channels // 2
        self.pre = Conv1d(self.half_channels, hidden_channels, 1)
        self.enc_block = nn.ModuleList([DiTConVBlock(hidden_channels, 2, mlp_ratio=4.0, kernel=5, p_dropout=p_dropout)
                                        for _ in range(n_layers)])
        self.post = Conv1d(hidden_channels, self.half_channels, 1)
        self.flipped = False

    def set_flipped(self, flipped: bool):
        """``flipped``: this layer is called on a tensor whose channel axis is stored REVERSED -- the Flip in front of it
        (modules.py:270-277; hierspeechpp_speechsynthesizer.py:80-86 runs `Flip, coupling` pairs in reverse) was not
        launched.  flip(x)[:, :half] is x[:, half:] reversed and flip(x)[:, half:] is x[:, :half] reversed, so the layer
        reads x[:, half:] through a `pre` packed with reversed input columns and updates x[:, :half] through a `post`
        packed with reversed rows: flip(result) is what the reference's layer returns on flip(x).  Same products, the
        input-channel sum of `pre` runs in the opposite order (not bit-identical, equally accurate)."""
        self.flipped = bool(flipped)
        self.pre.pack_flipped(inputs=self.flipped)
        self.post.pack_flipped(outputs=self.flipped)

    def forward(self, x, x_mask, g=None, reverse=False, inplace=False, c_silu=None, mods=None, modq=None):
        """``mods`` [B, n_layers * 6 * hidden, 1]: the adaLN outputs of this layer's blocks, stacked; ``modq`` [B, n_layers *
        6 * hidden, 1]: their qkv layers' (c1_b | bias_b) rows (DiTConVBlock.forward), or None."""
        if not reverse:
            raise NotImplementedError("training direction (logdet) is out of scope")
        half = self.half_channels
        rd, wr = (slice(half, None), slice(0, half)) if self.flipped else (slice(0, half), slice(half, None))
        h = self.pre(x[:, rd], mask=x_mask, mask_mode=L.MASK_PRE)
        R = 6 * self.hidden_channels
        for j, blk in enumerate(self.enc_block):
            # `pre` and every block's last launch multiply by the mask before the residual add: h stays masked
            h = blk(h, g, x_mask, c_silu=c_silu, mod=None if mods is None else mods[:, j * R:(j + 1) * R],
                    premasked=True, modq=None if modq is None else modq[:, j * R:(j + 1) * R])
        out = x if inplace else x.clone()
        # x1 <- (x1 - post(h) * mask) * mask          (modules.py:473,486)
        self.post(h, mask=x_mask, mask_mode=L.MASK_BOTH, scale=-1.0, res=x[:, wr], out=out[:, wr])
        return out
