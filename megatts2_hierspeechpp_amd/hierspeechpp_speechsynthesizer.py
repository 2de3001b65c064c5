"""HierSpeech++ synthesizer with the reference's call surface (reference:
hierspeechpp_speechsynthesizer.py): same class names, constructor arguments, method
signatures and state-dict keys, so ``inference_plm.py``-style harnesses and reference
checkpoints work unchanged -- but every tensor op is a launch into libhsp.so.

Typical use::

    net_g = SynthesizerTrn(spec_channels, segment_size, **hps.model)
    net_g.load_state_dict(checkpoint)        # reference checkpoint or synth weights
    net_g.finalize("cuda:0")                 # fold weight-norm, pack, upload (once)
    o, e_ = net_g.infer(mel, w2v, length, f0)

Training-only sub-modules of the reference (``enc_p``, ``enc_q``, ``mel_decoder``,
discriminators) are not instantiated; their checkpoint keys are skipped on load.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
from torch import nn

from . import _lib as L
from . import commons
from . import functional as Fh
from . import hip_layers
from . import activations
from . import modules
from .alias_free_torch import Activation1d
from .commons import get_padding
from .hip_layers import Conv1d, ConvTranspose1d, Linear, ModulatedNormRows, StackedLinearCT, entry as _entry, finalize as _finalize
from .styleencoder import StyleEncoder

UNUSED_PREFIXES = ("enc_p.", "enc_q.", "mel_decoder.", "emb.")  # training / analysis only

# Activation1d placement.  The anti-aliased activation can run as the LDS prologue of the
# following conv (no HBM round trip) or as its own HBM-bound launch in front of a plain conv.
# On MI355X the fp32 MFMA shares the SIMD's fp32 datapath with the VALU, so prologue arithmetic
# is paid in MFMA time; layers with more input channels than this run the activation unfused.
# Measured again with every later kernel set, always the same order: round 4 (DESIGN.md §5.4
# item 23, wave-per-segment activation kernel at 4.8 TB/s, frequency-domain k = 7 / 11 convs):
# 80.5 / 84.1 / 78.3 ms per 32 x 4 s step for thresholds 32 / 64 / 0 (round 1, DESIGN.md §5
# item 6: 102.5 / 104.4 / 101.3).  So the default is 0 -- the direct k = 3 convs read an
# activation tensor written by its own launch, the frequency-domain convs fuse theirs into the
# forward transform's staging (hsp_dftseg_args.act_*) -- and the fused conv prologue stays
# available (and tested) behind this knob.
FUSE_ACT_MAX_CHANNELS = int(os.environ.get("HSP_FUSE_ACT_MAX_C", "0"))


class ResidualCouplingBlock_Transformer(nn.Module):
    """hierspeechpp_speechsynthesizer.py:53-88 (reverse direction)."""

    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers=3, n_flows=4, gin_channels=0):
        super().__init__()
        self.channels, self.hidden_channels, self.n_flows = channels, hidden_channels, n_flows
        # nn.Sequential(Linear, SiLU, Linear): parameters at indices 0 and 2
        self.cond_block = nn.ModuleList([Linear(gin_channels, 4 * hidden_channels), nn.Identity(),
                                         Linear(4 * hidden_channels, hidden_channels)])
        self.flows = nn.ModuleList()
        for _ in range(n_flows):
            self.flows.append(modules.ResidualCouplingLayer_Transformer_simple(
                channels, hidden_channels, kernel_size, dilation_rate, n_layers, mean_only=True))
            self.flows.append(modules.Flip())
        # The adaLN_modulation Linears of all n_flows * n_layers DiT blocks read the same SiLU(c): their rows are
        # stacked into ONE GEMM per forward (12 launches -> 1); the blocks keep the parameters (checkpoint keys).
        lins = [blk.adaLN_modulation[1] for i in range(n_flows) for blk in self.flows[2 * i].enc_block]
        for lin in lins:
            lin.__dict__["_stacked_elsewhere"] = True
        self._mod_rows = lins[0].cout * n_layers   # rows per coupling layer
        # (round 6) ... and, behind them, per block the (c1_b | bias_b) rows its qkv layer needs to run norm1 + modulate
        # inside its own GEMM (hip_layers.ModulatedNormRows: 6 * hidden rows per block as well)
        self._modq_base = None
        if modules.FOLD_LN:
            blocks = [blk for i in range(n_flows) for blk in self.flows[2 * i].enc_block]
            self._modq_base = sum(l.cout for l in lins)
            lins = lins + [ModulatedNormRows(blk.attn.qkv, blk.adaLN_modulation[1]) for blk in blocks]
        self.adaln_all = StackedLinearCT(lins)
        # (round 6)  The Flips cost no launch: reverse runs `Flip, coupling` for i = n_flows - 1 ... 0, so the tensor
        # reaches coupling i after n_flows - i flips; the layers that see an odd number of them work on the reversed
        # channel axis through their packed weights (set_flipped).  n_flows odd leaves one real Flip at the end.
        if modules.FOLD_FLIP:
            for i in range(n_flows):
                self.flows[2 * i].set_flipped((n_flows - i) % 2 == 1)

    def forward(self, x, x_mask, g=None, reverse=False, owned=False):
        """``owned``: the caller hands x over (a temporary of its own): the first coupling layer may update it in place."""
        if not reverse:
            raise NotImplementedError("training direction is out of scope")
        c = self.cond_block[0](g.reshape(g.shape[0], -1), act=L.ACT_SILU)
        # every DiT block consumes SiLU(c) (adaLN_modulation = Sequential(SiLU, Linear)): apply it
        # once in the epilogue of the producing Linear instead of 12 times per flow
        c_silu = self.cond_block[2](c, act=L.ACT_SILU)   # [B, hidden, 1]
        mods = self.adaln_all(c_silu)                    # [B, n_flows * n_layers * 6 * hidden, 1]
        R = self._mod_rows
        flipped = False                                   # is the stored channel axis the reverse of the true one?
        for i in reversed(range(self.n_flows)):
            layer = self.flows[2 * i]
            if layer.flipped != (not flipped):
                x = self.flows[2 * i + 1](x, x_mask, reverse=True)                              # Flip -> fresh tensor
                owned = True
            else:
                flipped = not flipped                                                           # Flip, not launched
            q0 = self._modq_base
            x = layer(x, x_mask, g=None, mods=mods[:, i * R:(i + 1) * R], reverse=True, inplace=owned,
                      modq=None if q0 is None else mods[:, q0 + i * R:q0 + (i + 1) * R])                 # coupling
            owned = True
        return Fh.flip_channels(x) if flipped else x


class PosteriorSFEncoder(nn.Module):
    """hierspeechpp_speechsynthesizer.py:168-203."""

    def __init__(self, src_channels, out_channels, hidden_channels, kernel_size, dilation_rate, n_layers,
                 gin_channels=0):
        super().__init__()
        self.out_channels, self.hidden_channels = out_channels, hidden_channels
        self.pre_source = Conv1d(src_channels, hidden_channels, 1)
        self.pre_filter = Conv1d(1, hidden_channels, 9, stride=4, padding=4)
        self.source_enc = modules.WN(hidden_channels, kernel_size, dilation_rate, n_layers // 2, gin_channels=gin_channels)
        self.filter_enc = modules.WN(hidden_channels, kernel_size, dilation_rate, n_layers // 2, gin_channels=gin_channels)
        self.enc = modules.WN(hidden_channels, kernel_size, dilation_rate, n_layers // 2, gin_channels=gin_channels)
        self.proj = Conv1d(hidden_channels, out_channels * 2, 1)

    def stats(self, x_src, x_ftr, x_mask, g=None):
        x_src = self.pre_source(x_src, mask=x_mask, mask_mode=L.MASK_PRE)
        x_ftr = self.pre_filter(x_ftr, mask=x_mask, mask_mode=L.MASK_PRE)
        x_src = self.source_enc(x_src, x_mask, g=g)
        x_ftr = self.filter_enc(x_ftr, x_mask, g=g)
        x = self.enc(Fh.axpby(x_src, x_ftr, 1.0, 1.0), x_mask, g=g)
        return self.proj(x, mask=x_mask, mask_mode=L.MASK_PRE)  # [B, 2*out, T] = (m, logs)

    def forward(self, x_src, x_ftr, x_mask, g=None, noise=None):
        stats = self.stats(x_src, x_ftr, x_mask, g)
        C = self.out_channels
        if noise is None:
            noise = torch.randn(stats.shape[0], C, stats.shape[2], dtype=torch.float32, device=stats.device)
        z = Fh.sample_prior(stats, noise, x_mask, 1.0)
        return z, stats[:, :C], stats[:, C:]


class AMPBlock1(nn.Module):
    """hierspeechpp_speechsynthesizer.py:344-386.  Six launches: each conv carries its
    Activation1d as prologue; the second conv of every pair adds the residual, and the
    last one also folds the mean over the parallel blocks of the stage."""

    def __init__(self, channels, kernel_size=3, dilation=(1, 3, 5), activation=None):
        super().__init__()
        mk = lambda d: Conv1d(channels, channels, kernel_size, dilation=d, padding=get_padding(kernel_size, d),
                              weight_norm=True)
        self.convs1 = nn.ModuleList([mk(d) for d in dilation])
        self.convs2 = nn.ModuleList([mk(1) for _ in dilation])
        # (round 4) long kernels over many channels also carry their frequency-domain form (hip_layers.Conv1d.enable_fft:
        # overlap-save with a 128-point DFT, 2.2 multiply-adds per output and channel pair instead of k); forward()
        # takes it where it is the faster one (fft_wins)
        for c in list(self.convs1) + list(self.convs2):
            if fft_eligible(channels, kernel_size, c.dilation):
                c.enable_fft()
        self.num_layers = len(self.convs1) + len(self.convs2)
        self.activations = nn.ModuleList([
            Activation1d(activation=activations.SnakeBeta(channels, alpha_logscale=True)) for _ in range(self.num_layers)])

    def forward(self, x, *, out=None, accumulate=False, post_scale=1.0, before_last=None):
        """``before_last``: an event the current stream waits on before the last launch (the one
        that accumulates into the shared ``out`` of the stage)."""
        n = len(self.convs1)
        if FFT_THROUGH and fft_chain_ok(self, x):
            return amp_block_fft_chain(self, x, out=out, accumulate=accumulate, post_scale=post_scale, before_last=before_last)
        for i, (c1, c2) in enumerate(zip(self.convs1, self.convs2)):
            last = i == n - 1
            x = amp_pair(c1, c2, self.activations[2 * i], self.activations[2 * i + 1], x,
                         before_last=before_last if last else None, res=x, out=out if last else None,
                         accumulate=accumulate and last, post_scale=post_scale if last else 1.0)
        return x


# (round 6)  A whole AMP block in the frequency domain as ONE chain of spectra: the seam between two
# iterations -- inverse of c2's product + bias + residual -> x_new, then a1'(x_new) -> forward for the next c1 -- is one
# launch too (hsp_dftseg_pair_f32, pass-through form: x_new is written once and never read back by a transform): 7 transform
# launches per block instead of 9.  MEASURED AND NOT KEPT AS THE DEFAULT: same-box 56.84 (per-iteration form, amp_pair) against
# 57.10 ms per 32 x 4 s step with the chain (profiles/r06_ab_fft_through.json) -- the pair kernel runs its phases one after
# the other on one workgroup per CU, and the seam's residual read and x_new write join a phase that nothing overlaps, while
# the separate inverse and forward launches run two workgroups per CU each.  HSP_FFT_THROUGH=1 switches it on; the parity
# tests run both forms (tests/test_gpu_parity.py::test_amp_block_as_one_chain_of_spectra).
FFT_THROUGH = os.environ.get("HSP_FFT_THROUGH", "0") == "1"


def fft_chain_ok(block, x) -> bool:
    """Every conv of the block takes the frequency-domain form for an input like x, every pair and every seam fuses."""
    if not (FFT_PAIR and fft_act(x)) or x.shape[1] <= FUSE_ACT_MAX_CHANNELS:
        return False
    key = (x.shape[0], x.shape[2], x.stride(0), x.stride(1), x.data_ptr() & 15)
    memo = block.__dict__.setdefault("_chain_ok", {})
    ok = memo.get(key)
    if ok is None:
        n = len(block.convs1)
        ok = all(fft_wins(c, x) for c in list(block.convs1) + list(block.convs2)) and \
            all(block.convs1[i].fft_pair_ok(block.convs2[i], x) for i in range(n)) and \
            all(block.convs2[i].fft_through_ok(block.convs1[i + 1], x) for i in range(n - 1))
        if len(memo) >= 256:
            memo.clear()
        memo[key] = ok
        return ok
    # (fft_wins also guards the capture-before-first-use case: ask it again, it is cheap once the geometry is cached)
    return ok and all(fft_wins(c, x) for c in list(block.convs1) + list(block.convs2))


def amp_block_fft_chain(block, x, *, out=None, accumulate=False, post_scale=1.0, before_last=None):
    """AMPBlock1.forward (hierspeechpp_speechsynthesizer.py:377-386) with every conv in the frequency domain:
    forward(a1(x)) -> [product c1 -> inverse + a2 + forward -> product c2 -> inverse + residual + a1' + forward] x (n - 1)
    -> product c1 -> inverse + a2 + forward -> product c2 -> inverse with the block's epilogue."""
    n = len(block.convs1)
    B, Cc, Lx = x.shape
    acts = block.activations
    hook = hip_layers.LAUNCH_HOOK
    xf, e_first = block.convs1[0]._fft_forward(x, acts[0])
    for i in range(n):
        c1, c2 = block.convs1[i], block.convs2[i]
        xf2, _ = c1._fft_pair_launch(c2, c1._fft_product(xf), B, Lx, acts[2 * i + 1], x)
        yf2 = c2._fft_product(xf2)
        if i < n - 1:
            x_new = torch.empty_like(x)
            xf, _ = c2._fft_pair_launch(block.convs1[i + 1], yf2, B, Lx, acts[2 * i + 2], x, res=x, y=x_new)
            x = x_new
        else:
            out, e_last = c2._fft_inverse(yf2, B, Lx, res=x, out=out, accumulate=accumulate, post_scale=post_scale,
                                          before_inverse=before_last)
    if hook is not None:
        ks = sum(c.k for c in list(block.convs1) + list(block.convs2))
        nio = 2 * n + 1 + 2 * bool(accumulate)            # every iteration reads and writes x once more than a lone conv
        hook("hsp_fftconv", 2 * B * Cc * Cc * ks * Lx, 4 * B * Cc * Lx * nio + 4 * ks * Cc * Cc, e_first, e_last, 2 * n)
    return out


def amp_pair(c1, c2, a1, a2, x, *, form=None, before_last=None, **kw):
    """One iteration of AMPBlock1.forward (hierspeechpp_speechsynthesizer.py:380-384): c2(a2(c1(a1(x)))) with c2's epilogue
    ``kw`` (res, out, accumulate, post_scale).  ``form`` None = the policy below picks per conv (fft_wins) and per pair
    (fft_pair_ok); "direct" / "fft" / "pair" force one form -- tools/fftconv_table.py measures the policy against them.
    ``before_last``: an event the stream waits on before the launch that touches ``out``."""
    if c1.cin <= FUSE_ACT_MAX_CHANNELS and form is None:
        xt = c1(x, act1d=a1)
        if before_last is not None:
            torch.cuda.current_stream(x.device).wait_event(before_last)
        return c2(xt, act1d=a2, **kw)
    # a conv in its frequency-domain form applies its activation while the forward transform stages the input
    # (hsp_dftseg_args.act_*): no launch, no act(x) in HBM; two such convs in a row also meet in one launch (inverse of c1
    # + a2 + forward of c2, hsp_dftseg_pair_f32): xt never in HBM either
    w1 = fft_wins(c1, x) if form is None else form != "direct"
    w2 = fft_wins(c2, x) if form is None else form != "direct"
    if w1 and w2 and fft_act(x) and ((FFT_PAIR and form is None) or form == "pair") and c1.fft_pair_ok(c2, x):
        return c1.forward_fft_pair(c2, x, act_first=a1, act_second=a2, before_inverse=before_last, **kw)
    if form == "pair":
        raise L.HspError("amp_pair(form='pair'): hsp_dftseg_pair_supported says no for this geometry")
    if w1:
        xt = c1.forward_fft(x, act1d=a1) if fft_act(x) else c1.forward_fft(a1(x))
    else:
        xt = c1(a1(x))
    if before_last is not None and not w2:
        torch.cuda.current_stream(x.device).wait_event(before_last)
    if w2:
        return c2.forward_fft(xt, act1d=a2, before_inverse=before_last, **kw) if fft_act(xt) \
            else c2.forward_fft(a2(xt), before_inverse=before_last, **kw)
    return c2(a2(xt), **kw)


# Where the frequency-domain form of an AMP conv beats the direct MFMA conv: fft_eligible (by shape) and fft_min_cols (by
# samples per channel) below, both from tables measured on one box (tools/fftconv_bench.py; profiles/r04_fftconv_bench.txt,
# r04_fftconv_batch.txt).  The channel product shrinks 5.0 x (k = 11) / 3.3 x (k = 7); what the two transforms cost grows
# with the tensor, not with C^2, so the gain rises with the channel count.  HSP_FFT_CONV=0 switches the form off (A/B
# runs, parity tests of both forms).
FFT_CONV = os.environ.get("HSP_FFT_CONV", "1") == "1"


def fft_eligible(channels: int, k: int, dilation: int) -> bool:
    """Where the frequency-domain form wins at the Generator's shapes (profiles/r04_fftconv_bench.txt, 32 x 4 s): k = 11
    from 64 channels (1.05-1.18 x there, 1.7-2.5 x from 128), k = 7 from 128 (1.09-1.21 x; 1.2-1.6 x from 256); k = 3
    never (0.4-0.8 x) and k = 7 at 64 channels not (0.7-0.8 x)."""
    del dilation  # (every dilation of the blocks wins where dilation 1 does)
    return channels >= 64 if k >= 11 else (k >= 7 and channels >= 128)


def fft_min_cols(channels: int, k: int) -> int:
    """Samples per channel (batch x length) from which the frequency-domain form wins: its product launch reads one
    matrix set per bin (201 MB per conv at 512 channels) and there are three launches, a floor that a short single
    utterance does not always amortise.  From profiles/r05_fftconv_dispatch_table.txt (504 cells: B 1 ... 64 x T 50 / 200 /
    1000 x every (C, k, d) of the blocks, each AMP iteration as direct / frequency-domain / pair launch): with the
    three-product channel mix (round 5) the form wins from far fewer samples than with round 4's block product -- k = 11 at
    512 channels at ANY size (1.2-1.4 x on a single 1-s utterance), k = 7 there from 3 200; the dilation does not move
    the thresholds."""
    if FFT_MIN_COLS is not None:
        return FFT_MIN_COLS
    if k >= 11:
        return 0 if channels >= 512 else 8000 if channels >= 256 else 32000 if channels >= 128 else 128000
    return 3200 if channels >= 512 else 8000 if channels >= 256 else 64000


# HSP_FFT_MIN_COLS=<n> (or the module attribute) replaces the measured thresholds above; 0 forces the form on every
# eligible conv whatever the batch is -- the parity tests run the reference-generated fixtures through the frequency-domain
# kernels that way (tests/test_gpu_parity.py::test_golden_frequency_domain_forced).
FFT_MIN_COLS = int(os.environ["HSP_FFT_MIN_COLS"]) if os.environ.get("HSP_FFT_MIN_COLS") else None
_WARNED_CAPTURE = False


def fft_wins(conv, x) -> bool:
    """Does this call take the conv's frequency-domain form?  Eligible by shape (enable_fft), enough samples per channel
    for the form to pay (fft_min_cols), a geometry the transform kernels address (hsp_dftseg_supported: the reference has no
    batch or length limit, hierspeechpp_speechsynthesizer.py:377-386,635-651, so beyond it the direct conv runs), and --
    inside a stream capture -- per-bin matrices that already exist (they are derived on first eager use)."""
    if not (FFT_CONV and conv.__dict__.get("_fft") and x.stride(2) == 1):
        return False
    B, _, Lx = x.shape
    if B * Lx < fft_min_cols(conv.cin, conv.k):
        return False
    ok = conv.__dict__.setdefault("_fft_ok", {})
    sup = ok.get((B, Lx))
    if sup is None:
        if len(ok) >= 1024:          # a long-running process with ragged batches: the cache is a convenience, not a log
            ok.clear()
        sup = ok[(B, Lx)] = conv.fft_supported(B, Lx)
    if sup and (conv._wf is None or conv._wf_form != conv.fft_form()) and torch.cuda.is_current_stream_capturing():
        global _WARNED_CAPTURE
        if not _WARNED_CAPTURE:
            _WARNED_CAPTURE = True
            import warnings
            warnings.warn("a stream capture met a conv whose frequency-domain matrices were never derived: it takes the "
                          "direct form inside this graph (run one eager call or hip_layers.prepare_fft(model) first)")
        return False
    return sup


FFT_ACT = os.environ.get("HSP_FFT_ACT", "1") == "1"   # 0: the activation stays its own launch (A/B runs)
FFT_PAIR = os.environ.get("HSP_FFT_PAIR", "1") == "1"  # 0: inverse and forward transform between the convs of a pair stay two launches


def fft_act(x) -> bool:
    return FFT_ACT and hip_layers.fft_act_fusable(x)


# The parallel AMP blocks of a stage are independent chains of six launches each.  They are
# issued on separate HIP streams so that the GPU can co-schedule them: an HBM-bound activation
# launch of one chain runs under the MFMA-bound conv of another, and the partial last wave of
# a conv (e.g. 448 workgroups on 256 CUs at stage 1) is back-filled.  Only the last conv of each
# chain touches the shared accumulator; events serialise those three launches in block order,
# so the sum is formed in the reference's order ((b0 + b1) + b2) / 3.
AMP_STREAMS = int(os.environ.get("HSP_AMP_STREAMS", "1"))
# (round 6; SURVEY.md §7 "stage ordering for cache")  HSP_GEN_GROUPS = G > 1: the stages of the
# Generator with at most HSP_GEN_GROUP_MAX_C channels (default 128: L = 16 000 ... 64 000, 262-MB tensors at 32 x 4 s, just
# past the 256-MB Infinity Cache) run as G SEQUENTIAL utterance groups -- each group walks ups -> AMP stage -> ... ->
# conv_post before the next one starts, on the same three AMP streams -- so that a group's intermediates (65 MB at G = 4)
# can stay cache-resident between the launch that writes them and the one that reads them.  Same launches per utterance,
# bit-identical results (utterances are independent); measured in DESIGN.md §5.6.
GEN_GROUPS = int(os.environ.get("HSP_GEN_GROUPS", "1"))
GEN_GROUP_MAX_C = int(os.environ.get("HSP_GEN_GROUP_MAX_C", "128"))
FRONT_SPLITS = int(os.environ.get("HSP_FRONT_SPLITS", "4"))
# measurement mode (bench.py's per-launch pass, tools/pmc_traffic.sh): the SAME launches as the product step -- the front
# part still cut into FRONT_SPLITS batch groups, the AMP chains unchanged -- but issued one after the other on the
# current stream, so that an event pair or a profiler row brackets one kernel alone
SERIAL_STREAMS = os.environ.get("HSP_SERIAL_STREAMS", "0") == "1"
_SIDE_STREAMS = {}


def _side_streams(device, n):
    key = (device.type, device.index)
    pool = _SIDE_STREAMS.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def _amp_stage(resblocks, first, num_kernels, x):
    """xs = sum_j block_j(x); x = xs / num_kernels (hierspeechpp_speechsynthesizer.py:440-446)."""
    if not AMP_STREAMS or SERIAL_STREAMS or num_kernels == 1:
        xs = None
        for j in range(num_kernels):
            last = j == num_kernels - 1
            xs = resblocks[first + j](x, out=xs, accumulate=j > 0, post_scale=(1.0 / num_kernels) if last else 1.0)
        return xs
    main = torch.cuda.current_stream(x.device)
    side = _side_streams(x.device, num_kernels - 1)
    xs = torch.empty_like(x)
    fork = torch.cuda.Event()
    fork.record(main)
    done = [torch.cuda.Event() for _ in range(num_kernels)]
    for j in range(num_kernels):
        st = main if j == 0 else side[j - 1]
        last = j == num_kernels - 1
        with torch.cuda.stream(st):
            if j > 0:
                st.wait_event(fork)
            gate = done[j - 1] if j > 0 else None
            resblocks[first + j](x, out=xs, accumulate=j > 0, post_scale=(1.0 / num_kernels) if last else 1.0,
                                 before_last=gate)
            done[j].record(st)
    for ev in done[1:]:
        main.wait_event(ev)
    return xs


class DBlock(nn.Module):
    """hierspeechpp_speechsynthesizer.py:317-339.  F.interpolate(nearest, 1/factor) is a
    strided view of the input (the reference convolves at full rate, then drops 3/4)."""

    def __init__(self, input_size, hidden_size, factor):
        super().__init__()
        self.factor = factor
        self.residual_dense = Conv1d(input_size, hidden_size, 1, weight_norm=True)
        self.conv = nn.ModuleList([
            Conv1d(input_size, hidden_size, 3, dilation=1, padding=1, weight_norm=True),
            Conv1d(hidden_size, hidden_size, 3, dilation=2, padding=2, weight_norm=True),
            Conv1d(hidden_size, hidden_size, 3, dilation=4, padding=4, weight_norm=True)])

    def forward(self, x):
        if x.shape[-1] % self.factor != 0:
            raise L.HspError("DBlock needs a length divisible by its factor (pitch is 4 frames per w2v frame)")
        xs = x[..., ::self.factor]
        res = self.residual_dense(xs)
        h = self.conv[0](xs, lrelu=modules.LRELU_SLOPE)
        h = self.conv[1](h, lrelu=modules.LRELU_SLOPE)
        return self.conv[2](h, lrelu=modules.LRELU_SLOPE, res=res)


class SourceNetwork(nn.Module):
    """hierspeechpp_speechsynthesizer.py:251-308."""

    def __init__(self, upsample_initial_channel=256):
        super().__init__()
        ks, us, uks = [3, 5, 7], [2, 2], [4, 4]
        ds = [[1, 3, 5]] * 3
        self.num_kernels, self.num_upsamples = len(ks), len(us)
        c0 = upsample_initial_channel
        self.conv_pre = Conv1d(192, c0, 7, padding=3, weight_norm=True)
        self.ups = nn.ModuleList([ConvTranspose1d(c0 // 2 ** i, c0 // 2 ** (i + 1), k, u, padding=(k - u) // 2,
                                                  weight_norm=True) for i, (u, k) in enumerate(zip(us, uks))])
        self.resblocks = nn.ModuleList()
        for i in range(len(self.ups)):
            ch = c0 // 2 ** (i + 1)
            for k, d in zip(ks, ds):
                self.resblocks.append(AMPBlock1(ch, k, d, activation="snakebeta"))
        self.activation_post = Activation1d(activation=activations.SnakeBeta(ch, alpha_logscale=True))
        self.conv_post = Conv1d(ch, 1, 7, padding=3, bias=False)
        self.cond = Conv1d(256, c0, 1)

    def forward(self, x, g):
        x = self.conv_pre(x, cbias=self.cond(g))
        for i in range(self.num_upsamples):
            x = self.ups[i](x)
            x = _amp_stage(self.resblocks, i * self.num_kernels, self.num_kernels, x)
        x = self.activation_post(x)
        return x, self.conv_post(x)


class Generator(nn.Module):
    """hierspeechpp_speechsynthesizer.py:394-451."""

    def __init__(self, initial_channel, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, gin_channels=256):
        super().__init__()
        self.num_kernels, self.num_upsamples = len(resblock_kernel_sizes), len(upsample_rates)
        c0 = upsample_initial_channel
        self.conv_pre = Conv1d(initial_channel, c0, 7, padding=3, weight_norm=True)
        self.ups = nn.ModuleList([ConvTranspose1d(c0 // 2 ** i, c0 // 2 ** (i + 1), k, u, padding=(k - u) // 2,
                                                  weight_norm=True)
                                  for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes))])
        self.resblocks = nn.ModuleList()
        for i in range(len(self.ups)):
            ch = c0 // 2 ** (i + 1)
            for k, d in zip(resblock_kernel_sizes, resblock_dilation_sizes):
                self.resblocks.append(AMPBlock1(ch, k, d, activation="snakebeta"))
        self.activation_post = Activation1d(activation=activations.SnakeBeta(ch, alpha_logscale=True))
        self.conv_post = Conv1d(ch, 1, 7, padding=3, bias=False)
        if gin_channels != 0:
            self.cond = Conv1d(gin_channels, c0, 1)
        self.downs = DBlock(c0 // 8, c0, 4)
        self.proj = Conv1d(c0 // 8, c0 // 2, 7, padding=3)

    @_entry
    def forward(self, x, pitch, g=None):
        x = self.conv_pre(x, cbias=self.cond(g), res=self.downs(pitch))
        first_grouped = self.num_upsamples
        if GEN_GROUPS > 1 and x.shape[0] >= 2 * GEN_GROUPS:
            # the first stage whose channel count is at most GEN_GROUP_MAX_C: from there on the batch walks the rest of
            # the Generator as sequential utterance groups (SURVEY.md §7 "stage ordering for cache")
            chans = [self.ups[i].cout for i in range(self.num_upsamples)]
            first_grouped = next((i for i, c in enumerate(chans) if c <= GEN_GROUP_MAX_C), self.num_upsamples)
        for i in range(first_grouped):
            x = self.ups[i](x)
            if i == 0:
                x = self.proj(pitch, res=x, out=x)
            x = _amp_stage(self.resblocks, i * self.num_kernels, self.num_kernels, x)
        if first_grouped == self.num_upsamples:
            x = self.activation_post(x)
            return self.conv_post(x, act=L.ACT_TANH)
        B = x.shape[0]
        total_up = 1
        for i in range(first_grouped, self.num_upsamples):
            total_up *= self.ups[i].up
        out = torch.empty(B, 1, x.shape[2] * total_up, dtype=torch.float32, device=x.device)
        bounds = [(B * j) // GEN_GROUPS for j in range(GEN_GROUPS + 1)]
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            xg = x[lo:hi]
            for i in range(first_grouped, self.num_upsamples):
                xg = self.ups[i](xg)
                if i == 0:
                    xg = self.proj(pitch[lo:hi], res=xg, out=xg)
                xg = _amp_stage(self.resblocks, i * self.num_kernels, self.num_kernels, xg)
            xg = self.activation_post(xg)
            self.conv_post(xg, act=L.ACT_TANH, out=out[lo:hi])
        return out


class SynthesizerTrn(nn.Module):
    """hierspeechpp_speechsynthesizer.py:562-699 (inference methods)."""

    def __init__(self, spec_channels, segment_size, inter_channels, hidden_channels, filter_channels, n_heads,
                 n_layers, kernel_size, p_dropout, resblock, resblock_kernel_sizes, resblock_dilation_sizes,
                 upsample_rates, upsample_initial_channel, upsample_kernel_sizes, gin_channels=256, prosody_size=20,
                 uncond_ratio=0., cfg=False, **kwargs):
        super().__init__()
        self.spec_channels, self.segment_size = spec_channels, segment_size
        self.inter_channels, self.hidden_channels = inter_channels, hidden_channels
        self.upsample_rates = upsample_rates
        self.enc_p_l = PosteriorSFEncoder(1024, inter_channels, hidden_channels, 5, 1, 16, gin_channels=gin_channels)
        self.flow_l = ResidualCouplingBlock_Transformer(inter_channels, hidden_channels, 5, 1, 3, gin_channels=gin_channels)
        self.flow = ResidualCouplingBlock_Transformer(inter_channels, hidden_channels, 5, 1, 3, gin_channels=gin_channels)
        self.dec = Generator(inter_channels, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                             upsample_initial_channel, upsample_kernel_sizes, gin_channels=gin_channels)
        self.sn = SourceNetwork(upsample_initial_channel // 2)
        self.emb_g = StyleEncoder(in_dim=80, hidden_dim=256, out_dim=gin_channels)
        if cfg:
            # classifier-free null speaker embedding (:628-633): torch.nn.Embedding(1, 256), key "emb.weight"
            from .ttv_v1.t2w2v_transformer import Embedding
            self.emb = Embedding(1, 256)
            self.uncond_ratio = uncond_ratio
        self.cfg = cfg

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """Accepts the reference's checkpoints: bare state dict (inference_plm.py:218) or
        {'model': ...} (utils.py:19-44); keys of training-only sub-modules are skipped."""
        if "model" in state_dict and not any(k.startswith("dec.") for k in state_dict):
            state_dict = state_dict["model"]
        skip = tuple(p for p in UNUSED_PREFIXES if not (self.cfg and p == "emb."))
        sd = {k: v for k, v in state_dict.items() if not k.startswith(skip)}
        return super().load_state_dict(sd, strict=strict, **kw)

    def finalize(self, device, materialize: bool = True):
        """Fold weight-norm and pack all weights into one device arena (call once after
        loading weights; again after changing them)."""
        return _finalize(self, device, materialize)

    # ---------------------------------------------------------------- inference
    def _prior(self, w2v, f0, y_mask, g, noise, noise_scale):
        stats = self.enc_p_l.stats(w2v, f0, y_mask, g)
        C = self.inter_channels
        if noise is None:
            noise = torch.randn(stats.shape[0], C, stats.shape[2], dtype=torch.float32, device=stats.device)
        return Fh.sample_prior(stats, noise, y_mask, noise_scale)

    def _latent(self, w2v, f0, y_mask, g, noise, noise_scale):
        """prior sample -> both reversed flows.  Everything here works on 50 Hz frames (a few
        hundred columns per utterance): ~250 small, latency-bound launches.  Utterances are
        independent, so the batch is cut into FRONT_SPLITS groups issued on separate streams;
        their launches overlap on the GPU instead of each one draining the chip."""
        B = w2v.shape[0]
        n = min(FRONT_SPLITS, B)
        if n <= 1:
            z = self._prior(w2v, f0, y_mask, g, noise, noise_scale)
            return self.flow(self.flow_l(z, y_mask, g=g, reverse=True, owned=True), y_mask, g=g, reverse=True, owned=True)
        if noise is None:
            noise = torch.randn(B, self.inter_channels, w2v.shape[2], dtype=torch.float32, device=w2v.device)
        main = torch.cuda.current_stream(w2v.device)
        side = [main] * (n - 1) if SERIAL_STREAMS else _side_streams(w2v.device, n - 1)
        fork = torch.cuda.Event()
        fork.record(main)
        z = torch.empty(B, self.inter_channels, w2v.shape[2], dtype=torch.float32, device=w2v.device)
        bounds = [(B * i) // n for i in range(n + 1)]
        for i in range(n):
            lo, hi = bounds[i], bounds[i + 1]
            st = main if i == 0 else side[i - 1]
            with torch.cuda.stream(st):
                if i > 0 and st is not main:
                    st.wait_event(fork)
                zi = self._prior(w2v[lo:hi], f0[lo:hi], y_mask[lo:hi], g[lo:hi], noise[lo:hi], noise_scale)
                zi = self.flow_l(zi, y_mask[lo:hi], g=g[lo:hi], reverse=True, owned=True)
                zi = self.flow(zi, y_mask[lo:hi], g=g[lo:hi], reverse=True, owned=True)
                z[lo:hi].copy_(zi)
                if i > 0 and st is not main:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    main.wait_event(ev)
        return z

    def _decode(self, z, g):
        e, e_ = self.sn(z, g)
        return self.dec(z, e, g=g), e_

    @_entry
    @torch.no_grad()
    def infer(self, x_mel, w2v, length, f0, noise: Optional[torch.Tensor] = None):
        """:635-651 -> (o [B,1,320T], e_ [B,1,4T]).  ``noise`` ([B,192,T]) replaces the
        randn_like draw of :202 for reproducible parity runs."""
        x_mask = commons.sequence_mask(length, x_mel.size(2))
        g = self.emb_g(x_mel, x_mask).unsqueeze(-1)
        z = self._latent(w2v, f0, x_mask, g, noise, 1.0)
        return self._decode(z, g)

    @_entry
    @torch.no_grad()
    def voice_conversion(self, src, src_length, trg_mel, trg_length, f0, noise_scale=0.333, uncond=False,
                         noise: Optional[torch.Tensor] = None):
        """:652-673.  ``uncond`` (a model built with cfg=True): the source network and the generator are conditioned
        on the null embedding ``emb(0) * sqrt(256)`` instead of the prompt's style vector (:669-671); the prior encoder
        and the flows still see the prompt."""
        trg_mask = commons.sequence_mask(trg_length, trg_mel.size(2))
        g = self.emb_g(trg_mel, trg_mask).unsqueeze(-1)
        y_mask = commons.sequence_mask(src_length, src.size(2))
        z = self._latent(src, _f0_3d(f0), y_mask, g, noise, noise_scale)
        if uncond:
            g = self._null_g(src.shape[0])
        return self._decode(z, g)[0]

    def _null_g(self, B):
        """[B, 256, 1] = emb(0) * sqrt(256), one copy per utterance (the reference broadcasts a [1, 256, 1] tensor)."""
        if not self.cfg:
            raise AttributeError("'SynthesizerTrn' object has no attribute 'emb' (uncond needs a model built with cfg=True)")
        w = self.emb._w.view(1, -1, 1).expand(B, -1, 1).contiguous()
        return Fh.axpby(w, w, math.sqrt(256.0), 0.0)

    @_entry
    @torch.no_grad()
    def voice_conversion_noise_control(self, src, src_length, trg_mel, trg_length, f0, noise_scale=0.333,
                                       uncond=False, denoise_ratio=0, noise: Optional[torch.Tensor] = None):
        """:674-699.  trg_mel holds two prompts (original, denoised); their style vectors
        are interpolated with ``denoise_ratio`` (B = 1 by construction in the reference, SURVEY.md
        App. B1).  With B source utterances trg_mel is [2B, 80, T]: the B prompts, then the B denoised
        prompts, and f0 is [B, 1, 4T]."""
        if uncond and not self.cfg:   # the reference evaluates self.emb here (:693-695) and then does not use the result
            raise AttributeError("'SynthesizerTrn' object has no attribute 'emb' (uncond needs a model built with cfg=True)")
        B = src.shape[0]
        assert trg_mel.shape[0] == 2 * B
        trg_mask = commons.sequence_mask(trg_length, trg_mel.size(2))
        g = self.emb_g(trg_mel, trg_mask)  # [2B, 256]
        g = Fh.axpby(g[:B], g[B:], 1.0 - denoise_ratio, float(denoise_ratio)).unsqueeze(-1)
        y_mask = commons.sequence_mask(src_length, src.size(2))
        z = self._latent(src, _f0_3d(f0), y_mask, g, noise, noise_scale)
        return self._decode(z, g)[0]


def _f0_3d(f0):
    """inference_plm.py:172 passes f0 as [1, 4T]; torch treats that as an unbatched
    (C=1, L) input of Conv1d."""
    return f0.unsqueeze(0) if f0.dim() == 2 else f0
