"""Host-side layer objects over the libhsp C ABI.

``Conv1d`` / ``ConvTranspose1d`` / ``Linear`` keep the reference's parameter names
(``weight`` or ``weight_g``/``weight_v``, ``bias``) so reference checkpoints load
unchanged, and own a *packed* copy of the weight-norm-folded weights in the layout
the kernels read (``w[K][Cin][M]``).  Packing runs once on the GPU
(``hsp_fold_weight_norm_f32`` + ``hsp_gather_f32`` with a host-built index map); the
reference instead re-derives ``g*v/||v||`` on every forward.

All packed tensors of a model live in one contiguous fp32 arena so that a multi-GPU
job can ship them with a single RCCL broadcast.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import nn

from . import _lib as L
from .synth import kaiser_sinc_filter12


# Optional measurement hook (bench.py): called as hook(kind, flops, bytes, ev_start, ev_end, args)
# with torch.cuda.Events recorded around the launch on the launch stream.
LAUNCH_HOOK = None
# hsp_conv1d_args.debug of every conv launch.  Always 0 in the product: the release libhsp.so refuses anything
# else.  tools/ set it programmatically together with HSP_LIB=.../libhsp_tune.so (kernel decomposition runs).
DEBUG_FLAGS = 0


_ZEROS = {}


def _zeros(device) -> torch.Tensor:
    """64 zero floats per device: the source of out-of-range LDS-DMA lanes (hsp_conv1d_args.zeros)."""
    z = _ZEROS.get(device)
    if z is None:
        z = _ZEROS[device] = torch.zeros(64, dtype=torch.float32, device=device)
    return z


_DFT = {}


def _dft_tables(device):
    """(fwd, inv): the constant tables of csrc/hsp_dftseg.hip (include/hsp.h hsp_dftseg_tables_f32: the 64 x 64 matrix of
    the radix-2 half-length transform + the 128-point twiddles), filled by the library, once per device."""
    t = _DFT.get(device)
    if t is None:
        n = 4160  # HSP_DFTSEG_TABLE_FLOATS
        f = np.zeros(n, dtype=np.float32)
        finv = np.zeros(n, dtype=np.float32)
        L.check(L.lib().hsp_dftseg_tables_f32(f.ctypes.data, finv.ctypes.data), "hsp_dftseg_tables_f32")
        t = _DFT[device] = (torch.from_numpy(f).to(device), torch.from_numpy(finv).to(device))
    return t


_TW = {}


def _wspec_twiddles(device):
    """256 float64 values in device memory, cos(2 pi n / 128) then sin, n < 128: the table of
    hsp_dftseg_weight_spectrum_f32 (include/hsp.h), generated here in float64."""
    t = _TW.get(device)
    if t is None:
        ang = 2.0 * np.pi * np.arange(128, dtype=np.float64) / 128.0
        t = _TW[device] = torch.from_numpy(np.concatenate([np.cos(ang), np.sin(ang)])).to(device)
    return t


# The channel product between the two transforms: "three" = hsp_cprod3_f32 (round 5: three real C x C products per bin,
# 6 C^2 flops per column and 3 C^2 weights per bin), "block" = round 4's [2C x 2C] real block matrix per bin on the conv
# kernel (hsp_conv1d_args.w_bs; 8 C^2 / 4 C^2) -- kept for A/B runs and for channel counts that are not multiples of 64.
FFT_PRODUCT = os.environ.get("HSP_FFT_PRODUCT", "three")


def fft_act_fusable(x) -> bool:
    """hsp_dftseg_args.act_*: the fused activation reads 16-B groups of every row."""
    return (x.stride(2) == 1 and x.shape[2] % 4 == 0 and x.stride(0) % 4 == 0 and x.stride(1) % 4 == 0
            and x.data_ptr() % 16 == 0)


# SURVEY.md §8(b) names its minimum C ABI (hsp_conv1d_f32, hsp_convtr1d_f32, hsp_wn_layer_f32,
# hsp_layernorm_modulate_f32).  Those entry points dispatch to the kernel-level ones this module calls by
# default; with SURVEY_ABI set (HSP_SURVEY_ABI=1) every launch goes through them instead - same kernels, same
# results (tests/test_gpu_parity.py::test_survey_abi_names_give_identical_results).
SURVEY_ABI = os.environ.get("HSP_SURVEY_ABI", "0") == "1"
_DEFER = None  # a list while a caller collects the argument structs of several layers for ONE fused entry point
# (modules.WN -> hsp_wn_layer_f32, modules.DiTConVBlock -> hsp_ffn_conv_f32): _launch() then records instead of launching


class deferred:
    """``with deferred() as args:`` -- conv layers called inside only build their hsp_conv1d_args (outputs are
    allocated as usual); the caller passes the collected structs to a fused entry point with launch_group()."""

    def __enter__(self):
        global _DEFER
        assert _DEFER is None
        _DEFER = []
        return _DEFER

    def __exit__(self, *exc):
        global _DEFER
        _DEFER = None
        return False


def launch_group(kind: str, fn, structs, *extra):
    """One call into a fused entry point taking several hsp_conv1d_args; measurement hook as for _launch().
    ``structs``: the (args, flops, bytes) entries collected by deferred(); a None entry is passed as NULL."""
    flops = sum(e[1] for e in structs if e is not None)
    nbytes = sum(e[2] for e in structs if e is not None)
    ptrs = [C.byref(e[0]) if e is not None else None for e in structs]
    hook = LAUNCH_HOOK
    if hook is None:
        L.check(fn(*ptrs, *extra, L.stream_ptr()), kind)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(fn(*ptrs, *extra, L.stream_ptr()), kind)
    e1.record()
    hook(kind, flops, nbytes, e0, e1, structs[0][0])


def _launch(kind: str, fn, a, flops: int, nbytes: int, soft: bool = False, keep=()):
    """``soft``: return the status instead of raising on HSP_EINVAL (a shape the requested fusion does not
    cover; the caller then issues the un-fused launches)."""
    a.debug = DEBUG_FLAGS
    if _DEFER is not None:
        # the struct holds raw device pointers: `keep` pins the tensors behind them until launch_group() has run
        if soft or a.ln_c1 or a.split_row:    # not an assert: python -O would pass a fused-LayerNorm struct through
            raise L.HspError("deferred(): no soft / fused-LayerNorm / split launches")
        _DEFER.append((a, flops, nbytes, keep))
        return 0
    if SURVEY_ABI:
        fn = L.lib().hsp_convtr1d_f32 if a.rows == L.ROWS_SHUFFLE else L.lib().hsp_conv1d_f32
    hook = LAUNCH_HOOK
    if hook is None:
        rc = fn(C.byref(a), L.stream_ptr())
        if soft and rc == L.EINVAL:
            return rc
        L.check(rc, kind)
        return 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(C.byref(a), L.stream_ptr())
    if soft and rc == L.EINVAL:
        return rc
    L.check(rc, kind)
    e1.record()
    hook(kind, flops, nbytes, e0, e1, a)
    return 0


# ------------------------------------------------------------------ index maps (host logic)
def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def plain_rows(cout: int) -> np.ndarray:
    """packed row -> source output channel (-1 = zero padding); M multiple of 4."""
    m = _round_up(cout, 4)
    rows = np.full(m, -1, np.int64)
    rows[:cout] = np.arange(cout)
    return rows


def gated_rows(half: int) -> np.ndarray:
    """Row order of the GATE modes: 32-row blocks alternate between the 'a' half
    (tanh / linear) and the matching 'b' half (sigmoid) so that one wave holds both
    accumulators of an output channel (hsp_conv1d_mfma.hip epilogue)."""
    assert half % 32 == 0, "gated convs need a half-width that is a multiple of 32"
    m = np.arange(2 * half)
    return ((m >> 5) & 1) * half + (m >> 6) * 32 + (m & 31)


def conv_pack_map(cout: int, cin: int, k: int, rows: np.ndarray) -> np.ndarray:
    """index map src[Cout, Cin, K] -> dst[K][Cin][M]."""
    M = rows.shape[0]
    j = np.arange(k)[:, None, None]
    ci = np.arange(cin)[None, :, None]
    co = rows[None, None, :]
    idx = (co * cin + ci) * k + j
    idx = np.where(co >= 0, idx, -1)
    return np.ascontiguousarray(np.broadcast_to(idx, (k, cin, M))).astype(np.int32).reshape(-1)


def convtr_pack_map(cin: int, cout: int, k: int, up: int) -> Tuple[np.ndarray, int, int]:
    """ConvTranspose1d(stride=up) as a polyphase conv: returns (map, K', M).

    y[co, n] = sum_ci sum_i x[ci, i] * w[ci, co, n + p - up*i]; writing n + p = up*q + r
    gives, for packed row m = co*up + r and tap j' (input index q + j' - (K'-1)),
    the source tap r + (K'-1-j')*up (absent taps are zero).  src is w[Cin, Cout, K]."""
    kp = -(-k // up)
    M = _round_up(cout * up, 4)
    m = np.arange(M)
    co, r = m // up, m % up
    jp = np.arange(kp)[:, None, None]
    ci = np.arange(cin)[None, :, None]
    tap = r[None, None, :] + (kp - 1 - jp) * up
    idx = (ci * cout + co[None, None, :]) * k + tap
    ok = (tap < k) & (co[None, None, :] < cout)
    return np.where(ok, idx, -1).astype(np.int32).reshape(-1), kp, M


# ---------------------------------------------------------------------------- arena
class WeightArena:
    """One contiguous fp32 device buffer holding every packed tensor of a model."""

    ALIGN = 64  # floats (256 B): keeps float4 weight loads aligned

    def __init__(self):
        self.specs: List[Tuple[object, str, int]] = []
        self.buffer: Optional[torch.Tensor] = None
        self.views: Dict[Tuple[int, str], torch.Tensor] = {}
        self._offsets: List[int] = []
        self.total = 0

    def request(self, owner, name: str, numel: int):
        self._offsets.append(self.total)
        self.specs.append((owner, name, numel))
        self.total += _round_up(numel, self.ALIGN)

    def allocate(self, device):
        self.buffer = torch.zeros(max(self.total, 1), dtype=torch.float32, device=device)
        for (owner, name, numel), off in zip(self.specs, self._offsets):
            self.views[(id(owner), name)] = self.buffer[off:off + numel]

    def view(self, owner, name: str) -> torch.Tensor:
        return self.views[(id(owner), name)]


# Bumped whenever a HipLayer's parameters move (``.cuda()`` / ``.to(device)``) or are reloaded
# (``load_state_dict``): a top-level mirror compares its own stamp with this counter on every entry call -- one
# integer compare on the hot path -- and only walks its layers when something changed anywhere (ensure_ready()).
_EPOCH = 0


def _bump_epoch():
    global _EPOCH
    _EPOCH += 1


class HipLayer(nn.Module):
    """Base of modules that own packed device state.  ``_hsp_stale`` says the packed copy no longer matches the
    parameters (never packed, parameters moved to another device, or reloaded)."""

    _hsp_stale = True

    def hsp_requests(self) -> List[Tuple[str, int]]:
        return []

    def hsp_fill(self, arena: WeightArena, materialize: bool) -> None:
        pass

    def _apply(self, fn, *args, **kw):
        before = [(q.device, q.data_ptr()) for q in self._parameters.values() if q is not None]
        r = super()._apply(fn, *args, **kw)
        after = [(q.device, q.data_ptr()) for q in self._parameters.values() if q is not None]
        if before != after:
            self.__dict__["_hsp_stale"] = True
            _bump_epoch()
        return r

    def _load_from_state_dict(self, *args, **kw):
        r = super()._load_from_state_dict(*args, **kw)
        self.__dict__["_hsp_stale"] = True
        _bump_epoch()
        return r


def ensure_ready(model: nn.Module) -> None:
    """Drop-in behaviour of the reference's call sites (``Model(...).cuda()``, ``load_state_dict``, ``.eval()``,
    then inference methods: inference_plm.py:215-262): the first inference call after the parameters moved or
    changed folds / packs the weights on the device the parameters live on.  An explicit ``finalize(device)`` stays
    available (multi-GPU jobs lay the arena out without materialising it).  Raises on CPU: no CPU fallback."""
    if model.__dict__.get("_hsp_epoch") == _EPOCH:
        return
    if any(m._hsp_stale for m in model.modules() if isinstance(m, HipLayer)):
        p = next(model.parameters(), None)
        dev = p.device if p is not None else torch.device("cpu")
        if dev.type != "cuda":
            raise L.HspError(f"{type(model).__name__}: parameters are on {dev}; move the model to a ROCm device "
                             "(.cuda() / .to(device)) -- the product path has no CPU fallback")
        fin = getattr(model, "finalize", None)
        if fin is not None:
            fin(dev)
        else:
            finalize(model, dev)
    model.__dict__["_hsp_epoch"] = _EPOCH


def entry(fn):
    """Decorator of a mirror's public inference methods: pack the weights on first use (ensure_ready)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kw):
        ensure_ready(self)
        return fn(self, *args, **kw)
    return wrapped


def _gather(src: torch.Tensor, idx_map: np.ndarray, dst: torch.Tensor):
    mp = torch.from_numpy(idx_map).to(src.device)
    L.check(L.lib().hsp_gather_f32(L.fptr(src), L.ptr(mp), L.fptr(dst), dst.numel(), L.stream_ptr()), "hsp_gather_f32")
    # `mp` must outlive the asynchronous gather on the current stream
    torch.cuda.current_stream().synchronize()


def _fold(v: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    rows = v.shape[0]
    cols = v.numel() // rows
    w = torch.empty_like(v)
    L.check(L.lib().hsp_fold_weight_norm_f32(L.fptr(v.contiguous()), L.fptr(g.contiguous()), L.fptr(w), rows, cols,
                                             L.stream_ptr()), "hsp_fold_weight_norm_f32")
    return w


def _spectrum(Cc: int, Np: int, device) -> torch.Tensor:
    """[64 bins][2 C][Np] spectrum buffer of a frequency-domain conv.  (Round 6 measured a padded plane stride -- 64 or 1088
    floats between the bins, so that the 128 scattered 128-B runs of a column do not share an address pattern modulo the HBM
    channel interleave: no effect on any transform kernel, 147.3 / 147.7 / 147.9 us per inverse launch,
    profiles/r06_plane_pad.txt; the knob is gone.)"""
    return torch.empty(64, 2 * Cc, Np, dtype=torch.float32, device=device)


class _ConvBase(HipLayer):
    def __init__(self, weight_shape, rows0: int, bias: bool, weight_norm: bool):
        super().__init__()
        if weight_norm:
            self.weight_g = nn.Parameter(torch.ones(rows0, 1, 1), requires_grad=False)
            self.weight_v = nn.Parameter(torch.zeros(*weight_shape), requires_grad=False)
        else:
            self.weight = nn.Parameter(torch.zeros(*weight_shape), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(bias), requires_grad=False) if bias else None
        self.has_bias = bool(bias)
        self.wn = weight_norm
        self._w: Optional[torch.Tensor] = None
        self._b: Optional[torch.Tensor] = None
        self._wf: Optional[torch.Tensor] = None     # per-bin matrices of the frequency-domain form (Conv1d.ensure_wf)

    def _folded(self) -> torch.Tensor:
        if self.wn:
            return _fold(self.weight_v.data, self.weight_g.data)
        return self.weight.data.contiguous()

    def _bias_src(self) -> torch.Tensor:
        return self.bias.data

    def _require_ready(self):
        if self._w is None:
            raise L.HspError(f"{type(self).__name__} used before finalize(): call model.finalize(device) after "
                             "loading weights")


class Conv1d(_ConvBase):
    """torch.nn.Conv1d (optionally weight-normed) with fused prologue/epilogue."""

    def __init__(self, cin, cout, k, stride=1, padding=0, dilation=1, bias=True, weight_norm=False,
                 rows=L.ROWS_PLAIN, weight_2d=False):
        super().__init__((cout, cin) if weight_2d else (cout, cin, k), cout, cout if bias else 0, weight_norm)
        self.cin, self.cout, self.k, self.stride, self.padding, self.dilation = cin, cout, k, stride, padding, dilation
        self.rows = rows
        self.trim_right = 0   # outputs dropped at the end (an even kernel with padding k // 2 yields one too many)
        if rows in (L.ROWS_GATE_WN, L.ROWS_GATE_GLU):
            self.row_map = gated_rows(cout // 2)
        else:
            self.row_map = plain_rows(cout)
        self.M = int(self.row_map.shape[0])

    def fuse_input_layernorm(self, norm):
        """Run ``norm`` (an affine LayerNorm over the input channels: attributes ``weight``/``gamma``,
        ``bias``/``beta``, ``eps``) inside this 1x1 layer's GEMM: the packed weight becomes W diag(gamma), the
        bias W beta + b, and ``ln_c1`` the row sums of the packed weight (hsp_conv1d_args.ln_c1).  The layer is
        then called on the UN-normalised input."""
        assert self.k == 1 and self.rows == L.ROWS_PLAIN
        self.__dict__["_pre_norm"] = norm   # not registered: the norm stays where the reference keeps it

    def _ln_params(self):
        n = self._pre_norm
        g = (n.weight if hasattr(n, "weight") else n.gamma).data
        b = (n.bias if hasattr(n, "weight") else n.beta).data
        return g, b

    def keep_rowmajor_weight(self):
        """Also keep the folded weight as stored -- [cout][cin] row-major -- in the arena (``_wt``): the A operand of
        the projection inside hsp_mha_proj_f32 (attention + output projection in one launch)."""
        assert self.k == 1 and self.rows == L.ROWS_PLAIN
        self.__dict__["_rowmajor"] = True

    def pack_flipped(self, inputs: bool = False, outputs: bool = False):
        """Pack this layer for a tensor whose channel axis is stored REVERSED: ``inputs`` -- the input channels arrive in
        reverse order (the packed weight's input columns are reversed), ``outputs`` -- the output channels are to be
        written in reverse order (rows and bias reversed).  The parameters and their state-dict keys stay as the
        reference stores them; only the packed copy changes.  This is how modules.Flip (modules.py:270-277) between two
        coupling layers costs no launch: the layer after a Flip reads and writes the un-flipped tensor through a packed
        weight that has the permutation in it (modules.ResidualCouplingLayer_Transformer_simple.flipped)."""
        assert self.rows == L.ROWS_PLAIN and not self.__dict__.get("_pre_norm") and not self.__dict__.get("_rowmajor")
        self.__dict__["_flip_in"], self.__dict__["_flip_out"] = bool(inputs), bool(outputs)
        self.__dict__["_hsp_stale"] = True
        _bump_epoch()

    def hsp_requests(self):
        if self.__dict__.get("_stacked_elsewhere"):   # rows live in a StackedLinearCT: parameters only
            return []
        fused = self.__dict__.get("_pre_norm") is not None
        return [("w", self.k * self.cin * self.M)] + ([("b", self.cout)] if self.has_bias or fused else []) + \
            ([("c1", self.cout)] if fused else []) + \
            ([("wt", self.cout * self.cin)] if self.__dict__.get("_rowmajor") else [])

    def enable_fft(self):
        """This same-length stride-1 conv may also run in its frequency-domain form (round 4, csrc/hsp_dftseg.hip):
        ``forward_fft`` computes it as forward DFT -> one batched channel product over the 64 bins -> inverse DFT, 1.7 (2.2
        in the block form) multiply-adds per output and channel pair instead of k.  The per-bin matrices of conj(rfft(w,
        128)) are DERIVED data: they are not part of the weight arena (what a multi-GPU job broadcasts) but
        a side buffer that ``ensure_wf`` fills from the packed taps on first use, on every rank, with
        hsp_dftseg_weight_spectrum_f32 -- a model that never meets a batch large enough for the form (fft_wins) never
        allocates them (201 MB per conv at 512 channels)."""
        assert self.stride == 1 and self.rows == L.ROWS_PLAIN and self.cin == self.cout and 2 <= self.k <= 64
        assert self.padding * 2 == (self.k - 1) * self.dilation, "same-length conv"
        self.__dict__["_fft"] = True

    def fft_form(self) -> str:
        return "three" if FFT_PRODUCT == "three" and self.cin % 64 == 0 else "block"

    def ensure_wf(self):
        """The per-bin matrices of this conv's channel product, derived on the device from the packed taps (``_w``:
        [k][C][M]) in float64 and rounded once: [64][3][C][C] = (a + b, a, b) of conj(W) = a + i b for hsp_cprod3_f32, or
        round 4's [64][2C][2C] block matrices.  Deterministic: every rank derives the same bits from the same taps."""
        if self._wf is not None and self._wf_form == self.fft_form():
            return self._wf
        self._require_ready()
        if torch.cuda.is_current_stream_capturing():
            raise L.HspError("ensure_wf() inside a stream capture: run one eager call first (or prepare_fft(model))")
        Cc, form = self.cin, self.fft_form()
        wf = torch.empty(64 * (3 if form == "three" else 4) * Cc * Cc, dtype=torch.float32, device=self._w.device)
        L.check(L.lib().hsp_dftseg_weight_spectrum_f32(
            L.fptr(self._w), self.k, Cc, self.M, L.ptr(_wspec_twiddles(self._w.device)), L.fptr(wf),
            L.WSPEC_THREE if form == "three" else L.WSPEC_BLOCK, L.stream_ptr()), "hsp_dftseg_weight_spectrum_f32")
        self._wf, self._wf_form = wf, form
        return wf

    def hsp_fill(self, arena, materialize):
        if self.__dict__.get("_stacked_elsewhere"):
            return
        fused = self.__dict__.get("_pre_norm") is not None
        self._w = arena.view(self, "w")
        self._b = arena.view(self, "b") if self.has_bias or fused else None
        self._c1 = arena.view(self, "c1") if fused else None
        self._wt = arena.view(self, "wt").view(self.cout, self.cin) if self.__dict__.get("_rowmajor") else None
        self._wf = None                                   # derived from _w on first use (ensure_wf)
        if materialize:
            w = self._folded()
            if self.__dict__.get("_flip_in"):
                w = w.flip(1).contiguous()
            if self.__dict__.get("_flip_out"):
                w = w.flip(0).contiguous()
            if self._wt is not None:
                assert not fused
                self._wt.copy_(w.reshape(self.cout, self.cin))
            if fused:
                g, beta = self._ln_params()
                w2 = w.reshape(self.cout, self.cin).double()
                wg = w2 * g.double()[None, :]
                self._c1.copy_(wg.sum(1).float())
                c2 = w2 @ beta.double()
                if self.has_bias:
                    c2 = c2 + self._bias_src().double()
                self._b.copy_(c2.float())
                w = wg.float().reshape(w.shape).contiguous()
            _gather(w, conv_pack_map(self.cout, self.cin, self.k, self.row_map), self._w)
            if self._b is not None and not fused:
                self._b.copy_(self._bias_src().flip(0) if self.__dict__.get("_flip_out") else self._bias_src())

    # The frequency-domain form in pieces (forward_fft strings them together; an AMP pair fuses the inverse of its first
    # conv with the forward transform of its second: forward_fft_pair)
    def _fft_args(self, B, Lx):
        """hsp_dftseg_args of this conv for a [B, C, Lx] tensor: geometry only, no pointers."""
        d, k = self.dilation, self.k
        hop = 129 - k
        nseg = -(-(-(-Lx // d)) // hop)
        da = L.DftSegArgs()
        da.B, da.C, da.L, da.k, da.dil, da.pad, da.nseg = B, self.cin, Lx, k, d, self.padding, nseg
        da.Np = _round_up(B * d * nseg, 4)
        da.post_scale = 1.0
        da.prod3 = int(self.fft_form() == "three")
        return da

    def fft_supported(self, B, Lx) -> bool:
        """Do the transform kernels take this geometry (hsp_dftseg_supported: dilation <= 8, a spectrum below 4 GiB per
        launch, item counts within int)?  fft_wins asks before choosing the form; beyond it the direct conv runs."""
        if not self.__dict__.get("_fft"):
            return False
        da = self._fft_args(B, Lx)
        da.xf_bs = 2 * self.cin * da.Np
        return bool(L.lib().hsp_dftseg_supported(C.byref(da)))

    @staticmethod
    def _fft_set_act(da, x_like, act1d):
        if act1d._ea is None:
            raise L.HspError("Activation1d used before finalize()")
        da.act_alpha_exp, da.act_beta_inv, da.act_filt = L.fptr(act1d._ea), L.fptr(act1d._binv), L.fptr(act1d._filt)

    def _fft_forward(self, x, act1d=None):
        """x [B, C, L] -> spectrum [64][2 C][Np] (hsp_dftseg_fwd_f32), the activation applied on the way if given."""
        B, Cc, Lx = x.shape
        assert Cc == self.cin and x.stride(2) == 1 and self.__dict__.get("_fft")
        da = self._fft_args(B, Lx)
        xf = _spectrum(Cc, da.Np, x.device)
        da.x, da.x_bs, da.x_cs = L.fptr(x), x.stride(0), x.stride(1)
        da.xf, da.xf_bs, da.dft = L.fptr(xf), xf.stride(0), L.fptr(_dft_tables(x.device)[0])
        if act1d is not None:
            if not fft_act_fusable(x):
                raise L.HspError("forward_fft(act1d=...) needs 16-B addressable rows")
            self._fft_set_act(da, x, act1d)
        hook, ev = LAUNCH_HOOK, None
        if hook is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        L.check(L.lib().hsp_dftseg_fwd_f32(C.byref(da), L.stream_ptr()), "hsp_dftseg_fwd_f32")
        if hook is not None:
            ev[1].record()
            hook("hsp_dftseg_fwd_f32", 2 * 2 * 64 * 64 * B * Cc * da.dil * da.nseg, 4 * (B * Cc * Lx + 128 * Cc * da.Np), ev[0], ev[1], None)
        return xf, (ev[0] if ev else None)

    def _fft_product(self, xf):
        """ONE batched launch over the 64 bins: yf[bin] = conj(W)[bin] xf[bin] -- three real C x C products per bin
        (hsp_cprod3_f32) or the [2C x 2C] block matrix [[Wr, Wi], [-Wi, Wr]] on the conv kernel."""
        Cc, Np = self.cin, xf.shape[2]
        wf = self.ensure_wf()
        yf = _spectrum(Cc, Np, xf.device)
        if self._wf_form == "three":
            pa = L.Cprod3Args()
            pa.xf, pa.yf, pa.w, pa.zeros = L.fptr(xf), L.fptr(yf), L.fptr(wf), L.fptr(_zeros(xf.device))
            pa.xf_bs, pa.yf_bs, pa.bins, pa.C, pa.Np, pa.debug = xf.stride(0), yf.stride(0), 64, Cc, Np, DEBUG_FLAGS
            hook = LAUNCH_HOOK
            if hook is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            L.check(L.lib().hsp_cprod3_f32(C.byref(pa), L.stream_ptr()), "hsp_cprod3_f32")
            if hook is not None:
                e1.record()
                hook("hsp_cprod3_f32", 2 * 64 * 3 * Cc * Cc * Np, 4 * (64 * 2 * 2 * Cc * Np + 64 * 3 * Cc * Cc), e0, e1, None)
            return yf
        a = L.Conv1dArgs()
        a.x, a.x_bs, a.x_cs, a.x_ts = L.fptr(xf), xf.stride(0), xf.stride(1), 1
        a.B, a.Cin, a.Lin = 64, 2 * Cc, Np
        a.w, a.K, a.dil, a.pad, a.stride = L.fptr(wf), 1, 1, 0, 1
        a.M, a.w_ld, a.w_bs = 2 * Cc, 2 * Cc, 4 * Cc * Cc
        a.zeros = L.fptr(_zeros(xf.device))
        _set_out(a, yf, 64, 2 * Cc, Np)
        a.ncols, a.rows, a.scale, a.post_scale = Np, L.ROWS_PLAIN, 1.0, 1.0
        # the launches are booked with the flops / bytes they EXECUTE; the conv they stand for is reported once more, as
        # a whole, under the kind "hsp_fftconv" (algorithmic flops and bytes of the direct form over all its launches)
        _launch("hsp_conv1d_mfma_f32", L.lib().hsp_conv1d_mfma_f32, a, 2 * 64 * 4 * Cc * Cc * Np,
                4 * (64 * 2 * 2 * Cc * Np + 64 * 4 * Cc * Cc))
        return yf

    def _fft_inverse_args(self, yf, B, Lx):
        da = self._fft_args(B, Lx)
        da.xf, da.xf_bs, da.dft = L.fptr(yf), yf.stride(0), L.fptr(_dft_tables(yf.device)[1])
        da.bias = L.fptr(self._b) if self._b is not None else None
        return da

    def _fft_inverse(self, yf, B, Lx, *, res=None, out=None, accumulate=False, post_scale=1.0, before_inverse=None):
        Cc = self.cin
        if out is None:
            out = torch.empty(B, Cc, Lx, dtype=torch.float32, device=yf.device)
        assert out.shape == (B, Cc, Lx) and out.stride(2) == 1 and (res is None or (res.shape == out.shape and res.stride(2) == 1))
        if before_inverse is not None:
            torch.cuda.current_stream(yf.device).wait_event(before_inverse)
        da = self._fft_inverse_args(yf, B, Lx)
        da.y, da.y_bs, da.y_cs = L.fptr(out), out.stride(0), out.stride(1)
        if res is not None:
            da.res, da.res_bs, da.res_cs = L.fptr(res), res.stride(0), res.stride(1)
        da.accumulate, da.post_scale = int(bool(accumulate)), float(post_scale)
        hook, e1 = LAUNCH_HOOK, None
        if hook is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        L.check(L.lib().hsp_dftseg_inv_f32(C.byref(da), L.stream_ptr()), "hsp_dftseg_inv_f32")
        if hook is not None:
            e1.record()
            nio = 2 + bool(res is not None) + 2 * bool(accumulate)
            hook("hsp_dftseg_inv_f32", 2 * 2 * 64 * 64 * B * Cc * da.dil * da.nseg, 4 * (B * Cc * Lx * (nio - 1) + 128 * Cc * da.Np), e0, e1, None)
        return out, e1

    def forward_fft(self, x, *, res=None, out=None, accumulate=False, post_scale=1.0, act1d=None, before_inverse=None):
        """The conv in its frequency-domain form (enable_fft()): three launches -- forward DFT of 128-sample segments,
        ONE 1x1 product over the 64 bins on the conv kernel (hsp_conv1d_args.w_bs), inverse DFT with the conv's epilogue
        (bias, residual, running sum, post_scale).  x [B, C, L] with unit time stride.  ``act1d``: the Activation1d in
        front of the conv, applied by the forward transform while it stages its input (needs 16-B addressable rows:
        fft_act_fusable).  ``before_inverse``: an event the stream waits on before the inverse transform, the only launch
        that touches ``out``."""
        self._require_ready()
        B, Cc, Lx = x.shape
        xf, e_first = self._fft_forward(x, act1d)
        yf = self._fft_product(xf)
        out, e_last = self._fft_inverse(yf, B, Lx, res=res, out=out, accumulate=accumulate, post_scale=post_scale,
                                        before_inverse=before_inverse)
        if LAUNCH_HOOK is not None:
            nio = 2 + bool(res is not None) + 2 * bool(accumulate)
            LAUNCH_HOOK("hsp_fftconv", 2 * B * Cc * Cc * self.k * Lx, 4 * B * Cc * Lx * nio + 4 * self.k * Cc * Cc, e_first, e_last, None)
        return out

    def fft_pair_ok(self, second, x) -> bool:
        """Can the inverse transform of this conv be fused with the activation and the forward transform of ``second``
        (hsp_dftseg_pair_f32) for an input like x?  Both unchunked and their two LDS stretches within one CU's 160 KB."""
        if not self.__dict__.get("_fft") or not second.__dict__.get("_fft") or second.cin != self.cin or x.shape[2] % 4:
            return False
        B, _, Lx = x.shape
        ia, fa = self._fft_args(B, Lx), second._fft_args(B, Lx)
        dummy = L.fptr(_zeros(x.device))                          # the predicate reads geometry only; pointers must be non-null
        ia.xf = ia.dft = fa.xf = fa.dft = fa.act_alpha_exp = fa.act_beta_inv = fa.act_filt = dummy
        ia.xf_bs, fa.xf_bs = 2 * self.cin * ia.Np, 2 * self.cin * fa.Np
        return bool(L.lib().hsp_dftseg_pair_supported(C.byref(ia), C.byref(fa)))

    def _fft_pair_launch(self, second, yf, B, Lx, act_second, like, res=None, y=None):
        """ONE launch (hsp_dftseg_pair_f32): inverse transform of this conv's product ``yf`` + bias [+ ``res``, the result
        written to ``y``: the pass-through form] -> ``act_second`` -> forward transform for ``second``.  Returns second's
        spectrum and the launch's (start, end) events when a LAUNCH_HOOK is set."""
        Cc = self.cin
        ia = self._fft_inverse_args(yf, B, Lx)
        fa = second._fft_args(B, Lx)
        xf2 = _spectrum(Cc, fa.Np, yf.device)
        fa.xf, fa.xf_bs, fa.dft = L.fptr(xf2), xf2.stride(0), L.fptr(_dft_tables(yf.device)[0])
        self._fft_set_act(fa, like, act_second)
        if y is not None:
            assert y.shape == (B, Cc, Lx) and y.stride(2) == 1
            ia.y, ia.y_bs, ia.y_cs = L.fptr(y), y.stride(0), y.stride(1)
            if res is not None:
                assert res.shape == y.shape and res.stride(2) == 1
                ia.res, ia.res_bs, ia.res_cs = L.fptr(res), res.stride(0), res.stride(1)
        else:
            assert res is None
        hook, ev = LAUNCH_HOOK, (None, None)
        if hook is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        L.check(L.lib().hsp_dftseg_pair_f32(C.byref(ia), C.byref(fa), L.stream_ptr()), "hsp_dftseg_pair_f32")
        if hook is not None:
            ev[1].record()
            nio = 0 if y is None else 1 + (res is not None)
            hook("hsp_dftseg_pair_f32", 2 * 2 * 64 * 64 * B * Cc * (ia.dil * ia.nseg + fa.dil * fa.nseg),
                 4 * 128 * Cc * (ia.Np + fa.Np) + 4 * B * Cc * Lx * nio, ev[0], ev[1], None)
        return xf2, ev

    def fft_through_ok(self, nxt, x) -> bool:
        """Can the inverse transform of this conv WITH its residual epilogue be fused with the activation and the forward
        transform of ``nxt`` -- the seam between two iterations of an AMP block (hsp_dftseg_pair_f32, pass-through form)?"""
        if not self.__dict__.get("_fft") or not nxt.__dict__.get("_fft") or nxt.cin != self.cin or x.shape[2] % 4:
            return False
        if x.stride(2) != 1 or (x.stride(0) | x.stride(1)) & 3 or x.data_ptr() & 15:
            return False
        B, _, Lx = x.shape
        ia, fa = self._fft_args(B, Lx), nxt._fft_args(B, Lx)
        dummy = L.fptr(_zeros(x.device))
        ia.xf = ia.dft = fa.xf = fa.dft = fa.act_alpha_exp = fa.act_beta_inv = fa.act_filt = dummy
        ia.y = ia.res = L.fptr(x)                                # geometry + alignment of rows like x's
        ia.y_bs = ia.res_bs = x.stride(0)
        ia.y_cs = ia.res_cs = x.stride(1)
        ia.xf_bs, fa.xf_bs = 2 * self.cin * ia.Np, 2 * self.cin * fa.Np
        return bool(L.lib().hsp_dftseg_pair_supported(C.byref(ia), C.byref(fa)))

    def forward_fft_pair(self, second, x, *, act_first, act_second, res=None, out=None, accumulate=False, post_scale=1.0,
                         before_inverse=None):
        """second(act_second(self(act_first(x)))) [+ res ...] with both convs in the frequency domain and the tensor between
        them never in HBM: forward(x) -> product(self) -> [inverse + bias + act_second + forward] -> product(second) ->
        inverse with second's epilogue.  Five launches instead of six (eight with the activations)."""
        self._require_ready()
        second._require_ready()
        B, Cc, Lx = x.shape
        xf, e_first = self._fft_forward(x, act_first)
        yf = self._fft_product(xf)
        hook = LAUNCH_HOOK
        xf2, _ = self._fft_pair_launch(second, yf, B, Lx, act_second, x)
        yf2 = second._fft_product(xf2)
        out, e_last = second._fft_inverse(yf2, B, Lx, res=res, out=out, accumulate=accumulate, post_scale=post_scale,
                                          before_inverse=before_inverse)
        if hook is not None:
            nio = 2 + bool(res is not None) + 2 * bool(accumulate)
            hook("hsp_fftconv", 2 * B * Cc * Cc * (self.k + second.k) * Lx, 4 * B * Cc * Lx * nio + 4 * (self.k + second.k) * Cc * Cc,
                 e_first, e_last, 2)   # (two convs)
        return out

    # ----------------------------------------------------------------------------
    def forward(self, x, *, act1d=None, lrelu: Optional[float] = None, silu_in=False, act=L.ACT_NONE, cbias=None,
                mask=None, mask_mode=L.MASK_NONE, cscale=None, scale=1.0, res=None, out=None, accumulate=False,
                post_scale=1.0, force_direct=False, row_range=None, split_out=None, mask_mode2=L.MASK_NONE, ln_mod=None):
        """``row_range=(r0, r1)`` computes only output channels [r0, r1) (PLAIN rows, r0 % 4 == 0):
        the WN res/skip layer is one parameter set feeding two differently-fused launches.
        ``split_out=(split_row, out2, accumulate2)``: ONE launch for both halves of such a layer
        (hsp_conv1d_args.split_row): rows [0, split_row) -> the usual output with this call's epilogue, rows
        [split_row, cout) -> ``out2`` (+= if accumulate2; allocated when None).  Returns (out, out2), or None when
        the library has no fused kernel for the shape (the caller then launches the halves separately).
        ``ln_mod=(scale, c1, bias_b, mask, eps)``: the modulated input LayerNorm of a DiT block inside this 1x1 layer's
        GEMM (hsp_conv1d_args.ln_scale): x is the UN-normalised input, ``scale`` [B, Cin], ``c1`` / ``bias_b`` [B, cout]
        (unit inner stride) the per-utterance vectors of include/hsp.h, ``mask`` [B, 1, T] or None.  Returns None when the
        library has no kernel for the shape (the caller then runs LayerNorm + modulate as its own launch)."""
        self._require_ready()
        B, Cin, Lin = x.shape
        assert Cin == self.cin, (Cin, self.cin)
        gated = self.rows in (L.ROWS_GATE_WN, L.ROWS_GATE_GLU)
        cout = self.cout // 2 if gated else self.cout
        r0 = 0
        if row_range is not None:
            r0, r1 = row_range
            assert not gated and r0 % 4 == 0 and 0 <= r0 < r1 <= self.cout
            cout = r1 - r0
        Lout = (Lin + 2 * self.padding - self.dilation * (self.k - 1) - 1) // self.stride + 1 - self.trim_right
        out2 = None
        if split_out is not None:
            split_row, out2, acc2 = split_out
            assert row_range is None and not gated and self.k == 1 and 0 < split_row < self.cout
            cout = split_row
            if out2 is None:
                assert not acc2
                out2 = torch.empty(B, self.cout - split_row, Lout, dtype=torch.float32, device=x.device)
        if out is None:
            out = torch.empty(B, cout, Lout, dtype=torch.float32, device=x.device)
        a = L.Conv1dArgs()
        a.x, a.x_bs, a.x_cs, a.x_ts = L.fptr(x), x.stride(0), x.stride(1), x.stride(2)
        a.B, a.Cin, a.Lin = B, Cin, Lin
        a.w, a.K, a.dil, a.pad, a.stride = L.fptr(self._w) + 4 * r0, self.k, self.dilation, self.padding, self.stride
        a.M = self.M if row_range is None else _round_up(cout, 4)
        a.w_ld = self.M
        a.zeros = L.fptr(_zeros(x.device))
        _set_out(a, out, B, cout, Lout)
        a.ncols = Lout
        a.rows, a.gate_half = self.rows, (cout if gated else 0)
        a.bias = (L.fptr(self._b) + 4 * r0) if self._b is not None else None
        _set_epilogue(a, act, cbias, mask, mask_mode, cscale, scale, res, accumulate, post_scale, out)
        if act1d is not None:
            a.prologue = L.PRO_ACT1D
            a.alpha_exp, a.beta_inv, a.filt = L.fptr(act1d._ea), L.fptr(act1d._binv), L.fptr(act1d._filt)
        elif lrelu is not None:
            a.prologue, a.slope = L.PRO_LRELU, float(lrelu)
        elif silu_in:
            a.prologue = L.PRO_SILU
        if self.__dict__.get("_pre_norm") is not None:
            a.ln_c1, a.ln_eps = L.fptr(self._c1), float(self._pre_norm.eps)
            assert not force_direct and row_range is None
        if ln_mod is not None:
            sc, c1, bb, lmask, eps = ln_mod
            assert self.k == 1 and not gated and row_range is None and split_out is None and cbias is None and \
                self.__dict__.get("_pre_norm") is None and act1d is None
            assert sc.shape == (B, Cin) and c1.shape == (B, cout) and bb.shape == (B, cout)
            assert sc.stride(1) == 1 and c1.stride(1) == 1 and bb.stride(1) == 1
            a.ln_c1, a.ln_c1_bs, a.ln_eps = L.fptr(c1), c1.stride(0), float(eps)
            a.ln_scale, a.ln_scale_bs = L.fptr(sc), sc.stride(0)
            a.bias, a.cbias, a.cbias_bs = None, L.fptr(bb), bb.stride(0)
            if lmask is not None:
                assert lmask.stride(-1) == 1 and lmask.shape[0] == B
                a.ln_mask, a.ln_mask_bs = L.fptr(lmask), lmask.stride(0)
            rc = _launch("hsp_conv1d_mfma_f32", L.lib().hsp_conv1d_mfma_f32, a, 2 * B * cout * Cin * Lout,
                         4 * (B * Cin * Lin + B * cout * Lout + cout * Cin), soft=True)
            return None if rc else out
        direct = (force_direct or self.stride != 1 or self.cin < 8 or cout < 8 or Lout < 8 or silu_in) \
            and not gated and act1d is None and self.__dict__.get("_pre_norm") is None
        if direct and a.res_ts > 1:
            # the VALU kernel reads its residual at unit column stride (hsp_conv1d_args.res_ts is the register-path
            # token GEMM's): gather it first -- e.g. the PLM's last layer at B = 4, whose [D, 4] product is below the MFMA path
            from . import functional as Fh
            res = Fh.copy_strided(res)
            a.res, a.res_bs, a.res_cs, a.res_ts = L.fptr(res), res.stride(0), res.stride(1), 0
        rows_full = cout * (2 if gated else 1)
        flops = 2 * B * rows_full * Cin * self.k * Lout
        # algorithmic traffic: input once, output once (+ residual / accumulate reads), weights once
        nbytes = 4 * (B * Cin * Lin + B * cout * Lout * (1 + (res is not None) + bool(accumulate))
                      + rows_full * Cin * self.k)
        if split_out is not None:
            a.Cout = self.cout                      # rows [split_row, cout) go to the second output
            a.split_row, a.accumulate2, a.mask_mode2 = split_row, int(bool(acc2)), mask_mode2   # (the second output shares `mask`)
            a.y2, a.y2_bs, a.y2_cs = L.fptr(out2), out2.stride(0), out2.stride(1)
            flops = 2 * B * self.cout * Cin * Lout
            nbytes += 4 * B * (self.cout - split_row) * Lout * (1 + bool(acc2))
            rc = _launch("hsp_conv1d_mfma_f32", L.lib().hsp_conv1d_mfma_f32, a, flops, nbytes, soft=True)
            return None if rc else (out, out2)
        if direct:
            _launch("hsp_conv1d_direct_f32", L.lib().hsp_conv1d_direct_f32, a, flops, nbytes)
        elif self.__dict__.get("_pre_norm") is not None and _DEFER is not None:
            # inside deferred(): the normalisation (affine part folded into the packed weights) runs now as its own
            # launch, the plain GEMM joins the group
            from . import functional as Fh
            xn = Fh.layernorm_mod(x, float(self._pre_norm.eps))
            a.x, a.x_bs, a.x_cs, a.x_ts = L.fptr(xn), xn.stride(0), xn.stride(1), xn.stride(2)
            a.ln_c1 = None
            _launch("hsp_conv1d_mfma_f32", L.lib().hsp_conv1d_mfma_f32, a, flops, nbytes,
                    keep=(xn, out, res, cbias, mask, cscale))
        elif self.__dict__.get("_pre_norm") is not None:
            # fused input LayerNorm: token-GEMM shapes only (16-B addressable columns).  Any other shape runs the
            # normalisation as its own launch -- WITHOUT the affine part, which is folded into the packed weights
            # (W diag(gamma), W beta + b) -- and then the same GEMM: identical arithmetic, one launch more.
            rc = _launch("hsp_conv1d_mfma_f32", L.lib().hsp_conv1d_mfma_f32, a, flops, nbytes, soft=True)
            if rc:
                from . import functional as Fh
                xn = Fh.layernorm_mod(x, float(self._pre_norm.eps))
                a.x, a.x_bs, a.x_cs, a.x_ts = L.fptr(xn), xn.stride(0), xn.stride(1), xn.stride(2)
                a.ln_c1 = None
                _launch("hsp_conv1d_mfma_f32", L.lib().hsp_conv1d_mfma_f32, a, flops, nbytes)
        else:
            _launch("hsp_conv1d_mfma_f32", L.lib().hsp_conv1d_mfma_f32, a, flops, nbytes,
                    keep=(x, out, res, cbias, mask, cscale))
        return out


class ConvTranspose1d(_ConvBase):
    """torch.nn.ConvTranspose1d (weight-normed) as a polyphase conv on the MFMA kernel."""

    def __init__(self, cin, cout, k, stride, padding=0, bias=True, weight_norm=False):
        super().__init__((cin, cout, k), cin, cout if bias else 0, weight_norm)
        self.cin, self.cout, self.k, self.up, self.padding = cin, cout, k, stride, padding
        self.pack_map, self.kp, self.M = convtr_pack_map(cin, cout, k, stride)

    def hsp_requests(self):
        return [("w", self.kp * self.cin * self.M)] + ([("b", self.cout)] if self.bias is not None else [])

    def hsp_fill(self, arena, materialize):
        self._w = arena.view(self, "w")
        self._b = arena.view(self, "b") if self.bias is not None else None
        if materialize:
            _gather(self._folded(), self.pack_map, self._w)
            if self._b is not None:
                self._b.copy_(self.bias.data)

    def forward(self, x, *, res=None, out=None, lrelu: Optional[float] = None):
        self._require_ready()
        B, Cin, Lin = x.shape
        assert Cin == self.cin
        Lout = (Lin - 1) * self.up - 2 * self.padding + self.k
        if out is None:
            out = torch.empty(B, self.cout, Lout, dtype=torch.float32, device=x.device)
        a = L.Conv1dArgs()
        a.x, a.x_bs, a.x_cs, a.x_ts = L.fptr(x), x.stride(0), x.stride(1), x.stride(2)
        a.B, a.Cin, a.Lin = B, Cin, Lin
        a.w, a.K, a.M, a.dil, a.pad, a.stride = L.fptr(self._w), self.kp, self.M, 1, self.kp - 1, 1
        a.w_ld = self.M
        a.zeros = L.fptr(_zeros(x.device))
        _set_out(a, out, B, self.cout, Lout)
        a.ncols = (Lout - 1 + self.padding) // self.up + 1
        a.rows, a.up, a.shuf_pad = L.ROWS_SHUFFLE, self.up, self.padding
        a.bias = L.fptr(self._b)
        _set_epilogue(a, L.ACT_NONE, None, None, L.MASK_NONE, None, 1.0, res, False, 1.0, out)
        if lrelu is not None:
            a.prologue, a.slope = L.PRO_LRELU, float(lrelu)
        flops = 2 * B * Cin * self.cout * self.k * Lin
        nbytes = 4 * (B * Cin * Lin + B * self.cout * Lout * (1 + (res is not None)) + Cin * self.cout * self.k)
        _launch("hsp_conv1d_mfma_f32", L.lib().hsp_conv1d_mfma_f32, a, flops, nbytes)
        return out


class _SubArena:
    """arena views of a layer that lives inside another HipLayer (not registered as a sub-module)"""

    def __init__(self, arena, owner, prefix):
        self.arena, self.owner, self.prefix = arena, owner, prefix

    def view(self, _layer, name):
        return self.arena.view(self.owner, self.prefix + name)


class PolyphaseConv1d(HipLayer):
    """nn.Conv1d(cin, cout, k, stride = s, padding = 0) on the unit-stride MFMA kernel: taps are grouped by phase
    p = j mod s, and phase p is a unit-stride conv with ceil((k - p) / s) taps over the strided view x[..., p::s]
    (hsp_conv1d_args.x_ts = s); the phases accumulate into one output.  Parameters keep nn.Conv1d's names and
    shapes (``weight`` [cout, cin, k], ``bias``): the feature-encoder layers of wav2vec2 (HF
    Wav2Vec2LayerNormConvLayer.conv: k 3 / stride 2 and k 2 / stride 2)."""

    def __init__(self, cin, cout, k, stride, bias=True):
        super().__init__()
        self.cin, self.cout, self.k, self.stride = cin, cout, k, stride
        self.weight = nn.Parameter(torch.zeros(cout, cin, k), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False) if bias else None
        ph = [Conv1d(cin, cout, len(range(p, k, stride)), bias=(bias and p == 0)) for p in range(min(stride, k))]
        self.__dict__["_phases"] = ph     # not registered: no duplicate state_dict keys

    def hsp_requests(self):
        return [(f"p{i}.{n}", numel) for i, ph in enumerate(self._phases) for n, numel in ph.hsp_requests()]

    def hsp_fill(self, arena, materialize):
        for i, ph in enumerate(self._phases):
            if materialize:
                ph.weight.data = self.weight.data[:, :, i::self.stride].contiguous()
                if ph.bias is not None:
                    ph.bias.data = self.bias.data
                ph.to(self.weight.device)
            ph.hsp_fill(_SubArena(arena, self, f"p{i}."), materialize)

    def forward(self, x, **kw):
        B, Cin, Lin = x.shape
        Lout = (Lin - self.k) // self.stride + 1
        out = torch.empty(B, self.cout, Lout, dtype=torch.float32, device=x.device)
        for i, ph in enumerate(self._phases):
            xp = x[:, :, i: i + self.stride * (Lout + ph.k - 1): self.stride]     # exactly Lout + K_p - 1 samples
            ph(xp, out=out, accumulate=i > 0, **(kw if i == len(self._phases) - 1 else {}))
        return out


class GroupedPosConv1d(HipLayer):
    """HF Wav2Vec2PositionalConvEmbedding.conv + SamePad + GELU + the residual add of the encoder:
    Conv1d(C, C, k = 128, padding = 64, groups = 16) with weight_norm over dim 2 (one gain per TAP: g [1, 1, k]), last
    output dropped (even kernel), y = x + gelu(conv(x)).  One group = two MFMA launches on channel-slice views, taps
    [0, 64) and [64, 128): the conv kernel stages all taps of a channel chunk at once, and 128 taps x 64 rows do not
    fit its LDS double buffer.  Parameter names: ``weight_g`` / ``weight_v`` (torch <= 2.0 checkpoints;
    ``parametrizations.weight.original0 / original1`` of newer ones are renamed on load by the owner) and ``bias``."""

    TAPS = 64

    def __init__(self, channels, k, groups):
        super().__init__()
        assert k % 2 == 0 and channels % groups == 0 and k % self.TAPS == 0
        self.channels, self.k, self.groups, self.cg = channels, k, groups, channels // groups
        self.weight_g = nn.Parameter(torch.ones(1, 1, k), requires_grad=False)
        self.weight_v = nn.Parameter(torch.zeros(channels, self.cg, k), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(channels), requires_grad=False)
        subs = []
        for _ in range(groups):
            for j0 in range(0, k, self.TAPS):
                # taps j0 .. j0 + 63 read x[t + j - k / 2]: a conv with padding k / 2 - j0 (negative = a shifted window;
                # the kernel bounds-checks every read) whose first L outputs are kept: natural length L + 2 pad - 63
                pad = k // 2 - j0
                c = Conv1d(self.cg, self.cg, self.TAPS, padding=pad, bias=(j0 == 0))
                c.trim_right = 2 * pad - (self.TAPS - 1)
                subs.append(c)
        self.__dict__["_subs"] = subs

    def hsp_requests(self):
        return [(f"s{i}.{n}", numel) for i, c in enumerate(self._subs) for n, numel in c.hsp_requests()]

    def hsp_fill(self, arena, materialize):
        w = None
        nb = self.k // self.TAPS
        if materialize:
            # weight_norm(dim = 2): w[:, :, j] = g[j] * v[:, :, j] / ||v[:, :, j]||_F  == the dim-0 fold of v laid out
            # [k][C * cg] (a permutation, no arithmetic); folded by the same kernel as every other weight-normed layer
            vp = self.weight_v.data.permute(2, 0, 1).contiguous().reshape(self.k, -1)
            w = _fold(vp, self.weight_g.data.reshape(self.k)).reshape(self.k, self.channels, self.cg).permute(1, 2, 0)
        for i, c in enumerate(self._subs):
            gi, bi = divmod(i, nb)
            if materialize:
                c.weight.data = w[gi * self.cg:(gi + 1) * self.cg, :, bi * self.TAPS:(bi + 1) * self.TAPS].contiguous()
                if c.bias is not None:
                    c.bias.data = self.bias.data[gi * self.cg:(gi + 1) * self.cg].contiguous()
            c.hsp_fill(_SubArena(arena, self, f"s{i}."), materialize)

    def forward(self, x):
        from . import functional as Fh
        nb = self.k // self.TAPS
        conv = torch.empty_like(x)
        for i, c in enumerate(self._subs):
            gi, bi = divmod(i, nb)
            sl = slice(gi * self.cg, (gi + 1) * self.cg)
            c(x[:, sl], out=conv[:, sl], accumulate=bi > 0)
        return Fh.axpby(x, Fh.act(conv, L.ACT_GELU_ERF), 1.0, 1.0)


class Linear(Conv1d):
    """torch.nn.Linear on a (B, C) or (B, C, 1) style vector == 1x1 conv with L = 1.
    Parameters keep nn.Linear's 2-D ``weight`` shape for checkpoint compatibility."""

    def __init__(self, cin, cout, bias=True):
        super().__init__(cin, cout, 1, bias=bias, weight_2d=True)

    def forward(self, x, **kw):
        x3 = x.reshape(x.shape[0], self.cin, 1)
        return super().forward(x3, force_direct=True, **kw)


class LinearCT(Conv1d):
    """torch.nn.Linear applied along the channel axis of a channel-major [B, C, T] tensor (a 1x1
    conv on the MFMA path).  ``packed=False`` keeps only the parameters (nn.Linear names and
    shapes) for a layer whose rows are packed into a :class:`StackedLinearCT`."""

    def __init__(self, cin, cout, bias=True, packed=True):
        super().__init__(cin, cout, 1, bias=bias, weight_2d=True)
        self.packed = packed

    def hsp_requests(self):
        return super().hsp_requests() if self.packed else []

    def hsp_fill(self, arena, materialize):
        if self.packed:
            super().hsp_fill(arena, materialize)


class StackedLinearCT(Conv1d):
    """Several nn.Linear layers reading the same input, run as ONE GEMM with their output rows
    stacked (q/k/v projections of transformer_mega.MultiHeadAttention, ttv_v1/transformer_mega.py:54-56).
    Owns no parameters: the packed weight is gathered from the parts' ``weight`` / ``bias``."""

    def __init__(self, parts):
        super().__init__(parts[0].cin, sum(p.cout for p in parts), 1, bias=True, weight_2d=True)
        del self._parameters["weight"], self._parameters["bias"]
        self.__dict__["_parts"] = tuple(parts)  # not registered as sub-modules (no duplicate state_dict keys)

    def _folded(self):
        return torch.cat([p.stack_weight() if hasattr(p, "stack_weight") else p.weight.data for p in self._parts], 0).contiguous()

    def _bias_src(self):
        return torch.cat([p.stack_bias() if hasattr(p, "stack_bias") else p.bias.data for p in self._parts], 0)


class ModulatedNormRows:
    """Rows of a stacked conditioning GEMM that deliver, per utterance, the two vectors a modulated input LayerNorm needs
    inside a following 1x1 layer `lin` (include/hsp.h ln_scale; modules.DiTConVBlock, modules.py:346-347,406-409):
        c1_b   = lin.W (1 + scale_b)        bias_b = lin.W shift_b + lin.b
    where (shift_b, scale_b) = rows [0, C) and [C, 2C) of an `ada` Linear applied to the SAME conditioning input s_b.
    Both are linear in s_b:   c1_b = (W A_scale) s_b + (W a_scale + W 1),   bias_b = (W A_shift) s_b + (W a_shift + b)
    -- so they are 2 cout extra rows of the GEMM that evaluates `ada` anyway (products formed in float64 when the
    weights are packed, rounded once).  Not a module: owns no parameters, a part of a StackedLinearCT."""

    def __init__(self, lin: "Conv1d", ada: "Conv1d"):
        assert lin.k == 1 and ada.k == 1 and ada.cout >= 2 * lin.cin
        self.lin, self.ada, self.cin, self.cout = lin, ada, ada.cin, 2 * lin.cout

    def _mats(self):
        Cc = self.lin.cin
        W = self.lin._folded().reshape(self.lin.cout, Cc).double()
        A = self.ada.weight.data.reshape(self.ada.cout, self.ada.cin).double()
        ab = self.ada.bias.data.double()
        return W, A[:Cc], A[Cc:2 * Cc], ab[:Cc], ab[Cc:2 * Cc]          # W, A_shift, A_scale, a_shift, a_scale

    def stack_weight(self):
        W, A_sh, A_sc, _, _ = self._mats()
        return torch.cat([W @ A_sc, W @ A_sh], 0).float()

    def stack_bias(self):
        W, _, _, a_sh, a_sc = self._mats()
        b = self.lin.bias.data.double() if self.lin.bias is not None else 0.0
        return torch.cat([W @ a_sc + W.sum(1), W @ a_sh + b], 0).float()


def _set_out(a, out, B, cout, Lout):
    assert out.shape == (B, cout, Lout), (tuple(out.shape), (B, cout, Lout))
    assert out.stride(2) == 1 or Lout == 1
    a.y, a.y_bs, a.y_cs = L.fptr(out), out.stride(0), out.stride(1)
    a.Cout, a.Lout = cout, Lout


def _set_epilogue(a, act, cbias, mask, mask_mode, cscale, scale, res, accumulate, post_scale, out):
    a.act = act
    if cbias is not None:
        assert cbias.dim() >= 2 and cbias.stride(1) == 1
        a.cbias, a.cbias_bs = L.fptr(cbias), cbias.stride(0)
    if mask is not None and mask_mode != L.MASK_NONE:
        assert mask.stride(-1) == 1
        a.mask, a.mask_bs, a.mask_mode = L.fptr(mask), mask.stride(0), mask_mode
    if cscale is not None:
        assert cscale.stride(1) == 1
        a.cscale, a.cscale_bs = L.fptr(cscale), cscale.stride(0)
    a.scale = float(scale)
    if res is not None:
        assert res.shape == out.shape
        a.res, a.res_bs, a.res_cs = L.fptr(res), res.stride(0), res.stride(1)
        if res.stride(2) != 1 and res.shape[2] != 1:
            if res.stride(2) < 1:         # a time-broadcast (.expand) residual: every kernel reads res_ts 0 as unit stride
                raise L.HspError("residual with time stride %d: materialise it (copy_strided) before the launch" % res.stride(2))
            a.res_ts = res.stride(2)      # hsp_conv1d_args.res_ts: the register-path token GEMM only (else HSP_EINVAL)
    a.accumulate = 1 if accumulate else 0
    a.post_scale = float(post_scale)


def prepare_fft(model: nn.Module) -> int:
    """Derive the per-bin matrices of every conv of ``model`` that carries the frequency-domain form now (Conv1d.ensure_wf
    does it on first use otherwise) -- before a hipGraph capture, or to pay the start-up cost at a chosen moment.  Returns
    the number of floats the side buffers hold."""
    n = 0
    for m in model.modules():
        if isinstance(m, Conv1d) and m.__dict__.get("_fft"):
            n += m.ensure_wf().numel()
    return n


# ------------------------------------------------------------------ finalisation
def finalize(model: nn.Module, device, materialize: bool = True) -> WeightArena:
    """Pack every HipLayer of ``model`` into one arena on ``device``.

    ``materialize=False`` only lays the arena out (same offsets on every rank) so that
    the contents can arrive by broadcast (parallel.broadcast_weights)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise L.HspError("finalize() needs a ROCm device: the product path has no CPU fallback")
    L.lib()
    if materialize:
        model.to(device)
    arena = WeightArena()
    layers = [m for m in model.modules() if isinstance(m, HipLayer)]
    for m in layers:
        for name, numel in m.hsp_requests():
            arena.request(m, name, numel)
    with torch.cuda.device(device):
        arena.allocate(device)
        for m in layers:
            m.hsp_fill(arena, materialize)
            m.__dict__["_hsp_stale"] = False
        torch.cuda.current_stream().synchronize()
    model._hsp_arena = arena
    model.__dict__["_hsp_epoch"] = _EPOCH
    return arena
