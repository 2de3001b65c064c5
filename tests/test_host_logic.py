"""CPU: host-side logic of the product path (no compute calls): weight packing maps,
arena layout, state-dict compatibility with the reference, the C ABI of libhsp.so."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_keys_match_reference():
    """Key names and shapes of every inference-path module equal the reference's
    (captured from the reference's own state_dict by tools/make_golden.py)."""
    for name in H.fixture_names():
        meta, arrays = H.load_fixture(name)
        mod = H.build_module(meta)
        mine = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        if meta["kind"] == "speechsr_real":     # the reference's shipped checkpoint itself: its dec.* keys and shapes
            ref = {k[2:]: v.shape for k, v in arrays.items() if k.startswith("w:")}
        else:
            ref = {k: tuple(s) for k, s in meta["shapes"]}
        # the t2w2v class also carries the legacy prosody convs, which only the ttv_infer fixtures list
        extra = {k for k in mine.keys() - ref.keys() if ".".join(k.split(".")[:2]).replace("ttv.", "").startswith("plm_conv")}
        if meta["kind"] in ("ttv_front", "ttv_gen", "tts_e2e"):
            mine = {k: v for k, v in mine.items() if k not in extra}
        assert mine == ref, name


def test_reference_checkpoint_wrappers_and_unused_keys():
    from megatts2_hierspeechpp_amd.hierspeechpp_speechsynthesizer import SynthesizerTrn
    from oracle.hsp_oracle import default_config
    meta, _ = H.load_fixture("infer_config1")
    net = SynthesizerTrn(641, 192, **default_config())
    sd = H.synth_sd(meta)
    sd["enc_q.pre.weight"] = torch.zeros(1)          # training-only modules of the reference are skipped
    sd["mel_decoder.proj.weight"] = torch.zeros(1)
    net.load_state_dict(sd)                            # bare state dict (inference_plm.py:218)
    net.load_state_dict({"model": sd, "iteration": 3})  # utils.load_checkpoint wrapper (utils.py:19-44)
    with pytest.raises(RuntimeError):
        net.load_state_dict({k: v for k, v in sd.items() if k != "dec.conv_post.weight"})


def test_product_path_refuses_cpu():
    """No CPU fallback: finalize and the kernels raise on CPU tensors / devices."""
    from megatts2_hierspeechpp_amd import _lib, functional as Fh
    from megatts2_hierspeechpp_amd.hip_layers import Conv1d, finalize
    conv = Conv1d(8, 8, 3, padding=1)
    with pytest.raises(_lib.HspError):
        finalize(conv, "cpu")
    with pytest.raises(_lib.HspError):
        conv(torch.zeros(1, 8, 16))
    with pytest.raises(_lib.HspError):
        Fh.flip_channels(torch.zeros(1, 4, 4))


def test_conv_pack_map_plain_and_gated():
    from megatts2_hierspeechpp_amd.hip_layers import conv_pack_map, gated_rows, plain_rows
    cout, cin, k = 6, 5, 3
    w = np.arange(cout * cin * k, dtype=np.float32).reshape(cout, cin, k)
    rows = plain_rows(cout)
    assert rows.shape[0] == 8 and list(rows[6:]) == [-1, -1]
    m = conv_pack_map(cout, cin, k, rows).reshape(k, cin, 8)
    packed = np.where(m >= 0, w.reshape(-1)[np.maximum(m, 0)], 0.0)
    for j in range(k):
        assert np.array_equal(packed[j, :, :cout], w[:, :, j].T) and not packed[j, :, cout:].any()
    g = gated_rows(64)  # packed 32-row blocks alternate tanh-half / sigmoid-half of the same channels
    assert list(g[:3]) == [0, 1, 2] and g[32] == 64 and g[63] == 95 and g[64] == 32 and g[96] == 96
    assert sorted(g.tolist()) == list(range(128))


def test_convtr_pack_map_is_the_polyphase_identity():
    """ConvTranspose1d == polyphase conv with the packed taps (numpy check vs torch)."""
    from megatts2_hierspeechpp_amd.hip_layers import convtr_pack_map
    rng = np.random.default_rng(0)
    for (k, u) in [(8, 4), (11, 5), (4, 2), (3, 3)]:
        cin, cout, L, p = 3, 2, 7, (k - u) // 2
        w = rng.standard_normal((cin, cout, k)).astype(np.float32)
        x = rng.standard_normal((1, cin, L)).astype(np.float32)
        ref = torch.nn.functional.conv_transpose1d(torch.from_numpy(x), torch.from_numpy(w), stride=u, padding=p)[0].numpy()
        m, kp, M = convtr_pack_map(cin, cout, k, u)
        wp = np.where(m >= 0, w.reshape(-1)[np.maximum(m, 0)], 0.0).reshape(kp, cin, M)
        Lout = (L - 1) * u - 2 * p + k
        y = np.zeros((cout, Lout), np.float32)
        ncols = (Lout - 1 + p) // u + 1
        for q in range(ncols):
            for mm in range(cout * u):
                co, r = divmod(mm, u)
                n = u * q + r - p
                if not (0 <= n < Lout):
                    continue
                acc = 0.0
                for jp in range(kp):
                    i = q + jp - (kp - 1)
                    if 0 <= i < L:
                        acc += float(wp[jp, :, mm] @ x[0, :, i])
                y[co, n] = acc
        assert np.abs(y - ref).max() < 1e-5, (k, u)


def test_flip_folds_into_the_next_coupling_layers_packed_weights():
    """The algebra behind modules.ResidualCouplingLayer_Transformer_simple.set_flipped (round 6): a coupling layer called on
    flip(x) (modules.py:270-277, hierspeechpp_speechsynthesizer.py:80-86) equals flip of the same layer called on x with
    `pre` reading the OTHER half through reversed input columns and `post` updating the first half through reversed rows
    and bias -- checked on the CPU oracle's own coupling layer (float64), not on the GPU path (the `flow` golden does that)."""
    from oracle import hsp_oracle as O
    g = torch.Generator().manual_seed(11)
    C_, Hd, T = 8, 16, 9
    name = "cl"
    sd = {}
    def rnd(*shape): return torch.randn(*shape, generator=g, dtype=torch.float64)
    sd[f"{name}.pre.weight"], sd[f"{name}.pre.bias"] = rnd(Hd, C_ // 2, 1), rnd(Hd)
    sd[f"{name}.post.weight"], sd[f"{name}.post.bias"] = rnd(C_ // 2, Hd, 1), rnd(C_ // 2)
    x = rnd(2, C_, T)
    mask = (torch.rand(2, 1, T, generator=g) > 0.2).double()
    half = C_ // 2
    # the reference's order with zero DiT blocks in between (their input and output live in the hidden space: untouched by the fold)
    want = O.coupling_reverse(sd, name, torch.flip(x, [1]), mask, None, n_layers=0)
    # the folded layer on the un-flipped tensor
    pre_w = sd[f"{name}.pre.weight"].flip(1)
    post_w, post_b = sd[f"{name}.post.weight"].flip(0), sd[f"{name}.post.bias"].flip(0)
    h = torch.nn.functional.conv1d(x[:, half:], pre_w, sd[f"{name}.pre.bias"]) * mask
    m = torch.nn.functional.conv1d(h, post_w, post_b) * mask
    got = torch.cat([(x[:, :half] - m) * mask, x[:, half:]], 1)
    assert torch.allclose(torch.flip(got, [1]), want, rtol=0, atol=1e-12)
    # and the block's bookkeeping: with n_flows = 4 the layers 3 and 1 work on the reversed axis, no Flip is left over
    from megatts2_hierspeechpp_amd.hierspeechpp_speechsynthesizer import ResidualCouplingBlock_Transformer
    blk = ResidualCouplingBlock_Transformer(192, 192, 5, 1, 3, gin_channels=256)
    assert [blk.flows[2 * i].flipped for i in range(4)] == [False, True, False, True]
    assert [blk.flows[2 * i].pre.__dict__.get("_flip_in", False) for i in range(4)] == [False, True, False, True]
    assert [blk.flows[2 * i].post.__dict__.get("_flip_out", False) for i in range(4)] == [False, True, False, True]


def test_arena_layout_is_deterministic_and_aligned():
    """Every rank lays the arena out identically (the broadcast relies on it)."""
    from megatts2_hierspeechpp_amd.hip_layers import HipLayer, WeightArena
    meta, _ = H.load_fixture("generator")
    layouts = []
    for _ in range(2):
        mod = H.build_module(meta)
        arena = WeightArena()
        for m in mod.modules():
            if isinstance(m, HipLayer):
                for name, numel in m.hsp_requests():
                    arena.request(m, name, numel)
        layouts.append((arena.total, list(arena._offsets)))
        assert all(o % WeightArena.ALIGN == 0 for o in arena._offsets)
    assert layouts[0] == layouts[1] and layouts[0][0] > 50_000_000  # ~59 M generator weights


def test_shard_range_partitions_exactly():
    from megatts2_hierspeechpp_amd.parallel import shard_range
    for n in (1, 7, 32, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


# ------------------------------------------------------------------------- C ABI
def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "hsp.h")).read()
    return sorted(set(re.findall(r"\b(hsp_[a-z0-9_]+)\s*\(", hdr)))


def test_library_loads_and_exports_every_declared_symbol():
    from megatts2_hierspeechpp_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 17
    assert set(declared) == set(_lib.SIGNATURES), set(declared) ^ set(_lib.SIGNATURES)
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert lib.hsp_version() == 102 and lib.hsp_arch() == b"gfx950"


def test_dynamic_symbol_table_is_exactly_the_header():
    """`nm -D --defined-only libhsp.so` lists the functions include/hsp.h declares and nothing else (-fvisibility=hidden,
    the header's visibility push, csrc/hsp.map): no mangled helper, kernel handle or __hip_cuid_* leaks out of the boundary."""
    import shutil
    import subprocess
    from megatts2_hierspeechpp_amd import _lib
    if shutil.which("nm") is None:
        pytest.skip("binutils nm not available")
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == sorted(_declared_symbols()), set(exported) ^ set(_declared_symbols())


def test_isa_lint_passes_on_the_real_kernel_sources():
    """The lint is part of `make all` (csrc/Makefile: build/isa/.checked) -- so a library built by build() has passed it;
    here it runs again over the device assembly of the REAL translation units (not the synthetic kernels of the test
    below) whenever hipcc is present, compiling the assembly first if a bare `make lib` left it out."""
    import glob
    import subprocess
    csrc = os.path.join(ROOT, "megatts2_hierspeechpp_amd", "csrc")
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    res = subprocess.run(["make", "-C", csrc, "-j4", "check-isa"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-4000:] + res.stderr[-4000:]
    files = sorted(glob.glob(os.path.join(csrc, "build", "isa", "*.s")))
    assert len(files) >= 15, files
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa
    reads = 0
    for f in files:
        if not f.endswith(("hsp_cprod3.s", "hsp_dftseg.s", "hsp_conv1d_tile_M128.s")):
            continue                                     # the three largest hand-scheduled units again, in-process (~15 s)
        findings, n = check_isa.lint(f)
        assert not findings, (f, findings[:3])
        reads += n
    assert reads > 200, reads


def test_ctypes_structs_match_the_header(tmp_path):
    """sizeof / offsetof of the argument structs as gcc sees include/hsp.h == the ctypes mirrors."""
    from megatts2_hierspeechpp_amd import _lib
    src = tmp_path / "abi.c"
    fields_c = [f for f, _ in _lib.Conv1dArgs._fields_]
    fields_m = [f for f, _ in _lib.MhaArgs._fields_]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "hsp.h"', "int main(void){",
             'printf("%zu\\n", sizeof(hsp_conv1d_args));']
    lines += [f'printf("%zu\\n", offsetof(hsp_conv1d_args, {f}));' for f in fields_c]
    lines += ['printf("%zu\\n", sizeof(hsp_mha_args));']
    lines += [f'printf("%zu\\n", offsetof(hsp_mha_args, {f}));' for f in fields_m]
    fields_p = [f for f, _ in _lib.MhaProjArgs._fields_]
    lines += ['printf("%zu\\n", sizeof(hsp_mha_proj_args));']
    lines += [f'printf("%zu\\n", offsetof(hsp_mha_proj_args, {f}));' for f in fields_p]
    fields_d = [f for f, _ in _lib.DftSegArgs._fields_]
    lines += ['printf("%zu\\n", sizeof(hsp_dftseg_args));']
    lines += [f'printf("%zu\\n", offsetof(hsp_dftseg_args, {f}));' for f in fields_d]
    fields_3 = [f for f, _ in _lib.Cprod3Args._fields_]
    lines += ['printf("%zu\\n", sizeof(hsp_cprod3_args));']
    lines += [f'printf("%zu\\n", offsetof(hsp_cprod3_args, {f}));' for f in fields_3]
    lines += ["return 0;}"]
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    vals = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(_lib.Conv1dArgs)] + [getattr(_lib.Conv1dArgs, f).offset for f in fields_c]
    want += [ctypes.sizeof(_lib.MhaArgs)] + [getattr(_lib.MhaArgs, f).offset for f in fields_m]
    want += [ctypes.sizeof(_lib.MhaProjArgs)] + [getattr(_lib.MhaProjArgs, f).offset for f in fields_p]
    want += [ctypes.sizeof(_lib.DftSegArgs)] + [getattr(_lib.DftSegArgs, f).offset for f in fields_d]
    want += [ctypes.sizeof(_lib.Cprod3Args)] + [getattr(_lib.Cprod3Args, f).offset for f in fields_3]
    assert vals == want


def test_write_wav_round_trip(tmp_path):
    """inference_plm.py:195-200 writes int16 mono PCM with scipy.io.wavfile.write; the mirror's writer must
    produce a file scipy reads back bit-exactly."""
    from scipy.io import wavfile
    from megatts2_hierspeechpp_amd.inference_plm import write_wav
    pcm = (np.random.default_rng(0).integers(-32768, 32767, 4801)).astype(np.int16)
    for sr in (16000, 24000, 48000):
        p = tmp_path / f"a{sr}.wav"
        write_wav(p, sr, torch.from_numpy(pcm))
        rate, back = wavfile.read(p)
        assert rate == sr and back.dtype == np.int16 and np.array_equal(back, pcm)
    with pytest.raises(ValueError):
        write_wav(tmp_path / "bad.wav", 16000, pcm.astype(np.float32))


def test_mel_front_end_host_side():
    """MelSpectrogramFixed mirror: constructor surface of inference_plm.py:204-213, DFT basis rows, refusal of
    CPU input / unsupported options (no fallback)."""
    from megatts2_hierspeechpp_amd import _lib
    from megatts2_hierspeechpp_amd.Mels_preprocess import MelSpectrogramFixed, melscale_fbanks_htk
    m = MelSpectrogramFixed(sample_rate=16000, n_fft=1280, win_length=1280, hop_length=320, f_min=0, f_max=8000,
                            n_mels=80, window_fn=torch.hann_window)
    w = m.dft.weight.detach().numpy().reshape(1282, 1280)
    n = np.arange(1280)
    assert np.abs(w[3] - np.cos(2 * np.pi * 3 * n / 1280)).max() < 1e-6
    assert np.abs(w[641 + 7] + np.sin(2 * np.pi * 7 * n / 1280)).max() < 1e-6
    assert np.allclose(w[0], 1.0) and np.allclose(w[641], 0.0)
    fb = melscale_fbanks_htk(641, 0.0, 8000.0, 80, 16000)
    assert fb.shape == (641, 80) and (fb >= 0).all() and ((fb > 0).sum(1) <= 2).all()   # triangles overlap pairwise
    with pytest.raises(_lib.HspError):
        m(torch.zeros(1, 16000))                       # not finalized / CPU tensor: refuse
    with pytest.raises(_lib.HspError):
        MelSpectrogramFixed(n_fft=1280, win_length=640)
    with pytest.raises(_lib.HspError):
        MelSpectrogramFixed(n_fft=1280, power=1.0)


# ------------------------------------------------------------------ conv dispatch (no GPU: hsp_conv1d_mfma_plan only validates and selects)
def _plan_args(B, Cin, L, K, cout, *, rows=0, up=0, dil=1, act=0, res=False, accumulate=False, pro=0, mask=False,
               cscale=False, scale=1.0, unaligned=False):
    from megatts2_hierspeechpp_amd import _lib as L_
    a = L_.Conv1dArgs()
    gated = rows in (L_.ROWS_GATE_WN, L_.ROWS_GATE_GLU)
    M = cout * up if rows == L_.ROWS_SHUFFLE else (2 * cout if gated else (cout + 3) // 4 * 4)
    lout = L * up if rows == L_.ROWS_SHUFFLE else L
    a.x, a.x_bs, a.x_cs, a.x_ts, a.B, a.Cin, a.Lin = 0x10000, Cin * L, L, 1, B, Cin, L
    a.w, a.K, a.M, a.dil, a.pad, a.stride, a.zeros, a.w_ld = 0x20000, K, M, dil, (K - 1) * dil // 2, 1, 0x30000, M
    a.y, a.y_bs, a.y_cs, a.Cout, a.Lout = 0x40000 + (4 if unaligned else 0), cout * lout, lout, cout, lout
    a.ncols = L + 1 if rows == L_.ROWS_SHUFFLE else L
    a.rows, a.up, a.gate_half, a.scale, a.post_scale, a.act, a.prologue = rows, up, cout if gated else 0, scale, 1.0, act, pro
    if pro == L_.PRO_ACT1D:
        a.alpha_exp, a.beta_inv, a.filt = 0x50000, 0x51000, 0x52000
    if res:
        a.res, a.res_bs, a.res_cs = 0x60000, cout * lout, lout
    if mask:
        a.mask, a.mask_bs, a.mask_mode = 0x70000, lout, L_.MASK_PRE
    if cscale:
        a.cscale, a.cscale_bs = 0x80000, cout
    a.accumulate = int(accumulate)
    return a


def test_conv_dispatch_covers_every_host_side_combination():
    """Every (tile shape, epilogue kind, activation prologue) the dispatcher can route to must exist in the
    library: walk short / long sequences x channel widths x row modes x epilogue features through
    hsp_conv1d_mfma_plan (validation + selection, no launch)."""
    from megatts2_hierspeechpp_amd import _lib as L_
    lib = L_.lib()
    plan = (ctypes.c_int32 * 4)()
    seen = set()
    for B, L in ((1, 37), (2, 200), (32, 200), (32, 4000)):
        for C_ in (32, 48, 64, 128, 192, 256, 512):
            for K, dil in ((1, 1), (3, 1), (7, 3), (11, 5)):
                combos = [dict(), dict(res=True), dict(res=True, accumulate=True), dict(act=L_.ACT_TANH),
                          dict(mask=True, res=True), dict(cscale=True, res=True), dict(scale=0.5),
                          dict(act=L_.ACT_GELU_TANH, unaligned=True), dict(pro=L_.PRO_LRELU, res=True),
                          dict(pro=L_.PRO_ACT1D), dict(pro=L_.PRO_ACT1D, res=True, accumulate=True),
                          dict(pro=L_.PRO_ACT1D, act=L_.ACT_TANH),
                          dict(rows=L_.ROWS_GATE_WN), dict(rows=L_.ROWS_GATE_GLU, mask=True, res=True),
                          dict(rows=L_.ROWS_SHUFFLE, up=2), dict(rows=L_.ROWS_SHUFFLE, up=5, res=True)]
                for kw in combos:
                    if kw.get("rows") in (L_.ROWS_GATE_WN, L_.ROWS_GATE_GLU) and C_ % 32:
                        continue
                    a = _plan_args(B, C_, L, K, C_, dil=dil, **kw)
                    rc = lib.hsp_conv1d_mfma_plan(ctypes.byref(a), ctypes.byref(plan))
                    assert rc == 0, (B, L, C_, K, dil, kw)
                    seen.add((plan[0], plan[1]))
    # the sweep reaches every tile shape, and the token GEMM (reported as 64 x 64 with KC = 0)
    assert {(128, 128), (64, 256), (32, 512), (64, 64), (64, 128), (32, 128)} <= seen, seen


def test_wide_halo_convs_have_a_tile_shape_plain_and_gated():
    """Halos beyond the 64-column window slack: the 64-tap blocks of the wav2vec2 positional conv (plain rows) and WN
    in-layers with dilation_rate > 1 (gated rows: k = 5, dilation 16 and 25 -- modules.WN accepts any dilation_rate, as
    the reference does) both find a shape; past the widest pitch the library refuses and modules.WN says so at
    construction."""
    from megatts2_hierspeechpp_amd import _lib as L_
    from megatts2_hierspeechpp_amd import modules
    lib = L_.lib()
    plan = (ctypes.c_int32 * 4)()
    for B, L in ((1, 50), (32, 200), (32, 4000)):
        for kw in (dict(rows=L_.ROWS_GATE_WN), dict(rows=L_.ROWS_GATE_GLU, mask=True, res=True), dict(res=True)):
            for K, dil in ((5, 16), (5, 25), (64, 1), (3, 60)):
                a = _plan_args(B, 192, L, K, 192, dil=dil, **kw)
                assert lib.hsp_conv1d_mfma_plan(ctypes.byref(a), ctypes.byref(plan)) == 0, (B, L, K, dil, kw)
                assert (plan[0], plan[1]) == ((64, 128) if "rows" in kw else (64, 64))
        a = _plan_args(B, 192, L, 5, 192, dil=32, rows=L_.ROWS_GATE_WN)      # halo 131 > 125
        assert lib.hsp_conv1d_mfma_plan(ctypes.byref(a), ctypes.byref(plan)) == L_.EINVAL
    modules.WN(192, 5, 2, 5, gin_channels=256)                               # dilations 1 ... 16
    with pytest.raises(L_.HspError):
        modules.WN(192, 5, 2, 6, gin_channels=256)                           # dilation 32


def test_release_library_refuses_the_tuning_word():
    """hsp_conv1d_args.debug selects kernel tuning switches that exist only in libhsp_tune.so; the shipped
    library must reject any non-zero value instead of silently producing wrong audio."""
    from megatts2_hierspeechpp_amd import _lib as L_
    if os.environ.get("HSP_LIB"):
        pytest.skip("HSP_LIB selects a non-default library")
    lib = L_.lib()
    plan = (ctypes.c_int32 * 4)()
    a = _plan_args(32, 128, 4000, 3, 128, res=True)
    assert lib.hsp_conv1d_mfma_plan(ctypes.byref(a), ctypes.byref(plan)) == 0
    for bits in (1, 2, 16, 256, 32768):
        a.debug = bits
        assert lib.hsp_conv1d_mfma_plan(ctypes.byref(a), ctypes.byref(plan)) == L_.EINVAL
    assert "HSP_CONV_DEBUG" not in open(os.path.join(os.path.dirname(L_.__file__), "hip_layers.py")).read()


def test_wn_layers_and_dit_ffn_reach_their_named_entry_points_only_under_survey_abi(monkeypatch):
    """modules.WN / DiTConVBlock issue their own launches by default; under HSP_SURVEY_ABI a WN layer goes through
    hsp_wn_layer_f32 (in-layer, res, skip: three structs, the last layer two) and the DiT FFN through hsp_ffn_conv_f32
    (SURVEY.md 8(b) names).  Host logic only: device pointers replaced by host addresses, nothing is launched."""
    from megatts2_hierspeechpp_amd import _lib as L_, hip_layers as HL, modules as M
    monkeypatch.setattr(L_, "ptr", lambda t: None if t is None else t.data_ptr())
    monkeypatch.setattr(L_, "fptr", lambda t: None if t is None else t.data_ptr())
    monkeypatch.setattr(HL, "_zeros", lambda dev: torch.zeros(64))
    seen, plain = [], []
    monkeypatch.setattr(HL, "launch_group", lambda kind, fn, structs, *extra: seen.append((kind, [e is not None for e in structs])))
    monkeypatch.setattr(M.Fh, "mask_mul", lambda x, m: x)
    monkeypatch.setattr(M.Fh, "layernorm_mod", lambda x, *a, **k: torch.empty_like(x))
    monkeypatch.setattr(M.Fh, "mha", lambda q, k, v, *a, **kw: torch.empty_like(q))
    monkeypatch.setattr(M.Fh, "mha_proj_supported", lambda *a: False)   # the attention launches are not this test's subject
    monkeypatch.setattr(HL, "_launch", lambda kind, fn, a, fl, nb, soft=False, keep=(): (
        HL._DEFER.append((a, fl, nb, keep)) if HL._DEFER is not None else plain.append(kind)) and 0 or 0)
    wn = M.WN(192, 5, 1, 3, gin_channels=0)
    blk = M.DiTConVBlock(192, 2, mlp_ratio=4.0, kernel=5)
    for m in list(wn.modules()) + list(blk.modules()):
        if isinstance(m, HL.Conv1d):
            m._w = torch.zeros(m.k * m.cin * m.M)
            m._b = torch.zeros(m.cout)
    x, mask = torch.zeros(2, 192, 36), torch.ones(2, 1, 36)
    monkeypatch.setattr(HL, "SURVEY_ABI", False)
    wn(x, mask)
    blk(x, None, mask, mod=torch.zeros(2, 6 * 192, 1), premasked=True)
    assert seen == [] and len(plain) >= 3 * 2 + 2           # no grouped call: every layer launches itself
    monkeypatch.setattr(HL, "SURVEY_ABI", True)
    plain.clear()
    wn(x, mask)
    blk(x, None, mask, mod=torch.zeros(2, 6 * 192, 1), premasked=True)
    assert [k for k, _ in seen] == ["hsp_wn_layer_f32"] * 3 + ["hsp_ffn_conv_f32"], seen
    assert seen[0][1] == [True, True, True] and seen[2][1] == [True, False, True]   # last WN layer: skip only


def test_dftseg_tables_give_the_128_point_real_transform_and_its_inverse():
    """hsp_dftseg_tables_f32 (host side, no GPU): the radix-2 tables of csrc/hsp_dftseg.hip, recombined exactly as the two
    kernels do it (forward: X[k] = E[k] + W^k O[k], X[64 - k] = conj(E[k] - W^k O[k]), bin 0's lane carrying DC, Nyquist
    and bin 32; inverse: E^ = X[k] + conj(X[64 - k]), O^ = (X[k] - conj(X[64 - k])) W^-k), are numpy's rfft / irfft."""
    from megatts2_hierspeechpp_amd import _lib as L

    f = np.zeros(4160, np.float32)
    fi = np.zeros(4160, np.float32)
    assert L.lib().hsp_dftseg_tables_f32(f.ctypes.data, fi.ctypes.data) == 0
    M = f[:4096].reshape(64, 64).astype(np.float64)
    c, s = f[4096:4128].astype(np.float64), f[4128:4160].astype(np.float64)
    assert np.array_equal(f[4096:], fi[4096:])
    x = np.random.default_rng(0).standard_normal(128)
    E, O = M @ x[0::2], M @ x[1::2]
    Xr, Xi = np.zeros(65), np.zeros(65)
    for wh in range(2):
        for i in range(16):
            k = 16 * wh + i
            er, ei, orr, oi = E[32 * wh + i], E[32 * wh + 16 + i], O[32 * wh + i], O[32 * wh + 16 + i]
            tr, ti = c[k] * orr + s[k] * oi, c[k] * oi - s[k] * orr
            if k == 0:
                Xr[0], Xr[64], Xr[32], Xi[32] = er + tr, er - tr, ei, -oi
            else:
                Xr[k], Xi[k], Xr[64 - k], Xi[64 - k] = er + tr, ei + ti, er - tr, ti - ei
    ref = np.fft.rfft(x)
    assert np.abs(Xr - ref.real).max() < 2e-6 and np.abs(Xi - ref.imag).max() < 2e-6
    Mi = fi[:4096].reshape(64, 64).astype(np.float64)
    Eb, Ob = np.zeros(64), np.zeros(64)
    for k in range(32):
        if k == 0:
            xr, xi, yr, yi = Xr[0], Xr[64], Xr[32], Xi[32]
            Eb[0], Eb[32], Ob[0], Ob[32] = xr + xi, 2 * yr, xr - xi, -2 * yi
        else:
            xr, xi, yr, yi = Xr[k], Xi[k], Xr[64 - k], Xi[64 - k]
            dr, di = xr - yr, xi + yi
            Eb[k], Eb[32 + k], Ob[k], Ob[32 + k] = xr + yr, xi - yi, dr * c[k] - di * s[k], dr * s[k] + di * c[k]
    y = np.zeros(128)
    y[0::2], y[1::2] = Mi @ Eb, Mi @ Ob
    assert np.abs(y - x).max() < 1e-6


def test_frequency_domain_form_has_a_supported_predicate():
    """hsp_dftseg_supported (include/hsp.h): the transform kernels address a launch's spectrum with 32-bit byte offsets
    (C Np <= 8.4 M), take dilations up to 8 and int item counts; the reference has no batch or length limit
    (hierspeechpp_speechsynthesizer.py:377-386,635-651), so fft_wins consults the predicate and keeps the direct conv
    beyond it.  Host logic only: the predicate reads geometry, no device."""
    import ctypes as C
    from megatts2_hierspeechpp_amd import _lib as L
    from megatts2_hierspeechpp_amd.hip_layers import Conv1d

    def geom(C_, k, d, B, Lx):
        lay = Conv1d(C_, C_, k, dilation=d, padding=(k - 1) * d // 2, weight_norm=True)
        lay.enable_fft()
        da = lay._fft_args(B, Lx)
        da.xf_bs = 2 * C_ * da.Np
        return lay, da

    lib = L.lib()
    for C_, k, d, B, Lx in ((512, 11, 5, 32, 800), (128, 7, 1, 32, 16000), (64, 11, 3, 32, 32000), (256, 11, 1, 1, 50)):
        lay, da = geom(C_, k, d, B, Lx)
        assert lib.hsp_dftseg_supported(C.byref(da)) == 1 and lay.fft_supported(B, Lx)
    # B = 482 x 4 s at the 128-channel stage: 482 x 136 segments x 128 channels > 8.4 M spectrum columns x rows
    lay, da = geom(128, 11, 1, 482, 16000)
    assert 128 * da.Np > 0xffffffff // 1024 and lib.hsp_dftseg_supported(C.byref(da)) == 0 and not lay.fft_supported(482, 16000)
    lay, da = geom(128, 11, 1, 481, 16000)
    assert lib.hsp_dftseg_supported(C.byref(da)) == 1
    # one 17-minute utterance at 64 channels; a dilation the kernels do not carry; a wrong segment count
    assert not geom(64, 11, 1, 1, 16_200_000)[0].fft_supported(1, 16_200_000)
    assert not geom(64, 11, 9, 2, 4000)[0].fft_supported(2, 4000)
    lay, da = geom(64, 11, 1, 2, 4000)
    da.nseg += 1
    assert lib.hsp_dftseg_supported(C.byref(da)) == 0
    assert lib.hsp_dftseg_supported(None) == 0


def test_three_product_weights_reproduce_the_complex_product():
    """The algebra of hsp_cprod3_f32 in numpy (float64): with conj(W) = a + i b and the matrices (a + b, a, b),
    k1 - k3 / k1 + k2 are the real / imaginary part of conj(W) X; slot 0 with (-E0, -O0) in the two planes and the
    matrices (0, W_nyquist, W_dc) gives (W_dc DC, W_nyquist Nyquist) -- what hsp_dftseg_inv_f32 expects there."""
    rng = np.random.default_rng(5)
    C_, k, n = 6, 11, 9
    w = rng.standard_normal((C_, C_, k))                        # [co, ci, tap]
    W = np.fft.rfft(np.pad(w, ((0, 0), (0, 0), (0, 128 - k))), axis=2)   # [co, ci, 65]
    X = rng.standard_normal((C_, n)) + 1j * rng.standard_normal((C_, n))
    for b in (1, 17, 63):
        a_, b_ = W[:, :, b].real, -W[:, :, b].imag              # conj(W) = a + i b
        k1, k2, k3 = (a_ + b_) @ X.real, a_ @ (X.imag - X.real), b_ @ (X.real + X.imag)
        ref = np.conj(W[:, :, b]) @ X
        assert np.allclose(k1 - k3, ref.real) and np.allclose(k1 + k2, ref.imag)
    E0, O0 = rng.standard_normal((C_, n)), rng.standard_normal((C_, n))
    Wdc, Wny = W[:, :, 0].real, W[:, :, 64].real
    xr, xi = -E0, -O0
    k1, k2, k3 = 0.0 * xr, Wny @ (xi - xr), Wdc @ (xr + xi)
    assert np.allclose(k1 - k3, Wdc @ (E0 + O0)) and np.allclose(k1 + k2, Wny @ (E0 - O0))


def test_isa_lint_finds_a_fragment_used_before_its_wait(tmp_path):
    """tools/check_isa.py (make check-isa): an instruction that touches the destination of an inline-asm ds_read before the
    s_waitcnt that validates it is reported; the same code with the wait in place, a read issued on the loop-continue side
    of a scalar test only, and a partial lgkmcnt(N) wait are not."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa
    head = "kern:\n"
    rd = lambda reg, addr: f"\t;;#ASMSTART\n\tds_read_b32 {reg}, {addr} offset:0\n\t;;#ASMEND\n"
    wait = lambda n: f"\t;;#ASMSTART\n\ts_waitcnt lgkmcnt({n})\n\t;;#ASMEND\n"
    cases = {
        "copy_before_wait": (head + rd("v10", "v1") + "\tv_mov_b32_e32 v20, v10\n" + wait(0) + "\ts_endpgm\n", 1),
        "spill_before_wait": (head + rd("v10", "v1") + "\tscratch_store_dword off, v10, s0\n" + wait(0) + "\ts_endpgm\n", 1),
        "mfma_before_wait": (head + rd("v10", "v1") + "\tv_mfma_f32_32x32x2_f32 v[20:35], v10, v2, v[20:35]\n" + wait(0)
                             + "\ts_endpgm\n", 1),
        "clean": (head + rd("v10", "v1") + "\tv_add_f32_e32 v3, v4, v5\n" + wait(0) + "\tv_mov_b32_e32 v20, v10\n\ts_endpgm\n", 0),
        "partial_wait": (head + rd("v10", "v1") + rd("v11", "v1") + wait(1) + "\tv_mov_b32_e32 v20, v10\n" + wait(0)
                         + "\tv_mov_b32_e32 v21, v11\n\ts_endpgm\n", 0),
        "partial_wait_too_early": (head + rd("v10", "v1") + rd("v11", "v1") + wait(1) + "\tv_mov_b32_e32 v21, v11\n" + wait(0)
                                   + "\ts_endpgm\n", 1),
        # the consumer loops: the next chunk's first reads sit behind `c + 1 < nchunks`, the same scalars as the loop exit
        "loop_exit": (head + ".LBB0_1:\n\ts_add_i32 s8, s8, 1\n\ts_cmp_lt_i32 s8, s56\n\ts_cbranch_scc0 .LBB0_2\n"
                      + rd("v10", "v1") + ".LBB0_2:\n\tv_add_f32_e32 v3, v4, v5\n\ts_cmp_eq_u32 s8, s56\n\ts_cbranch_scc1 .LBB0_3\n"
                      + wait(0) + "\tv_mov_b32_e32 v20, v10\n\ts_branch .LBB0_1\n.LBB0_3:\n\tv_mov_b32_e32 v10, v7\n\ts_endpgm\n", 0),
    }
    # ... and the same with the exit test's operands the other way round (what hipcc emits for hsp_cprod3.hip)
    cases["loop_exit_swapped"] = (cases["loop_exit"][0].replace("s_cmp_eq_u32 s8, s56", "s_cmp_eq_u32 s56, s8")
                                  .replace("s_cmp_lt_i32 s8, s56\n\ts_cbranch_scc0", "s_cmp_ge_u32 s8, s56\n\ts_cbranch_scc1"), 0)
    for name, (text, want) in cases.items():
        f = tmp_path / f"{name}.s"
        f.write_text(text)
        findings, n = check_isa.lint(str(f))
        assert n >= 1 and len(findings) == want, (name, findings)
