"""GPU (-m gpu): the HIP product path, called through the C ABI of libhsp.so, against
 (a) the golden vectors captured from the reference (tests/golden),
 (b) the CPU oracle on seeded inputs at sizes that cross tile / chunk boundaries,
 (c) size-independent properties at the full BASELINE.json size (32 x 4 s).
Tolerance: BASELINE.json north_star -- fp32 1e-4 (scaled by the peak magnitude for
tensors that are not audio)."""
import ctypes as C

import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu


def _close(got, ref, name=""):
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert np.isfinite(got).all(), name
    err = float(np.abs(got - ref).max())
    assert err <= H.tol_for(ref), f"{name}: max|hip - ref| = {err:.3e} > {H.tol_for(ref):.1e}"


# ------------------------------------------------------------------ (a) golden vectors
@pytest.mark.parametrize("name", H.fixture_names())
def test_golden(name, device):
    meta, arrays = H.load_fixture(name)
    outs = H.run_hip(meta, arrays, device)
    for i, (o, r) in enumerate(zip(outs, H.outputs(arrays))):
        _close(o, r, f"{name}[{i}]")


# ------------------------------------------------------------ (b) oracle, other sizes
def _amp_case(device, C_, k, L, B, seed):
    from megatts2_hierspeechpp_amd import synth
    from megatts2_hierspeechpp_amd.hierspeechpp_speechsynthesizer import AMPBlock1
    from megatts2_hierspeechpp_amd.hip_layers import finalize
    from oracle import hsp_oracle as O
    mod = AMPBlock1(C_, k, (1, 3, 5))
    sd = {kk: torch.from_numpy(synth.synth_tensor("t." + kk, tuple(v.shape), seed)) for kk, v in mod.state_dict().items()}
    mod.load_state_dict(sd)
    finalize(mod, device)
    x = torch.from_numpy(np.random.default_rng(seed).standard_normal((B, C_, L)).astype(np.float32))
    with torch.no_grad():
        got = mod(x.to(device)).cpu().numpy()
        ref = O.amp_block({"t." + kk: v for kk, v in sd.items()}, "t", x, k).numpy()
    _close(got, ref, f"amp C={C_} k={k} L={L}")


@pytest.mark.parametrize("C_,k,L,B", [
    (32, 3, 1300, 2),    # 32 x 512 tiles, several per utterance, ragged last tile
    (32, 11, 700, 1),
    (64, 7, 900, 2),     # 64 x 256 tiles
    (128, 11, 450, 2),   # 128 x 128 tiles, dil 5 halo 25
    (256, 3, 300, 1),    # 256 x 128 tiles, chunk loop over 256 channels
    (512, 7, 160, 1),    # two row tiles
    (40, 5, 130, 3),     # channel count that is no multiple of the chunk / MFMA block
    (8, 3, 7, 2),        # shorter than the resampler support
])
def test_amp_block_vs_oracle(C_, k, L, B, device):
    _amp_case(device, C_, k, L, B, seed=100 + C_ + k)


@pytest.mark.parametrize("cin,cout,k,u,L", [(64, 32, 8, 4, 300), (48, 24, 11, 5, 77), (16, 8, 4, 2, 1000), (8, 4, 3, 3, 5)])
def test_conv_transpose_vs_oracle(cin, cout, k, u, L, device):
    from megatts2_hierspeechpp_amd import synth
    from megatts2_hierspeechpp_amd.hip_layers import ConvTranspose1d, finalize
    from oracle import hsp_oracle as O
    mod = ConvTranspose1d(cin, cout, k, u, padding=(k - u) // 2, weight_norm=True)
    sd = {kk: torch.from_numpy(synth.synth_tensor("ct." + kk, tuple(v.shape), 5)) for kk, v in mod.state_dict().items()}
    mod.load_state_dict(sd)
    finalize(mod, device)
    x = torch.from_numpy(np.random.default_rng(k).standard_normal((2, cin, L)).astype(np.float32))
    with torch.no_grad():
        got = mod(x.to(device)).cpu().numpy()
        ref = O.conv_transpose1d({"ct." + kk: v for kk, v in sd.items()}, "ct", x, u, (k - u) // 2).numpy()
    _close(got, ref, "convtr")


@pytest.mark.parametrize("cin,cout,k,stride,dil,L", [(1, 192, 9, 4, 1, 803), (32, 1, 7, 1, 1, 999), (5, 3, 3, 2, 2, 41),
                                                    (256, 1024, 1, 1, 1, 1)])
def test_direct_conv_vs_torch(cin, cout, k, stride, dil, L, device):
    """The VALU conv (Cin=1 / Cout=1 / strided / L=1 shapes) against torch's conv1d on CPU."""
    from megatts2_hierspeechpp_amd.hip_layers import Conv1d, finalize
    pad = (k - 1) * dil // 2
    mod = Conv1d(cin, cout, k, stride=stride, padding=pad, dilation=dil)
    g = torch.Generator().manual_seed(3)
    mod.weight.data = torch.randn(cout, cin, k, generator=g) / (cin * k) ** 0.5
    mod.bias.data = torch.randn(cout, generator=g) * 0.1
    finalize(mod, device)
    x = torch.randn(2, cin, L, generator=g)
    ref = torch.nn.functional.conv1d(x, mod.weight.data.cpu(), mod.bias.data.cpu(), stride, pad, dil).numpy()
    got = mod(x.to(device), force_direct=True).cpu().numpy()
    _close(got, ref, "direct")


def test_wn_and_attention_longer_ragged(device):
    """WN (gated MFMA epilogue) and the DiT block (LayerNorm + attention + conv FFN) at
    T = 333 with ragged lengths: more than one tile per utterance, masked tails."""
    from megatts2_hierspeechpp_amd import functional as Fh, modules, synth
    from megatts2_hierspeechpp_amd.hip_layers import finalize
    from oracle import hsp_oracle as O
    T, lens = 333, np.array([333, 201, 64], np.int64)
    rng = np.random.default_rng(9)
    mask_c = O.sequence_mask(torch.from_numpy(lens), T).unsqueeze(1).float()
    wn = modules.WN(192, 5, 1, 4, gin_channels=256)
    sd = {k: torch.from_numpy(synth.synth_tensor("w." + k, tuple(v.shape), 11)) for k, v in wn.state_dict().items()}
    wn.load_state_dict(sd)
    finalize(wn, device)
    x = torch.from_numpy(rng.standard_normal((3, 192, T)).astype(np.float32)) * mask_c
    g = torch.from_numpy(rng.standard_normal((3, 256, 1)).astype(np.float32))
    mask = Fh.sequence_mask(torch.from_numpy(lens).to(device), T)
    with torch.no_grad():
        got = wn(x.to(device), mask, g=g.to(device)).cpu().numpy()
        ref = O.wavenet({"w." + k: v for k, v in sd.items()}, "w", x, mask_c, g, 192, 5, 4).numpy()
    _close(got, ref, "wn T=333")
    blk = modules.DiTConVBlock(192, 2, mlp_ratio=4.0, kernel=5)
    sd = {k: torch.from_numpy(synth.synth_tensor("d." + k, tuple(v.shape), 12)) for k, v in blk.state_dict().items()}
    blk.load_state_dict(sd)
    finalize(blk, device)
    c = torch.from_numpy(rng.standard_normal((3, 192)).astype(np.float32))
    with torch.no_grad():
        got = blk(x.to(device), c.to(device), mask).cpu().numpy()
        ref = O.dit_conv_block({"d." + k: v for k, v in sd.items()}, "d", x.transpose(1, 2), c,
                               mask_c.transpose(1, 2)).transpose(1, 2).numpy()
    _close(got, ref, "dit T=333")


def test_linear_interp_long_sequence_matches_torch_cpu(device):
    """SpeechSR's x3 linear interpolation at the full 4-s length: torch-CPU evaluates the source
    index with a single-rounding fp32 FMA; mul+sub is off by 2e-3 on white noise at L = 64000
    (SURVEY.md §8a A15).  Also the x1.5 factor of speechsr24k."""
    from megatts2_hierspeechpp_amd import functional as Fh
    x = torch.randn(2, 3, 64000, generator=torch.Generator().manual_seed(4))
    for out_len in (192000, 96000):
        ref = torch.nn.functional.interpolate(x, out_len, mode="linear").numpy()
        got = Fh.linear_interp(x.to(device), out_len).cpu().numpy()
        assert np.abs(got - ref).max() < 1e-5


# --------------------------------------------------- (c) properties at the full size
@pytest.fixture(scope="module")
def full_model(device):
    from megatts2_hierspeechpp_amd import synth
    from megatts2_hierspeechpp_amd.hierspeechpp_speechsynthesizer import SynthesizerTrn
    from oracle.hsp_oracle import default_config
    net = SynthesizerTrn(641, 192, **default_config())
    net.load_state_dict({k: torch.from_numpy(synth.synth_tensor(k, tuple(v.shape), 0)) for k, v in net.state_dict().items()})
    net.finalize(device)
    return net


def test_full_size_batch_properties(full_model, device):
    """BASELINE.json configs[1] (32 x 4 s).  The oracle needs minutes per utterance at this
    size, so parity is carried by properties: outputs are finite and inside tanh's range,
    two runs are bit-identical, and every utterance of the batch equals the same
    utterance synthesised alone (utterances are independent: SURVEY.md §8e)."""
    from megatts2_hierspeechpp_amd import synth
    inp = synth.synth_inputs(32, 200, seed=20240)
    d = {k: torch.from_numpy(v).to(device) for k, v in inp.items()}
    with torch.no_grad():
        o1, e1 = full_model.infer(d["mel"], d["w2v"], d["length"], d["f0"], noise=d["noise"])
        o2, _ = full_model.infer(d["mel"], d["w2v"], d["length"], d["f0"], noise=d["noise"])
        assert o1.shape == (32, 1, 64000) and e1.shape == (32, 1, 800)
        assert bool(torch.isfinite(o1).all()) and float(o1.abs().max()) <= 1.0
        assert torch.equal(o1, o2), "the path must be deterministic"
        for b in (0, 17, 31):
            ob, _ = full_model.infer(d["mel"][b:b + 1], d["w2v"][b:b + 1], d["length"][b:b + 1], d["f0"][b:b + 1],
                                     noise=d["noise"][b:b + 1])
            err = float((ob - o1[b:b + 1]).abs().max())
            assert err <= 2e-5, f"utterance {b}: batch vs alone differ by {err:.2e}"


def test_full_size_one_utterance_vs_oracle(full_model, device):
    """One 4-s utterance of the full-size batch against the oracle (about 10 s of CPU)."""
    from megatts2_hierspeechpp_amd import synth
    from oracle import hsp_oracle as O
    torch.set_num_threads(min(16, torch.get_num_threads()))
    inp = synth.synth_inputs(1, 200, seed=77)
    t = {k: torch.from_numpy(v) for k, v in inp.items()}
    sd = {k: v.detach().cpu() for k, v in full_model.state_dict().items()}
    with torch.no_grad():
        ro, re_ = O.synth_infer(sd, O.default_config(), t["mel"], t["w2v"], t["length"], t["f0"], t["noise"])
        go, ge = full_model.infer(*(t[k].to(device) for k in ("mel", "w2v", "length", "f0")), noise=t["noise"].to(device))
    _close(go.cpu().numpy(), ro.numpy(), "infer 1x4s audio")
    _close(ge.cpu().numpy(), re_.numpy(), "infer 1x4s source")


def test_conv_linearity_full_width(device):
    """conv(a*x + b*y) == a*conv(x) + b*conv(y) for the plain MFMA conv at a stage-2 shape."""
    from megatts2_hierspeechpp_amd.hip_layers import Conv1d, finalize
    mod = Conv1d(256, 256, 7, padding=9, dilation=3, bias=False)
    g = torch.Generator().manual_seed(1)
    mod.weight.data = torch.randn(256, 256, 7, generator=g) / (256 * 7) ** 0.5
    finalize(mod, device)
    x, y = (torch.randn(4, 256, 4000, generator=g).to(device) for _ in range(2))
    lhs = mod(1.5 * x - 0.25 * y)
    rhs = 1.5 * mod(x) - 0.25 * mod(y)
    assert float((lhs - rhs).abs().max()) < 2e-5 * float(rhs.abs().max() + 1)


# ------------------------------------------------------------- C-ABI error behaviour
def test_abi_rejects_bad_arguments(device):
    from megatts2_hierspeechpp_amd import _lib as L
    lib = L.lib()
    a = L.Conv1dArgs()   # all-zero: null pointers
    assert lib.hsp_conv1d_mfma_f32(C.byref(a), None) == L.EINVAL
    assert lib.hsp_conv1d_direct_f32(C.byref(a), None) == L.EINVAL
    assert lib.hsp_act1d_snakebeta_f32(None, None, 1, 1, 1, None, None, None, None) == L.EINVAL
    m = L.MhaArgs()
    assert lib.hsp_mha_f32(C.byref(m), None) == L.EINVAL
    x = torch.zeros(1, 4, 8, device=device)
    assert lib.hsp_flip_channels_f32(x.data_ptr(), x.data_ptr(), 1, 4, 8, None) == L.EINVAL  # in-place flip
