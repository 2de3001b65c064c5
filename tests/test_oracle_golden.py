"""CPU: the oracle (oracle/hsp_oracle.py) against the golden vectors that
tools/make_golden.py captured from the REFERENCE itself.  This is what pins the oracle."""
import numpy as np
import pytest
import torch

import helpers as H

# the heaviest full-width cases run in a few seconds each; keep the CPU suite to minutes
CASES = H.fixture_names()


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(name):
    meta, arrays = H.load_fixture(name)
    torch.set_num_threads(min(8, torch.get_num_threads()))
    with torch.no_grad():
        outs = H.run_oracle(meta, arrays)
    refs = H.outputs(arrays)
    assert len(outs) == len(refs)
    for i, (o, r) in enumerate(zip(outs, refs)):
        o = o.numpy()
        assert o.shape == r.shape
        # oracle == reference up to fp32 re-association (weight-norm fold order): well inside the 1e-4 bar
        tol = 2e-5 * max(1.0, np.abs(r).max())
        if meta["kind"] == "tts_e2e" and i == 1:
            tol = 1.0   # int16 samples: fp32 re-association may move a value across an integer boundary (1 LSB)
        assert np.abs(o - r).max() <= tol, name


def test_act1d_closed_form_matches_oracle():
    """The index-level polyphase statement the HIP kernels implement."""
    from oracle import hsp_oracle as O
    meta, arrays = H.load_fixture("act1d_c4_l37")
    sd = H.oracle_sd(meta)
    x = torch.from_numpy(arrays["x"])[:1, :2]
    pre = meta["prefix"]
    y = O.act1d_closed_form(x, sd[pre + ".act.alpha"][:2], sd[pre + ".act.beta"][:2])
    assert np.abs(y.numpy() - arrays["out0"][:1, :2]).max() < 5e-6


def test_kaiser_filter_matches_reference_buffer():
    """Closed-form 12-tap filter == the buffer values the survey measured in the reference's
    real SpeechSR checkpoint."""
    from megatts2_hierspeechpp_amd.synth import kaiser_sinc_filter12
    h = kaiser_sinc_filter12()
    want = [0.0020289647, 0.0093894657, -0.0255434588, -0.0576573834, 0.1285725832, 0.4432097971]
    assert np.allclose(h[:6], want, atol=2e-7) and np.allclose(h[6:], h[:6][::-1], atol=1e-8)
    assert abs(h.sum() - 1.0) < 1e-6
