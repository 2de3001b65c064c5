#!/usr/bin/env python3
"""PLM loop (B = 16, T = 200) as ONE chain against two / four utterance groups on separate streams inside one hipGraph
(do independent chains fill each other's launch phases?).  Prints ms per call."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import synth  # noqa: E402
from megatts2_hierspeechpp_amd.ttv_v1.t2w2v_transformer import Megatts2PLM1  # noqa: E402

dev = torch.device("cuda:0")
m = Megatts2PLM1()
m.load_state_dict({k: torch.from_numpy(synth.synth_tensor("plm." + k, tuple(v.shape), 7)) for k, v in m.state_dict().items()})
m.finalize(dev)
tc = torch.from_numpy(np.random.default_rng(1).standard_normal((16, 256, 200)).astype(np.float32)).to(dev)
ref = m.infer(tc)
torch.cuda.synchronize()
side = [torch.cuda.Stream() for _ in range(3)]


def run(groups):
    main = torch.cuda.current_stream()
    if groups == 1:
        return [m.infer(tc)]
    n = 16 // groups
    fork = torch.cuda.Event()
    fork.record(main)
    outs = []
    for g in range(groups):
        st = main if g == 0 else side[g - 1]
        with torch.cuda.stream(st):
            if g:
                st.wait_event(fork)
            outs.append(m.infer(tc[g * n:(g + 1) * n]))
            if g:
                ev = torch.cuda.Event()
                ev.record(st)
                main.wait_event(ev)
    return outs


for groups in (1, 2, 4):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run(groups)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            outs = run(groups)
    g.replay()
    torch.cuda.synchronize()
    ok = bool((torch.cat(outs) == ref).all())
    t0 = time.perf_counter()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print(f"{groups} group(s): {1e3 * (time.perf_counter() - t0) / 3:.1f} ms per call; codes equal the one-chain run: {ok}", flush=True)
