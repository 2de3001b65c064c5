#!/usr/bin/env python3
"""Run every golden fixture through the HIP path on cuda:0 and print the error table
(does not stop at the first failure; for kernel bring-up).  `python tools/gpu_check.py [names...]`"""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh  # noqa: E402

names = sys.argv[1:] or Hh.fixture_names()
dev = torch.device("cuda:0")
bad = 0
for n in names:
    meta, arrays = Hh.load_fixture(n)
    try:
        t0 = time.time()
        outs = Hh.run_hip(meta, arrays, dev)
        dt = time.time() - t0
        refs = Hh.outputs(arrays)
        row = []
        ok = True
        for o, r in zip(outs, refs):
            if o.shape != r.shape:
                row.append(f"SHAPE {o.shape} vs {r.shape}")
                ok = False
                continue
            err = float(np.abs(o - r).max())
            nan = bool(np.isnan(o).any())
            ok &= (err <= Hh.tol_for(r)) and not nan
            row.append(f"err {err:.3e} (tol {Hh.tol_for(r):.1e}{', NaN' if nan else ''})")
        print(f"{'PASS' if ok else 'FAIL'} {n:24s} {'; '.join(row)}  [{dt:.1f}s]", flush=True)
        bad += not ok
    except Exception:
        bad += 1
        print(f"ERROR {n}\n{traceback.format_exc()}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
