#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc passes of tools/pmc_dftseg.sh into one table per kernel (sums over all launches of the pass):
    python tools/pmc_dftseg_summarize.py gpurun_out profiles/r05_dftseg_pmc.txt [prefix=pmc_dftseg]
Derived columns (MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 holds a SIMD 64 cycles, a VALU instruction 4; GRBM_GUI_ACTIVE
is summed over the 8 XCDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles):
  mfma_util  = SQ_INSTS_MFMA x 64 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)     share of the matrix pipe's time that is MFMA issue
  valu_util  = (SQ_INSTS_VALU - SQ_INSTS_MFMA) x 4 / the same denominator  share that is other vector issue
  wait_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES                                 share of wave lifetime parked (s_waitcnt / barrier)
  stall_share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                           share stalled at issue"""
import csv
import glob
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
prefix = sys.argv[3] if len(sys.argv) > 3 else "pmc_dftseg"
KEYS = ["dftseg_fwd_kernel<true>", "dftseg_fwd_kernel<false>", "dftseg_fwd_kernel", "dftseg_inv_kernel", "dftseg_pair_kernel",
        "cprod3_kernel", "conv1d_mfma_kernel", "act1d_seg_kernel"]


def cls(name):
    for k in KEYS:
        if k in name:
            return k
    return None


tab, cols = {}, []
for d in sorted(glob.glob(f"{src}/{prefix}_?")):
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    seen = set()
    for row in csv.DictReader(open(max(files, key=os.path.getmtime))):
        k = cls(row["Kernel_Name"])
        if k is None:
            continue
        c = row["Counter_Name"]
        if c not in cols:
            cols.append(c)
        t = tab.setdefault(k, {"launches": {}})
        t[c] = t.get(c, 0.0) + float(row["Counter_Value"])
        t["launches"].setdefault(d, set()).add(row.get("Dispatch_Id", len(seen)))
        seen.add(row.get("Dispatch_Id"))

lines = ["kernel launches " + " ".join(cols) + " VALU_per_MFMA mfma_util valu_util wait_share stall_share"]
for k, t in tab.items():
    n = max(len(s) for s in t["launches"].values())
    g = lambda c: t.get(c, 0.0)
    den = g("GRBM_GUI_ACTIVE") / 8.0 * 1024.0
    mf, va = g("SQ_INSTS_MFMA"), g("SQ_INSTS_VALU")
    extra = [f"{(va - mf) / mf:.1f}" if mf else "-",
             f"{mf * 64 / den:.3f}" if den else "-", f"{(va - mf) * 4 / den:.3f}" if den else "-",
             f"{g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.3f}" if g("SQ_WAVE_CYCLES") else "-",
             f"{g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f}" if g("SQ_WAVE_CYCLES") else "-"]
    lines.append(f"{k} {n} " + " ".join(f"{g(c):.4g}" for c in cols) + " " + " ".join(extra))
head = f"# tools/pmc_dftseg.sh: rocprofv3 --pmc over tools/dftseg_eager.py (B = 32, the Generator's stage shapes; one eager pass per counter set), summed per kernel.\n"
open(dst, "w").write(head + "\n".join(lines) + "\n")
print("\n".join(lines))
