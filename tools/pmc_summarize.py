#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into profiles/r01_traffic.json
(KB summed per kernel class; bench.py reads conv1d_mfma_bytes_per_step for `roofline.traffic`).
    python tools/pmc_summarize.py gpurun_out profiles/r01_traffic.json"""
import csv
import glob
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
CLASSES = ["conv1d_mfma_kernel", "tokgemm_kernel", "act1d_seg_kernel", "act1d_kernel", "mha_mfma_kernel", "mha_kernel", "layernorm",
           "conv1d_direct_kernel"]


def cls(name):
    for c in CLASSES:
        if c in name:
            return c
    return "other"


out = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"{src}/traffic_{counter}/**/*counter_collection.csv", recursive=True)
    assert files, f"no counter csv for {counter}"
    for row in csv.DictReader(open(sorted(files)[-1])):
        if row["Counter_Name"] != counter:
            continue
        k = out.setdefault(cls(row["Kernel_Name"]), {"launches": 0, "FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0})
        k[f"{counter}_KB"] += float(row["Counter_Value"])
        if counter == "FETCH_SIZE":
            k["launches"] += 1
steps = 2  # bench.py --steps 1 --warmup 0 --no-graph executes the pre-capture step + 1


def _cal(counter, sub):
    """average counter value (bytes) per conv launch of the calibration case"""
    files = glob.glob(f"{src}/{sub}/**/*counter_collection.csv", recursive=True)
    vals = [float(r["Counter_Value"]) * 1024 for r in csv.DictReader(open(sorted(files)[-1]))
            if r["Counter_Name"] == counter and "conv1d_mfma_kernel" in r["Kernel_Name"]]
    return sum(vals) / len(vals)


# calibration case of tools/pmc_traffic.sh: conv 128 -> 128, k = 3, L = 16000, B = 8, no activation, no residual,
# one row tile: reads = input once + 2 halo columns per 128-column tile + weights, writes = output once
CAL_READ = 8 * 128 * 16000 * 4 * (130 / 128) + 3 * 128 * 128 * 4
CAL_WRITE = 8 * 128 * 16000 * 4
fetch_factor = CAL_READ / _cal("FETCH_SIZE", "traffic_cal_f")
write_factor = CAL_WRITE / _cal("WRITE_SIZE", "traffic_cal_w")
conv = out["conv1d_mfma_kernel"]
res = {
    "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 0 --no-graph "
            "--no-roofline` (= 2 executed steps); values are KB summed over all launches of a kernel class, as the "
            "counters report them.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE under-reports wide coalesced reads "
            "(exactly 1/2 for 16-B/lane streaming reads) and other access shapes must be calibrated in the kernel's own "
            "pattern.  Calibration case (same kernel, known bytes: conv 128->128 k3, L=16000, B=8, no activation): the "
            "factors below = expected / reported; conv1d_mfma_bytes_per_step applies them (FETCH x fetch_factor + "
            "WRITE x write_factor)",
    "fetch_factor": fetch_factor, "write_factor": write_factor,
    "steps_in_run": steps, "batch_per_gpu": 32, "frames": 200, "kernels": out,
    "conv1d_mfma_bytes_per_step": (conv["FETCH_SIZE_KB"] * fetch_factor + conv["WRITE_SIZE_KB"] * write_factor) * 1024 / steps,
    "conv1d_mfma_bytes_per_step_uncorrected": (conv["FETCH_SIZE_KB"] + conv["WRITE_SIZE_KB"]) * 1024 / steps,
    "conv1d_mfma_launches_per_step": conv["launches"] / steps,
}
act = out.get("act1d_seg_kernel")
if act:
    # 16-B/lane streaming global_load: FETCH_SIZE reports exactly 1/2 (MI355X_MICROARCH.md, HBM); stores read exactly
    res["act1d_seg_bytes_per_step"] = (2.0 * act["FETCH_SIZE_KB"] + act["WRITE_SIZE_KB"]) * 1024 / steps
    res["act1d_seg_launches_per_step"] = act["launches"] / steps
json.dump(res, open(dst, "w"), indent=1)
print({k: res[k] for k in ("fetch_factor", "write_factor", "conv1d_mfma_bytes_per_step", "conv1d_mfma_launches_per_step")})
