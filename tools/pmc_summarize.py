#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into profiles/r05_traffic.json.
bench.py quotes `roofline.traffic` from that file ONLY when the kernel-source hash, the workload size and the launch mix
recorded here equal the run's own (the library is identified by the hash of its kernel sources).

    python tools/pmc_summarize.py gpurun_out profiles/r05_traffic.json [bench line of the same build]

FETCH_SIZE on gfx950 under-reports wide coalesced reads (exactly 1/2 for 16-B-per-lane streams, MI355X_MICROARCH.md);
three figures are given for every kernel class: raw (as counted), x2 (the guide's literal correction) and calibrated
(x the factor that makes act1d_seg_kernel's FETCH equal its independently known read bytes: B C L 4 per launch)."""
import csv
import glob
import hashlib
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASSES = ["conv1d_mfma_kernel", "cprod3_kernel", "wspec_kernel", "rgemm_kernel", "bgemm_kernel", "mha_proj_kernel",
           "dftseg_fwd_kernel", "dftseg_inv_kernel", "dftseg_pair_kernel", "act1d_seg_kernel", "act1d_kernel",
           "mha_mfma_kernel", "mha_tok_kernel", "mha_kernel", "layernorm", "conv1d_cout1_kernel", "conv1d_direct_kernel"]
B, T = 32, 200
STEPS = 2  # bench.py --steps 1 --warmup 0 --no-graph executes the pre-capture step + 1


def cls(name):
    for c in CLASSES:
        if c in name:
            return c
    return "other"


out = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"{src}/traffic_{counter}/**/*counter_collection.csv", recursive=True)
    assert files, f"no counter csv for {counter}"
    for row in csv.DictReader(open(max(files, key=os.path.getmtime))):  # newest pass (gpurun_out keeps older ones)
        if row["Counter_Name"] != counter:
            continue
        k = out.setdefault(cls(row["Kernel_Name"]), {"launches": 0, "FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0})
        k[f"{counter}_KB"] += float(row["Counter_Value"])
        if counter == "FETCH_SIZE":
            k["launches"] += 1

# algorithmic bytes of the stand-alone activation launches of one vocoder step (one read, one write per element): the
# Activation1d launches that are NOT applied by a forward transform (hsp_dftseg_args.act_*).  Their count and bytes are
# what bench.py's ACT_HOOK summed in the per-launch pass of the SAME build (`roofline_activation` of the line written by
# tools/profile_final.sh, or any bench line given as third argument); the counter rows must show the same launch count.
bench_line = sys.argv[3] if len(sys.argv) > 3 else os.path.join(src, "prof_final_bench.json")
ra = json.loads(open(bench_line).read().strip().splitlines()[-1])["roofline_activation"]
act = out["act1d_seg_kernel"]
n_act = ra["launches_per_step"]
assert act["launches"] == n_act * STEPS, (act["launches"], n_act, "the launch mix is not the one the bench line prices")
act_elems_bytes = ra["algorithmic_mb_per_step"] * 1e6 / 2          # read bytes per step (the hook books read + write)
act_read_bytes = act_elems_bytes * STEPS
fetch_factor = act_read_bytes / (act["FETCH_SIZE_KB"] * 1024)
write_check = act_read_bytes / (act["WRITE_SIZE_KB"] * 1024)


def three(k):
    f, w = k["FETCH_SIZE_KB"] * 1024 / STEPS, k["WRITE_SIZE_KB"] * 1024 / STEPS
    return {"raw": f + w, "x2": 2 * f + w, "calibrated": fetch_factor * f + w, "fetch_raw": f, "write_raw": w}


sys.path.insert(0, ROOT)
from megatts2_hierspeechpp_amd.build import source_id  # noqa: E402
sha = source_id()
conv = out["conv1d_mfma_kernel"]
res = {
    "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `bench.py --steps 1 --warmup 0 --no-graph "
            "--no-roofline` (2 executed steps); per kernel class KB as counted.  FETCH calibrated on the step's own "
            "act1d_seg_kernel launches (known bytes: one fp32 read per element); write_check = known / counted write "
            "bytes of the same launches (the guide: WRITE_SIZE is exact for 16-B stores).",
    "kernel_source_sha16": sha, "batch_per_gpu": B, "frames": T, "steps_in_run": STEPS,
    "fetch_factor": fetch_factor, "write_check": write_check, "kernels": out,
    "conv1d_mfma_launches_per_step": conv["launches"] // STEPS,
    "act1d_launches_per_step": act["launches"] // STEPS,
    "conv1d_mfma_bytes_per_step": three(conv),
    "act1d_seg_bytes_per_step": three(act),
}
for name in ("cprod3_kernel", "rgemm_kernel", "bgemm_kernel", "mha_proj_kernel", "dftseg_fwd_kernel", "dftseg_inv_kernel",
             "dftseg_pair_kernel"):
    if name in out:
        res[name + "_bytes_per_step"] = three(out[name])
        res[name + "_launches_per_step"] = out[name]["launches"] // STEPS
# (round 5) the whole step: calibrated bytes of every class that runs once per step -- the weight
# packing of finalize() (copy kernels in `other`) and the once-per-process derivation of the per-bin matrices
# (wspec_kernel) are NOT per-step traffic and stay out -- against SURVEY.md 8(d)'s algorithmic bytes of the full infer
# path (405 MB per audio-second).  Per class: measured over algorithmic where the bench line of the same build carries the
# algorithmic figure (the direct convs, the channel products, the stand-alone activation).
line = json.loads(open(bench_line).read().strip().splitlines()[-1])
rf = line.get("roofline", {})
per_step = {k: three(v)["calibrated"] for k, v in out.items() if k not in ("other", "wspec_kernel")}
res["step_traffic_gb"] = sum(per_step.values()) / 1e9
res["step_traffic_by_class_gb"] = {k: v / 1e9 for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])}
res["step_algorithmic_gb"] = 405e6 * B * (T / 50.0) / 1e9
res["step_traffic_over_algorithmic"] = res["step_traffic_gb"] / res["step_algorithmic_gb"]
res["not_per_step_gb"] = {k: three(out[k])["calibrated"] * STEPS / 1e9 for k in ("other", "wspec_kernel") if k in out}
alg = {"conv1d_mfma_kernel": rf.get("algorithmic_mb_per_step"), "act1d_seg_kernel": ra["algorithmic_mb_per_step"],
       "cprod3_kernel": rf.get("channel_products", {}).get("algorithmic_mb_per_step")}
res["measured_over_algorithmic"] = {k: per_step[k] / (v * 1e6) for k, v in alg.items() if v and k in per_step}
json.dump(res, open(dst, "w"), indent=1)
print({k: res[k] for k in ("kernel_source_sha16", "fetch_factor", "write_check", "conv1d_mfma_launches_per_step", "act1d_launches_per_step")})
print("conv1d_mfma GB/step", {k: round(v / 1e9, 2) for k, v in res["conv1d_mfma_bytes_per_step"].items()})
print("act1d_seg GB/step", {k: round(v / 1e9, 2) for k, v in res["act1d_seg_bytes_per_step"].items()})
print("step GB", round(res["step_traffic_gb"], 1), "algorithmic", round(res["step_algorithmic_gb"], 1), {k: round(v, 2) for k, v in res["step_traffic_by_class_gb"].items()})
print("measured / algorithmic", {k: round(v, 2) for k, v in res["measured_over_algorithmic"].items()})
