#!/usr/bin/env python3
"""Fold tools/pmc_traffic_extra.sh's passes into profiles/r06_traffic_extra.json (keyed by the kernel-source hash):
per workload and kernel class the launches and the FETCH_SIZE / WRITE_SIZE KB of ONE pass (the script runs two).
tools/bench_extra.py quotes `roofline.traffic` of extra_configs from it when the hash matches."""
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megatts2_hierspeechpp_amd.build import source_id  # noqa: E402

CLASSES = ["conv1d_mfma_kernel", "cprod3_kernel", "wspec_kernel", "dftseg_fwd_kernel", "dftseg_inv_kernel", "dftseg_pair_kernel",
           "rgemm_kernel", "bgemm_kernel", "mha_proj_kernel", "act1d_seg_kernel", "act1d_kernel",
           "mha_tok_kernel", "mha_mfma_kernel", "mha_kernel", "layernorm", "conv1d_cout1_kernel", "conv1d_direct_kernel",
           "linear_interp", "plm_embed", "argmax"]
PASSES = 2
WORKLOADS = ("tts", "sr48", "vc_w2v", "denoiser")


def cls(name):
    for c in CLASSES:
        if c in name:
            return c
    return "other"


res = {"kernel_source_sha16": source_id(), "passes_in_run": PASSES,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/pmc_extra.py; KB per ONE pass of the stage. "
               "FETCH_SIZE under-reports 16-B-per-lane streams by the factor calibrated in r05_traffic.json (conv / LDS-DMA token "
               "GEMM); the register-path GEMM reads 4 B per lane, where round 1 measured 0.90 of the true bytes."}
for w in WORKLOADS:
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(f"{src}/trafficx_{w}_{counter}/**/*counter_collection.csv", recursive=True)
        assert files, f"no counter csv for {w} {counter}"
        for row in csv.DictReader(open(max(files, key=os.path.getmtime))):
            if row["Counter_Name"] != counter:
                continue
            k = out.setdefault(cls(row["Kernel_Name"]), {"launches": 0, "FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0})
            k[f"{counter}_KB"] += float(row["Counter_Value"]) / PASSES
            if counter == "FETCH_SIZE":
                k["launches"] += 1
    for k in out.values():
        k["launches"] //= PASSES
    res[w] = out
json.dump(res, open(dst, "w"), indent=1)
for w in WORKLOADS:
    print(w, {k: (v["launches"], round(v["FETCH_SIZE_KB"] / 1e6, 3), round(v["WRITE_SIZE_KB"] / 1e6, 3)) for k, v in res[w].items()})
