#!/usr/bin/env python3
"""Static instruction mix of every kernel of a device-assembly file (make build/isa/<unit>.s): MFMA, other VALU, SALU,
readlane / writelane (SGPR spills into VGPR lanes), 64-bit address arithmetic, global loads / stores, LDS instructions,
and the metadata's register / spill counts.  `VALU per MFMA` here is STATIC (all paths, prologue and epilogue included);
the executed ratio comes from the SQ_INSTS_VALU / SQ_INSTS_MFMA counters (tools/pmc_dftseg.sh).

    python tools/isa_hist.py megatts2_hierspeechpp_amd/csrc/build/isa/hsp_dftseg.s [substring-of-kernel-name]"""
import collections
import re
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
lines = open(path).read().split("\n")
kern, body = None, collections.defaultdict(list)
for ln in lines:
    m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
    if m:
        kern = m.group(1)
        continue
    if kern and ln.startswith("\t") and not ln.startswith("\t.") and not ln.startswith("\t;"):
        body[kern].append(ln.split()[0])
        if ln.split()[0] == "s_endpgm":
            kern = None
meta = {}
for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)",
                     "\n".join(lines)):
    meta[m.group(1)] = tuple(int(x) for x in m.groups()[1:])
for k, ins in body.items():
    if want not in k:
        continue
    c = collections.Counter(ins)
    mfma = sum(v for n, v in c.items() if n.startswith("v_mfma"))
    valu = sum(v for n, v in c.items() if n.startswith("v_") and not n.startswith("v_mfma"))
    salu = sum(v for n, v in c.items() if n.startswith("s_"))
    lanes = c["v_readlane_b32"] + c["v_writelane_b32"]
    u64 = c["v_lshl_add_u64"] + c["v_mad_u64_u32"] + c["v_mad_i64_i32"]
    gl = sum(v for n, v in c.items() if n.startswith(("global_load", "buffer_load")))
    gs = sum(v for n, v in c.items() if n.startswith(("global_store", "buffer_store")))
    ds = sum(v for n, v in c.items() if n.startswith("ds_"))
    sg, sgs, vg, vgs = meta.get(k, (0, 0, 0, 0))
    print(f"{k[:90]}\n   mfma {mfma}  valu {valu} ({valu / max(mfma, 1):.1f} per mfma, static)  salu {salu}  read/writelane {lanes}  "
          f"64-bit addr ops {u64}  global ld/st {gl}/{gs}  lds {ds}  sgpr {sg} (spilled {sgs})  vgpr {vg} (spilled {vgs})")
