# HBM traffic of one bench step from the PMC counters (MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE cannot share a
# pass; rocprofv3 wraps python3 bench.py directly -- no env / shell hop behind `--`).  Run on the GPU box:
#     bash tools/pmc_traffic.sh          then, back in the build container:
#     python tools/pmc_summarize.py gpurun_out profiles/r06_traffic.json
# The step's own stand-alone activation launches (act1d_seg_kernel: exactly one float4 read and one float4 write per
# element, 16-B lanes like the conv kernel's LDS-DMA) are the calibration for FETCH_SIZE -- a kernel whose bytes are
# known independently of the kernel being judged.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# the same launch mix as the timed step and as bench.py's own per-launch pass, serialised on one stream
export HSP_SERIAL_STREAMS=1   # the product step's own launches, one after the other on one stream
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/traffic_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/traffic_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-roofline --no-extra > $R/gpurun_out/traffic_$c.log 2>&1
done
ls $R/gpurun_out | grep traffic
