# PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the dominant stage of extra_configs.tts_b16, sr48_b32, vc_b1_4s (the
# wav2vec2 producer) and tts_prompt_denoise (the denoiser):
#     bash tools/pmc_traffic_extra.sh        then, in the build container:
#     python tools/pmc_summarize_extra.py gpurun_out profiles/r06_traffic_extra.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in tts sr48 vc_w2v denoiser; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/trafficx_${w}_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/trafficx_${w}_$c -- python3 $R/tools/pmc_extra.py $w > $R/gpurun_out/trafficx_${w}_$c.log 2>&1
    tail -1 $R/gpurun_out/trafficx_${w}_$c.log
  done
done
ls $R/gpurun_out | grep trafficx
