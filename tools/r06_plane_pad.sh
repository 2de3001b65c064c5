# (round 6 experiment) transform kernels under three plane strides of the spectrum (HSP_FFT_PLANE_PAD floats of padding per bin)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pad in 0 64 1088; do
  export HSP_FFT_PLANE_PAD=$pad
  rm -rf $R/gpurun_out/dftseg_pad_$pad
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dftseg_pad_$pad -- python3 $R/tools/dftseg_eager.py --reps 2 > $R/gpurun_out/dftseg_pad_$pad.log 2>&1 || exit 1
  f=$(ls $R/gpurun_out/dftseg_pad_$pad/*/*kernel_stats.csv | head -1)
  echo "== pad $pad"; grep -E "dftseg|cprod3" $f | cut -c1-150
done
