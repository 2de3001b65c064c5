# kernel stats of tools/dftseg_eager.py under the current library (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/dftseg_quick
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dftseg_quick -- python3 $R/tools/dftseg_eager.py --reps 2 > $R/gpurun_out/dftseg_quick.log 2>&1 || exit 1
f=$(ls $R/gpurun_out/dftseg_quick/*/*kernel_stats.csv | head -1)
grep -E "dftseg|cprod3" $f | cut -c1-150
