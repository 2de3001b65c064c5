# (round 6) same-box comparison of the transform kernels before / after the index-arithmetic rewrite:
#   gpurun -- 'bash tools/r06_dftseg_ab.sh'
# kernel stats of tools/dftseg_eager.py under the baseline library (megatts2_hierspeechpp_amd/libhsp_base.so, a build of the
# round-5 kernel sources) and under the current one, then the step A/B (tools/lib_ab.py).
# The baseline library is not kept in the tree: build it from the commit to compare against, e.g.
#   git stash; git checkout 81413a4 -- megatts2_hierspeechpp_amd/csrc include; make -C megatts2_hierspeechpp_amd/csrc lib
#   cp megatts2_hierspeechpp_amd/libhsp.so megatts2_hierspeechpp_amd/libhsp_base.so; git checkout HEAD -- megatts2_hierspeechpp_amd/csrc include; git stash pop; make ... lib
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in base new; do
  if [ $tag = base ]; then export HSP_LIB=$R/megatts2_hierspeechpp_amd/libhsp_base.so; else export HSP_LIB=$R/megatts2_hierspeechpp_amd/libhsp.so; fi
  rm -rf $R/gpurun_out/dftseg_stats_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dftseg_stats_$tag -- python3 $R/tools/dftseg_eager.py --reps 2 > $R/gpurun_out/dftseg_stats_$tag.log 2>&1 || exit 1
  f=$(ls $R/gpurun_out/dftseg_stats_$tag/*/*kernel_stats.csv | head -1)
  cp $f $R/gpurun_out/r06_dftseg_kernel_stats_$tag.csv
  echo "== $tag"; grep -E "dftseg|cprod3" $f | cut -c1-160
done
unset HSP_LIB
cd $R && python tools/lib_ab.py --libs megatts2_hierspeechpp_amd/libhsp_base.so megatts2_hierspeechpp_amd/libhsp.so --rounds 3 --json gpurun_out/r06_ab_dftseg_index.json
