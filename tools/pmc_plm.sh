# MFMA / LDS counters of the PLM loop's kernels (eager launches, one call): two --pmc passes -> gpurun_out/pmc_plm_{a,b}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_plm_a $R/gpurun_out/pmc_plm_b
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_plm_a -- python3 $R/tools/plm_bench.py --no-graph --reps 1 > $R/gpurun_out/pmc_plm_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_plm_b -- python3 $R/tools/plm_bench.py --no-graph --reps 1 > $R/gpurun_out/pmc_plm_b.log 2>&1
ls $R/gpurun_out/pmc_plm_a/* $R/gpurun_out/pmc_plm_b/* | head
tail -2 $R/gpurun_out/pmc_plm_a.log
