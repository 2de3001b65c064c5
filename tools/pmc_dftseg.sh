# MFMA / VALU / LDS counters of the transform kernels of the frequency-domain convs: three --pmc passes
# over tools/dftseg_eager.py (rocprofv3 wraps python3 directly) -> gpurun_out/pmc_dftseg_{a,b,c}; fold with
#     python tools/pmc_dftseg_summarize.py gpurun_out profiles/r05_dftseg_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_dftseg_a $R/gpurun_out/pmc_dftseg_b $R/gpurun_out/pmc_dftseg_c
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_dftseg_a -- python3 $R/tools/dftseg_eager.py --reps 1 > $R/gpurun_out/pmc_dftseg_a.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_dftseg_b -- python3 $R/tools/dftseg_eager.py --reps 1 > $R/gpurun_out/pmc_dftseg_b.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_dftseg_c -- python3 $R/tools/dftseg_eager.py --reps 1 > $R/gpurun_out/pmc_dftseg_c.log 2>&1
ls $R/gpurun_out/pmc_dftseg_a/* $R/gpurun_out/pmc_dftseg_b/* $R/gpurun_out/pmc_dftseg_c/* | head
tail -n 2 $R/gpurun_out/pmc_dftseg_a.log
