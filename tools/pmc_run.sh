cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in "--cin 512 --cout 512 --k 11 --len 800" "--cin 512 --cout 512 --k 3 --len 800" "--cin 128 --cout 128 --k 11 --len 16000" "--cin 32 --cout 32 --k 7 --len 64000" "--cin 512 --cout 512 --k 11 --len 800 --act 0" "--cin 512 --cout 512 --k 11 --len 1024 --batch 16"; do
  python3 $R/tools/conv_bench.py $shape
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/tools/conv_bench.py --cin 512 --cout 512 --k 11 --len 800 --reps 3 > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc2 -- python3 $R/tools/conv_bench.py --cin 512 --cout 512 --k 11 --len 800 --reps 3 > $R/gpurun_out/pmc2.log 2>&1
ls $R/gpurun_out/pmc1/* | head
