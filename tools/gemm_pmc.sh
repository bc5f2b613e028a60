# LDS / issue PMC passes over the GEMM kernels on the Base shapes (GPU box): bash tools/gemm_pmc.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/gemm_pmc
rm -rf $D; : > $D.log
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 250 rocprofv3 --pmc $set -d $D/p$i -- python3 tools/gemm_bench.py >> $D.log 2>&1 || { echo "pass $i failed"; tail -n 3 $D.log; continue; }
  python3 tools/pmc_dump.py $D/p$i gemm
done > gpurun_out/gemm_pmc.txt 2>&1
rm -rf $D
