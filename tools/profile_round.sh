cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02f/trace -- $CMD > gpurun_out/prof_r02f_trace.log 2>&1 || exit 1
CMD1="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline"
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_r02f/fetch -- $CMD1 > gpurun_out/prof_r02f_fetch.log 2>&1 || exit 1
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_r02f/write -- $CMD1 > gpurun_out/prof_r02f_write.log 2>&1 || exit 1
timeout -k 10 280 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/prof_r02f/mfma -- $CMD1 > gpurun_out/prof_r02f_mfma.log 2>&1 || exit 1
echo profiles done
