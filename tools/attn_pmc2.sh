# (the TA_* counters are split over two passes: four of them in one pass exceed the block's counter slots and rocprofv3 aborts)
# memory-path PMC passes over the attention kernels (GPU box): bash tools/attn_pmc2.sh [lib tag]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ -n "$1" ] && export NVIT_LIB=$PWD/nvit_amd/libnvit_hip.so.$1
D=gpurun_out/attn_pmc
rm -rf $D; : > $D.log
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ" \
           "TCP_TCC_READ_REQ_LATENCY TCP_TCP_LATENCY TCP_TOTAL_CACHE_ACCESSES TCP_READ_TAGCONFLICT_STALL_CYCLES" \
           "TCC_HIT TCC_MISS TCC_REQ TCC_TAG_STALL" \
           "TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES" \
           "TA_TA_BUSY TA_FLAT_READ_LDS_WAVEFRONTS" \
           "TCP_TCR_TCP_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TA_TCP_STATE_READ TCP_GATE_EN1"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set -d $D/p$i -- python3 tools/attn_once.py >> $D.log 2>&1 || { echo "pass $i failed"; tail -n 3 $D.log; continue; }
  python3 tools/pmc_dump.py $D/p$i attn
done > gpurun_out/attn_pmc2.txt 2>&1
rm -rf $D
