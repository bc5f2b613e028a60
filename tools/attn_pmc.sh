# PMC passes over the attention kernels (GPU box): bash tools/attn_pmc.sh [lib tag]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ -n "$1" ] && export NVIT_LIB=$PWD/nvit_amd/libnvit_hip.so.$1
D=gpurun_out/attn_pmc
rm -rf $D
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $D/p1 -- python3 tools/attn_once.py > $D.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC -d $D/p2 -- python3 tools/attn_once.py >> $D.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_WAVES -d $D/p3 -- python3 tools/attn_once.py >> $D.log 2>&1 || exit 1
for p in p1 p2 p3; do python3 tools/pmc_dump.py $D/$p attn; done > gpurun_out/attn_pmc.txt 2>&1
rm -rf $D
