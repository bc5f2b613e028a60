# times the attention backward with each timing-probe variant of the hand-placed dK/dV loops (GPU box); MODES selects the form
python3 tools/attn_dkv_asm_ab.py
for t in "$@"; do echo "== variant $t"; TIME_ONLY=1 NVIT_LIB=$PWD/nvit_amd/libnvit_hip.so.dkv_$t python3 tools/attn_dkv_asm_ab.py 2>/dev/null | grep "hand-placed"; done
