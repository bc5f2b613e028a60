# exact tuning variants of the hand-placed dK/dV loop (GPU box): checks bit-exactness and times each; MODES selects the form
for t in "$@"; do echo "== variant $t"; NVIT_LIB=$PWD/nvit_amd/libnvit_hip.so.dkv_$t python3 tools/attn_dkv_asm_ab.py 2>/dev/null | grep -E "BIT-EXACT|MISMATCH|backward"; done
echo "== product library"; python3 tools/attn_dkv_asm_ab.py 2>/dev/null | grep -E "BIT-EXACT|MISMATCH|backward"
