#!/bin/bash
# Timing probes of the hand-placed dK/dV loop (build container): variant libraries nvit_amd/libnvit_hip.so.dkv_<tag> whose
# generated loop lacks one ingredient (RESULTS ARE GARBAGE BY DESIGN - timing only).  bash tools/dkv_variants.sh tag:probe,probe ...
set -e
cd "$(dirname "$0")/../nvit_amd/csrc"
make -j8 >/dev/null
for spec in "$@"; do
  tag=${spec%%:*}; probes=${spec#*:}
  opts=""; case "$probes" in opt=*) opts=${probes#opt=}; probes="";; esac
  rm -rf build_dkv && mkdir -p build_dkv
  for f in core gemm gemm_p gemm_tn_p kohonen rowops weights optim attn_ref misc xgmi patch_embed; do cp -p build/$f.o build_dkv/$f.o; done
  GEN_PROBE=$probes GEN_OPT=$opts python3 gen/gen_attn_dkv32_asm.py > attn_dkv32_asm.inc
  make BUILD=build_dkv OUT=../libnvit_hip.so.dkv_$tag EXTRA="$DKV_EXTRA" >/dev/null
  echo "built libnvit_hip.so.dkv_$tag ($probes)"
done
python3 gen/gen_attn_dkv32_asm.py > attn_dkv32_asm.inc
rm -rf build_dkv
make -j8 >/dev/null
