#!/bin/bash
# Builds the stamped variant of the hand-placed dK/dV kernel (build container): nvit_amd/libnvit_hip.so.dkv_stamps =
# the product objects with attn_mfma.o replaced by tools/probes/attn_dkv_stamps.hip (the product TU + the stamped copy).
# Read with tools/dkv_stamps.py on the GPU box.      [GEN_PROBE=inloop,nodma,...] [GEN_OPT=ring4,...] [PROBE_ALONE=1] [TAG=_x] bash tools/dkv_stamps.sh   (inloop: also the per-tile waits)
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root/nvit_amd/csrc"
make -j8 >/dev/null
GEN_PROBE=stamps${GEN_PROBE:+,$GEN_PROBE} GEN_OPT=$GEN_OPT python3 gen/gen_attn_dkv32_asm.py > "$root/tools/probes/attn_dkv32_stamps.inc"
python3 "$root/tools/probes/make_dkv_stamps_tu.py"
rm -rf build_dkv && mkdir -p build_dkv
objs=""
for f in core gemm gemm_p gemm_tn_p kohonen rowops weights optim attn_ref misc xgmi patch_embed; do objs="$objs build/$f.o"; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DNVIT_PRODUCT_BUILD -Wall -Wno-unused-function -Wno-unused-variable \
  -fno-slp-vectorize -I. -I"$root/tools/probes" -c "$root/tools/probes/attn_dkv_stamps.hip" -o build_dkv/attn_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnvit_hip.so.dkv_stamps${TAG} $objs build_dkv/attn_stamps.o
rm -rf build_dkv
echo "built nvit_amd/libnvit_hip.so.dkv_stamps${TAG}"
