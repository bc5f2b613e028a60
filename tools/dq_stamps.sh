#!/bin/bash
# Builds the stamped variant of the compiler-built dQ kernel (build container): nvit_amd/libnvit_hip.so.dq_stamps.
# Read with tools/dq_stamps.py on the GPU box.      [PROBE_PRIO=1 TAG=_prio] bash tools/dq_stamps.sh
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root/nvit_amd/csrc"
make -j8 >/dev/null
python3 "$root/tools/probes/make_dq_stamps_tu.py"
rm -rf build_dq && mkdir -p build_dq
objs=""
for f in core gemm gemm_p gemm_tn_p kohonen rowops weights optim attn_ref misc xgmi patch_embed; do objs="$objs build/$f.o"; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DNVIT_PRODUCT_BUILD -Wall -Wno-unused-function -Wno-unused-variable \
  -fno-slp-vectorize -I. -I"$root/tools/probes" -c "$root/tools/probes/attn_dq_stamps.hip" -o build_dq/attn_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnvit_hip.so.dq_stamps${TAG} $objs build_dq/attn_stamps.o
rm -rf build_dq
echo "built nvit_amd/libnvit_hip.so.dq_stamps${TAG}"
