#!/bin/bash
# Build a variant of the library that differs from the product build only in attn_mfma.hip:
#   bash tools/attn_variant.sh <tag> [extra hipcc flags...]   ->  nvit_amd/libnvit_hip.so.<tag>   (use with NVIT_LIB=...)
set -e
tag=$1; shift
cd "$(dirname "$0")/../nvit_amd/csrc"
make -j8 >/dev/null
mkdir -p build_var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DNVIT_PRODUCT_BUILD -Wno-unused-function -Wno-unused-value -Wno-unused-variable "$@" -c attn_mfma.hip -o build_var/attn_$tag.o
objs=$(for f in core gemm gemm_p gemm_tn_p kohonen rowops weights optim attn_ref misc xgmi patch_embed; do echo -n "build/$f.o "; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnvit_hip.so.$tag $objs build_var/attn_$tag.o
echo built nvit_amd/libnvit_hip.so.$tag
