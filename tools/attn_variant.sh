#!/bin/bash
# Build a variant of the library that differs from the product build only in attn_mfma.hip, with the SAME flags the
# Makefile gives that file (incl. its per-file -fno-slp-vectorize) plus the extra ones given here:
#   bash tools/attn_variant.sh <tag> [extra hipcc flags...]   ->  nvit_amd/libnvit_hip.so.<tag>   (use with NVIT_LIB=...)
set -e
tag=$1; shift
cd "$(dirname "$0")/../nvit_amd/csrc"
make -j8 >/dev/null
rm -rf build_var_$tag && mkdir -p build_var_$tag
for f in core gemm gemm_p gemm_tn_p kohonen rowops weights optim attn_ref misc xgmi patch_embed; do cp -p build/$f.o build_var_$tag/$f.o; done
make BUILD=build_var_$tag OUT=../libnvit_hip.so.$tag EXTRA="$*" build_var_$tag/attn_mfma.o ../libnvit_hip.so.$tag >/dev/null
echo built nvit_amd/libnvit_hip.so.$tag
