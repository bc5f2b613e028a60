#!/bin/bash
# Builds variants of libnvit_hip.so whose attention kernels lack one ingredient (LDS fragment reads, the exponential,
# the LDS-DMA after the first tiles) into nvit_amd/lib_attn_<variant>.so.var; time them with tools/attn_bench.py after
# copying one over nvit_amd/libnvit_hip.so ON THE GPU BOX (results of the cut-down builds are garbage by construction).
set -e
cd "$(dirname "$0")/../../nvit_amd/csrc"
for v in full:NVIT_PROBE_NONE nolds:NVIT_PROBE_ATTN_NOLDS noexp:NVIT_PROBE_ATTN_NOEXP nodma:NVIT_PROBE_ATTN_NODMA; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -D${v#*:} -c attn_mfma.hip -o build/attn_probe.o
  objs=$(for f in core gemm gemm_p gemm_tn_p kohonen rowops weights optim attn_ref misc xgmi patch_embed; do echo -n "build/$f.o "; done)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_attn_${v%%:*}.so.var $objs build/attn_probe.o
done
rm -f build/attn_probe.o
