#!/bin/bash
# Builds the four gemm_parts probe binaries into tools/probes/bin/ (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/probes/bin
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS -c nvit_amd/csrc/core.hip -o tools/probes/bin/core.o
for v in full:NVIT_PROBE_NONE nodma:NVIT_PROBE_NO_DMA nomfma:NVIT_PROBE_NO_MFMA noepi:NVIT_PROBE_NO_EPI plainst:NVIT_PROBE_PLAIN_STORE nobdma:NVIT_PROBE_NO_B_DMA nobread:NVIT_PROBE_NO_B_READ nob:NVIT_PROBE_NO_B_DMA+NVIT_PROBE_NO_B_READ; do
  /opt/rocm/bin/hipcc $FLAGS $(echo ${v#*:} | sed 's/+/ -D/g; s/^/-D/') -c tools/probes/gemm_parts.hip -o tools/probes/bin/gp_${v%%:*}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 tools/probes/bin/gp_${v%%:*}.o tools/probes/bin/core.o -o tools/probes/bin/gemm_parts_${v%%:*}
done
for v in full:NVIT_PROBE_NONE nodma:NVIT_PROBE_NO_DMA nomfma:NVIT_PROBE_NO_MFMA; do
  /opt/rocm/bin/hipcc $FLAGS -D${v#*:} -c tools/probes/gemm_tn_parts.hip -o tools/probes/bin/gtn_${v%%:*}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 tools/probes/bin/gtn_${v%%:*}.o tools/probes/bin/core.o -o tools/probes/bin/gemm_tn_parts_${v%%:*}
done
