#!/bin/bash
# One GPU-box call: the whole GPU test suite, smoke, the default bench with --check, and a 2-rank data-parallel rehearsal
# of bench.py (ranks share the GPU over gloo: NOT a benchmark, it shows the `dist` object the 8-GPU run will print).
#   bash tools/round_check.sh [tag]
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_chain.sh
step 1000 gpu_tests_$TAG.log python3 -m pytest tests -m gpu -q -s --durations=12
step 120 smoke_$TAG.log python3 -c "import __graft_entry__ as g; g.smoke()"
step 420 bench_check_$TAG.log python3 bench.py --check
for coll in rccl xgmi; do
  step 300 bench_dp2_${coll}_$TAG.log python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 2 --batch 32 --backend gloo --share-gpu --collective $coll
done
[ -f gpurun_out/parity_margins.json ] && cp gpurun_out/parity_margins.json gpurun_out/${TAG}_parity_margins.json
echo round_check done
