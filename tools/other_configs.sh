# round's numbers for the non-headline configs (GPU box):  bash tools/other_configs.sh
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['config']['workload'][:60], '| B', d['config']['per_gpu_batch'], '|', d['value'], 'img/s', d['ms_per_step'], 'ms', 'frac', d['step_mfma_frac'], 'graph' if d.get('graph') else '')"; }
run --config large --batch 64 --steps 5 --warmup 2
run --config base_k --batch 128 --steps 5 --warmup 2
run --config base_p16 --batch 128 --steps 10 --warmup 3
run --config tiny --batch 32 --steps 50 --warmup 10
run --config tiny --batch 32 --steps 50 --warmup 10 --graph
run --config base --batch 64 --steps 8 --warmup 3
run --config base --batch 256 --steps 5 --warmup 2
