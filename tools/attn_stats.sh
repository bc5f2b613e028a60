# per-kernel durations of the attention kernels (GPU box): bash tools/attn_stats.sh <lib tag> [ENV=VALUE ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
[ "$tag" != product ] && export NVIT_LIB=$PWD/nvit_amd/libnvit_hip.so.$tag
for kv in "$@"; do export "$kv"; done
D=gpurun_out/attn_stats_tmp
rm -rf $D
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 tools/attn_once.py > $D.log 2>&1 || { tail -n 5 $D.log; exit 1; }
f=$(find $D -name "*kernel_stats.csv" | head -n 1)
[ -z "$f" ] && { echo "no kernel_stats.csv under $D"; find $D | head -n 20; exit 1; }
echo "== $tag $@"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'attn' in r['Name']:
        n=r['Name']; k='dkv' if 'dkv' in n else 'dq' if 'bwd_dq' in n else 'fwd'
        print(f'  {k:4s} calls {r[\"Calls\"]:>3s} avg {float(r[\"AverageNs\"])/1e3:8.1f} us  min {float(r[\"MinNs\"])/1e3:8.1f}')
"
rm -rf $D
