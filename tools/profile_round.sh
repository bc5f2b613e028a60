# rocprofv3 passes of the default bench for the round's committed summaries (GPU box):  bash tools/profile_round.sh r04
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/prof_$TAG
CMD="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- $CMD > gpurun_out/prof_${TAG}_trace.log 2>&1 || exit 1
CMD1="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline"
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- $CMD1 > gpurun_out/prof_${TAG}_fetch.log 2>&1 || exit 1
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- $CMD1 > gpurun_out/prof_${TAG}_write.log 2>&1 || exit 1
timeout -k 10 280 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $D/mfma -- $CMD1 > gpurun_out/prof_${TAG}_mfma.log 2>&1 || exit 1


# summarise on the box (the raw rocprofv3 databases exceed what gpurun copies back) and leave only the summaries
python3 tools/summarize_profile.py $TAG --trace $D/trace --fetch $D/fetch --write $D/write --mfma $D/mfma \
  --cmd "$CMD (PMC passes: --steps 1 --warmup 1)" > gpurun_out/prof_${TAG}_summary.log 2>&1 || exit 1
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
# rocprofv3's OWN output next to the summaries (small): its --stats table verbatim, and the per-dispatch counter CSVs gzipped,
# so the summaries can be re-aggregated from the tool's files
f=$(find $D/trace -name '*kernel_stats.csv' | head -n 1); [ -n "$f" ] && cp "$f" gpurun_out/profiles_$TAG/${TAG}_rocprofv3_kernel_stats.csv
for c in fetch write mfma; do
  f=$(find $D/$c -name '*counter_collection.csv' | head -n 1); [ -n "$f" ] && gzip -9 -c "$f" > gpurun_out/profiles_$TAG/${TAG}_rocprofv3_pmc_$c.csv.gz
done
rm -rf $D
echo profiles done
