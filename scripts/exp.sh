cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/bb; mkdir -p $O
for f in 2 3 4; do
NTS_BIN_FAKE=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o s -- python scripts/bloom_bench.py --reps 3 > $O/prof.log 2>&1
F=$(find $O/st -name "*kernel_stats.csv" | head -1); echo fake $f; grep -E "k_bin1" $F | cut -c1-120 | head
rm -rf $O/st
done
