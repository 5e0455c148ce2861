cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/bb; mkdir -p $O
for FAM in uniform assembly-like; do
timeout 300 python scripts/bloom_bench.py --family $FAM --reps 3 --and-levels 2 > $O/bb_$FAM.log 2>&1; tail -8 $O/bb_$FAM.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$FAM -o s -- python scripts/bloom_bench.py --family $FAM --reps 3 --and-levels 2 > $O/prof_$FAM.log 2>&1
F=$(find $O/st_$FAM -name "*kernel_stats.csv" | head -1); cp $F $O/kernel_stats_$FAM.csv; python profiles/summarize.py $F "bloom build, $FAM family, 3 Gbp: 3 inserts + 2 cascade levels" | head -14
rm -rf $O/st_$FAM
done
