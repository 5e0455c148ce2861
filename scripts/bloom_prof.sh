cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/bb; mkdir -p $O
timeout 300 python scripts/bloom_bench.py --check > $O/bb.log 2>&1; tail -6 $O/bb.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o s -- python scripts/bloom_bench.py --reps 3 > $O/prof.log 2>&1
F=$(find $O/st -name "*kernel_stats.csv" | head -1); grep -E "k_bin|k_hash|Name" $F | cut -c1-200 | head
rm -rf $O/st
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q -k "bloom or binned or bf or scale" 2>&1 | tail -3
