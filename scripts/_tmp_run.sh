timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q 2>&1 | tail -2
LEGS="--no-cpu-baseline --no-e2e --no-c4-leg --no-cold-leg --no-dense-leg"
for c in 12 13 14 15 16; do
echo c=$c
timeout 120 python bench.py $LEGS --prune-c $c 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['other_kernels_avg_ms'], d['roofline']['uncovered_kmers'])" 2>&1 | tail -1
done
