LEGS="--no-cpu-baseline --no-e2e --no-cold-leg --no-dense-leg --no-c4-leg"
for i in 1 2 3; do timeout 200 python bench.py $LEGS 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], sum(d['roofline']['other_kernels_avg_ms'].values()))" 2>&1 | tail -1; done
