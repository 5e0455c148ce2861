timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q 2>&1 | tail -2
NTS_HI_TPW=3 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
LEGS="--no-cpu-baseline --no-e2e --no-c4-leg --no-cold-leg --no-dense-leg"
for wl in c3 c2; do
timeout 120 python bench.py $LEGS --workload $wl 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['prune_c'], d['roofline']['other_kernels_avg_ms'])" 2>&1 | tail -1
done
timeout 300 python scripts/frag_select.py 2>&1 | grep auto
