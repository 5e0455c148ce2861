cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
LEGS="--no-cpu-baseline --no-e2e --no-c4-leg --no-cold-leg --no-nruns-leg --no-c5-leg --no-valley-leg --no-dense-leg"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python bench.py --steps 3 --warmup 2 $LEGS > gpurun_out/tl.log 2>&1
F=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
python scripts/timeline.py $F | tail -45
rm -rf gpurun_out/tl
