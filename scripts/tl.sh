cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-dense-leg > gpurun_out/tl.log 2>&1
F=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
python scripts/timeline.py $F | tail -45
rm -rf gpurun_out/tl
