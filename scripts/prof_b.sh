cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/statsb -o s -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dense-leg > gpurun_out/statsb.log 2>&1
find gpurun_out/statsb -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_b.csv
rm -rf gpurun_out/statsb
python profiles/summarize.py gpurun_out/kernel_stats_b.csv batch | head -40
