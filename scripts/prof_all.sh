cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dense-leg"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats9 -o s -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/stats9.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_sq1 -o s -- $B > gpurun_out/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_sq2 -o s -- $B > gpurun_out/sq2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_sq3 -o s -- $B > gpurun_out/sq3.log 2>&1
python profiles/sq_summarize.py gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_sq3 > gpurun_out/sq_counters9.json
find gpurun_out/stats9 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats9.csv
rm -rf gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_sq3 gpurun_out/stats9
tail -2 gpurun_out/stats9.log | cut -c1-300
