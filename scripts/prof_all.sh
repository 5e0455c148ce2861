#!/bin/bash
# All profile passes behind profiles/ (run on the GPU box through gpurun): kernel stats, HBM traffic (FETCH_SIZE and
# WRITE_SIZE in separate passes), SQ instruction counters (three passes).  Each rocprofv3 run uses --kernel-trace only
# next to --pmc.  Summaries land in gpurun_out/ and are copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
B1="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dense-leg"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pw.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq1 -o s -- $B1 > $O/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_sq2 -o s -- $B1 > $O/sq2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_sq3 -o s -- $B1 > $O/sq3.log 2>&1
python profiles/sq_summarize.py $O/pmc_sq1 $O/pmc_sq2 $O/pmc_sq3 > $O/sq_counters.json
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python profiles/pmc_summarize.py "$F" "$W" $O/pmc_traffic "python bench.py --steps 1 --warmup 0 --no-cpu-baseline (3 x 100 Mbp sketched as one batch; pruned step + dense leg + Bloom build)"
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/pmc_sq1 $O/pmc_sq2 $O/pmc_sq3 $O/stats $O/pmc_fetch $O/pmc_write
ls -la $O | tail -12
