#!/bin/bash
# The config-4 part of scripts/prof_all.sh alone: SQ counters and HBM traffic of the sparse-filter select kernel (k_hash_accept4r),
# 8 x 3 Gbp at 10 % on one GPU.  Summaries land in gpurun_out/prof/ and are copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof
mkdir -p $O
LEGS="--no-cpu-baseline --no-e2e --no-c4-leg --no-cold-leg --no-nruns-leg"
C4="python bench.py --workload c4 --steps 1 --warmup 0 $LEGS --no-dense-leg"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/c4_sq1 -o s -- $C4 > $O/c4sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/c4_sq2 -o s -- $C4 > $O/c4sq2.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/c4_sq3 -o s -- $C4 > $O/c4sq3.log 2>&1
NTS_PROF_KMERS=3000000000 NTS_PROF_WORKLOAD="config 4 on one GPU: 8 x 3 Gbp at 10 %, one genome per launch" python profiles/sq_summarize.py $O/c4_sq1 $O/c4_sq2 $O/c4_sq3 > $O/c4_sq_counters.json
rm -rf $O/c4_sq1 $O/c4_sq2 $O/c4_sq3
cat $O/c4_sq_counters.json
