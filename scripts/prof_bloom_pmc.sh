#!/bin/bash
# SQ counters (VALU / LDS issue, LDS bank conflicts) and HBM traffic of the partitioned Bloom build's three kernels: 3 Gbp genome,
# scripts/bloom_bench.py.  Summaries land in gpurun_out/prof/ and are copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof
mkdir -p $O
CMD="python scripts/bloom_bench.py --family uniform --reps 2 --and-levels 1"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/b_sq1 -o s -- $CMD > $O/bsq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/b_sq2 -o s -- $CMD > $O/bsq2.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/b_sq3 -o s -- $CMD > $O/bsq3.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/b_f -o s -- $CMD > $O/bf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/b_w -o s -- $CMD > $O/bw.log 2>&1
NTS_PROF_KMERS=3000000000 NTS_PROF_WORKLOAD="Bloom build of one 3 Gbp genome (scripts/bloom_bench.py)" python profiles/sq_summarize.py $O/b_sq1 $O/b_sq2 $O/b_sq3 > $O/bloom_sq_counters.json
python profiles/pmc_summarize.py $(ls $O/b_f/*/*counter_collection.csv $O/b_f/*counter_collection.csv 2>/dev/null | head -1) $(ls $O/b_w/*/*counter_collection.csv $O/b_w/*counter_collection.csv 2>/dev/null | head -1) $O/bloom_pmc_traffic "Bloom build of one 3 Gbp genome (scripts/bloom_bench.py)" > /dev/null 2>&1
rm -rf $O/b_sq1 $O/b_sq2 $O/b_sq3 $O/b_f $O/b_w
cat $O/bloom_sq_counters.json
