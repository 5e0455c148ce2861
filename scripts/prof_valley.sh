#!/bin/bash
# SQ counters, HBM traffic and the kernel trace of the tiered selection (k_hash_tiers): one 3 Gbp genome against the common filter of
# three genomes at 10 % (scripts/valley_bench.py).  Summaries land in gpurun_out/prof/ and are copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof
mkdir -p $O
export MODES=tiers GENOMES=${GENOMES:-3} DIV=${DIV:-0.10}
CMD="python scripts/valley_bench.py"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/v_trace -o s -- $CMD > $O/v_trace.log 2>&1
python profiles/summarize.py $O/v_trace/*kernel_stats.csv > $O/valley_kernel_stats.md 2>/dev/null || cp $O/v_trace/*kernel_stats.csv $O/valley_kernel_stats.csv
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/v_sq1 -o s -- $CMD > $O/vsq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/v_sq2 -o s -- $CMD > $O/vsq2.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/v_sq3 -o s -- $CMD > $O/vsq3.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/v_f -o s -- $CMD > $O/vf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/v_w -o s -- $CMD > $O/vw.log 2>&1
NTS_PROF_KMERS=3000000000 NTS_PROF_WORKLOAD="one 3 Gbp genome against the common filter of $GENOMES genomes at $DIV" python profiles/sq_summarize.py $O/v_sq1 $O/v_sq2 $O/v_sq3 > $O/valley_sq_counters.json
python - <<PY > $O/valley_raw_counters.json
import csv, glob, json
out = {}
for d in ("v_sq1", "v_sq2", "v_sq3", "v_f", "v_w"):
    for f in glob.glob(f"$O/{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_hash_tiers" in r["Kernel_Name"]:
                e = out.setdefault(r["Counter_Name"], [])
                e.append(float(r["Counter_Value"]))
print(json.dumps({k: {"max": max(v), "launches": len(v)} for k, v in out.items()}, indent=1))
PY
rm -rf $O/v_sq1 $O/v_sq2 $O/v_sq3 $O/v_f $O/v_w $O/v_trace
cat $O/valley_sq_counters.json $O/valley_raw_counters.json
