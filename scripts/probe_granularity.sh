#!/bin/bash
# Which request size reaches HBM for a random 4-byte Bloom probe?  TCC_EA0_RDREQ by size (32 / 64 / 128 B) for the
# no-hashing probe microbenchmark (scripts/probe_ceiling.py) under rocprofv3 --pmc (counters only, --kernel-trace).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/granularity
mkdir -p $O
timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $O/p1 -o g -- python scripts/probe_ceiling.py > $O/probe.json 2> $O/p1.log
timeout 600 rocprofv3 --pmc TCC_BUBBLE_sum TCC_MISS_sum TCC_HIT_sum --kernel-trace --output-format csv -d $O/p2 -o g -- python scripts/probe_ceiling.py > /dev/null 2> $O/p2.log
python - <<'PY'
import csv, glob, json, collections
out = {}
for d in ("p1", "p2"):
    for f in glob.glob(f"gpurun_out/granularity/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_bench_probe" not in r["Kernel_Name"]:
                continue
            key = (r["Counter_Name"])
            out.setdefault(key, []).append(float(r["Counter_Value"]))
# launches come in groups of 4 per filter size (1 warm-up + 3 timed), 10^9 probes each
res = {k: [round(x) for x in v] for k, v in out.items()}
json.dump({"probes_per_launch": 1000000000, "per_launch_counters_in_launch_order": res}, open("gpurun_out/granularity/summary.json", "w"), indent=1)
print(json.dumps({k: v[-3:] for k, v in res.items()}))
PY
rm -rf $O/p1 $O/p2
