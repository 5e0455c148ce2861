#!/bin/bash
# SQ / cache counters of the sparse-filter select kernel (config 4 shape, 8 genomes on one GPU)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c4pmc
mkdir -p $O
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $O/p1 -o s -- python scripts/c4_probe.py 8 > $O/p1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/p2 -o s -- python scripts/c4_probe.py 8 > $O/p2.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $O/p3 -o s -- python scripts/c4_probe.py 8 > $O/p3.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/p4 -o s -- python scripts/c4_probe.py 8 > $O/p4.log 2>&1
python - <<'PY'
import csv, glob, json, collections
out = collections.defaultdict(dict)
for d in ("p1", "p2", "p3", "p4"):
    for f in glob.glob(f"gpurun_out/c4pmc/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "k_hash_accept" not in n and "k_hash_keys_sparse" not in n:
                continue
            k = "k_hash_accept4" if "accept4" in n else ("k_hash_accept" if "accept" in n else "k_hash_keys_sparse")
            c, v = r["Counter_Name"], float(r["Counter_Value"])
            out[k][c] = max(out[k].get(c, 0), v)
            if c in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES"):
                out[k]["us_" + c] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(json.dumps(out, indent=1))
PY
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
