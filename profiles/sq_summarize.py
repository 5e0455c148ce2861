#!/usr/bin/env python3
"""Instruction-issue counters per kernel from rocprofv3 PMC passes (each pass: --pmc <counters> --kernel-trace only):

  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_sq1 -o s -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dense-leg
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU ...   (pmc_sq2)
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT ...                                (pmc_sq3)
  python profiles/sq_summarize.py gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_sq3 > profiles/rNN_sq_counters.json

Largest launch of each kernel: the sketch kernels see the batch of three 100 Mbp genomes (300.19 M k-mers), the Bloom
build kernels one genome (100.06 M).  valu_issue_utilisation = wave-level VALU
instructions x 4 clocks / (256 CUs x 4 SIMDs) / launch duration at 2.4 GHz: the share of the chip's VALU issue slots
the kernel used (the launch duration is the one measured under PMC collection)."""
import csv
import glob
import json
import re
import sys

KEEP = ["k_hash_select", "k_hash_select_hi", "k_sparse_win", "k_cand_compact", "k_cand_compact_slots", "k_bin1", "k_bin2", "k_bin3", "k_hash<0>", "k_window_min",
        "k_hash_accept4r", "k_hash_accept4", "k_hash_accept", "k_hash_tiers"]
import os

# k-mers per launch: NTS_PROF_KMERS for the bench command's genomes (round 2: 3 Gbp genomes, one launch sequence each);
# round 1's defaults: a 100 Mbp genome for the Bloom build, the batch of three for the sketch kernels
KMERS = float(os.environ.get("NTS_PROF_KMERS", 100.06e6))
BATCH_KMERS = float(os.environ.get("NTS_PROF_KMERS", 300.19e6))
PER_GENOME = ("k_bin1", "k_bin2", "k_bin3")


def short(n):
    m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", n)
    if not m or "rocprim" in n:
        return None
    return m.group(1) if m.group(1) in ("k_hash_select", "k_hash_select_hi", "k_hash_accept4r", "k_bin1") else m.group(0)   # (template variants: one row)


def main():
    out, dur = {}, {}
    for d in sys.argv[1:]:
        for f in glob.glob(f"{d}/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if not k:
                    continue
                e = out.setdefault(k, {})
                c, v = r["Counter_Name"], float(r["Counter_Value"])
                if v >= e.get(c, 0):
                    e[c] = v
                    if c == "SQ_INSTS_VALU":
                        dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    what = os.environ.get("NTS_PROF_WORKLOAD") or (
        "one 3 Gbp genome per launch (bench.py's default family)" if "NTS_PROF_KMERS" in os.environ
        else "sketch kernels: the batch of three 100 Mbp genomes; k_bin*: one genome")
    res = {"workload": "python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dense-leg under rocprofv3 --pmc "
                       f"(three passes); largest launch of each kernel ({what}); k-mers per launch taken as {BATCH_KMERS:.0f}", "kernels": {}}
    for k in KEEP:
        if k not in out:
            continue
        e = out[k]
        row = {c: int(v) for c, v in e.items()}
        if "SQ_INSTS_VALU" in e:
            row["valu_wave_instructions_per_64_kmers"] = round(e["SQ_INSTS_VALU"] / ((KMERS if k in PER_GENOME else BATCH_KMERS) / 64), 1)
            if k in dur:
                row["launch_us_under_pmc"] = round(dur[k], 1)
                row["valu_issue_utilisation_at_2.4GHz"] = round(e["SQ_INSTS_VALU"] * 4 / (256 * 4) / (dur[k] * 1e-6 * 2.4e9), 3)
        res["kernels"][k] = row
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
