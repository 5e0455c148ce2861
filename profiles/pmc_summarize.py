#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM section: FETCH_SIZE and
WRITE_SIZE are collected in separate runs, with --kernel-trace only).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python bench.py ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python bench.py ...
  python profiles/pmc_summarize.py <f_counter_collection.csv> <w_counter_collection.csv> profiles/rNN_pmc_traffic "<workload>"

Writes <out>.json (read by bench.py for roofline.traffic) and <out>.md.  Values are raw counters (KB) / 1024 = MB,
the largest launch of each kernel (the whole-genome launches); the guide's gfx950 correction for wide coalesced
streams (FETCH_SIZE reports half of them) is applied by the reader, not here."""
import csv
import json
import re
import sys


def short(name):
    if "rocprim" in name:
        return "rocprim (sort/scan)"
    m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", name)
    if m:
        return m.group(1) if m.group(1) in ("k_hash_select", "k_hash_select_hi", "k_hash_accept4r") else m.group(0)   # (template variants: one row)
    m = re.search(r"(__amd_rocclr_[A-Za-z]+)", name)
    return m.group(1) if m else name[:60]


def collect(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        v = float(r["Counter_Value"]) / 1024.0
        a = out.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1
        a[1] = max(a[1], v)
        a[2] += v
    return out


def main():
    f_csv, w_csv, out, workload = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
    f, w = collect(f_csv, "FETCH_SIZE"), collect(w_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, [0, 0, 0])[1] + w.get(k, [0, 0, 0])[1])):
        kernels[k] = {"fetch_MB_per_launch_max": round(f.get(k, [0, 0, 0])[1], 1),
                      "write_MB_per_launch_max": round(w.get(k, [0, 0, 0])[1], 1),
                      "launches": max(f.get(k, [0])[0], w.get(k, [0])[0])}
    doc = {"workload": workload,
           "units": "MB = FETCH_SIZE/WRITE_SIZE (KB) / 1024, raw counter values, largest launch of each kernel",
           "kernels": kernels}
    with open(out + ".json", "w") as fh:
        json.dump(doc, fh, indent=1)
    with open(out + ".md", "w") as fh:
        fh.write(f"# HBM traffic per launch from PMC counters\n\n{workload}\n\n")
        fh.write("Separate `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` runs; largest launch "
                 "of each kernel, raw counter (KB) / 1024.\n\n| kernel | launches | FETCH_SIZE MB | WRITE_SIZE MB |\n|---|---:|---:|---:|\n")
        for k, v in kernels.items():
            fh.write(f"| {k} | {v['launches']} | {v['fetch_MB_per_launch_max']} | {v['write_MB_per_launch_max']} |\n")


if __name__ == "__main__":
    main()
