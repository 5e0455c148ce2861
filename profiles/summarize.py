#!/usr/bin/env python3
"""Turn a rocprofv3 `*_kernel_stats.csv` into a short markdown table (kernel names shortened).
usage: python profiles/summarize.py <kernel_stats.csv> "<title>" > profiles/<name>.md"""
import csv
import re
import sys


def short(name):
    m = None if "rocprim" in name else re.search(r"(k_[a-z_0-9]+)(<\d+>)?", name)
    if m:
        mode = {"k_hash<0>": " (keys: hash+probe or hash only)", "k_hash<1>": " (bloom insert)",
                "k_hash<2>": " (bloom cascade)"}.get(m.group(0), "")
        return m.group(0) + mode
    m = re.search(r"(radix_sort_[a-z_]+|merge_sort_[a-z_]+|transform_impl|__amd_rocclr_[A-Za-z]+|scan_[a-z_]+)", name)
    return ("rocprim::" if "rocprim" in name else "") + (m.group(1) if m else name[:60])


def main():
    path, title = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    agg = {}
    for r in csv.DictReader(open(path)):
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
    tot = sum(v[1] for v in agg.values())
    print(f"# {title}\n")
    print("| kernel | calls | avg us | total ms | % |")
    print("|---|---:|---:|---:|---:|")
    for k, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k} | {c} | {ns / c / 1e3:.1f} | {ns / 1e6:.2f} | {100 * ns / tot:.1f} |")


if __name__ == "__main__":
    main()
