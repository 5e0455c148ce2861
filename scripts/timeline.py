#!/usr/bin/env python3
"""Kernel timeline of one sketch call in a rocprofv3 --kernel-trace CSV: start offset, duration, idle gap before each
kernel (us).  usage: python scripts/timeline.py <..._kernel_trace.csv> [anchor-kernel-substring] [occurrence]
The call is the one starting at the given occurrence of the anchor kernel (default -2: with bench.py that is the last
TIMED step -- the very last call is bench.py's detail pass, where an event pair sits between all kernels)."""
import csv
import re
import sys


def main():
    path = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "k_hash_select"
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
    rows.sort()
    starts = [i for i, r in enumerate(rows) if anchor in r[2]]
    if not starts:
        print("anchor kernel not found")
        return
    which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
    i0 = starts[which]
    later = [i for i in starts if i > i0]
    i1 = later[0] if later else len(rows)
    t0 = rows[i0][0]
    prev_end = rows[i0 - 1][1] if i0 else t0
    busy = 0
    for s, e, n in rows[i0:i1]:
        m = re.search(r"(k_[a-z_0-9]+|rocprim::[a-zA-Z_:]+|__amd_rocclr_[A-Za-z]+)", n)
        name = m.group(1)[:40] if m else n[:40]
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  {name}")
        busy += e - s
        prev_end = max(prev_end, e)
    print(f"span {(prev_end - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")


if __name__ == "__main__":
    main()
