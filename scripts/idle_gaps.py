#!/usr/bin/env python3
"""GPU idle time inside the timed steps of bench.py, from a rocprofv3 --kernel-trace CSV: the launches between the first and the last
k_hash_select_hi of the trace, their busy time and the gaps between them.
  rocprofv3 --kernel-trace --output-format csv -d out -o t -- python bench.py --steps 5 --warmup 2 --no-... ; python scripts/idle_gaps.py out/*/t_kernel_trace.csv"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
sel = [i for i, r in enumerate(rows) if "k_hash_select_hi" in r[2]]
# the timed region: the last 15 select launches (5 steps x 3 genomes) up to the launch before the following select / end
first = sel[-15]
last = sel[-1]
# extend to the final mail of the last sketch: next k_mail x2 after the last select
end = last
mails = 0
for i in range(last, len(rows)):
    end = i
    if "k_mail" in rows[i][2]:
        mails += 1
        if mails == 2:
            break
seg = rows[first:end + 1]
t0, t1 = seg[0][0], max(r[1] for r in seg)
busy = 0
cur_s, cur_e = seg[0][0], seg[0][1]
gaps = []
for s, e, n in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"wall {(t1 - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms ({100 * (t1 - t0 - busy) / (t1 - t0):.1f} %), launches {len(seg)}")
agg = {}
for g, n in gaps:
    k = n.split("(")[0][-40:]
    a = agg.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += g
for k, (c, g) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  idle before {k:42s} x{c:3d}  {g / 1e3:8.1f} us  ({g / c / 1e3:.1f} us each)")
