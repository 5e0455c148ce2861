#!/usr/bin/env python3
"""Bloom build: the automatic choice against the one-atomic-per-k-mer build and the partitioned build, by genome size and filter size
(warm calls; insert into an empty filter, then one cascade level).  Lines on stdout.   python scripts/bloom_mode_sweep.py"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd.device import BloomFilter, Context, Genome, bf_size_bytes  # noqa: E402

ctx = Context(0)
for mbp in [float(x) for x in os.environ.get("MBPS", "3,10,30,100,300,1000,3000").split(",")]:
    n = int(mbp * 1e6)
    g0 = Genome.synth(ctx, n, 8, 7, 1, 0.005)
    g1 = Genome.synth(ctx, n, 8, 7, 2, 0.005)
    for fpr in (0.025, 0.3):
        _, nb = bf_size_bytes(g0.total_bp, fpr)
        row, pcs = {}, set()
        for mode in ("auto", "atomic", "binned"):
            ctx.bf_build_mode(mode)
            bf = BloomFilter(ctx, nb, 24)
            for rep in range(3):
                bf.clear()
                ctx.sync()
                t = time.time()
                bf.insert(g0)
                ctx.sync()
                t1 = time.time()
                bf.insert_and(g1)
                ctx.sync()
                t2 = time.time()
            row[mode] = (round((t1 - t) * 1e3, 3), round((t2 - t1) * 1e3, 3))
            pcs.add(bf.popcount())
            bf.free()
        ctx.bf_build_mode("auto")
        best_i = min(v[0] for v in row.values())
        best_a = min(v[1] for v in row.values())
        print(f"{mbp:g} Mbp fpr {fpr} filter {nb / 1e6:.1f} MB  (insert ms, level ms)", row, "same bits" if len(pcs) == 1 else "DIFFERENT BITS",
              "auto/best", round(row["auto"][0] / best_i, 2), round(row["auto"][1] / best_a, 2), flush=True)
    g0.free()
    g1.free()
