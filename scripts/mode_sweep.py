#!/usr/bin/env python3
"""Which way is fastest where?  One 3 Gbp genome, with and without its family's filter, windows from 16 to 400, every sketch mode forced
in turn next to the automatic choice (same lists whichever way: counts compared).   python scripts/mode_sweep.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd.device import BloomFilter, Context, Genome, bf_size_bytes, sketch  # noqa: E402

ctx = Context(0)
n = int(float(os.environ.get("MBP", "3000")) * 1e6)
g0 = Genome.synth(ctx, n, 24, 20240207, 1, 0.005)
g1 = Genome.synth(ctx, n, 24, 20240207, 2, 0.005)
_, nb = bf_size_bytes(g0.total_bp, 0.025)
bf = BloomFilter(ctx, nb, 24)
bf.insert(g0)
bf.insert_and(g1)
ctx.trim_bf_build()
out = {}
for filt, tag in ((None, "no filter"), (bf, "filter")):
    for w in [int(x) for x in os.environ.get("WS", "16,33,63,64,100,150,199,200,300,400").split(",")]:
        row = {}
        counts = set()
        for label, mode, tiers in (("auto", "auto", "auto"), ("pruned", "pruned", "never"), ("dense", "dense", "never"), ("tiers", "auto", "always")):
            ctx.sketch_mode(mode)
            ctx.sketch_tiers(tiers)
            try:
                sketch(ctx, g1, 24, w, filt).free()
                ctx.sync()
                t = time.time()
                mx = sketch(ctx, g1, 24, w, filt)
                c = len(mx)
                ctx.sync()
                dt = time.time() - t
                mx.free()
                counts.add(c)
                row[label] = round(dt * 1e3, 2)
            except Exception as exc:                             # noqa: BLE001
                row[label] = "failed: " + str(exc)[:80]
        row["same count"] = len(counts) == 1
        out[f"{tag} w={w}"] = row
        print(tag, w, row, flush=True)
ctx.sketch_mode("auto")
ctx.sketch_tiers("auto")
