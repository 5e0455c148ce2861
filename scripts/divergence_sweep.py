#!/usr/bin/env python3
"""The automatic choice against every forced way across divergence (the filter's accepted share from 0.9 down to next to nothing), for the
default window and a middle one, on whole and on fragmented genomes (3 Gbp).  Lines on stderr, one JSON object on stdout.
python scripts/divergence_sweep.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd.device import BloomFilter, Context, Genome, bf_size_bytes, sketch  # noqa: E402

ctx = Context(0)
n = int(float(os.environ.get("MBP", "3000")) * 1e6)
out = {}
for contigs in [int(x) for x in os.environ.get("CONTIGS", "24,200000").split(",")]:
    for div in [float(x) for x in os.environ.get("DIVS", "0.001,0.005,0.02,0.04,0.07,0.1").split(",")]:
        gs = [Genome.synth(ctx, n, contigs, 20240207, j, div) for j in range(3)]
        _, nb = bf_size_bytes(gs[0].total_bp, 0.025)
        bf = BloomFilter(ctx, nb, 24)
        bf.insert(gs[0])
        for g in gs[1:]:
            bf.insert_and(g)
        ctx.trim_bf_build()
        for w in [int(x) for x in os.environ.get("WS", "1000,250,64,20").split(",")]:
            row, counts = {}, set()
            for label, mode, tiers in (("auto", "auto", "auto"), ("pruned", "pruned", "never"), ("tiers", "auto", "always"), ("dense", "dense", "never")):
                if label == "pruned" and w < 200:
                    continue
                ctx.sketch_mode(mode)
                ctx.sketch_tiers(tiers)
                try:
                    sketch(ctx, gs[1], 24, w, bf).free()
                    ctx.sync()
                    t = time.time()
                    mx = sketch(ctx, gs[1], 24, w, bf)
                    c = len(mx)
                    ctx.sync()
                    row[label] = round((time.time() - t) * 1e3, 2)
                    counts.add(c)
                    mx.free()
                except Exception as exc:                         # noqa: BLE001
                    row[label] = "failed: " + str(exc)[:60]
            ctx.sketch_mode("auto")
            ctx.sketch_tiers("auto")
            best = min(v for v in row.values() if isinstance(v, float))
            row["same count"] = len(counts) == 1
            row["auto over best"] = round(row["auto"] / best, 2) if isinstance(row["auto"], float) else None
            out[f"contigs={contigs} div={div} w={w}"] = row
            print(f"contigs={contigs} div={div} w={w}", row, file=sys.stderr, flush=True)
        bf.free()
        for g in gs:
            g.free()
print(json.dumps(out, indent=1))
