#!/usr/bin/env python3
"""Short windows through the tiered selection (k_hash_tiers: probes in increasing hash order, only where a window is still open) against
the window tiles that hash and probe every k-mer (k_window_min<true>): where is the crossover?  Experiments build (NTS_TIER_MIN_W lets the
tiers below w = 64).   MBP=3000 DIV=0.005,0.02 WS=10,12,16,20,24,28,33,48,63 python scripts/tiers_small_w.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["NTS_TIER_MIN_W"] = "4"
os.environ.setdefault("NTS_TIER_SMALL_C", "100")       # (the product stops at an estimated 0.85 of the every-k-mer pass: here the tiers run wherever they can)
import numpy as np  # noqa: E402
from ntsynt_amd.device import BloomFilter, Context, Genome, bf_size_bytes, sketch  # noqa: E402

mbp = float(os.environ.get("MBP", "3000"))
ws = [int(x) for x in os.environ.get("WS", "10,12,16,20,24,28,33,48,63").split(",")]
ctx = Context(0, variant="experiments")
n = int(mbp * 1e6)
out = {}
for div in [float(x) for x in os.environ.get("DIV", "0.005,0.02").split(",")]:
    g0 = Genome.synth(ctx, n, 24, 20240207, 1, div)
    g1 = Genome.synth(ctx, n, 24, 20240207, 2, div)
    _, nb = bf_size_bytes(g0.total_bp, 0.025)
    bf = BloomFilter(ctx, nb, 24)
    bf.insert(g0)
    bf.insert_and(g1)
    ctx.trim_bf_build()
    for w in ws:
        ref = None
        row = {}
        for label in ("fused", "tiers"):
            ctx.sketch_mode("dense" if label == "fused" else "auto")
            ctx.sketch_tiers("never" if label == "fused" else "always")
            sketch(ctx, g1, 24, w, bf).free()
            ctx.sync()
            ctx.profile(1)
            t = time.time()
            mx = sketch(ctx, g1, 24, w, bf)
            cnt = len(mx)
            ctx.sync()
            dt = time.time() - t
            probes, rounds, tiers = ctx.sketch_tiers(None)
            if mbp <= 400:
                h = mx.to_numpy()
                if ref is None:
                    ref = h
                else:
                    row["same"] = all(np.array_equal(a, b) for a, b in zip(ref, h))
            kern = {nm: round(ctx.timing(nm)[0], 2) for nm in ("hash_tiers", "cand_compact", "sparse_win", "gather_winners", "finalize", "window_min", "hash_probe", "sort_minimizers", "pack_image") if ctx.timing(nm)[1]}
            ctx.profile(0)
            row[label] = {"kernel_ms": kern, "ms": round(dt * 1e3, 2), "Gbases_s": round(n / dt / 1e9, 1), "tiers": tiers, "probes_per_kmer": round(probes / n, 3), "minimizers": cnt}
            mx.free()
        out[f"per-genome divergence {div} w={w}"] = row
    bf.free()
    g0.free()
    g1.free()
print(json.dumps(out, indent=1))
