#!/usr/bin/env python3
"""The automatic choice against every k-mer probed across k and short / middle windows (one 3 Gbp genome, its family's filter, warm calls):
where the tiered selection takes the call, its probes per k-mer, and the time in each kernel group.  One JSON object on stdout.
KS=16,24,40,64,100 WS=16,33,63,100,150 python scripts/k_w_sweep.py"""
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
out = {}
for k in [int(x) for x in os.environ.get("KS", "16,24,40,64,100").split(",")]:
    _, nb = bf_size_bytes(g0.total_bp, 0.025)
    bf = BloomFilter(ctx, nb, k)
    bf.insert(g0)
    bf.insert_and(g1)
    ctx.trim_bf_build()
    for w in [int(x) for x in os.environ.get("WS", "16,33,63,100,150").split(",")]:
        row = {}
        for mode in ("auto", "dense"):
            ctx.sketch_mode(mode)
            sketch(ctx, g1, k, w, bf).free()
            ctx.sync()
            ctx.profile(1)
            t = time.time()
            mx = sketch(ctx, g1, k, w, bf)
            c = len(mx)
            ctx.sync()
            dt = time.time() - t
            kern = {nm: round(ctx.timing(nm)[0], 1) for nm in ("hash_tiers", "cand_compact", "sparse_win", "gather_winners", "finalize", "window_min", "hash_probe", "hash_select")
                    if ctx.timing(nm)[1]}
            ctx.profile(0)
            row[mode] = {"ms": round(dt * 1e3, 1), "minimizers": c, "tiers": ctx.sketch_tiers()[2], "probes_per_kmer": round(ctx.sketch_tiers()[0] / n, 3), "kernel_ms": kern}
            mx.free()
        ctx.sketch_mode("auto")
        out[f"k={k} w={w}"] = row
        print(f"k={k} w={w}", row["auto"]["ms"], row["dense"]["ms"], row["auto"]["tiers"], row["auto"]["probes_per_kmer"], file=sys.stderr, flush=True)
    bf.free()
print(json.dumps(out, indent=1))
