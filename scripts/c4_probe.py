#!/usr/bin/env python3
"""Config 4 on one GPU, per sketch policy: ms per 3 Gbp genome with the folded copy in LDS, without it, and through the key /
window kernels (scripts/c4_probe.py [n_genomes])."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntsynt_amd.device import BloomFilter, Context, Genome, bf_size_bytes, sketch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = Context(0)
fam = [Genome.synth(ctx, 3_000_000_000, 24, 20240207, 1000 + g, 0.05) for g in range(n)]
_, nb = bf_size_bytes(fam[0].total_bp, 0.025)
c = BloomFilter(ctx, nb, 24)
c.insert(fam[0])
t = BloomFilter(ctx, nb, 24)
for g in fam[1:]:
    t.clear()
    t.insert(g)
    c.and_(t)
t.free()
for smode in ("auto", "no-lds"):
    ctx.sketch_summary(smode)
    for mode in ("auto", "dense"):
        ctx.sketch_mode(mode)
        sketch(ctx, fam[0], 24, 1000, c).free()
        ctx.sync()
        t0 = time.time()
        for g in fam:
            sketch(ctx, g, 24, 1000, c).free()
        ctx.sync()
        dt = time.time() - t0
        print(smode, mode, round(dt / n * 1e3, 2), "ms per genome", round(3 * n / dt, 1), "Gbases/s", flush=True)
