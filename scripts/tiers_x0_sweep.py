import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd.device import BloomFilter, Context, Genome, bf_size_bytes, sketch
ctx = Context(0)
n = 3_000_000_000
for div in (0.005, 0.015):
    g0 = Genome.synth(ctx, n, 24, 20240207, 1, div); g1 = Genome.synth(ctx, n, 24, 20240207, 2, div)
    _, nb = bf_size_bytes(g0.total_bp, 0.025)
    bf = BloomFilter(ctx, nb, 24); bf.insert(g0); bf.insert_and(g1); ctx.trim_bf_build()
    for w in (10, 16, 24, 33, 48):
        row = []
        for x0 in (0.6, 0.9, 1.2, 1.7, 2.4, 3.4):
            ctx.sketch_tiers("auto", x0)
            sketch(ctx, g1, 24, w, bf).free(); ctx.sync()
            t = time.time(); mx = sketch(ctx, g1, 24, w, bf); c = len(mx); ctx.sync(); dt = time.time() - t
            pr, rd, nt = ctx.sketch_tiers(None); mx.free()
            row.append((x0, round(dt * 1e3, 1), nt, round(pr / n, 3)))
        print(div, w, row, flush=True)
    bf.free(); g0.free(); g1.free()
