"""A/B of two builds of the library on the valley workload in ONE process, alternating: the product build against the experiments build
(a candidate change compiled under #ifdef NTS_EXPERIMENTS).  Box-to-box and run-to-run spread is +-5 %; alternation in one process is
what makes a 3 % difference visible.   GENOMES=3 DIV=0.10 REPS=8 python scripts/valley_ab.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd.device import Context, Genome, BloomFilter, bf_size_bytes, sketch
n_g, div, mbp = int(os.environ.get("GENOMES", "3")), float(os.environ.get("DIV", "0.10")), float(os.environ.get("MBP", "3000"))
k, w, reps = 24, int(os.environ.get("W", "1000")), int(os.environ.get("REPS", "8"))
total = int(mbp * 1e6)
side = {}
for name, variant in (("product", None), ("experiments", "experiments")):
    ctx = Context(0, variant=variant)
    g0 = Genome.synth(ctx, total, 24, 20240207, 1000, div / 2)
    _, nb = bf_size_bytes(g0.total_bp, 0.025)
    bf = BloomFilter(ctx, nb, k)
    bf.insert(g0)
    for j in range(1, n_g):
        g = Genome.synth(ctx, total, 24, 20240207, 1000 + j, div / 2)
        bf.insert_and(g)
        g.free()
    ctx.sketch_tiers(os.environ.get("TIERS", "auto"))
    for _ in range(2):
        sketch(ctx, g0, k, w, bf).free()
    ctx.sync()
    side[name] = (ctx, g0, bf, [])
for r in range(reps):
    for name in ("product", "experiments") if r % 2 == 0 else ("experiments", "product"):
        ctx, g0, bf, ts = side[name]
        ctx.sync(); t = time.time(); mx = sketch(ctx, g0, k, w, bf); n = len(mx); mx.free(); ctx.sync()
        ts.append(time.time() - t)
for name, (ctx, g0, bf, ts) in side.items():
    ts = sorted(ts)
    print(f"{name:12s}: median {ts[len(ts) // 2] * 1e3:7.3f} ms, min {ts[0] * 1e3:7.3f}, max {ts[-1] * 1e3:7.3f}  ({g0.total_bp / ts[len(ts) // 2] / 1e9:6.1f} Gbases/s at the median)", flush=True)
