"""Pruned sketch of a 3 Gbp genome against the common filter of a pair at growing divergence: the candidate share c/w grows with
1/p; which select kernel the library picks and what it costs (k_hash_select_hi serves up to ~180 listed k-mers per tile)."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd.device import Context, Genome, BloomFilter, bf_size_bytes, sketch
ctx = Context(0)
for div in [float(x) for x in os.environ.get("DIVS", "0.01,0.02,0.03,0.04,0.05,0.06").split(",")]:
    g = Genome.synth(ctx, 3_000_000_000, 24, 20240207, 1000, div / 2)
    g2 = Genome.synth(ctx, 3_000_000_000, 24, 20240207, 1001, div / 2)
    _, nb = bf_size_bytes(g.total_bp, 0.025)
    bf = BloomFilter(ctx, nb, 24)
    bf.insert(g)
    tmp = BloomFilter(ctx, nb, 24)
    tmp.insert(g2)
    bf.and_(tmp)
    tmp.free(); g2.free()
    for impl in os.environ.get("IMPLS", "auto,full").split(","):
        ctx.sketch_select(impl)
        for i in range(3):
            ctx.sync(); t = time.time(); mx = sketch(ctx, g, 24, 1000, bf); n = len(mx); mx.free(); ctx.sync()
            dt = time.time() - t
        ctx.sketch_stats()
        ctx.profile(True)
        mx = sketch(ctx, g, 24, 1000, bf); mx.free(); ctx.sync()
        hs = ctx.timing("hash_select")
        ctx.profile(False)
        print(f"div {div:.2f} {impl:5s} c={ctx.last_prune_c:3d} sketch {dt*1e3:6.2f} ms = {3.0/dt:6.1f} Gbases/s, select {hs[0]/max(hs[1],1):.3f} ms, minimizers {n}", flush=True)
    bf.free(); g.free()
