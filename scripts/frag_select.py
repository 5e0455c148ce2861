"""Pruned sketch of one 3 Gbp genome cut into more and more contigs: the two select kernels side by side (tiles that span runs
take the listed-k-mer path of k_hash_select_hi)."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd.device import Context, Genome, BloomFilter, bf_size_bytes, sketch
ctx = Context(0)
for contigs in (24, 5000, 100000, 1000000):
    g = Genome.synth(ctx, 3_000_000_000, contigs, 20240207, 1000, 0.005)
    g2 = Genome.synth(ctx, 3_000_000_000, contigs, 20240207, 1001, 0.005)
    _, nb = bf_size_bytes(g.total_bp, 0.025)
    bf = BloomFilter(ctx, nb, 24)
    bf.insert(g)
    tmp = BloomFilter(ctx, nb, 24)
    tmp.insert(g2)
    bf.and_(tmp)
    tmp.free()
    g2.free()
    for impl in ("auto", "full"):
        ctx.sketch_select(impl)
        for i in range(3):
            ctx.sync(); t = time.time(); mx = sketch(ctx, g, 24, 1000, bf); n = len(mx); mx.free(); ctx.sync()
            dt = time.time() - t
        ctx.profile(True)
        mx = sketch(ctx, g, 24, 1000, bf); mx.free(); ctx.sync()
        hs = ctx.timing("hash_select")
        ctx.profile(False)
        print(contigs, impl, "sketch ms", round(dt * 1e3, 2), "minimizers", n, "hash_select ms", round(hs[0] / max(hs[1], 1), 3), flush=True)
    bf.free(); g.free()
