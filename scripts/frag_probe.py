import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd.device import Context, Genome, BloomFilter, bf_size_bytes, sketch
ctx = Context(0)
for contigs in (24, 5000):
    g = Genome.synth(ctx, 3_000_000_000, contigs, 20240207, 1000, 0.005)
    _, nb = bf_size_bytes(g.total_bp, 0.025)
    bf = BloomFilter(ctx, nb, 24)
    for i in range(3):
        ctx.sync(); t = time.time(); bf.insert(g); ctx.sync(); print(contigs, "insert", i, round(time.time() - t, 4), flush=True)
    for i in range(2):
        t = time.time(); bf.popcount(); print(contigs, "popcount", i, round(time.time() - t, 4))
        bf.and_(bf)
    for i in range(3):
        ctx.sync(); t = time.time(); mx = sketch(ctx, g, 24, 1000, bf); n = len(mx); mx.free(); print(contigs, "sketch", i, round(time.time() - t, 4), n, flush=True)
    bf.free(); g.free()
