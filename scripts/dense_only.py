import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd.device import BloomFilter, Context, Genome, bf_size_bytes, sketch
ctx = Context(0)
g = [Genome.synth(ctx, 100_000_000, 4, 1, 10 + j, 0.005) for j in range(2)]
_, nb = bf_size_bytes(100_000_000, 0.025)
bf = BloomFilter(ctx, nb, 24); bf.insert(g[0]); t = BloomFilter(ctx, nb, 24); t.insert(g[1]); bf.and_(t)
ctx.sketch_mode("dense")
for _ in range(3):
    sketch(ctx, g[1], 24, 1000, bf).free()
