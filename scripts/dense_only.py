import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd.device import BloomFilter, Context, Genome, bf_size_bytes, sketch
ctx = Context(0)
g = Genome.synth(ctx, 3_000_000_000, 24, 20240207, 1000, 0.005)
r = Genome.synth(ctx, 3_000_000_000, 24, 20240207, 1001, 0.005)
_, nb = bf_size_bytes(g.total_bp, 0.025)
bf = BloomFilter(ctx, nb, 24); bf.insert(g); bf.insert_and(r)
ctx.sketch_mode("dense"); ctx.profile(1)
for _ in range(3): sketch(ctx, g, 24, 1000, bf).free()
ctx.sync()
print("hash_probe", ctx.timing("hash_probe"), "window", ctx.timing("window_min"))
