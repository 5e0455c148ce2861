import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd.device import BloomFilter, Context, Genome, bf_size_bytes
ctx = Context(0)
g = Genome.synth(ctx, 300_000_000, 8, 20240207, 1000, 0.005)
r = Genome.synth(ctx, 300_000_000, 8, 20240207, 1001, 0.005)
_, nb = bf_size_bytes(g.total_bp, 0.025)
ctx.profile(1)
bf = BloomFilter(ctx, nb, 24); bf.insert(g); a = bf.popcount(); bf.insert_and(r); b = bf.popcount()
print("popcount kernel launches (0 = both counts came with the builds):", ctx.timing("bf_popcount"), a, b)
ctx.bf_build_mode("atomic")
c = BloomFilter(ctx, nb, 24); c.insert(g); assert c.popcount() == a; t = BloomFilter(ctx, nb, 24); t.insert(r); c.and_(t); assert c.popcount() == b
print("same as the atomic build's")
