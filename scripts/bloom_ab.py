"""A/B of two builds of the library on the Bloom build of one 3 Gbp genome in ONE process, alternating (scripts/valley_ab.py for the why):
insert into an empty filter, then one fused cascade level.   REPS=8 [FAMILY=assembly-like] python scripts/bloom_ab.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd.device import Context, Genome, BloomFilter, bf_size_bytes
mbp, reps, k = float(os.environ.get("MBP", "3000")), int(os.environ.get("REPS", "8")), 24
total = int(mbp * 1e6)
side = {}
for name, variant in (("product", None), ("experiments", "experiments")):
    ctx = Context(0, variant=variant)
    if os.environ.get("FAMILY"):                      # FAMILY=assembly-like: bench.py's c5-like genomes (scaffold tails, N gaps, repeats)
        import argparse, bench
        a = argparse.Namespace(family=os.environ["FAMILY"], substitutions_only=False)
        g0, g1 = (bench.family_genome(ctx, a, total, 24, j, 0.0065) for j in range(2))
    else:
        g0 = Genome.synth(ctx, total, 24, 20240207, 1000, 0.005)
        g1 = Genome.synth(ctx, total, 24, 20240207, 1001, 0.005)
    _, nb = bf_size_bytes(g0.total_bp, 0.025)
    bf = BloomFilter(ctx, nb, k)
    for _ in range(2):
        bf.clear(); bf.insert(g0); bf.insert_and(g1)
    ctx.sync()
    side[name] = (ctx, g0, g1, bf, [], [])
pops = {}
for r in range(reps):
    for name in ("product", "experiments") if r % 2 == 0 else ("experiments", "product"):
        ctx, g0, g1, bf, ti, ta = side[name]
        bf.clear(); ctx.sync()
        t = time.time(); bf.insert(g0); ctx.sync(); ti.append(time.time() - t)
        t = time.time(); bf.insert_and(g1); ctx.sync(); ta.append(time.time() - t)
        pops[name] = bf.popcount()
for name, (ctx, g0, g1, bf, ti, ta) in side.items():
    ti, ta = sorted(ti), sorted(ta)
    print(f"{name:12s}: insert median {ti[len(ti) // 2] * 1e3:7.3f} ms (min {ti[0] * 1e3:7.3f}) = {g0.total_bp / ti[len(ti) // 2] / 1e9:6.1f} Gbases/s; "
          f"fused AND level median {ta[len(ta) // 2] * 1e3:7.3f} ms (min {ta[0] * 1e3:7.3f}); popcount {pops[name]}", flush=True)
assert len(set(pops.values())) == 1, "the two builds disagree on the filter"
