"""Whole-genome sketch at small windows (64 <= w < 200: the refinement rounds' values, and `ntSynt -w 100`): the tiered selection
against every k-mer probed.  W=100 DIV=0.01 [MODES=auto,never,always] python scripts/small_w_bench.py
(any w: with W=250 ... 700 this is the check of where one threshold hands over to the tiers)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd.device import Context, Genome, BloomFilter, bf_size_bytes, sketch
ctx = Context(0)
div, w, k, total = float(os.environ.get("DIV", "0.01")), int(os.environ.get("W", "100")), 24, int(float(os.environ.get("MBP", "3000")) * 1e6)
g0 = Genome.synth(ctx, total, 24, 20240207, 1000, div / 2)
_, nb = bf_size_bytes(g0.total_bp, 0.025)
bf = BloomFilter(ctx, nb, k)
bf.insert(g0)
for j in (1, 2):
    g = Genome.synth(ctx, total, 24, 20240207, 1000 + j, div / 2)
    bf.insert_and(g); g.free()
for mode in os.environ.get("MODES", "auto,never").split(","):
    ctx.sketch_tiers(mode.split(':')[0], x0=float(mode.split(':')[1]) if ':' in mode else 0.0)   # (always:6 = forced, first tier aimed at 6 accepted k-mers per window)
    for _ in range(2):
        ctx.sync(); t = time.time(); mx = sketch(ctx, g0, k, w, bf); n = len(mx); mx.free(); ctx.sync(); dt = time.time() - t
    pr, rounds, tiers = ctx.sketch_tiers()
    print(f"w={w} div={div}: tiers {mode:9s} (planned {tiers}): {dt * 1e3:7.2f} ms = {g0.total_bp / dt / 1e9:6.1f} Gbases/s, probes per k-mer {pr / g0.valid_kmers(k):.3f}, minimizers {n}", flush=True)
