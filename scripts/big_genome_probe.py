#!/usr/bin/env python3
"""Does every launch still fit when one genome is larger than 2^32 bases?  (A launch that would run truncated is refused: the call fails.)
A synthetic genome of MBP bases: filter build, cascade level, sketches the default way and every k-mer probed, counts compared.
MBP=6000 python scripts/big_genome_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd.device import BloomFilter, Context, Genome, bf_size_bytes, sketch  # noqa: E402

ctx = Context(0)
n = int(float(os.environ.get("MBP", "6000")) * 1e6)
t = time.time()
contigs = int(os.environ.get("CONTIGS", "40"))            # (1: a single record beyond 2^32 bases -- positions need their 64 bits)
g0 = Genome.synth(ctx, n, contigs, 20240207, 1, 0.005)
g1 = Genome.synth(ctx, n, contigs, 20240207, 2, 0.005)
ctx.sync()
print("synth", round(time.time() - t, 2), "s; bases", g0.total_bp, flush=True)
_, nb = bf_size_bytes(g0.total_bp, 0.025)
bf = BloomFilter(ctx, nb, 24)
bf.insert(g0)
print("insert ok, popcount", bf.popcount(), flush=True)
bf.insert_and(g1)
pc = bf.popcount()
print("insert_and ok, popcount", pc, "share of bits", round(pc / (nb * 8), 5), flush=True)
ctx.trim_bf_build()
import numpy as np  # noqa: E402
for w in (1000, 100):
    got = {}
    for mode in ("auto", "dense"):
        ctx.sketch_mode(mode)
        t = time.time()
        mx = sketch(ctx, g1, 24, w, bf)
        c = len(mx)
        ctx.sync()
        dt = time.time() - t
        got[mode] = mx.to_numpy()
        mx.free()
        print("w", w, mode, "minimizers", c, round(dt * 1e3, 1), "ms; largest position", int(got[mode][2].max()), flush=True)
    print("   whole lists:", "SAME" if all(np.array_equal(a, b) for a, b in zip(got["auto"], got["dense"])) else "DIFFERENT", flush=True)
ctx.sketch_mode("auto")
