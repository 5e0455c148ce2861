#!/usr/bin/env python3
"""Full-size consistency sweep (no oracle at this size: the paths against each other).  One family of two 3 Gbp genomes; for every k and
filter size: the partitioned Bloom build against the one-atomic-per-k-mer build (popcounts after insert and after the cascade level), and
the default sketch against every k-mer probed (whole lists compared) at several windows.   python scripts/scale_consistency.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd.device import BloomFilter, Context, Genome, bf_size_bytes, sketch  # noqa: E402

ctx = Context(0)
n = int(float(os.environ.get("MBP", "3000")) * 1e6)
contigs = int(os.environ.get("CONTIGS", "24"))
g0 = Genome.synth(ctx, n, contigs, 20240207, 1, 0.005)
g1 = Genome.synth(ctx, n, contigs, 20240207, 2, 0.005)
bad = 0
for k in [int(x) for x in os.environ.get("KS", "16,24,31,32,33,64,100,128,129").split(",")]:
    for fpr in [float(x) for x in os.environ.get("FPRS", "0.025,0.3,0.01").split(",")]:
        _, nb = bf_size_bytes(g0.total_bp, fpr)
        pcs = {}
        for mode in ("binned", "atomic"):
            ctx.bf_build_mode(mode)
            bf = BloomFilter(ctx, nb, k)
            t = time.time()
            bf.insert(g0)
            p1 = bf.popcount()
            bf.insert_and(g1)
            p2 = bf.popcount()
            ctx.sync()
            pcs[mode] = (p1, p2, round(time.time() - t, 3))
            if mode == "atomic":
                bf.free()
        ctx.bf_build_mode("auto")
        ok = pcs["binned"][:2] == pcs["atomic"][:2]
        bad += not ok
        print("k", k, "fpr", fpr, "GB", round(nb / 1e9, 2), "binned", pcs["binned"], "atomic", pcs["atomic"], "SAME" if ok else "DIFFERENT", flush=True)
        if fpr == 0.025:
            ctx.bf_build_mode("binned")
            bfk = BloomFilter(ctx, nb, k)
            bfk.insert(g0)
            bfk.insert_and(g1)
            ctx.bf_build_mode("auto")
            ctx.trim_bf_build()
            for w in (1000, 250, 100, 33):
                lists = {}
                for mode in ("auto", "dense"):
                    ctx.sketch_mode(mode)
                    t = time.time()
                    mx = sketch(ctx, g1, k, w, bfk)
                    lists[mode] = (mx.to_numpy(), round((time.time() - t) * 1e3, 1))
                    mx.free()
                ctx.sketch_mode("auto")
                a, b = lists["auto"][0], lists["dense"][0]
                same = len(a[0]) == len(b[0]) and all(np.array_equal(x, y) for x, y in zip(a, b))
                bad += not same
                print("   k", k, "w", w, "minimizers", len(a[0]), "auto ms", lists["auto"][1], "dense ms", lists["dense"][1], "SAME" if same else "DIFFERENT", flush=True)
            bfk.free()
        bf.free()
print("ok" if not bad else f"{bad} DIFFERENCES")
