#!/usr/bin/env python3
"""Full-size consistency of the ways a sketch can go, whole lists compared (3 Gbp; no oracle at this size): forced selections with next to
nothing listed (millions of uncovered ranges: the gap paths, the window kernel in several launches), the tiers forced, masks over most of
the genome (a refinement round's shape), a family in hundreds of thousands of records, a batch of genomes against its parts.
python scripts/scale_modes.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd.device import BloomFilter, Context, Genome, bf_size_bytes, sketch  # noqa: E402

ctx = Context(0)
n = int(float(os.environ.get("MBP", "3000")) * 1e6)
bad = 0


def lists(g, k, w, bf, mode, c=0, tiers="auto", masks=None):
    ctx.sketch_mode(mode, c)
    ctx.sketch_tiers(tiers)
    t = time.time()
    mx = sketch(ctx, g, k, w, bf, masks)
    out = mx.to_numpy()
    mx.free()
    ctx.sketch_mode("auto")
    ctx.sketch_tiers("auto")
    return out, round((time.time() - t) * 1e3, 1)


def same(a, b):
    return len(a[0]) == len(b[0]) and all(np.array_equal(x, y) for x, y in zip(a, b))


for contigs in (24, 300000):
    g0 = Genome.synth(ctx, n, contigs, 20240207, 1, 0.005)
    g1 = Genome.synth(ctx, n, contigs, 20240207, 2, 0.005)
    _, nb = bf_size_bytes(g0.total_bp, 0.025)
    bf = BloomFilter(ctx, nb, 24)
    bf.insert(g0)
    bf.insert_and(g1)
    ctx.trim_bf_build()
    for w in (1000, 200):
        ref, t_ref = lists(g1, 24, w, bf, "dense")
        for label, kw in (("auto", dict(mode="auto")), ("one threshold, c = 3", dict(mode="pruned", c=3)), ("one threshold, c = 3, gaps the dense way", dict(mode="pruned", c=3, tiers="never")),
                          ("one threshold, c = 40", dict(mode="pruned", c=40)), ("tiers forced", dict(mode="auto", tiers="always"))):
            got, t = lists(g1, 24, w, bf, **kw)
            ok = same(ref, got)
            bad += not ok
            print(f"contigs {contigs} w {w}: {label}: {len(got[0])} minimizers, {t} ms (every k-mer probed: {t_ref} ms)", "SAME" if ok else "DIFFERENT", "ranges", ctx.sketch_stats()[1], flush=True)
    # a refinement round's shape: nine tenths of every record masked, in stretches
    rng = np.random.default_rng(5)
    rec_len = g1.rec_len if hasattr(g1, "rec_len") else None
    masks = []
    n_rec = g1.n_rec if hasattr(g1, "n_rec") else contigs
    for r in range(min(n_rec, 2000)):
        ln = n // contigs
        for s in range(0, ln - 100, max(ln // 40, 200)):
            masks.append((r, s, s + int(max(ln // 40, 200) * 0.9)))
    for w in (100, 10):
        a, ta = lists(g1, 24, w, bf, "auto", masks=masks)
        b, tb = lists(g1, 24, w, bf, "dense", masks=masks)
        ok = same(a, b)
        bad += not ok
        print(f"contigs {contigs} w {w} with {len(masks)} masks: auto {ta} ms, every k-mer probed {tb} ms, {len(a[0])} minimizers", "SAME" if ok else "DIFFERENT", flush=True)
    bf.free()
    g0.free()
    g1.free()
# a batch against its parts
parts = [Genome.synth(ctx, 400_000_000, 50, 99, j, 0.01) for j in range(3)]
_, nb = bf_size_bytes(parts[0].total_bp, 0.025)
bf = BloomFilter(ctx, nb, 24)
bf.insert(parts[0])
for p in parts[1:]:
    bf.insert_and(p)
batch = Genome.concat(ctx, parts)
for w in (1000, 100, 20):
    whole, _ = lists(batch, 24, w, bf, "auto")
    split = batch.split_minimizers(*whole)
    ok = True
    for p, s in zip(parts, split):
        one, _ = lists(p, 24, w, bf, "auto")
        ok &= same(one, s)
    bad += not ok
    print(f"batch of 3 x 400 Mbp, w {w}:", "SAME as its parts" if ok else "DIFFERENT", flush=True)
print("ok" if not bad else f"{bad} DIFFERENCES")
