#!/usr/bin/env python3
"""Sketch / Bloom-build time on a genome with satellite arrays (a short unit repeated tandemly over a share of the
sequence): repeats make identical k-mers, i.e. identical hashes -- runs of candidates, runs of one Bloom bucket.
python scripts/repeat_bench.py [--mbp 100 --share 0.2 --unit 171]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbp", type=float, default=100.0)
    ap.add_argument("--share", type=float, default=0.2)
    ap.add_argument("--unit", type=int, default=171)
    ap.add_argument("--array-kbp", type=float, default=50.0)
    args = ap.parse_args()
    from ntsynt_amd.device import BloomFilter, Context, Genome, bf_size_bytes, sketch
    from oracle import nts_oracle as O
    ctx = Context(0)
    rng = np.random.default_rng(3)
    n = int(args.mbp * 1e6)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = {}
    for share in (0.0, args.share):
        seq = acgt[rng.integers(0, 4, size=n)]
        unit = acgt[rng.integers(0, 4, size=args.unit)]
        alen = int(args.array_kbp * 1e3)
        n_arrays = int(share * n / alen)
        for s in rng.integers(0, n - alen, size=n_arrays):
            seq[s:s + alen] = np.resize(unit, alen)
        per = n // 4
        off = (np.arange(4, dtype=np.uint64) * np.uint64(per)).astype(np.uint64)
        lens = np.full(4, per, dtype=np.uint64)
        g = Genome(ctx, [f"c{i}" for i in range(4)], seq[:per * 4], off, lens)
        k, w = 24, 1000
        _, nbytes = bf_size_bytes(per * 4, 0.025)
        ctx.profile(True)
        bf = BloomFilter(ctx, nbytes, k)
        bf.insert(g)
        ctx.sync()
        ins_ms, ins_n = ctx.timing("bf_insert")
        res = {"bf_insert_ms": round(ins_ms / max(ins_n, 1), 3)}
        for mode in ("auto", "dense"):
            ctx.sketch_mode(mode, 0)
            sketch(ctx, g, k, w, bf).free()
            ctx.sync()
            t = time.perf_counter()
            reps = 5
            for _ in range(reps):
                mx = sketch(ctx, g, k, w, bf)
                cnt = len(mx)
                mx.free()
            ctx.sync()
            res[mode + "_ms"] = round((time.perf_counter() - t) / reps * 1e3, 3)
            res[mode + "_minimizers"] = cnt
            res[mode + "_stats"] = list(ctx.sketch_stats())
        ctx.sketch_mode("auto", 0)
        # parity on a slice that includes an array
        if share > 0:
            s0 = int(rng.integers(0, n - alen, size=1)[0])
        sl = seq[:3_000_000].tobytes()
        og = O.Genome(["c0"], [sl])
        dg = Genome(ctx, ["c0"], np.frombuffer(sl, dtype=np.uint8), np.zeros(1, np.uint64), np.array([len(sl)], np.uint64))
        exp = O.minimize(og, k, w)
        got = sketch(ctx, dg, k, w).to_numpy()
        res["slice_parity"] = bool(np.array_equal(got[0], exp[0][0].astype(got[0].dtype)) and np.array_equal(got[2], exp[0][1].astype(got[2].dtype))) if len(exp) else None
        out[f"share_{share:g}"] = res
        g.free()
        bf.free()
        dg.free()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
