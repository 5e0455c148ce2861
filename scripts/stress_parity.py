#!/usr/bin/env python3
"""Randomised parity run: device sketch / Bloom build / batch sketch against the CPU oracle over random genomes, k, w,
sketch policies and fragmentation.  python scripts/stress_parity.py [--seconds 240] [--seed 1]; exits non-zero on the
first mismatch and prints the configuration that produced it."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nts_oracle as O  # noqa: E402
from tests.helpers import oracle_flat, random_records, to_device, to_oracle  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args(argv)
    from ntsynt_amd.device import BloomFilter, Context, Genome, sketch
    ctx = Context(0)
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    n_cases = n_sketches = 0
    while time.time() < t_end:
        k = int(rng.choice([16, 20, 24, 31, 32, 40, 64, 100]))
        style = rng.choice(["few", "many", "tiny", "mixed"])
        total = int(rng.integers(80_000, 400_000))
        lengths = []
        while sum(lengths) < total:
            if style == "few":
                lengths.append(int(rng.integers(20_000, 150_000)))
            elif style == "many":
                lengths.append(int(rng.integers(300, 4000)))
            elif style == "tiny":
                lengths.append(int(rng.integers(k // 2, 4 * k)))
            else:
                lengths.append(int(rng.choice([0, 5, k - 1, k, k + 1, 1000, 17000, 60000])))
        n_frac = float(rng.choice([0.0, 0.001, 0.01]))
        seqs = random_records(rng, lengths, n_frac=n_frac)
        names = [f"r{i}" for i in range(len(seqs))]
        og, dg = to_oracle(names, seqs), to_device(ctx, names, seqs)
        # a relative, for a common filter that accepts part of the k-mers
        rel = []
        for s in seqs:
            a = np.frombuffer(s, dtype=np.uint8).copy()
            hit = rng.random(a.size) < float(rng.choice([0.005, 0.02, 0.06, 0.15, 0.3]))   # the last two: sparse common filters
            a[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(hit.sum()))]
            rel.append(a.tobytes())
        og2, dg2 = to_oracle(names, rel), to_device(ctx, names, rel)
        nbytes = int(rng.choice([1 << 18, 1 << 20, 3 << 20]))
        want = O.bf_build(og2, k, nbytes, prev=O.bf_build(og, k, nbytes))
        for mode in ("atomic", "binned"):
            ctx.bf_build_mode(mode)
            a = BloomFilter(ctx, nbytes, k)
            a.insert(dg)
            b = BloomFilter(ctx, nbytes, k)
            b.insert(dg2)
            a.and_(b)
            if not np.array_equal(a.to_numpy(), want):
                print("BLOOM MISMATCH", dict(k=k, style=style, total=total, nbytes=nbytes, mode=mode, seed=args.seed, case=n_cases))
                sys.exit(1)
            b.free()
            if mode == "atomic":
                a.free()
        ctx.bf_build_mode("auto")
        bf = a
        batch = Genome.concat(ctx, [dg, dg2])
        for _ in range(4):
            w = int(rng.choice([10, 33, 64, 100, 250, 500, 1000, 2500]))
            mode, c = [("auto", 0), ("pruned", int(rng.choice([1, 4, 12, 40, 300]))), ("dense", 0)][int(rng.integers(0, 3))]
            use_bf = bool(rng.integers(0, 2))
            ctx.sketch_mode(mode, c)
            ctx.sketch_select(str(rng.choice(["auto", "hi", "hi", "full"])))      # both candidate-selection kernels
            os.environ["NTS_HI_TPW"] = str(int(rng.choice([1, 2, 5])))            # one and several tiles per wave
            exp = [oracle_flat(O.minimize(o, k, w, want if use_bf else None)) for o in (og, og2)]
            got = [sketch(ctx, d, k, w, bf if use_bf else None).to_numpy() for d in (dg, dg2)]
            bmx = sketch(ctx, batch, k, w, bf if use_bf else None)
            parts = batch.split_minimizers(*bmx.to_numpy())
            dev_parts = bmx.split(batch.rec_base)                  # the same split on the device
            for hp, dp in zip(parts, dev_parts):
                if not all(np.array_equal(x, y) for x, y in zip(hp, dp.to_numpy())):
                    print("SPLIT MISMATCH", dict(k=k, w=w, seed=args.seed, case=n_cases))
                    sys.exit(1)
                dp.free()
            bmx.free()
            n_sketches += 3
            for e, g1, g2 in zip(exp, got, parts):
                for x, y, z in zip(e, g1, g2):
                    if not (np.array_equal(y, x.astype(y.dtype)) and np.array_equal(z, y)):
                        print("SKETCH MISMATCH", dict(k=k, w=w, style=style, total=total, mode=mode, c=c, use_bf=use_bf, nbytes=nbytes,
                                                      n_frac=n_frac, seed=args.seed, case=n_cases))
                        sys.exit(1)
        ctx.sketch_mode("auto", 0)
        ctx.sketch_select("auto")
        batch.free()
        bf.free()
        dg.free()
        dg2.free()
        n_cases += 1
    print(f"ok: {n_cases} genomes pairs, {n_sketches} sketches, {2 * n_cases} filter builds x 2 modes, seed {args.seed}")


if __name__ == "__main__":
    main()
