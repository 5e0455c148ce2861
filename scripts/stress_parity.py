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
    ctx = Context(0, variant="experiments")        # (the environment switches this script sets exist in that build only: csrc/nts_knobs.h)
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    n_cases = n_sketches = n_literal = n_levels = n_tiered = 0
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
        if rng.random() < 0.4:        # (round 4) a satellite-like array: the copies of a unit pile into a few Bloom buckets
            unit = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(rng.integers(5, 200)))])
            seqs.append(unit * int(rng.integers(200, 40000 // len(unit) * 20 + 201)))
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
            # (round 4) the same level fused into the build's last pass, the parking list sometimes cut short
            cap = rng.choice(["", "0", "7", "100000"])
            if cap:
                os.environ["NTS_BIN_LATE_CAP"] = str(cap)
            else:
                os.environ.pop("NTS_BIN_LATE_CAP", None)
            f = BloomFilter(ctx, nbytes, k)
            f.insert(dg)
            if rng.random() < 0.5:
                f.popcount()
            f.insert_and(dg2)
            os.environ.pop("NTS_BIN_LATE_CAP", None)
            if not np.array_equal(f.to_numpy(), want) or f.popcount() != int(np.unpackbits(want).sum()):
                print("FUSED AND MISMATCH", dict(k=k, style=style, total=total, nbytes=nbytes, mode=mode, cap=cap, seed=args.seed, case=n_cases))
                sys.exit(1)
            f.free()
            if mode == "atomic":
                a.free()
        ctx.bf_build_mode("auto")
        bf = a
        batch = Genome.concat(ctx, [dg, dg2])
        for _ in range(4):
            w = int(rng.choice([8, 10, 10, 13, 24, 33, 47, 63, 64, 100, 250, 500, 1000, 2500]))   # (below 64: tiers or the window tiles, by the filter)
            mode, c = [("auto", 0), ("pruned", int(rng.choice([1, 4, 12, 40, 300]))), ("dense", 0)][int(rng.integers(0, 3))]
            use_bf = bool(rng.integers(0, 2))
            ctx.sketch_mode(mode, c)
            ctx.sketch_select(str(rng.choice(["auto", "hi", "hi", "full"])))      # both candidate-selection kernels
            os.environ["NTS_HI_TPW"] = str(int(rng.choice([1, 2, 5])))            # one and several tiles per wave
            # (round 5) the tiered selection: by the library's own choice, forced wherever it applies, or forbidden; any schedule
            ctx.sketch_tiers(str(rng.choice(["auto", "auto", "always", "always", "never"])), x0=float(rng.choice([0.0, 0.0, 0.3, 1.0, 2.4, 6.0])),
                             half_steps=bool(rng.random() < 0.3))
            exp = [oracle_flat(O.minimize(o, k, w, want if use_bf else None)) for o in (og, og2)]
            got = []
            for d in (dg, dg2):
                got.append(sketch(ctx, d, k, w, bf if use_bf else None).to_numpy())
                n_tiered += ctx.sketch_tiers()[2] > 0
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
        ctx.sketch_tiers("auto")
        # (round 4) a cascade of several relatives: once the running filter is all but empty the level goes the literal way
        # (bf_level_sparse) -- by the library's own choice, or forced at any occupancy, through each of the accept kernels; every
        # level against the oracle's cascade, and a sketch with the tables the last level left behind
        if rng.random() < 0.6:
            forced = str(rng.choice(["", "", "1.0", "0.01"]))
            acc_reg = str(rng.choice(["", "0"]))
            ctx.sketch_summary(str(rng.choice(["auto", "auto", "no-lds"])))
            big = int(rng.choice([nbytes, 8 << 20, 64 << 20]))     # (folded tables need >= 2^19 bits: 64 KiB)
            want_l = O.bf_build(og, k, big)
            chain = BloomFilter(ctx, big, k)
            chain.insert(dg)
            went = []
            last = (og, dg)
            for lvl in range(int(rng.integers(2, 5))):
                d = float(rng.choice([0.01, 0.03, 0.06, 0.1]))
                rl = []
                for s in seqs:
                    a = np.frombuffer(s, dtype=np.uint8).copy()
                    hit = rng.random(a.size) < d
                    a[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(hit.sum()))]
                    rl.append(a.tobytes())
                ol, dl = to_oracle(names, rl), to_device(ctx, names, rl)
                want_l = O.bf_build(ol, k, big, prev=want_l)
                for var, val in (("NTS_BF_SPARSE_MAX_OCC", forced), ("NTS_ACCEPT_REG", acc_reg)):
                    if val:
                        os.environ[var] = val
                    else:
                        os.environ.pop(var, None)
                if rng.random() < 0.7:
                    chain.popcount()
                chain.insert_and(dl)
                went.append(ctx.bf_level_stats()["sparse_level"])
                for var in ("NTS_BF_SPARSE_MAX_OCC", "NTS_ACCEPT_REG"):
                    os.environ.pop(var, None)
                if not np.array_equal(chain.to_numpy(), want_l) or chain.popcount() != int(np.unpackbits(want_l).sum()):
                    print("CASCADE LEVEL MISMATCH", dict(k=k, style=style, total=total, nbytes=big, lvl=lvl, went=went, forced=forced, acc_reg=acc_reg,
                                                         seed=args.seed, case=n_cases))
                    sys.exit(1)
                if last[1] is not dg:
                    last[1].free()
                last = (ol, dl)
            n_literal += sum(went)
            n_levels += len(went)
            w = int(rng.choice([10, 64, 150]))
            ctx.sketch_mode("auto", 0)
            e = oracle_flat(O.minimize(last[0], k, w, want_l))
            mx = sketch(ctx, last[1], k, w, chain)
            g = mx.to_numpy()
            mx.free()
            if not all(np.array_equal(y, x.astype(y.dtype)) for x, y in zip(e, g)):
                print("SKETCH AFTER CASCADE MISMATCH", dict(k=k, w=w, style=style, total=total, nbytes=big, went=went, forced=forced, seed=args.seed, case=n_cases))
                sys.exit(1)
            if last[1] is not dg:
                last[1].free()
            chain.free()
            ctx.sketch_summary("auto")
        # (round 4) record shards: the shards' filters OR to the genome's, their lists concatenate to the genome's
        if len(names) >= 2:
            from ntsynt_amd.device import Minimizers
            from ntsynt_amd.pipeline import shard_plan
            n_sh = int(rng.integers(2, 6))
            _, ranges = shard_plan([[len(x) for x in seqs], [1]], 2 * n_sh)
            w = int(rng.choice([33, 250, 1000]))
            whole_bf = BloomFilter(ctx, nbytes, k)
            whole_bf.insert(dg)
            union = BloomFilter(ctx, nbytes, k)
            parts = []
            for r in range(0, 2 * n_sh, 2):
                sub = dg.slice(ranges[r][2], ranges[r][3])
                union.insert(sub)
                parts.append(sketch(ctx, sub, k, w, bf))
                sub.free()
            cat = Minimizers.concat(ctx, parts, [ranges[r][2] for r in range(0, 2 * n_sh, 2)])
            full = sketch(ctx, dg, k, w, bf)
            same = np.array_equal(union.to_numpy(), whole_bf.to_numpy()) and all(np.array_equal(x, y) for x, y in zip(cat.to_numpy(), full.to_numpy()))
            for m in parts + [cat, full]:
                m.free()
            union.free()
            whole_bf.free()
            if not same:
                print("SHARD MISMATCH", dict(k=k, w=w, style=style, n_sh=n_sh, seed=args.seed, case=n_cases))
                sys.exit(1)
        batch.free()
        bf.free()
        dg.free()
        dg2.free()
        n_cases += 1
    print(f"ok: {n_cases} genomes pairs, {n_sketches} sketches, {2 * n_cases} filter builds x 2 modes, {n_levels} further cascade levels "
          f"({n_literal} of them the literal way), {n_tiered} sketches through the tiered selection, seed {args.seed}")


if __name__ == "__main__":
    main()
