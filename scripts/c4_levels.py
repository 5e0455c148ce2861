#!/usr/bin/env python3
"""BASELINE configs[3]'s cascade on one GPU, level by level, both ways: the partitioned build with the AND in its last pass
(nts_bf_insert_and's build) and the literal level over a sparse running filter (bf_level_sparse: every k-mer looked up through the
accept kernels, the hit bits kept).  Prints one JSON line: per level the running filter's popcount before it, the milliseconds
either way, which way the library's own choice went -- where the crossover lies (the thresholds in bf_level_sparse).

  python scripts/c4_levels.py [--mbp 3000] [--genomes 8] [--divergence 0.10] [--force-from 3]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from ntsynt_amd.device import BloomFilter, Context, bf_size_bytes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbp", type=float, default=3000.0)
    ap.add_argument("--genomes", type=int, default=8)
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--divergence", type=float, default=0.10)
    ap.add_argument("--force-from", type=int, default=3, help="the forced run takes the literal level from this level on")
    ap.add_argument("--modes", default="build,auto,forced,auto again", help="which runs (comma separated; under rocprofv3: one)")
    a = ap.parse_args()
    ctx = Context(0, variant="experiments")        # (the environment switches this script sets exist in that build only: csrc/nts_knobs.h)
    args = argparse.Namespace(family="structural", substitutions_only=False, k=24, w=1000, fpr=0.025)
    total = int(a.mbp * 1e6)
    fam = [bench.family_genome(ctx, args, total, a.contigs, j, a.divergence / 2.0) for j in range(a.genomes)]
    _, nbytes = bf_size_bytes(fam[0].total_bp, 0.025)
    rows = {}
    pops = {}
    for how in a.modes.split(","):
        for v in ("NTS_BF_SPARSE_LEVEL", "NTS_BF_SPARSE_MAX_OCC"):
            os.environ.pop(v, None)
        if how == "build":
            os.environ["NTS_BF_SPARSE_LEVEL"] = "0"
        bf = BloomFilter(ctx, nbytes, 24)
        bf.insert(fam[0])
        ctx.sync()
        t_all = time.time()
        for lvl, g in enumerate(fam[1:], start=1):
            if how == "forced" and lvl >= a.force_from:
                os.environ["NTS_BF_SPARSE_MAX_OCC"] = "1.0"
            before = bf.popcount()
            ctx.sync()
            t = time.time()
            bf.insert_and(g)
            ctx.sync()
            ms = (time.time() - t) * 1e3
            st = ctx.bf_level_stats()
            rows.setdefault(lvl, {"popcount_before": before})[how] = {"ms": round(ms, 2), "literal": bool(st["sparse_level"]),
                                                                      "accepted_kmers": st["accepted_kmers"]}
            assert rows[lvl]["popcount_before"] == before, (how, lvl, before, rows[lvl])      # same bits whichever way
        rows.setdefault("total_ms", {})[how] = round((time.time() - t_all) * 1e3, 2)
        pops[how] = bf.popcount()
        bf.free()
    assert len(set(pops.values())) == 1, pops
    print(json.dumps({"what": f"{a.genomes} x {a.mbp:g} Mbp at {a.divergence:g}: cascade levels on one GPU", "filter_bytes": nbytes,
                      "final_popcount": next(iter(pops.values())), "levels": rows}))


if __name__ == "__main__":
    main()
