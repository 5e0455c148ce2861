#!/usr/bin/env python3
"""Partitioned Bloom build on one synthetic genome: wall time per insert, and (with --check) the same bits as one atomic per k-mer.

  python scripts/bloom_bench.py --mbp 3000 --reps 4 --check
  rocprofv3 --kernel-trace --stats ... -- python scripts/bloom_bench.py   # k_bin1 / k_bin2 / k_bin3 per launch
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ntsynt_amd.device import Context, Genome, BloomFilter, bf_size_bytes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbp", type=float, default=3000.0)
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("-k", type=int, default=24)
    ap.add_argument("--fpr", type=float, default=0.025)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--family", choices=["uniform", "assembly-like"], default="uniform")
    ap.add_argument("--and-levels", type=int, default=0, help="after the inserts: this many cascade levels (insert_and of relatives)")
    a = ap.parse_args()
    ctx = Context(0)

    def genome(j):
        if a.family == "assembly-like":
            from ntsynt_amd import synth
            plan = synth.realistic_plan(a.contigs, int(a.mbp * 1e6) // a.contigs, j, 20240207)
            return Genome.synth_plan(ctx, plan, 20240207, 1000 + j, 0.0065, rep=synth.REPEATS, names=plan[2])
        return Genome.synth(ctx, int(a.mbp * 1e6), a.contigs, 20240207, 1000 + j, 0.005)
    g = genome(0)
    _, nb = bf_size_bytes(g.total_bp, a.fpr)
    bf = BloomFilter(ctx, nb, a.k)
    for i in range(a.reps):
        bf.clear()
        ctx.sync()
        t = time.time()
        bf.insert(g)
        ctx.sync()
        dt = time.time() - t
        print("insert", i, round(dt * 1e3, 3), "ms", round(g.total_bp / dt / 1e9, 1), "Gbases/s", flush=True)
        print("   path stats", ctx.path_stats(), flush=True)
    for j in range(1, a.and_levels + 1):
        r = genome(j)
        bf.popcount()
        ctx.sync()
        t = time.time()
        bf.insert_and(r)
        ctx.sync()
        dt = time.time() - t
        print("insert_and", j, round(dt * 1e3, 3), "ms", round(r.total_bp / dt / 1e9, 1), "Gbases/s", ctx.path_stats(), flush=True)
        r.free()
    if a.check and not a.and_levels:
        pc = bf.popcount()
        ctx.bf_build_mode("atomic")
        at = BloomFilter(ctx, nb, a.k)
        at.insert(g)
        pa = at.popcount()
        at.and_(bf)
        both = at.popcount()
        print("popcount binned", pc, "atomic", pa, "intersection", both, "SAME" if pc == pa == both else "DIFFERENT", flush=True)
        at.free()
        if not (pc == pa == both):
            sys.exit(1)
    bf.free()
    g.free()


if __name__ == "__main__":
    main()
