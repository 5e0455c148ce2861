#!/usr/bin/env python3
"""Bloom-insert rate against the size of the filter, for the two builds of nts_bf_insert: one atomic OR per k-mer
(is the atomic unit faster when the target range stays resident in the memory-side cache?  no) and the partitioned
build (csrc/nts_bloom_bin.inc).  python scripts/atomic_rate.py [--mbp 200]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbp", type=float, default=200.0)
    args = ap.parse_args()
    from ntsynt_amd.device import BloomFilter, Context, Genome
    ctx = Context(0)
    g = Genome.synth(ctx, int(args.mbp * 1e6), 4, 1, 2, 0.01)
    ctx.profile(True)
    rows = []
    for mb in (1, 4, 16, 32, 64, 128, 256, 512, 1024, 4096):
        row = {"filter_MB": mb}
        for mode in ("atomic", "binned"):
            ctx.bf_build_mode(mode)
            bf = BloomFilter(ctx, mb << 20, 24)
            bf.insert(g)
            ms0, n0 = ctx.timing("bf_insert")  # warm-up launch excluded (timings are cumulative)
            for _ in range(3):
                bf.insert(g)
            ms1, n1 = ctx.timing("bf_insert")
            per = (ms1 - ms0) / max(n1 - n0, 1)
            row[mode + "_ms"] = round(per, 3)
            row[mode + "_G_inserts_s"] = round(g.total_bp / per / 1e6, 2)
            bf.free()
        rows.append(row)
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
