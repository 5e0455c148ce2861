#!/usr/bin/env python3
"""Empirical ceiling for random Bloom probes on this GPU: filters of the sizes BASELINE's configs use,
10^9 probes each, no hashing (nts_bench_random_probe), next to a streaming read of the same filter."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntsynt_amd.device import BloomFilter, Context, bf_size_bytes  # noqa: E402

ctx = Context(0)
out = []
for bp in (29_058_289, 100_000_000, 1_000_000_000, 3_000_000_000):
    _, nbytes = bf_size_bytes(bp, 0.025)
    bf = BloomFilter(ctx, nbytes, 24)
    n = 1_000_000_000
    ms = bf.bench_random_probe(n, 3)
    ctx.profile(True)
    for _ in range(3):
        bf.clear()                            # also invalidates the cached popcount
        ctx.lib.nts_bf_upload                 # (no-op reference; keeps the loop body obvious)
        bf.and_(bf)                           # any write invalidates the cache
        bf.popcount()
    pms, pn = ctx.timing("bf_popcount")
    out.append({"genome_bp": bp, "filter_MB": round(nbytes / 1e6, 1), "random_probe_G_per_s": round(n / ms / 1e6, 2),
                "random_probe_GBs_at_64B": round(64 * n / ms / 1e6, 1),
                "stream_read_GBs": round(nbytes / (pms / max(pn, 1)) / 1e6, 1) if pn else None})
    bf.free()
print(json.dumps(out, indent=1))
