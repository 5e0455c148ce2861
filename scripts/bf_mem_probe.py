#!/usr/bin/env python3
"""Does the kind of memory behind the Bloom filter change what a random probe costs?  NTS_BF_MEM = (default) | uncached | finegrained:
random-probe microbenchmark (10^9 probes of a 14.8 GB filter, no hashing), the every-k-mer-probed sketch, the pruned sketch and the
partitioned build of one 3 Gbp genome.  One process per setting (the variable is read when a filter is created)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    from ntsynt_amd.device import BloomFilter, Context, Genome, bf_size_bytes, sketch
    ctx = Context(0, variant="experiments")        # (the environment switches this script sets exist in that build only: csrc/nts_knobs.h)
    g = Genome.synth(ctx, 3_000_000_000, 24, 20240207, 1000, 0.005)
    r = Genome.synth(ctx, 3_000_000_000, 24, 20240207, 1001, 0.005)
    _, nb = bf_size_bytes(g.total_bp, 0.025)
    bf = BloomFilter(ctx, nb, 24)
    out = {"mem": os.environ.get("NTS_BF_MEM", "default")}
    t = []
    for _ in range(3):
        bf.clear()
        ctx.sync()
        t0 = time.time()
        bf.insert(g)
        ctx.sync()
        t.append(time.time() - t0)
    out["insert_ms"] = round(min(t) * 1e3, 2)
    t0 = time.time()
    bf.insert_and(r)
    ctx.sync()
    out["insert_and_ms"] = round((time.time() - t0) * 1e3, 2)
    out["random_probe_G_per_s"] = round(1e9 / bf.bench_random_probe(1_000_000_000, 3) / 1e6, 2)
    for mode in ("pruned", "dense"):
        ctx.sketch_mode(mode)
        sketch(ctx, g, 24, 1000, bf).free()
        ctx.sync()
        t0 = time.time()
        n = 3 if mode == "pruned" else 2
        for _ in range(n):
            sketch(ctx, g, 24, 1000, bf).free()
        ctx.sync()
        out[f"sketch_{mode}_Gbases_s"] = round(g.total_bp * n / (time.time() - t0) / 1e9, 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for kind in ("", "uncached", "finegrained"):
            env = dict(os.environ)
            env.pop("NTS_BF_MEM", None)
            if kind:
                env["NTS_BF_MEM"] = kind
            r = subprocess.run([sys.executable, __file__, "one"], env=env, capture_output=True, text=True, timeout=600)
            print(r.stdout.strip() or r.stderr[-500:], flush=True)
