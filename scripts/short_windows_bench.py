#!/usr/bin/env python3
"""Short windows (w < 64) on one 3 Gbp genome: the product's choice (tiered selection where the filter accepts enough), the window tiles hashing and probing their own k-mers (k_window_min<true>, the product)
against the key array (k_hash<MODE_KEYS> + k_window_min<false>; experiments build, NTS_WIN_FUSE=0).  One JSON object on stdout."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntsynt_amd.device import BloomFilter, Context, Genome, bf_size_bytes, sketch  # noqa: E402


def main():
    out = {}
    for label, variant, env in (("product", None, None), ("fused", None, None), ("key_array", "experiments", "0")):
        if env is not None:
            os.environ["NTS_WIN_FUSE"] = env
        ctx = Context(0, variant=variant) if variant else Context(0)
        ctx.sketch_mode("auto" if label == "product" else "dense")      # product: tiers where the filter accepts enough, else the window tiles
        g0 = Genome.synth(ctx, 3_000_000_000, 24, 20240207, 1, 0.005)
        g1 = Genome.synth(ctx, 3_000_000_000, 24, 20240207, 2, 0.005)
        _, nb = bf_size_bytes(g0.total_bp, 0.025)
        bf = BloomFilter(ctx, nb, 24)
        bf.insert(g0)
        bf.insert_and(g1)
        ctx.trim_bf_build()
        for w in (10, 33, 63):
            for filt in (bf, None):
                sketch(ctx, g1, 24, w, filt).free()
                ctx.sync()
                ctx.mem_reset_peak()
                ctx.profile(1)
                t = time.time()
                mx = sketch(ctx, g1, 24, w, filt)
                n = len(mx)
                ctx.sync()
                dt = time.time() - t
                tag = "hash_probe" if filt else "hash_only"
                out[f"{label} w={w} {'filter' if filt else 'no filter'}"] = {
                    "ms": round(dt * 1e3, 2), "Gbases_s": round(3.0 / dt, 1), "minimizers": n, "peak_HBM_GB": round(ctx.mem_stats()["peak"] / 1e9, 1),
                    "tiers": ctx.sketch_tiers()[2], "probes_per_kmer": round(ctx.sketch_tiers()[0] / 3e9, 3),
                    "kernel_ms": {nm: round(ctx.timing(nm)[0], 2) for nm in (tag, "window_min", "hash_tiers", "sparse_win", "cand_compact", "gather_winners", "finalize") if ctx.timing(nm)[1]}}
                ctx.profile(0)
                mx.free()
        bf.free()
        g0.free()
        g1.free()
        ctx.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
