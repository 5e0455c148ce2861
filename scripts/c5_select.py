"""The assembly-like family (BASELINE configs[4]'s parameter set, -d 1.3) against its common filter: the pruned sketch at the library's
own c and at forced ones; select kernel time, candidates, uncovered ranges.   CS=0,14,16,18,21 python scripts/c5_select.py"""
import os, sys, time, argparse
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from ntsynt_amd.device import Context, BloomFilter, bf_size_bytes, sketch
ctx = Context(0, variant=os.environ.get("NTS_LIB_VARIANT_ARG") or None)
args = argparse.Namespace(family=os.environ.get("FAMILY", "assembly-like"), substitutions_only=False)
div, total, contigs, k, w = float(os.environ.get("DIV", "0.013")), int(float(os.environ.get("MBP", "3000")) * 1e6), 24, 24, 1000
gens = [bench.family_genome(ctx, args, total, contigs, j, div / 2.0) for j in range(3)]
_, nb = bf_size_bytes(gens[0].total_bp, 0.025)
bf = BloomFilter(ctx, nb, k)
bf.insert(gens[0])
for g in gens[1:]:
    bf.insert_and(g)
print(f"occupancy {bf.get_fpr():.4e}", flush=True)
for c in [int(x) for x in os.environ.get("CS", "0,14,16,18,21").split(",")]:
    ctx.sketch_mode("pruned" if c else "auto", c)
    for g in gens:
        sketch(ctx, g, k, w, bf).free()
    ctx.sync(); t = time.time()
    n = 0
    for _ in range(3):
        for g in gens:
            mx = sketch(ctx, g, k, w, bf); n = len(mx); mx.free()
    ctx.sync(); dt = (time.time() - t) / 3
    st = ctx.sketch_stats()
    ctx.profile(2)
    for g in gens:
        sketch(ctx, g, k, w, bf).free()
    ctx.sync()
    hs = ctx.timing("hash_select")
    ctx.profile(1)                                        # (every stage timed, each with a synchronisation behind it)
    for g in gens:
        sketch(ctx, g, k, w, bf).free()
    ctx.sync()
    stages = {nm: ctx.timing(nm) for nm in ("hash_select", "cand_compact", "sparse_win", "gather_winners", "hash_tiers", "hash_probe", "window_min", "finalize", "pack_image")}
    ctx.profile(0)
    print(f"c={c or ctx.last_prune_c}{' (auto)' if not c else ''}: {sum(g.total_bp for g in gens) / dt / 1e9:7.1f} Gbases/s, select {hs[0] / max(hs[1], 1):.3f} ms, "
          f"candidates {st[0]}, uncovered ranges {st[1]} ({st[2]} k-mers), many-listed {ctx.path_stats()['sketch_many_listed']}, minimizers {n}; ms per genome: " + ", ".join(f"{nm} {v[0] / 3:.3f}" for nm, v in stages.items() if v[1]) + f"; wall per genome {dt / 3 * 1e3:.3f} ms", flush=True)
