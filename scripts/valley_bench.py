"""The valley: common filters that accept a few per cent of a genome's k-mers (three genomes at ~10 %, eight at ~4 %, the reference's
eleven-genome row, README.md:158).  Sketch of genome 0 of a synthetic family (substitutions at div/2 per genome) against the
family's common filter: tiered selection (k_hash_tiers) against one threshold and against every k-mer probed.
  GENOMES=3 DIV=0.10 MBP=3000 MODES=tiers,never python scripts/valley_bench.py        (X0=2.4 HALF=0 for the schedule)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd.device import Context, Genome, BloomFilter, bf_size_bytes, sketch
ctx = Context(0)
n_g, div, mbp = int(os.environ.get("GENOMES", "3")), float(os.environ.get("DIV", "0.10")), float(os.environ.get("MBP", "3000"))
k, w = int(os.environ.get("K", "24")), int(os.environ.get("W", "1000"))
total = int(mbp * 1e6)
g0 = Genome.synth(ctx, total, 24, 20240207, 1000, div / 2)
_, nb = bf_size_bytes(g0.total_bp, 0.025)
bf = BloomFilter(ctx, nb, k)
bf.insert(g0)
for j in range(1, n_g):
    g = Genome.synth(ctx, total, 24, 20240207, 1000 + j, div / 2)
    bf.insert_and(g)
    g.free()
occ = bf.get_fpr()
print(f"{n_g} genomes of {mbp:g} Mbp at {div * 100:g} %: common filter occupancy {occ:.3e}", flush=True)
n_k = g0.valid_kmers(k)
for mode in os.environ.get("MODES", "tiers,never").split(","):
    x0s = [float(x) for x in os.environ.get("X0", "2.4").split(",")] if mode == "tiers" else [0.0]
    for x0 in x0s:
        for half in ([int(x) for x in os.environ.get("HALF", "0").split(",")] if mode == "tiers" else [0]):
            ctx.sketch_tiers({"tiers": "always", "auto": "auto"}.get(mode, "never"), x0=x0, half_steps=bool(half))
            ctx.sketch_mode("dense" if mode == "dense" else "auto")
            dt = 0
            for i in range(3):
                ctx.sync(); t = time.time(); mx = sketch(ctx, g0, k, w, bf); n = len(mx); mx.free(); ctx.sync()
                dt = time.time() - t
            probes, rounds, tiers = ctx.sketch_tiers()
            cand = ctx.sketch_stats()[0]
            ctx.profile(True)
            sketch(ctx, g0, k, w, bf).free(); ctx.sync()
            tm = {nm: ctx.timing(nm) for nm in ("hash_tiers", "hash_select", "hash_probe", "window_min", "hash_accept", "cand_compact", "sparse_win", "finalize")}
            ctx.profile(False)
            ks = ", ".join(f"{nm} {v[0] / v[1]:.3f}" for nm, v in tm.items() if v[1])
            print(f"  {mode:6s} x0={x0:g} half={half} tiers={tiers} c={ctx.last_prune_c}: sketch {dt * 1e3:7.2f} ms = {g0.total_bp / dt / 1e9:7.1f} Gbases/s; "
                  f"probes {probes} ({probes / max(n_k, 1):.3f} per k-mer), rounds/tile {rounds / max(1, (n_k + 14335) // 14336):.2f}, listed {cand}, minimizers {n}; ms: {ks}", flush=True)
ctx.sketch_mode("auto")
