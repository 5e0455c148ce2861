"""Where does the sparse-filter (summary-first) sketch stop paying?  Families of 4 / 5 / 6 genomes at 10 %: the library's choice against
the summary path forced (NTS_SUMMARY_MAX, experiments build).   python scripts/summary_switch.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd.device import Context, Genome, BloomFilter, bf_size_bytes, sketch
ctx = Context(0, variant="experiments")
k, w, total = 24, 1000, 3_000_000_000
for n_g, div in [(int(a), float(b)) for a, b in (x.split(":") for x in os.environ.get("FAMS", "4:0.10,5:0.10,6:0.10,7:0.06").split(","))]:
    g0 = Genome.synth(ctx, total, 24, 20240207, 1000, div / 2)
    _, nb = bf_size_bytes(g0.total_bp, 0.025)
    bf = BloomFilter(ctx, nb, k)
    bf.insert(g0)
    for j in range(1, n_g):
        g = Genome.synth(ctx, total, 24, 20240207, 1000 + j, div / 2)
        bf.insert_and(g); g.free()
    occ = bf.get_fpr()
    for smax in os.environ.get("SMAX", "0.3,1.0,3.0").split(","):
        os.environ["NTS_SUMMARY_MAX"] = smax
        for _ in range(2):
            ctx.sync(); t = time.time(); mx = sketch(ctx, g0, k, w, bf); n = len(mx); mx.free(); ctx.sync(); dt = time.time() - t
        print(f"{n_g} genomes at {div * 100:g} %: occupancy {occ:.3e} (x 4096 = {occ * 4096:.2f}), summary below {smax}: {dt * 1e3:7.2f} ms = {3.0 / dt:6.1f} Gbases/s, "
              f"summary shift {ctx.sketch_summary()}, tiers {ctx.sketch_tiers()[2]}, minimizers {n}", flush=True)
    bf.free(); g0.free()
