#!/usr/bin/env python3
"One 3 Gbp FASTA file -> resident genome (nts_genome_from_fasta), repeated: seconds per load for a few NTS_IO_THREADS settings."
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ntsynt_amd import fasta as fa
from ntsynt_amd.device import Context

ap = argparse.ArgumentParser(); ap.add_argument("--mbp", type=float, default=3000.0); a = ap.parse_args()
args = argparse.Namespace(substitutions_only=False, k=24, w=1000, fpr=0.025)
work = os.environ.get("TMPDIR", "/tmp")
ctx = Context(0)
g = bench.family_genome(ctx, args, int(a.mbp * 1e6), 24, 0, 0.005)
p = os.path.join(work, "ingest_diag.fa")
bench.write_fasta_from_device(g, p)
g.free()
size = os.path.getsize(p)
for thr in ("4", "8", "16", "24", "32", "8"):
    os.environ["NTS_IO_THREADS"] = thr
    c = Context(0)                                   # (the variable is read once, when a context is created: nts_init)
    ts = []
    for rep in range(3):
        t = time.time()
        gg, recs = fa.read_fasta_device(c, p)
        ts.append(time.time() - t)
        gg.free()
    c.close()
    print(f"NTS_IO_THREADS={thr}: " + " ".join(f"{x:.3f}" for x in ts) + f" s  ({size / min(ts) / 1e9:.1f} GB/s best)", flush=True)
os.remove(p)
