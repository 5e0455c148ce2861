#!/usr/bin/env python3
"""The device graph engine against its host-array twin at a size where the minimizer graph has tens of millions of vertices: short windows on
Gbp-scale genomes (the CPU restatement would take hours there; the twin -- numpy over the device graph build, held call by call against the
reference's own runs at small size, tests/test_refrun_product.py -- takes minutes).  Both through ntsynt_amd.pipeline.run on the same files;
every table must be byte-identical.   python scripts/engine_twin_scale.py [--mbp 1000] [--genomes 3] [-w 64] [--rounds 16 4]"""
import argparse
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ntsynt_amd import pipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbp", type=float, default=1000.0)
    ap.add_argument("--genomes", type=int, default=3)
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--divergence", type=float, default=0.01)
    ap.add_argument("--family", default="structural")
    ap.add_argument("-k", type=int, default=24)
    ap.add_argument("-w", type=int, default=64)
    ap.add_argument("--rounds", type=int, nargs="+", default=[16, 4])
    ap.add_argument("-n", type=int, default=0)
    a = ap.parse_args()
    fam = argparse.Namespace(family=a.family, substitutions_only=False, k=a.k, w=a.w, fpr=0.025)
    work = tempfile.mkdtemp(prefix="nts_twin_", dir=os.environ.get("TMPDIR", "/tmp"))
    cwd = os.getcwd()
    out = {"what": f"{a.genomes} x {a.mbp:g} Mbp {a.family} at {a.divergence}, k = {a.k}, w = {a.w}, rounds {a.rounds}, -n {a.n or a.genomes}"}
    try:
        paths = bench.e2e_inputs(fam, 0, a.genomes, int(a.mbp * 1e6), a.contigs, a.divergence, work)
        kw = dict(k=a.k, w=a.w, w_rounds=a.rounds, indel=500, merge="3w", block_size=200, n=a.n, prefix="p", write_mx_tsv=False, log=lambda *x: None)
        digests = {}
        for engine in os.environ.get("ENGINES", "device,host").split(","):
            os.makedirs(os.path.join(work, engine))
            os.chdir(os.path.join(work, engine))
            t = time.time()
            eng = pipeline.run(paths, engine=engine, **kw)
            out[f"{engine}_engine_s"] = round(time.time() - t, 2)
            digests[engine] = {nm: hashlib.md5(eng.outputs[nm].encode()).hexdigest() for nm in ("p.synteny_blocks.tsv", "p.pre-collinear-merge.synteny_blocks.tsv")}
            out[f"{engine}_blocks"] = len(eng.outputs["p.synteny_blocks.tsv"].splitlines()) // a.genomes
            out[f"{engine}_stats"] = {k_: int(v) for k_, v in getattr(eng, "stats", {}).items()}
            out[f"{engine}_stages_s"] = {nm: round(sec, 2) for nm, sec in getattr(eng, "stage_times", [])}
            if getattr(eng, "times", None):                      # NTS_ENGINE_TIMES=1
                out[f"{engine}_engine_times_s"] = {nm: round(v, 2) for nm, v in sorted(eng.times.items(), key=lambda kv: -kv[1])[:8]}
            os.chdir(cwd)
        out["identical"] = len({json.dumps(d, sort_keys=True) for d in digests.values()}) == 1
        out["md5"] = next(iter(digests.values()))
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(out))
    return 0 if out.get("identical") else 1


if __name__ == "__main__":
    sys.exit(main())
