#!/usr/bin/env python3
"""Mid-size end-to-end parity: families of 2-4 genomes of 40-250 Mbp (written from genomes generated in HBM, like bench.py's end-to-end
inputs: structural events, soft masks; or the assembly-like family) with SHORT windows and small thresholds -- the parameter corner where
refinement rounds run at w = 4 ... 33 over a good part of the genome, blocks number in the tens of thousands, -n may be below the number of
genomes -- through the HIP pipeline and through the CPU restatement on the box's host cores; every table must be byte-identical.
python scripts/stress_midsize.py [--seconds 600] [--seed 1] [--threads 32]"""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ntsynt_amd import pipeline  # noqa: E402
from oracle import synteny_oracle as SO  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=600.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--threads", type=int, default=32)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    n = blocks = stops = 0
    cwd = os.getcwd()
    while time.time() < t_end:
        G = int(rng.choice([2, 3, 3, 4]))
        mbp = float(rng.choice([40, 80, 150, 250])) * (2 if G == 2 else 1)
        fam = argparse.Namespace(family=str(rng.choice(["structural", "structural", "assembly-like"])), substitutions_only=False, k=24, w=1000, fpr=0.025)
        div = float(rng.choice([0.002, 0.01, 0.03]))
        contigs = int(rng.choice([4, 24, 400]))
        kw = dict(k=int(rng.choice([20, 24, 32])), w=int(rng.choice([64, 150, 400])), w_rounds=[int(x) for x in rng.choice([[20, 8], [33, 10], [50, 12], [16, 4]])],
                  indel=int(rng.choice([300, 2000, 20000])), merge=rng.choice(["2w", "10w", 5000]).item(), block_size=int(rng.choice([100, 300, 1000])))
        if isinstance(kw["merge"], str) and kw["merge"].isdigit():
            kw["merge"] = int(kw["merge"])
        if G >= 3 and rng.random() < 0.35:
            kw["n"] = int(rng.integers(2, G))
        if rng.random() < 0.2:
            kw["m"] = int(rng.choice([75, 60]))
        if rng.random() < 0.15:
            kw["common"] = False
        case = dict(genomes=G, mbp=mbp, family=fam.family, divergence=div, contigs=contigs, **kw)
        work = tempfile.mkdtemp(prefix="nts_mid_", dir=os.environ.get("TMPDIR", "/tmp"))
        try:
            paths = bench.e2e_inputs(fam, 0, G, int(mbp * 1e6), contigs, div, work)
            os.makedirs(os.path.join(work, "hip"))
            os.makedirs(os.path.join(work, "ora"))
            stop = {}
            os.chdir(os.path.join(work, "hip"))
            t = time.time()
            try:
                eng = pipeline.run(paths, prefix="p", log=lambda *x: None, **kw)
            except SystemExit:
                eng, stop["hip"] = None, "no paths"
            except Exception as exc:                             # noqa: BLE001
                if "ntsynt_synteny.py:330" not in str(exc) and "ntsynt_synteny.py:437" not in str(exc):
                    raise
                eng, stop["hip"] = None, "reference stop"
            t_hip = time.time() - t
            os.chdir(os.path.join(work, "ora"))
            t = time.time()
            try:
                ora = SO.run_pipeline(paths, prefix="p", threads=a.threads, **kw)
            except SystemExit:
                ora, stop["ora"] = None, "no paths"
            except (AssertionError, IndexError):
                ora, stop["ora"] = None, "reference stop"
            t_ora = time.time() - t
            os.chdir(cwd)
            if stop.get("hip") != stop.get("ora"):
                print("MISMATCH: the two sides stop differently", stop, case, "seed", a.seed, flush=True)
                sys.exit(1)
            if eng is None:
                stops += 1
            else:
                for name in ("p.synteny_blocks.tsv", "p.pre-collinear-merge.synteny_blocks.tsv"):
                    if eng.outputs[name] != ora.outputs[name]:
                        print("MISMATCH", name, case, "seed", a.seed, flush=True)
                        sys.exit(1)
                for p in paths:
                    nm = f"{os.path.basename(p)}.k{kw['k']}.w{kw['w']}.tsv"
                    with open(os.path.join(work, "hip", nm), "rb") as f1, open(os.path.join(work, "ora", nm), "rb") as f2:
                        if f1.read() != f2.read():
                            print("MISMATCH (minimizer TSV)", nm, case, "seed", a.seed, flush=True)
                            sys.exit(1)
                nb = len(eng.outputs["p.synteny_blocks.tsv"].splitlines()) // G
                blocks += nb
                print(f"same: {G} x {mbp:g} Mbp {fam.family} at {div}, {nb} blocks, product {t_hip:.2f} s, restatement {t_ora:.1f} s", {k: v for k, v in kw.items()}, flush=True)
            n += 1
        finally:
            os.chdir(cwd)
            shutil.rmtree(work, ignore_errors=True)
    print(f"ok: {n} mid-size families end to end ({stops} stopped on both sides), {blocks} synteny blocks, seed {a.seed}")


if __name__ == "__main__":
    main()
