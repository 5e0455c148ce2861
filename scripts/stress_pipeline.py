#!/usr/bin/env python3
"""Randomised end-to-end parity: synthetic genome families with random size, fragmentation, divergence and parameters
through the GPU pipeline and through the CPU oracle pipeline; the synteny-block TSVs must be byte-identical.
python scripts/stress_pipeline.py [--seconds 240] [--seed 1]"""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntsynt_amd import pipeline, synth  # noqa: E402
from oracle import synteny_oracle as SO  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args(argv)
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    cwd = os.getcwd()
    n = 0
    blocks = 0
    none_found = asserts = 0
    while time.time() < t_end:
        # 2-12 genomes (the reference's published rows end at eleven, README.md:158); many genomes: smaller ones, so that a case stays seconds
        n_g = int(rng.choice([2, 2, 3, 3, 4, 5, 6, 8, 9, 11, 12]))
        case = dict(n=n_g, bp=int(rng.integers(600_000, 3_000_000) if n_g <= 4 else rng.integers(300_000, 900_000)), ctg=int(rng.choice([1, 2, 5, 40, 300])),
                    div=float(rng.choice([0.002, 0.01, 0.03, 0.06, 0.10])), seed=int(rng.integers(1, 10_000)),
                    micro=int(rng.choice([0, 6, 12])), n_runs=bool(rng.integers(0, 2)))
        w = int(rng.choice([200, 250, 500, 1000]))
        kw = dict(k=int(rng.choice([20, 24, 32])), w=w, w_rounds=[int(x) for x in rng.choice([[100, 10], [250, 100], [50, 5]])],
                  indel=int(rng.choice([500, 5000, 50000])), merge=rng.choice([3000, 20000, "100w", "3w"]).item(),
                  block_size=int(rng.choice([200, 500, 1000])))
        if isinstance(kw["merge"], str) and kw["merge"].isdigit():
            kw["merge"] = int(kw["merge"])
        if rng.random() < 0.3 and n_g <= 5:
            # a tight family: small genomes, short windows, small thresholds -- where rounds end without blocks, paths turn around and
            # the last round's stops (S:330, S:437) are reached (tests/golden/refrun_stress.py's regime, here through the HIP path)
            case.update(bp=int(rng.integers(50_000, 400_000)), ctg=int(rng.choice([1, 2, 3])), div=float(rng.choice([0.002, 0.005, 0.02])), micro=int(rng.choice([8, 20])))
            kw.update(w=int(rng.choice([30, 60, 150, 400])), w_rounds=[int(x) for x in rng.choice([[10, 4], [20, 5], [12, 6]])],
                      indel=int(rng.choice([150, 500])), merge=rng.choice(["1w", "2w", 600]).item(), block_size=int(rng.choice([50, 100, 150])))
            if isinstance(kw["merge"], str) and kw["merge"].isdigit():
                kw["merge"] = int(kw["merge"])
        # -n (ntsynt_run.py:17): edges fewer than all assemblies support stay
        if n_g >= 3 and rng.random() < 0.3:
            kw["n"] = int(rng.integers(2, n_g))
        # -m (ntsynt_run.py): the share of agreeing position differences that orients a contig
        if rng.random() < 0.25:
            kw["m"] = int(rng.choice([100, 75, 60, 51]))
        # the hidden switches of the reference's CLI and the Snakefile's experimental repeat filter, now and then
        kw.update(common=bool(rng.random() >= 0.12), simplify=bool(rng.random() >= 0.2), repeat=bool(rng.random() < 0.1))
        tmp = tempfile.mkdtemp(prefix="nts_stress_")
        try:
            if rng.random() < 0.35:
                # (round 4) an assembly-like family: repeat families, satellite arrays, segmental duplications, a tail of short scaffolds,
                # N gaps, half of the bases lower case -- generated in HBM, written out like bench.py's e2e inputs
                import bench
                from ntsynt_amd.device import Context, Genome
                gctx = Context(0)
                case["kind"] = "assembly-like"
                n_chrom = int(rng.choice([1, 2, 5]))
                scaf = int(rng.choice([n_chrom, 10, 60, 400]))
                paths = []
                for j in range(case["n"]):
                    plan = synth.realistic_plan(n_chrom, max(case["bp"] * 3 // n_chrom, 400_000), j, seed=case["seed"], n_scaffolds=scaf, n_tail=scaf,
                                                n_gaps=2 * scaf, sat_scale=float(rng.choice([0.01, 0.1, 0.3])), indel_bp=(100, 20000))
                    g = Genome.synth_plan(gctx, plan, case["seed"], 1000 + j, case["div"] / 2.0, rep=synth.REPEATS, names=plan[2])
                    p = os.path.join(tmp, f"asm{j}.fa")
                    bench.write_fasta_from_device(g, p, soft_mask_seed=case["seed"] + j, half_lower=True, line_width=(0, 60, 80)[j % 3])
                    g.free()
                    paths.append(p)
                gctx.close()
            else:
                paths = synth.make_family(tmp, case["n"], case["bp"], case["ctg"], case["div"], seed=case["seed"], micro=case["micro"],
                                          n_runs=case["n_runs"], soft_mask=True, line_width=(60 if case["seed"] % 2 else 0))
            os.makedirs(os.path.join(tmp, "ora"))
            os.makedirs(os.path.join(tmp, "hip"))
            os.chdir(os.path.join(tmp, "ora"))
            # (distant families may share no chain of four minimizers: both sides then stop with "no paths found", exit 1)
            # (and a last-round erosion walk may reach a vertex with two ways on: the reference asserts there,
            # bin/ntsynt_synteny.py:330 -- the oracle raises the same AssertionError, the product fails with that line in its message)
            ora_stop = eng_stop = None
            try:
                ora = SO.run_pipeline(paths, prefix="p", **kw)
            except SystemExit:
                ora, ora_stop = None, "no paths"
            except AssertionError:
                ora, ora_stop = None, "erosion assert"
            except IndexError:                                 # (a last round without a block of at least z bases: bin/ntsynt_synteny.py:437)
                ora, ora_stop = None, "erosion assert"
            os.chdir(os.path.join(tmp, "hip"))
            try:
                eng = pipeline.run(paths, prefix="p", log=lambda *a: None, **kw)
            except SystemExit:
                eng, eng_stop = None, "no paths"
            except Exception as exc:                           # noqa: BLE001
                if "ntsynt_synteny.py:330" not in str(exc) and "ntsynt_synteny.py:437" not in str(exc):
                    raise
                eng, eng_stop = None, "erosion assert"
            if ora_stop != eng_stop:
                print("PIPELINE MISMATCH: the two sides stop differently", ora_stop, eng_stop, case, kw, "seed", args.seed, "case", n)
                sys.exit(1)
            if ora_stop == "erosion assert":
                asserts += 1
                n += 1
                continue
            for p in paths:                                    # the minimizer TSVs (indexlr --seq text), byte for byte
                name = f"{os.path.basename(p)}.k{kw['k']}.w{kw['w']}.tsv"
                if open(os.path.join(tmp, "hip", name)).read() != open(os.path.join(tmp, "ora", name)).read():
                    print("MINIMIZER TSV MISMATCH", name, case, kw, "seed", args.seed, "case", n)
                    sys.exit(1)
            if eng is None:
                n += 1
                none_found += 1
                continue
            for name in ("p.synteny_blocks.tsv", "p.pre-collinear-merge.synteny_blocks.tsv"):
                if eng.outputs[name] != ora.outputs[name]:
                    print("PIPELINE MISMATCH", name, case, kw, "seed", args.seed, "case", n)
                    sys.exit(1)
            blocks += len(eng.outputs["p.synteny_blocks.tsv"].splitlines()) // max(case["n"], 1)
        finally:
            os.chdir(cwd)
            shutil.rmtree(tmp, ignore_errors=True)
        n += 1
    print(f"ok: {n} families end to end ({none_found} without any path on both sides, {asserts} stopped by the reference's erosion assert on both sides), "
          f"{blocks} synteny blocks, seed {args.seed}")


if __name__ == "__main__":
    main()
