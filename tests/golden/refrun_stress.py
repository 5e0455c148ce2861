#!/usr/bin/env python3
"""Randomised comparison of the reference's OWN graph stage (bin/ntsynt_synteny.py run over the stand-ins of make_golden_refrun.py)
with the CPU restatement (oracle/synteny_oracle.py run_pipeline) AND the product's host-array engine (ntsynt_amd/synteny.py, graph build
and re-sketch from the CPU test doubles) on families and parameter sets drawn at random: the run's
pre-collinear-merge and final TSVs, its interarrival file and its --dev warnings must be identical.  Nothing is stored but the log
(profiles/r06_refrun_stress.log); the thirteen committed scenarios are what the test suite replays.  Runs ONLY where /root/reference
exists:  PYTHONHASHSEED=0 python tests/golden/refrun_stress.py [--seconds 300] [--seed 1]"""
import argparse
import contextlib
import io
import os
import random
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden_refrun as M  # noqa: E402
from oracle import synteny_oracle as SO  # noqa: E402


def oracle_outputs(sc, fastas):
    orig = SO.SyntenyOracle.__init__

    def init(self, *a, **k):
        orig(self, *a, **k)
        self.dev = True
    err = io.StringIO()
    SO.SyntenyOracle.__init__ = init
    try:
        with contextlib.redirect_stderr(err), contextlib.redirect_stdout(io.StringIO()):
            k, w = sc["k"], sc["w"]
            from oracle import nts_oracle as O
            genomes = {p: O.read_fasta(p) for p in fastas}
            bf = O.common_bf(genomes, k, 0.025) if sc.get("common", True) else None
            tables, by_tsv = {}, {}
            for p in fastas:
                tsv = f"{os.path.basename(p)}.k{k}.w{w}.tsv"
                tables[tsv] = SO.mx_tables_from_tokens(SO.mx_records_from_arrays(genomes[p].names, O.minimize(genomes[p], k, w, bf)))
                by_tsv[tsv] = genomes[p]
            if sc.get("filter"):
                rep = O.repeat_bf([genomes[p] for p in fastas], k, int(bf.size))          # (sized like the common filter, as make_golden_refrun.py does)
                if sc["filter"] == "Filter":
                    import tempfile as _tf
                    with _tf.TemporaryDirectory() as td:
                        for p in fastas:
                            tsv = f"{os.path.basename(p)}.k{k}.w{w}.tsv"
                            O.write_indexlr_tsv(os.path.join(td, tsv), genomes[p], O.minimize(genomes[p], k, w, bf), k)
                            tables[tsv] = SO.read_minimizers_tsv(os.path.join(td, tsv), repeat_bf=rep)
            eng = SO.SyntenyOracle(list(tables), by_tsv, k, w, sc["w_rounds"], sc["indel"], sc["merge"], sc["z"], "ora", bf=bf,
                                   n=sc.get("min_weight", 0), interarrivals=True, simplify=sc.get("simplify", True), m=sc.get("m", 90))
            if sc.get("filter") == "Indexlr":
                eng.refine_repeat = rep
            elif sc.get("filter") == "Filter":
                eng.screen_repeat = rep
            eng.load(tables)
            eng.main()
    finally:
        SO.SyntenyOracle.__init__ = orig
    return eng.outputs, [ln for ln in err.getvalue().splitlines() if ln.startswith("WARNING")]


class _Shim:
    "what tests.test_refrun_product.host_engine reads of a Scenario"

    def __init__(self, sc):
        self.meta = dict(sc, indel=sc["indel"], merge=sc["merge"], z=sc["z"])
        self.prefix = "eng"
        self.min_weight = sc.get("min_weight", 0)
        self.simplify = sc.get("simplify", True)
        self.m = sc.get("m", 90)
        self.filter_mode = sc.get("filter")

    def repeat_filter(self, genomes):
        from oracle import nts_oracle as O
        common = O.common_bf({i: g for i, g in enumerate(genomes)}, self.meta["k"], 0.025)
        return O.repeat_bf(genomes, self.meta["k"], int(common.size))


def product_outputs(sc, fastas):
    "the product's host-array engine (ntsynt_amd/synteny.py; graph build and re-sketch from the CPU test doubles) on the same family"
    from tests.test_refrun_product import host_engine
    eng, initial = host_engine(_Shim(sc), fastas)
    err = io.StringIO()
    with contextlib.redirect_stderr(err), contextlib.redirect_stdout(io.StringIO()):
        out = eng.run(initial)
    return out, [ln for ln in err.getvalue().splitlines() if ln.startswith("WARNING")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    ns, sb, ab = M.install()
    rng = random.Random(a.seed)
    M.OUT = tempfile.mkdtemp(prefix="refrun_stress_")
    t_end = time.time() + a.seconds
    n = same = stopped = 0
    seen = {"min_weight_below_G": 0, "no_common": 0, "blocks": 0, "warnings": 0, "not_oriented": 0}
    while time.time() < t_end:
        G = rng.choice([2, 2, 3, 3, 4, 5])
        sc = dict(name=f"t{n}", n=G, bp=rng.choice([50_000, 90_000, 140_000, 200_000]), ctg=rng.choice([1, 2, 3, 6]),
                  div=rng.choice([0.002, 0.005, 0.01, 0.02, 0.04]), seed=2000 + rng.randrange(10 ** 6), k=rng.choice([16, 20, 24, 32]),
                  w=rng.choice([30, 50, 80, 150, 400]), w_rounds=rng.choice([[10, 4], [15, 6], [20], [25, 8], [12, 5]]),
                  indel=rng.choice([150, 400, 1500, 10000]), merge=rng.choice(["1w", "3w", 60, 600, 5000]), z=rng.choice([40, 100, 300]),
                  micro=rng.choice([0, 8, 20, 40]), n_runs=rng.random() < 0.4)
        if sc["w_rounds"][0] >= sc["w"]:
            continue
        if rng.random() < 0.3 and G > 2:
            sc["min_weight"] = rng.randint(2, G - 1)
            seen["min_weight_below_G"] += 1
        if rng.random() < 0.15:
            sc["common"] = False
            seen["no_common"] += 1
        if rng.random() < 0.2:                                      # ntSynt --no-simplify-graph
            sc["simplify"] = False
            seen["no_simplify"] = seen.get("no_simplify", 0) + 1
        if rng.random() < 0.15 and sc.get("common", True):          # ntsynt_run.py --filter Indexlr | Filter --repeat <bf>
            sc["filter"] = rng.choice(["Indexlr", "Filter"])
            seen["repeat_filter"] = seen.get("repeat_filter", 0) + 1
        if rng.random() < 0.3:                                      # ntsynt_run.py -m
            sc["m"] = rng.choice([100, 75, 60, 51])
            seen["m_not_90"] = seen.get("m_not_90", 0) + 1
        n += 1
        cwd = os.getcwd()
        try:
            try:
                outputs, counts, meta = M.run_scenario(ns, sc)
            except (SystemExit, IndexError, AssertionError) as e:          # no paths / the reference's merge on an empty list / its erosion assert
                ref_stop = type(e).__name__
                outputs = None
            d = os.path.join(M.OUT, sc["name"])
            with tempfile.TemporaryDirectory() as tmp:
                os.chdir(tmp)
                fastas = []
                import gzip
                if outputs is None and sc.get("filter"):
                    stopped += 1                                           # (its files carried a repeated stretch the generator put in: not made again)
                    continue
                if outputs is None:                                        # (the scenario's files were not kept: make the family again)
                    from ntsynt_amd import synth
                    fastas = synth.make_family(tmp, sc["n"], sc["bp"], sc["ctg"], sc["div"], seed=sc["seed"], n_runs=sc["n_runs"], micro=sc["micro"], line_width=0)
                else:
                    for f in meta["fastas"]:
                        with gzip.open(os.path.join(d, f + ".gz")) as fi, open(f, "wb") as fo:
                            fo.write(fi.read())
                        fastas.append(os.path.join(tmp, f))
                try:
                    ora, warns = oracle_outputs(sc, fastas)
                    ora_stop = None
                except (SystemExit, IndexError, AssertionError) as e:
                    ora, ora_stop = None, type(e).__name__
                try:
                    prod, pwarns = product_outputs(sc, fastas)
                    prod_stop = None
                except (SystemExit, IndexError, AssertionError) as e:
                    prod, prod_stop = None, type(e).__name__
                if (prod is None) != (ora is None) or (prod is not None and not (
                        prod["eng.synteny_blocks.tsv"] == ora["ora.synteny_blocks.tsv"] and pwarns == warns and
                        prod["eng.pre-collinear-merge.synteny_blocks.tsv"] == ora["ora.pre-collinear-merge.synteny_blocks.tsv"] and
                        sorted(prod["eng.interarrivals.tsv"].splitlines()) == sorted(ora["ora.interarrivals.tsv"].splitlines()))):
                    print("MISMATCH between the product's host engine and the restatement", sc, prod_stop, ora_stop, flush=True)
                    sys.exit(1)
            if outputs is None or ora is None:
                ok = (outputs is None) == (ora is None)
                stopped += ok
                if not ok:
                    print("MISMATCH (one side stopped)", sc, "reference:", None if outputs is not None else ref_stop, "oracle:", ora_stop, flush=True)
                    sys.exit(1)
                continue
            pre = meta["prefix"]
            ok = (ora["ora.synteny_blocks.tsv"] == outputs[f"{pre}.synteny_blocks.tsv"].replace("", "") and
                  ora["ora.pre-collinear-merge.synteny_blocks.tsv"] == outputs[f"{pre}.pre-collinear-merge.synteny_blocks.tsv"] and
                  ora["ora.interarrivals.tsv"] == outputs[f"{pre}.interarrivals.tsv"] and warns == meta["warnings"])
            if not ok:
                print("MISMATCH", sc, flush=True)
                sys.exit(1)
            same += 1
            seen["blocks"] += len(outputs[f"{pre}.synteny_blocks.tsv"].splitlines()) // G
            seen["warnings"] += len(meta["warnings"])
            seen["not_oriented"] += meta["n_not_oriented"]
        finally:
            os.chdir(cwd)
            import shutil
            shutil.rmtree(os.path.join(M.OUT, sc["name"]), ignore_errors=True)
    print(f"ok: {n} random scenarios -- {same} with every output of the reference's run identical to the restatement's and the product's host engine's, {stopped} on which all three "
          f"stop (no paths / the reference's own IndexError on an empty final list / its erosion assert); {seen}, seed {a.seed}", flush=True)


if __name__ == "__main__":
    main()
