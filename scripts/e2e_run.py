#!/usr/bin/env python3
"""End-to-end wall clock of the GPU pipeline: FASTA files on disk -> final synteny TSV (stage times in
{prefix}.stage_times.tsv).  Synthetic family written to --dir first (not timed).

  python scripts/e2e_run.py --mbp 100 --genomes 3 -d 1
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbp", type=float, default=100.0)
    ap.add_argument("--genomes", type=int, default=3)
    ap.add_argument("--contigs", type=int, default=4)
    ap.add_argument("--divergence", type=float, default=0.01)
    ap.add_argument("--dir", default="/tmp/e2e")
    ap.add_argument("--oracle", action="store_true", help="also run the CPU oracle pipeline and compare the TSVs")
    ap.add_argument("--oracle-only", action="store_true", help="no GPU run: time the oracle pipeline and print the TSV's md5 "
                    "(the family is deterministic, so this can run on another machine than the GPU run)")
    ap.add_argument("--cprofile", default=None, help="write a cProfile summary of the GPU pipeline run to this file")
    ap.add_argument("--out", default=None, help="also write the JSON here (after the GPU run, again after the oracle)")
    args = ap.parse_args()
    from ntsynt_amd import cli, pipeline, synth
    os.makedirs(args.dir, exist_ok=True)
    if args.out:
        args.out = os.path.abspath(args.out)
    if args.cprofile:
        args.cprofile = os.path.abspath(args.cprofile)
    t = time.time()
    paths = synth.make_family(args.dir, args.genomes, int(args.mbp * 1e6), args.contigs, args.divergence, micro=20)
    t_gen = time.time() - t
    os.chdir(args.dir)
    pct = args.divergence * 100
    parser = cli.build_parser()
    a = parser.parse_args(paths + ["-d", str(pct), "-p", "e2e"])
    cli.resolve(parser, a)
    import hashlib
    out = {"workload": f"{args.genomes} x {args.mbp:g} Mbp FASTA in {args.contigs} records, -d {pct:g} (w_rounds {a.w_rounds}, "
                       f"indel {a.indel}, merge {a.merge}, block {a.block_size})", "generate_inputs_s": round(t_gen, 2)}
    eng = None
    if not args.oracle_only:
        if args.cprofile:
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        t = time.time()
        eng = pipeline.run(paths, k=a.k, w=a.w, fpr=a.fpr, prefix=a.prefix, w_rounds=a.w_rounds, indel=a.indel, merge=a.merge,
                           block_size=a.block_size, benchmark=True, log=lambda *x: None)
        wall = time.time() - t
        if args.cprofile:
            import io
            import pstats
            prof.disable()
            buf = io.StringIO()
            pstats.Stats(prof, stream=buf).sort_stats("cumulative").print_stats(70)
            with open(args.cprofile, "w") as fh:
                fh.write(buf.getvalue())
        tsv = eng.outputs["e2e.synteny_blocks.tsv"]
        out.update({"end_to_end_s": round(wall, 3), "stages_s": {n: round(s, 3) for n, s in eng.stage_times},
                    "blocks": len(tsv.splitlines()) // args.genomes, "engine_stats": eng.stats,
                    "tsv_md5": hashlib.md5(tsv.encode()).hexdigest()})
        if eng.times:
            out["engine_times_s"] = {n: round(v, 3) for n, v in sorted(eng.times.items(), key=lambda kv: -kv[1])}
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)
    if args.oracle or args.oracle_only:
        from oracle import synteny_oracle as SO
        os.makedirs("ora", exist_ok=True)
        os.chdir("ora")
        t = time.time()
        ora = SO.run_pipeline(paths, k=a.k, w=a.w, fpr=a.fpr, prefix="e2e", w_rounds=a.w_rounds, indel=a.indel, merge=a.merge,
                              block_size=a.block_size, threads=os.cpu_count(), write_mx_tsv=False)
        out["oracle_s"] = round(time.time() - t, 2)
        out["oracle_threads"] = os.cpu_count()
        out["oracle_tsv_md5"] = hashlib.md5(ora.outputs["e2e.synteny_blocks.tsv"].encode()).hexdigest()
        out["oracle_blocks"] = len(ora.outputs["e2e.synteny_blocks.tsv"].splitlines()) // args.genomes
        if eng is not None:
            out["identical_to_oracle"] = ora.outputs["e2e.synteny_blocks.tsv"] == eng.outputs["e2e.synteny_blocks.tsv"]
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
