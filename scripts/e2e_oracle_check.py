#!/usr/bin/env python3
"""One-off check of the headline end-to-end run against the CPU oracle: bench.py's e2e family (3 x 3 Gbp with structural
events, soft-masked stretches) through the product pipeline AND through oracle/synteny_oracle.py's pipeline on the host
cores, byte comparison of both synteny TSVs (and of the minimizer TSVs and the filter).  Tens of minutes of CPU at full
size, so it runs outside bench.py; its record (profiles/r03_e2e_oracle.json, copied from gpurun_out/) is what bench.py
quotes as e2e.oracle_md5.

  python scripts/e2e_oracle_check.py --mbp 3000 --threads 16 --out gpurun_out/r03_e2e_oracle.json
"""
import argparse
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def md5_file(path):
    h = hashlib.md5()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbp", type=float, default=3000.0)
    ap.add_argument("--genomes", type=int, default=3)
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--divergence", type=float, default=0.01)
    ap.add_argument("-k", type=int, default=24)
    ap.add_argument("-w", type=int, default=1000)
    ap.add_argument("--fpr", type=float, default=0.025)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--substitutions-only", action="store_true")
    ap.add_argument("--family", choices=["structural", "assembly-like"], default="structural")
    ap.add_argument("--out", default="gpurun_out/r03_e2e_oracle.json")
    args = ap.parse_args()
    import bench
    from oracle import synteny_oracle as SO
    total_bp = int(args.mbp * 1e6)
    work = tempfile.mkdtemp(prefix="nts_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        paths = bench.e2e_inputs(args, 0, args.genomes, total_bp, args.contigs, args.divergence, work)
        hip_dir, ora_dir = os.path.join(work, "hip"), os.path.join(work, "ora")
        os.makedirs(hip_dir)
        os.makedirs(ora_dir)
        product = bench.e2e_leg(args, 0, args.genomes, total_bp, args.contigs, args.divergence, hip_dir, paths=paths)
        a, _ = bench.e2e_params(args, paths, args.divergence)
        cwd = os.getcwd()
        os.chdir(ora_dir)
        try:
            t = time.time()
            ora = SO.run_pipeline(paths, k=a.k, w=a.w, fpr=a.fpr, prefix=a.prefix, w_rounds=a.w_rounds, indel=a.indel, merge=a.merge,
                                  block_size=a.block_size, threads=args.threads)
            ora_s = time.time() - t
            import resource
            ora_rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss * 1024
        finally:
            os.chdir(cwd)
        names = ["e2e.synteny_blocks.tsv", "e2e.pre-collinear-merge.synteny_blocks.tsv"]
        names += [f"{os.path.basename(p)}.k{a.k}.w{a.w}.tsv" for p in paths]
        same = {}
        for n in names:
            same[n] = md5_file(os.path.join(hip_dir, n)) == md5_file(os.path.join(ora_dir, n))
        # the filter: the oracle keeps it in memory, the product wrote <prefix>.common.bf (header + bits)
        import numpy as np
        with open(os.path.join(hip_dir, "e2e.common.bf"), "rb") as fh:
            blob = fh.read(4096)
            hdr_end = blob.index(b"[HeaderEnd]\n") + len(b"[HeaderEnd]\n")
        bits = np.fromfile(os.path.join(hip_dir, "e2e.common.bf"), dtype=np.uint8, offset=hdr_end)
        same["common filter bits"] = bool(bits.size == ora.bf.size and np.array_equal(bits, ora.bf))
        # the whole minimizer output and the filter as digests, for the suite's full-size identity test (tests/test_gpu_scale.py)
        from ntsynt_amd import fasta as fa
        digests = {}
        for p in paths:
            n = f"{os.path.basename(p)}.k{a.k}.w{a.w}.tsv"
            _, h1, pos, line = fa.read_indexlr_tsv(os.path.join(ora_dir, n))     # (lines are the FASTA's records, in order)
            digests[n] = bench.mx_digest(h1, line, pos)
        pad = (-ora.bf.size) % 8
        words = np.concatenate([ora.bf, np.zeros(pad, dtype=np.uint8)]).view(np.uint64) if pad else ora.bf.view(np.uint64)
        filter_popcount = int(np.bitwise_count(words).sum(dtype=np.uint64))
        tsv = ora.outputs["e2e.synteny_blocks.tsv"]
        rec = {"key": bench.e2e_key(args, args.genomes, total_bp, args.contigs, args.divergence),
               "oracle_md5": hashlib.md5(tsv.encode()).hexdigest(), "product_md5": product["tsv_md5"],
               "identical": same, "all_identical": all(same.values()),
               "oracle_minimizer_digests": digests, "oracle_filter_bytes": int(ora.bf.size), "oracle_filter_popcount": filter_popcount,
               "oracle_seconds": round(ora_s, 1), "oracle_threads": args.threads, "product_seconds": product["seconds"],
               "oracle_peak_rss_bytes": ora_rss, "product_peak_hbm_bytes": product.get("peak_hbm_bytes"),
               "blocks": product["blocks"], "engine_stats": product["engine_stats"],
               "what": product["what"] + "; oracle = oracle/synteny_oracle.py run_pipeline on the same files, host cores of the GPU box"}
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as fh:
            json.dump(rec, fh, indent=1)
        print(json.dumps(rec), flush=True)
        return 0 if rec["all_identical"] else 1
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
