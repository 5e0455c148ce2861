#!/usr/bin/env python3
"""FASTA files on disk -> final synteny TSV at full size with the family generated in HBM (what bench.py's e2e leg does),
per stage and, with NTS_ENGINE_TIMES=1, per step of the graph stage; --family assembly-like for the c5_like family.

  NTS_ENGINE_TIMES=1 python scripts/e2e_synth.py --mbp 3000 --genomes 3 --contigs 24 --divergence 0.01
"""
import argparse
import json
import os
import shutil
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbp", type=float, default=3000.0)
    ap.add_argument("--genomes", type=int, default=3)
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--divergence", type=float, default=0.01)
    ap.add_argument("-k", type=int, default=24)
    ap.add_argument("-w", type=int, default=1000)
    ap.add_argument("--fpr", type=float, default=0.025)
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--substitutions-only", action="store_true")
    ap.add_argument("--family", choices=["structural", "assembly-like"], default="structural")
    args = ap.parse_args()
    import bench
    work = tempfile.mkdtemp(prefix="nts_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        for _ in range(args.repeat):
            out = bench.e2e_leg(args, 0, args.genomes, int(args.mbp * 1e6), args.contigs, args.divergence, work)
            print(json.dumps(out), flush=True)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
