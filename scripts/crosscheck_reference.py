#!/usr/bin/env python3
"""Cross-check of this build against the REAL reference tools, for whoever has them.

The container this build was made in holds neither btllib / ntJoin sources nor the reference's C. elegans FASTAs, so eleven
behaviours are recalled, not verified.  This script settles them on a machine that has
btllib's `indexlr`, the reference's `ntsynt_make_common_bf` and (optionally) its `ntSynt` on PATH -- or under --ref-bin -- plus one
MI355X for this build's twins in bin/:

  1. writes a small synthetic family built to separate the assumptions: lower-case stretches (u3), N runs and lone Ns (u5), records
     shorter than k, shorter than w + k - 1 and exactly that long (u5), headers with descriptions behind the id (u4), a k-mer that
     occurs twice in a genome (u6), and a first genome whose size makes btllib's constructor rounding (u1) change the byte count;
  2. runs `ntsynt_make_common_bf`, `indexlr` and `ntSynt` of BOTH sides on it with the Snakefile's own command lines
     (bin/ntsynt_run_pipeline.smk:55-103);
  3. diffs, and says what each difference implicates:
       <prefix>.bf header text      -> the header layout (no output bytes depend on it; --bf-signature)
       <prefix>.bf size             -> u1 (constructor rounding: try --bf-rounding down / none)
       <prefix>.bf bits             -> u2 (bit index / bit order), u3 (case folding), ntHash itself if the KATs had not pinned it
       minimizer TSV rows           -> u5 (window rule, N handling, short records), u4 (record ids), u12 with -r
       synteny TSVs                 -> u6-u11 (ntJoin's graph semantics, bedtools / NCLS interval ends) once the TSVs agree

Nothing here runs in the build container (the reference tools are absent); `tests/test_gpu_celegans_demo.py` is the companion that
runs the reference's own demo when NTS_CELEGANS_DIR points at its three FASTA files.

    python scripts/crosscheck_reference.py [--ref-bin DIR] [--workdir DIR] [-k 24] [-w 1000] [--keep]
"""
import argparse
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = os.path.join(ROOT, "bin")
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def find_tool(name, ref_bin):
    "the reference's executable `name`: under --ref-bin, else the first one on PATH that is not this build's twin"
    if ref_bin:
        p = os.path.join(ref_bin, name)
        return p if os.access(p, os.X_OK) else None
    for d in os.environ.get("PATH", "").split(os.pathsep):
        p = os.path.join(d, name)
        if os.access(p, os.X_OK) and os.path.realpath(d) != os.path.realpath(OURS):
            return p
    return None


def make_family(workdir, k, w, seed=7):
    """three genomes of ~1.2 Mbp; returns their paths.  The first file's size is chosen so that approximate_bf_size's byte count
    (src/ntsynt_make_common_bf.cpp:28-40) is NOT a multiple of 8: the constructor's rounding (u1) then shows in the file size."""
    rng = np.random.default_rng(seed)
    anc = [ACGT[rng.integers(0, 4, size=n)] for n in (600_011, 400_000, 150_000, w + k - 1, w + k - 2, k, k - 1, 3)]
    anc[2][70_000:70_000 + k] = anc[2][10_000:10_000 + k]          # a k-mer twice in one record (u6)
    paths = []
    for j in range(3):
        recs = []
        for a in anc:
            b = a.copy()
            hit = rng.random(b.size) < 0.01 * j
            b[hit] = ACGT[rng.integers(0, 4, size=int(hit.sum()))]
            recs.append(b)
        r0 = recs[0]
        r0[100_000:100_300] = ord("N")                              # an N run, a lone N, N at a record's end (u5)
        r0[250_000] = ord("N")
        recs[1][-5:] = ord("N")
        for lo in (5_000, 300_000, 420_000):                        # soft-masked stretches (u3)
            r0[lo:lo + 2_000] |= 0x20
        p = os.path.join(workdir, f"g{j}.fa")
        with open(p, "wb") as fh:
            for i, r in enumerate(recs):
                fh.write(f">ctg{i + 1} some description len={r.size}\n".encode())     # id = up to the first blank (u4)
                for s in range(0, r.size, 60):
                    fh.write(r[s:s + 60].tobytes() + b"\n")
        paths.append(p)
    return paths


def run(cmd, cwd, log):
    log.write("$ " + " ".join(cmd) + "\n")
    log.flush()
    r = subprocess.run(cmd, cwd=cwd, stdout=log, stderr=subprocess.STDOUT)
    return r.returncode


def split_bf(path):
    "(header bytes, bit array bytes) of a btllib filter file; the header ends with the [HeaderEnd] line"
    raw = open(path, "rb").read()
    mark = b"[HeaderEnd]\n"
    at = raw.find(mark)
    if at < 0:
        return raw[:256], raw
    return raw[:at + len(mark)], raw[at + len(mark):]


def md5(b):
    return hashlib.md5(b).hexdigest()


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ref-bin", help="directory holding the reference's indexlr / ntsynt_make_common_bf / ntSynt [search PATH]")
    ap.add_argument("--workdir", help="where to work [a temp dir]")
    ap.add_argument("-k", type=int, default=24)
    ap.add_argument("-w", type=int, default=1000)
    ap.add_argument("--fpr", type=float, default=0.025)
    ap.add_argument("--keep", action="store_true", help="keep the work directory")
    args = ap.parse_args()
    tools = {n: find_tool(n, args.ref_bin) for n in ("ntsynt_make_common_bf", "indexlr", "ntSynt")}
    missing = [n for n in ("ntsynt_make_common_bf", "indexlr") if not tools[n]]
    if missing:
        print("reference tools not found: " + ", ".join(missing) + " (btllib's indexlr and the reference's ntsynt_make_common_bf must be on PATH "
              "or under --ref-bin).  Nothing to compare with; see the module docstring.", file=sys.stderr)
        return 2
    work = args.workdir or tempfile.mkdtemp(prefix="nts_crosscheck_")
    os.makedirs(work, exist_ok=True)
    ref_dir, our_dir = os.path.join(work, "reference"), os.path.join(work, "this_build")
    os.makedirs(ref_dir, exist_ok=True)
    os.makedirs(our_dir, exist_ok=True)
    paths = make_family(work, args.k, args.w)
    k, w = str(args.k), str(args.w)
    findings = []
    with open(os.path.join(work, "commands.log"), "w") as log:
        for side, cwd, exe in (("reference", ref_dir, lambda n: tools[n]), ("this build", our_dir, lambda n: os.path.join(OURS, n))):
            # rule make_common_bf (smk:55-63), rule indexlr (smk:74-85)
            rc = run([exe("ntsynt_make_common_bf"), "--genome"] + paths + ["-k", k, "--fpr", str(args.fpr), "-p", "x", "-t", "4"], cwd, log)
            if rc:
                findings.append(f"{side}: ntsynt_make_common_bf exited {rc} (see commands.log)")
                continue
            for p in paths:
                out = os.path.join(cwd, os.path.basename(p) + f".k{k}.w{w}.tsv")
                run([exe("indexlr"), "--long", "--pos", "--seq", "-k", k, "-w", w, "-t", "4", "-s", "x.bf", "-o", out, p], cwd, log)
        # ---- the filter file -------------------------------------------------------------------------------------------------
        a, b = os.path.join(ref_dir, "x.bf"), os.path.join(our_dir, "x.bf")
        if os.path.exists(a) and os.path.exists(b):
            ha, ba = split_bf(a)
            hb, bb = split_bf(b)
            if ha != hb:
                findings.append("x.bf: header text differs -> the recalled header layout (file only; rerun this build with --bf-signature / "
                                "adjust ntsynt_amd/pipeline.py bf_header).\n    reference: %r\n    this build: %r" % (ha, hb))
            if len(ba) != len(bb):
                findings.append(f"x.bf: bit array of {len(ba)} bytes against {len(bb)} -> u1, the constructor's rounding of the byte count "
                                f"(this build: --bf-rounding up; try down / none: every bit index is h0 mod (8 x bytes))")
            elif ba != bb:
                na, nb = np.frombuffer(ba, np.uint8), np.frombuffer(bb, np.uint8)
                pa, pb = int(np.unpackbits(na).sum()), int(np.unpackbits(nb).sum())
                rev = np.packbits(np.unpackbits(nb, bitorder="little"), bitorder="big")
                hint = "same popcount: bit ORDER inside the byte (u2)" if pa == pb and np.array_equal(na, rev) else \
                       "different popcount: case folding (u3) or which k-mers are hashed (N handling)" if pa != pb else \
                       "same popcount, other positions: the bit index (u2: h0 mod bits) or the hash"
                findings.append(f"x.bf: same size, bits differ (popcount {pa} against {pb}) -> {hint}")
            else:
                findings.append(f"x.bf: identical bit array ({len(ba)} bytes, md5 {md5(ba)}): u1, u2, u3 hold")
        # ---- minimizer TSVs ----------------------------------------------------------------------------------------------------
        tsv_ok = True
        for p in paths:
            name = os.path.basename(p) + f".k{k}.w{w}.tsv"
            fa, fb = os.path.join(ref_dir, name), os.path.join(our_dir, name)
            if not (os.path.exists(fa) and os.path.exists(fb)):
                findings.append(f"{name}: missing on one side")
                tsv_ok = False
                continue
            la, lb = open(fa).read().splitlines(), open(fb).read().splitlines()
            if la == lb:
                continue
            tsv_ok = False
            ids_a, ids_b = [x.split("\t")[0] for x in la], [x.split("\t")[0] for x in lb]
            if ids_a != ids_b:
                findings.append(f"{name}: record ids / lines differ ({len(la)} against {len(lb)} lines; first ids {ids_a[:3]} / {ids_b[:3]}) -> u4 "
                                "(id = header up to the first blank) or whether records without minimizers get a line")
            for x, y in zip(la, lb):
                if x != y:
                    ta, tb = x.split("\t")[1].split() if "\t" in x else [], y.split("\t")[1].split() if "\t" in y else []
                    findings.append(f"{name}: record {x.split(chr(9))[0]}: {len(ta)} against {len(tb)} minimizers -> u5 (window = last w valid k-mers, "
                                    f"`<=` tie rule, N handling, records shorter than w k-mers); first differing token "
                                    f"{next(((p_, q_) for p_, q_ in zip(ta, tb) if p_ != q_), ('(length)', ''))}")
                    break
        if tsv_ok:
            findings.append("minimizer TSVs: identical for all three genomes: u4, u5 hold (and B1-B4 against the real indexlr)")
        # ---- the whole pipeline ------------------------------------------------------------------------------------------------
        if tools["ntSynt"] and tsv_ok:
            for side, cwd, exe in (("reference", ref_dir, tools["ntSynt"]), ("this build", our_dir, os.path.join(OURS, "ntSynt"))):
                run([exe, "-d", "1", "-k", k, "-w", w, "--prefix", "full", "--force"] + paths, cwd, log)
            for name in ("full.synteny_blocks.tsv", "full.pre-collinear-merge.synteny_blocks.tsv"):
                fa, fb = os.path.join(ref_dir, name), os.path.join(our_dir, name)
                if os.path.exists(fa) and os.path.exists(fb):
                    same = open(fa, "rb").read() == open(fb, "rb").read()
                    findings.append(f"{name}: {'identical' if same else 'DIFFERS -> u6-u11 (ntJoin graph semantics, bedtools slop / NCLS interval ends): diff the two files'}")
                else:
                    findings.append(f"{name}: missing on one side (see commands.log)")
        elif not tools["ntSynt"]:
            findings.append("ntSynt (reference) not found: the graph stage (u6-u11) was not compared")
    print("\n".join(findings))
    print(f"work directory: {work}" if args.keep or args.workdir else "")
    if not (args.keep or args.workdir):
        shutil.rmtree(work, ignore_errors=True)
    return 0 if all("DIFFERS" not in f and "differ" not in f.split("->")[0] for f in findings) else 1


if __name__ == "__main__":
    sys.exit(main())
