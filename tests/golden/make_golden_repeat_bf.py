#!/usr/bin/env python3
"""Vectors made by RUNNING the reference's repeat-filter script -- build container only (/root/reference):

bin/ntsynt_make_repeat_bfs.py's main() (rule make_repeat_bf, bin/ntsynt_run_pipeline.smk:65-72; experimental in the reference), imported and
run on small families over a stand-in `btllib`:

  KmerBloomFilter(bytes, 1, k)   a byte array of the constructor-rounded size (multiple of 8: u1), bit `h0 mod bits`, LSB first in its byte (u2)
                                 -- the rules of oracle/nts_oracle.c, through its own `contains`; insert sets the same bit
  SeqReader(file, LONG_MODE, t)  the records of the file in order, sequence text as in the file
  NtHash(seq, 1, k)              roll() steps to the next k-mer without a non-ACGT base, hashes() = [its canonical ntHash2 value] (oracle hash_all)

What the run pins is the script's own logic: which genome sizes the filter (`--genome`'s first), `int(ceil(-n / ln(1 - fpr)) / 8)` bytes and its
message, `--bf <n>{B,k,M,G}` (decimal units; anything else: help + error, status 2), a FRESH per-genome filter for every genome, "seen before in this
genome -> into the repeat filter, else into the genome's own" k-mer by k-mer in file order (so a false positive of the genome's own filter counts
as a repeat), the default prefix `out.bf` -> `out.bf.bf`.  Kept per case: arguments, status / error, the message, the saved file's name, the filter's
size, popcount and SHA-1.  tests/test_repeat_bf_refrun.py holds oracle.nts_oracle.repeat_bf and bin/ntsynt_make_repeat_bfs' argument handling
against them; tests/test_gpu_stages.py the HIP build (nts_bf_insert_repeats) against the same digests.  No reference source text is stored.

  python tests/golden/make_golden_repeat_bf.py"""
import contextlib
import gzip
import hashlib
import importlib.machinery
import importlib.util
import io
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from ntsynt_amd import synth  # noqa: E402
from oracle import nts_oracle as O  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(HERE, "repeat_bf")
SAVED = {}


class KmerBloomFilter:
    def __init__(self, nbytes, hash_num, k):
        assert hash_num == 1
        self.bits = np.zeros(O.bf_ctor_bytes(int(nbytes)), dtype=np.uint8)
        self.k = k

    def contains(self, hashes):
        return O.bf_contains(self.bits, hashes[0])

    def insert(self, hashes):
        idx = int(hashes[0]) % (self.bits.size * 8)
        self.bits[idx >> 3] |= np.uint8(1 << (idx & 7))

    def save(self, path):
        SAVED[path] = self.bits.copy()


class SeqReader:
    def __init__(self, path, flag, threads):
        self.g = O.read_fasta(path)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def __iter__(self):
        for i in range(len(self.g.names)):
            yield types.SimpleNamespace(id=self.g.names[i], seq=bytes(self.g.record(i)).decode())


class NtHash:
    def __init__(self, seq, hash_num, k):
        assert hash_num == 1
        self.pos, self.h0 = O.hash_all(seq.encode(), k)
        self.i = -1

    def roll(self):
        self.i += 1
        return self.i < self.h0.size

    def hashes(self):
        return [int(self.h0[self.i])]


def load_reference():
    bt = types.ModuleType("btllib")
    bt.KmerBloomFilter, bt.SeqReader, bt.NtHash = KmerBloomFilter, SeqReader, NtHash
    bt.SeqReaderFlag = types.SimpleNamespace(LONG_MODE=2)
    sys.modules["btllib"] = bt
    loader = importlib.machinery.SourceFileLoader("ref_make_repeat_bfs", os.path.join(REF, "bin", "ntsynt_make_repeat_bfs.py"))
    spec = importlib.util.spec_from_loader("ref_make_repeat_bfs", loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


def last_error(stderr):
    for line in reversed(stderr.splitlines()):
        if ": error: " in line:
            return line.split(": error: ", 1)[1]
    return None


def family(tmp, name, n, bp, ctg, seed, n_runs):
    "a small family with something repeated inside every genome: a stretch of the first record once more further down, and a short tandem array"
    paths = synth.make_family(tmp, n, bp, ctg, 0.01, seed=seed, n_runs=n_runs, micro=6, line_width=0)
    out = []
    for j, p in enumerate(paths):
        g = O.read_fasta(p)
        recs = [bytes(g.record(i)) for i in range(len(g.names))]
        r0 = recs[0]
        a, b, at = len(r0) // 9, len(r0) // 9 + 1500, len(r0) // 2
        unit = r0[100:137]
        recs[0] = r0[:at] + r0[a:b] + r0[at:] + unit * 12
        q = os.path.join(tmp, f"{name}{j}.fa")
        with open(q, "wb") as fh:
            for nm, rec in zip(g.names, recs):
                fh.write(b">" + nm.encode() + b"\n" + rec + b"\n")
        os.remove(p)
        out.append(os.path.basename(q))
    return out


def run(mod, argv, cwd):
    SAVED.clear()
    so, se = io.StringIO(), io.StringIO()
    here, old = os.getcwd(), sys.argv
    os.chdir(cwd)
    sys.argv = ["ntsynt_make_repeat_bfs.py"] + argv
    rec = {"argv": argv}
    try:
        with contextlib.redirect_stdout(so), contextlib.redirect_stderr(se):
            try:
                mod.main()
                rec["end"] = "ran"
            except SystemExit as exc:
                rec.update(end="exit", status=exc.code, error=last_error(se.getvalue()), printed_help="usage:" in so.getvalue())
    finally:
        sys.argv = old
        os.chdir(here)
    rec["stdout"] = [ln for ln in so.getvalue().splitlines() if ln.startswith("Calculated")]
    if SAVED:
        (path, bits), = SAVED.items()
        rec.update(saved=path, bytes=int(bits.size), popcount=int(O.bf_popcount(bits)), sha1=hashlib.sha1(bits.tobytes()).hexdigest())
    return rec


def main():
    mod = load_reference()
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    cases = []
    with tempfile.TemporaryDirectory() as tmp:
        fa = family(tmp, "a", 3, 60_000, 2, 31, False)
        fb = family(tmp, "b", 2, 40_000, 3, 32, True)
        for f in fa + fb:
            with open(os.path.join(tmp, f), "rb") as fi, gzip.GzipFile(os.path.join(OUT, f + ".gz"), "wb", mtime=0) as fo:
                fo.write(fi.read())
        for argv in (["--genome"] + fa + ["-k", "24"],
                     ["--genome"] + fa + ["-k", "24", "--fpr", "0.05", "-p", "rep"],
                     ["--genome"] + list(reversed(fa)) + ["-k", "24", "-p", "rev"],          # the first genome sizes the filter; order changes the bits
                     ["--genome", fa[0], "-k", "20", "--bf", "64k", "-p", "one"],
                     ["--genome"] + fb + ["-k", "16", "--bf", "100001B", "-p", "odd"],        # (not a multiple of 8: the constructor's rounding)
                     ["--genome"] + fb + ["-k", "32", "--bf", "1M", "-p", "m", "-t", "7"],
                     ["--genome"] + fb + ["-k", "24", "--bf", "10"],
                     ["--genome"] + fb + ["-k", "24", "--bf", "1.5M"],
                     ["--genome"] + fb + ["-k", "24", "--bf", "5T"],
                     ["--genome"] + fb + ["-k", "24", "--bf", "2g"],
                     ["--genome"] + fb + ["-k", "24", "--bf", ""],                            # (empty: falsy, the size is calculated)
                     ["--genome"] + fb,                                                       # -k is required
                     ["--genome"] + fb + ["-k", "x"]):
            cases.append(run(mod, argv, tmp))
    units = {}
    parser = types.SimpleNamespace(print_help=lambda: None, error=lambda msg: (_ for _ in ()).throw(ValueError(msg)))
    for text in ("1B", "7k", "12M", "3G", "0B", "00012k", "1K", "1m", "12", "k", "1 k", "-1k", "1kB", "1e3B", "٣k"):
        try:
            units[text] = mod.parse_bf_size(text, parser)
        except ValueError as exc:
            units[text] = "error: " + str(exc)
    with open(os.path.join(OUT, "cases.json"), "w") as fh:
        json.dump({"cases": cases, "parse_bf_size": units, "families": {"a": fa, "b": fb}}, fh, indent=1, sort_keys=True, ensure_ascii=True)
        fh.write("\n")
    for c in cases:
        print(c["argv"][-6:], c["end"], c.get("status"), c.get("error"), c.get("stdout"), c.get("saved"), c.get("bytes"), c.get("popcount"))
    print(units)


if __name__ == "__main__":
    main()
