#!/usr/bin/env python3
"""Generates the fixtures under tests/golden/ -- run ONLY in the development container, where
/root/reference exists.  The fixtures are data (inputs + expected outputs); no reference source
text is stored.

  1. fixtures the reference's own tests hold (tests/expected_result/): the four synteny-block TSVs
     and three .fai files verbatim; the five indexlr minimizer TSVs as compact arrays
     (mx_<name>.npz: contig ids, h1, pos) plus a deterministic sample of hash:pos:kmer known
     answers (kat_nthash.tsv) -- SURVEY.md 8(c) pins P1, P2, P4.
  2. golden in/out vectors produced by importing the reference's Python
     (bin/ntsynt_synteny.py, bin/synteny_block.py, bin/assembly_block.py) with empty stand-in
     modules for its absent third-party imports, calling only functions that touch neither igraph
     nor subprocesses: sorted() -> merge_collinear_blocks -> z filter -> merge_collinear_blocks ->
     get_block_string(verbose=True) (pin P3), plus determine_orientations / max_difference /
     get_block_string on random blocks.
"""
import collections
import json
import os
import random
import shutil
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
EXP = os.path.join(REF, "tests", "expected_result")


def copy_reference_fixtures():
    for name in os.listdir(EXP):
        if name.endswith(".synteny_blocks.tsv") or name.endswith(".fai"):
            shutil.copy(os.path.join(EXP, name), os.path.join(OUT, name))
            os.chmod(os.path.join(OUT, name), 0o644)
    kat = []
    for name in sorted(os.listdir(EXP)):
        if not name.endswith(".w1000.tsv"):
            continue
        k = 24 if ".k24." in name else 20
        contigs, cidx, hs, ps = [], [], [], []
        n_tok = 0
        with open(os.path.join(EXP, name)) as fh:
            for line in fh:
                cid, rest = line.rstrip("\n").split("\t")
                contigs.append(cid)
                for tok in rest.split(" "):
                    h, p, s = tok.split(":")
                    cidx.append(len(contigs) - 1)
                    hs.append(int(h))
                    ps.append(int(p))
                    if n_tok % 30 == 0:
                        kat.append(f"{k}\t{h}\t{p}\t{s}\t{name}")
                    n_tok += 1
        np.savez_compressed(os.path.join(OUT, "mx_" + name[:-4] + ".npz"),
                            contigs=np.array(contigs), contig_idx=np.array(cidx, dtype=np.uint16),
                            h1=np.array(hs, dtype=np.uint64), pos=np.array(ps, dtype=np.uint32))
    with open(os.path.join(OUT, "kat_nthash.tsv"), "w") as fh:
        fh.write("\n".join(kat) + "\n")


def import_reference():
    Minimizer = collections.namedtuple("Minimizer", ["mx", "position"])
    Bed = collections.namedtuple("Bed", ["contig", "start", "end"])
    for name in ("intervaltree", "pybedtools", "btllib", "ncls", "ntjoin_utils", "ntjoin"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["intervaltree"].Interval = collections.namedtuple("Interval", ["begin", "end"])
    sys.modules["ntjoin_utils"].Minimizer = Minimizer
    sys.modules["ntjoin_utils"].Bed = Bed

    class _Base:
        def __init__(self, args):
            self.args = args
    sys.modules["ntjoin"].Ntjoin = _Base
    sys.path.insert(0, os.path.join(REF, "bin"))
    import ntsynt_synteny
    import synteny_block
    return ntsynt_synteny, synteny_block, Minimizer


def make_engine(ns, k, bp, cm, z, w=1000):
    args = types.SimpleNamespace(FILES=["a.fa.k1.w1.tsv", "b.fa.k1.w1.tsv"], n=2, p="x", k=k, w=w,
                                 btllib_t=1, w_rounds=[100, 10], m=90, z=z, collinear_merge=str(cm),
                                 common=None, repeat=None, fastas=[], bp=bp, dev=False)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return ns.NtSyntSynteny(args)


def blocks_to_json(blocks):
    return [{"reason": b.broken_reason,
             "asm": {a: {"contig": ab.contig_id, "ori": ab.ori,
                         "mx": [[m.mx, m.position] for m in ab.minimizers]}
                     for a, ab in b.assembly_blocks.items()}} for b in blocks]


def run_merge(eng, blocks, z):
    ordered = sorted(blocks)
    pre = "".join(b.get_block_string(i) for i, b in enumerate(ordered))
    merged = eng.merge_collinear_blocks(ordered)
    merged = [b for b in merged if all(ab.get_block_length() >= z for ab in b.assembly_blocks.values())]
    merged = eng.merge_collinear_blocks(merged)
    out, num = "", 0
    for b in merged:
        if not all(ab.get_block_length() >= z for ab in b.assembly_blocks.values()):
            continue
        out += b.get_block_string(num, verbose=True)
        num += 1
    return pre, out


def pseudo_blocks_from_tsv(sb, Minimizer, path, k, m=90):
    "rows of a pre-collinear-merge TSV -> SyntenyBlock objects with first/last positions + dummy interior"
    rows = collections.OrderedDict()
    with open(path) as fh:
        for line in fh:
            num, asm, ctg, start, end, ori, n = line.rstrip("\n").split("\t")
            rows.setdefault(int(num), []).append((asm, ctg, int(start), int(end), ori, int(n)))
    blocks = []
    for num, lst in rows.items():
        names = [a + ".k1.w1.tsv" for a, *_ in lst]
        blk = sb.SyntenyBlock(k, m, *sorted(names, reverse=True))
        for asm, ctg, start, end, ori, n in lst:
            ab = blk.assembly_blocks[asm + ".k1.w1.tsv"]
            ab.contig_id, ab.ori = ctg, ori
            first, last = (start, end - k) if ori == "+" else (end - k, start)
            ab.minimizers = [Minimizer(f"{num}_0", first)] + \
                [Minimizer(f"{num}_{i}", first) for i in range(1, n - 1)] + [Minimizer(f"{num}_{n}", last)]
        blocks.append(blk)
    return blocks


def random_blocks(sb, Minimizer, rng, k, n_asm, n_blocks):
    "a chain of roughly collinear blocks with occasional breaks of every kind"
    names = [f"g{i}.fa.k1.w1.tsv" for i in range(n_asm)]
    cursor = [rng.randint(0, 5000) for _ in names]
    ctg = ["c1"] * n_asm
    blocks = []
    for b in range(n_blocks):
        blk = sb.SyntenyBlock(k, 90, *sorted(names, reverse=True))
        n_mx = rng.randint(4, 12)
        ori = ["+"] * n_asm
        event = rng.random()
        if event < 0.15:
            ori[rng.randrange(1, n_asm)] = "-"
        elif event < 0.25:
            j = rng.randrange(1, n_asm)
            ctg[j] = "c2" if ctg[j] == "c1" else "c1"
        gaps = [rng.choice([5, 40, 200, 900, 2500, 6000])] * n_asm
        if event > 0.8:
            gaps[rng.randrange(n_asm)] += rng.choice([100, 480, 520, 3000])
        if 0.25 <= event < 0.3:
            gaps[rng.randrange(n_asm)] = -rng.randint(1, 300)
        steps = [rng.randint(30, 400) for _ in range(n_mx - 1)]
        for i, a in enumerate(names):
            start = cursor[i] + gaps[i]
            pos = [start]
            for s in steps:
                pos.append(pos[-1] + s)
            cursor[i] = pos[-1] + k
            if ori[i] == "-":
                pos = pos[::-1]
            ab = blk.assembly_blocks[a]
            ab.contig_id, ab.ori = ctg[i], ori[i]
            ab.minimizers = [Minimizer(f"{b}_{t}", p) for t, p in enumerate(pos)]
        blocks.append(blk)
    rng.shuffle(blocks)
    return blocks


def main():
    copy_reference_fixtures()
    ns, sb, Minimizer = import_reference()
    cases = []
    # pin P3: the demo data through the reference's own merge + formatting
    for stem, k in (("celegans-A-ntSynt", 24), ("celegans-A-B-ntSynt", 20)):
        blocks = pseudo_blocks_from_tsv(sb, Minimizer, os.path.join(EXP, stem + ".pre-collinear-merge.synteny_blocks.tsv"), k)
        eng = make_engine(ns, k, 500, 3000, 500)
        inp = blocks_to_json(blocks)
        pre, out = run_merge(eng, blocks, 500)
        with open(os.path.join(EXP, stem + ".synteny_blocks.tsv")) as fh:
            assert out == fh.read(), stem
        # (not stored: tests rebuild the same pseudo-blocks from the two TSVs, which are fixtures)
    rng = random.Random(7)
    for c in range(40):
        k = rng.choice([20, 24, 32])
        bp = rng.choice([500, 10000])
        cm = rng.choice([1000, 3000, "3w"])
        z = rng.choice([100, 500, 1000])
        blocks = random_blocks(sb, Minimizer, rng, k, rng.choice([2, 3, 4]), rng.randint(2, 14))
        eng = make_engine(ns, k, bp, cm, z)
        inp = blocks_to_json(blocks)
        pre, out = run_merge(eng, blocks, z)
        cases.append({"name": f"rand{c}", "k": k, "bp": bp, "collinear_merge": cm, "z": z,
                      "blocks": inp, "pre_text": pre, "final_text": out})
    with open(os.path.join(OUT, "merge_cases.json"), "w") as fh:
        json.dump(cases, fh)
    # orientation vote + interarrival spread on random position lists
    ocases = []
    for c in range(200):
        n = rng.randint(2, 40)
        mode = rng.random()
        pos = sorted(rng.sample(range(100000), n))
        if mode < 0.3:
            pos = pos[::-1]
        for _ in range(rng.choice([0, 0, 1, 2, 5])):
            i, j = rng.randrange(n), rng.randrange(n)
            pos[i], pos[j] = pos[j], pos[i]
        m = rng.choice([90, 90, 50, 75])
        blk = sb.SyntenyBlock(24, m, "x.fa.k1.w1.tsv")
        blk.assembly_blocks["x.fa.k1.w1.tsv"].minimizers = [Minimizer(str(i), p) for i, p in enumerate(pos)]
        blk.determine_orientations()
        ocases.append({"pos": pos, "m": m, "ori": blk.assembly_blocks["x.fa.k1.w1.tsv"].ori})
    dcases = []
    Node = collections.namedtuple("Node", ["mx", "positions"])
    for c in range(100):
        g = rng.randint(2, 5)
        p1 = [rng.randint(0, 10**6) for _ in range(g)]
        p2 = [p + rng.randint(-5000, 5000) for p in p1]
        dcases.append({"p1": p1, "p2": p2,
                       "spread": ns.NtSyntSynteny.max_difference(Node("a", p1), Node("b", p2))})
    with open(os.path.join(OUT, "block_cases.json"), "w") as fh:
        json.dump({"orientation": ocases, "max_difference": dcases}, fh)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
