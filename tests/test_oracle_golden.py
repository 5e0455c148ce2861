"""The CPU oracle against every pin the reference tree offers.

These tests are what makes oracle/ trustworthy as the checker for the HIP path."""
import collections
import json
import os

import numpy as np
import pytest

from oracle import nts_oracle as O
from oracle import synteny_oracle as SO

MX_FILES = {
    ("ref", 24): "mx_celegans-chrII-III.fa.k24.w1000.npz",
    ("A", 24): "mx_celegans-chrII-III.A.fa.k24.w1000.npz",
    ("ref", 20): "mx_celegans-chrII-III.fa.k20.w1000.npz",
    ("A", 20): "mx_celegans-chrII-III.A.fa.k20.w1000.npz",
    ("B", 20): "mx_celegans-chrII-III.B.fa.k20.w1000.npz",
}


def load_mx(golden_dir, key):
    z = np.load(os.path.join(golden_dir, MX_FILES[key]))
    contigs = [str(c) for c in z["contigs"]]
    recs = [(c, []) for c in contigs]
    for ci, h, p in zip(z["contig_idx"].tolist(), z["h1"].tolist(), z["pos"].tolist()):
        recs[ci][1].append((str(h), p))
    return recs


def test_nthash_known_answers(golden_dir):
    "P1: hash:pos:kmer tokens written by the reference's indexlr (every 30th of 295,028)"
    n = 0
    with open(os.path.join(golden_dir, "kat_nthash.tsv")) as fh:
        for line in fh:
            k, h1, _pos, kmer, _src = line.rstrip("\n").split("\t")
            h0, got = O.hash_kmer(kmer, int(k))
            assert got == int(h1)
            assert O.h1_from_h0(h0, int(k)) == int(h1)
            n += 1
    assert n > 9000


def test_nthash_canonical_and_case():
    rng = np.random.default_rng(1)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for k in (20, 24, 31, 64, 100):
        s = bytes(rng.choice(list(b"ACGT"), size=k).astype(np.uint8))
        rc = s.translate(comp)[::-1]
        assert O.hash_kmer(s, k) == O.hash_kmer(rc, k)
        assert O.hash_kmer(s.lower(), k) == O.hash_kmer(s, k)
        assert O.hash_kmer(s[:-1] + b"N", k) is None


def test_rolling_equals_direct():
    rng = np.random.default_rng(2)
    seq = bytearray(rng.choice(list(b"ACGT"), size=5000).astype(np.uint8))
    for p in (0, 17, 18, 900, 901, 902, 903, 2500, 4999):
        seq[p] = ord("N")
    seq[3000:3100] = b"n" * 100
    seq = bytes(seq)
    for k in (20, 24, 33):
        pos, h0 = O.hash_all(seq, k)
        exp = [(i, O.hash_kmer(seq[i:i + k], k)[0]) for i in range(len(seq) - k + 1)
               if O.hash_kmer(seq[i:i + k], k) is not None]
        assert pos.tolist() == [i for i, _ in exp]
        assert h0.tolist() == [h for _, h in exp]


def test_minimizer_tsv_counts(golden_dir):
    "P2: token counts / de-duplication of the k=24 reference genome fixture"
    recs = load_mx(golden_dir, ("ref", 24))
    assert sum(len(t) for _, t in recs) == 59243
    info, lists = SO.mx_tables_from_tokens(recs)
    assert len(info) == 54066
    assert sum(len(x) for x in lists) == 54066
    # structural facts of indexlr output: positions strictly increase, gap <= w (window = w k-mers)
    for _, toks in recs:
        pos = np.array([p for _, p in toks])
        assert (np.diff(pos) > 0).all() and np.diff(pos).max() <= 1000


def _h0_from_h1(h1, k):
    "inverse of ntHash's extension step (t = h0 * (1 ^ k * MULTISEED); t ^= t >> 27): SURVEY.md P1"
    seed, mask = 0x90b45d39fb6da1fa, (1 << 64) - 1
    x = h1.copy()
    for _ in range(3):                                   # undoes t ^= t >> 27 (27 * 3 >= 64)
        x = h1 ^ (x >> np.uint64(27))
    inv = pow((1 ^ (k * seed)) & mask, -1, 1 << 64)      # the multiplier is odd
    with np.errstate(over="ignore"):
        h0 = x * np.uint64(inv)
        back = h0 * np.uint64((1 ^ (k * seed)) & mask)
    assert np.array_equal(back ^ (back >> np.uint64(27)), h1)
    return h0


@pytest.mark.parametrize("key", sorted(MX_FILES))
def test_window_rule_pinned_to_reference_minimizers(golden_dir, key):
    """u5 against the reference's own indexlr output (295,028 minimizers over the five fixture TSVs): the window is w k-mers,
    the comparison key is hashes()[0] (not the printed hash), and of equal keys the rightmost wins.

    h0 of every listed minimizer is recovered from the printed h1.  A k-mer that was never a window's minimum can be
    deleted without changing any window's minimum, so the oracle's decision logic run over ONLY the listed minimizers
    (every other position empty) must emit exactly the list again.  Negative controls show the data discriminates:
    the same with the printed hash as key, with `<` as tie rule, or with a window of w + 1 does not reproduce it.  A window
    shorter than w cannot be told apart this way; that bound comes from the gaps: none exceeds w, and dozens per file equal w."""
    z = np.load(os.path.join(golden_dir, MX_FILES[key]))
    k, w = key[1], 1000
    h1, pos, ci = z["h1"], z["pos"].astype(np.int64), z["contig_idx"]
    h0 = _h0_from_h1(h1, k)
    n_w = n_ties = 0
    for c in np.unique(ci):
        m = ci == c
        p, a, b = pos[m], h0[m], h1[m]
        assert (np.diff(p) > 0).all()
        gaps = np.diff(p)
        assert gaps.max() <= w
        n_w += int((gaps == w).sum())
        n_ties += int(((a[1:] == a[:-1]) & (gaps < w)).sum())     # equal keys, the later one listed while the earlier is in its window
        n = int(p[-1]) + w
        dense = np.full(n, np.iinfo(np.uint64).max, dtype=np.uint64)
        dense[p] = a
        assert np.array_equal(O.minimize_keys(dense, w).astype(np.int64), p), "oracle window rule != reference list"
        assert not np.array_equal(O.minimize_keys(dense, w + 1).astype(np.int64), p), "a window of w + 1 k-mers fits as well"
        assert not np.array_equal(O.minimize_keys(dense, w, strict=True).astype(np.int64), p), "`<` fits as well"
        dense[p] = b
        assert not np.array_equal(O.minimize_keys(dense, w).astype(np.int64), p), "the printed hash as key fits as well"
        # the condition the verdict names, spelled out: for neighbours a < b with key[b] > key[a] (b took over when a left the
        # window), every later minimizer c within a's reach has key[c] > key[b]
        up = np.nonzero(a[1:] > a[:-1])[0]
        for i in up[:4000]:
            j = i + 2
            while j < len(p) and p[j] <= p[i] + w:
                assert a[j] > a[i + 1]
                j += 1
    assert n_w >= 20 and n_ties >= 500


def test_bf_size_arithmetic():
    "row A1 numbers (src/ntsynt_make_common_bf.cpp:38-39)"
    assert O.bf_approx_bytes(29058289, 0.025) == 143467638
    assert O.bf_ctor_bytes(143467638) == 143467640
    assert O.bf_approx_bytes(100000000, 0.025) == 493723627
    assert O.bf_approx_bytes(3000000000, 0.025) == 14811708827


def _blocks_from_json(case):
    blocks = []
    for b in case["blocks"]:
        blk = SO.SynBlock(case["k"], 90, list(b["asm"]))
        blk.broken_reason = b["reason"]
        for a, d in b["asm"].items():
            ab = blk.asm[a]
            ab.contig_id, ab.ori = d["contig"], d["ori"]
            ab.minimizers = [(h, p) for h, p in d["mx"]]
        blocks.append(blk)
    return blocks


def _merge_text(eng, blocks, z):
    ordered = sorted(blocks, key=SO.SynBlock.sort_key)
    pre = "".join(b.text(i) for i, b in enumerate(ordered))
    merged = eng.merge_collinear(ordered)
    merged = [b for b in merged if b.long_enough(z)]
    merged = eng.merge_collinear(merged)
    out, num = "", 0
    for b in merged:
        if b.long_enough(z):
            out += b.text(num, verbose=True)
            num += 1
    return pre, out


def _engine(k, bp, cm, z, files=("a.fa.k1.w1.tsv", "b.fa.k1.w1.tsv")):
    return SO.SyntenyOracle(list(files), {}, k, 1000, [100, 10], bp, cm, z, "x")


def test_merge_random_goldens(golden_dir):
    "P3: sort + collinear merge + reasons + formatting vs vectors made by the reference's own code"
    cases = json.load(open(os.path.join(golden_dir, "merge_cases.json")))
    assert len(cases) >= 40
    n_merged = 0
    for c in cases:
        eng = _engine(c["k"], c["bp"], c["collinear_merge"], c["z"])
        pre, out = _merge_text(eng, _blocks_from_json(c), c["z"])
        assert pre == c["pre_text"], c["name"]
        assert out == c["final_text"], c["name"]
        n_merged += len(c["pre_text"].splitlines()) != len(c["final_text"].splitlines())
    assert n_merged > 10     # the vectors do exercise merging


@pytest.mark.parametrize("stem,k", [("celegans-A-ntSynt", 24), ("celegans-A-B-ntSynt", 20)])
def test_merge_demo_goldens(golden_dir, stem, k):
    "P3: pre-collinear-merge TSV -> final TSV of the reference demo, byte for byte"
    rows = collections.OrderedDict()
    for line in open(os.path.join(golden_dir, stem + ".pre-collinear-merge.synteny_blocks.tsv")):
        num, asm, ctg, start, end, ori, n = line.rstrip("\n").split("\t")
        rows.setdefault(int(num), []).append((asm, ctg, int(start), int(end), ori, int(n)))
    blocks = []
    for num, lst in rows.items():
        names = sorted((a + ".k1.w1.tsv" for a, *_ in lst), reverse=True)
        blk = SO.SynBlock(k, 90, names)
        for asm, ctg, start, end, ori, n in lst:
            ab = blk.asm[asm + ".k1.w1.tsv"]
            ab.contig_id, ab.ori = ctg, ori
            first, last = (start, end - k) if ori == "+" else (end - k, start)
            ab.minimizers = [(f"{num}_0", first)] + [(f"{num}_{i}", first) for i in range(1, n - 1)] \
                + [(f"{num}_{n}", last)]
        blocks.append(blk)
    eng = _engine(k, 500, 3000, 500)
    pre, out = _merge_text(eng, blocks, 500)
    assert pre == open(os.path.join(golden_dir, stem + ".pre-collinear-merge.synteny_blocks.tsv")).read()
    assert out == open(os.path.join(golden_dir, stem + ".synteny_blocks.tsv")).read()


def test_orientation_and_spread_goldens(golden_dir):
    d = json.load(open(os.path.join(golden_dir, "block_cases.json")))
    for c in d["orientation"]:
        blk = SO.SynBlock(24, c["m"], ["x.fa.k1.w1.tsv"])
        blk.asm["x.fa.k1.w1.tsv"].minimizers = [(str(i), p) for i, p in enumerate(c["pos"])]
        blk.orient()
        assert blk.asm["x.fa.k1.w1.tsv"].ori == c["ori"]
    for c in d["max_difference"]:
        gaps = [abs(x - y) for x, y in zip(c["p1"], c["p2"])]
        assert max(gaps) - min(gaps) == c["spread"]


@pytest.mark.parametrize("keys,k,stem,simplify,n_paths,n_initial", [
    ([("ref", 24), ("A", 24)], 24, "celegans-A-ntSynt", False, 28, 29),
    ([("ref", 20), ("A", 20), ("B", 20)], 20, "celegans-A-B-ntSynt", False, 50, 54),
    ([("ref", 24), ("A", 24)], 24, "celegans-A-ntSynt", True, 11, 15),
    ([("ref", 20), ("A", 20), ("B", 20)], 20, "celegans-A-B-ntSynt", True, 12, 16),
])
def test_initial_round_graph_stage(golden_dir, keys, k, stem, simplify, n_paths, n_initial):
    """P4 (containment): rows C1-C9 on the reference's own minimizer TSVs.  Every initial-round
    block must lie inside an expected (post-refinement, pre-merge) block with equal contigs and
    orientation.  Counts without bubble removal (28/29, 50/54) are the survey's independent probe;
    with bubble removal (the reference default) 11 paths at k=24 equal the 11 expected blocks."""
    names = {"ref": "celegans-chrII-III.fa", "A": "celegans-chrII-III.A.fa", "B": "celegans-chrII-III.B.fa"}
    tables = {}
    for key in keys:
        tsv = f"{names[key[0]]}.k{k}.w1000.tsv"
        tables[tsv] = SO.mx_tables_from_tokens(load_mx(golden_dir, key))
    eng = SO.SyntenyOracle(list(tables), {}, k, 1000, [], 500, 3000, 500, "x")
    eng.load(tables)
    eng.list_mxs = SO.filter_minimizers(eng.list_mxs)
    eng.graph = SO.build_graph(eng.list_mxs, eng.weights)
    n_v, n_e = len(eng.graph.adj), len(eng.graph.edges)
    n_full = sum(e[2] == len(keys) for e in eng.graph.edges)
    if k == 24:
        assert (n_v, n_e, n_full) == (53491, 53523, 53455)
    else:
        assert (n_v, n_e, n_full) == (51307, 51372, 51238)
    if simplify:
        eng.graph = eng.simplify_graph(eng.graph)
    eng.graph = SO.filter_graph_global(eng.graph, eng.n, eng.weights)
    paths = SO.find_paths(eng.graph, eng.list_mx_info[eng.files[-1]])
    assert len(paths) == n_paths
    blocks = eng.drop_small(eng.split_indels(eng.blocks_of_paths(paths)), 4)
    assert len(blocks) == n_initial
    exp = collections.defaultdict(dict)
    for line in open(os.path.join(golden_dir, stem + ".pre-collinear-merge.synteny_blocks.tsv")):
        num, asm, ctg, start, end, ori, _ = line.rstrip("\n").split("\t")
        exp[int(num)][asm] = (ctg, int(start), int(end), ori)
    for blk in blocks:
        mine = {SO.MX_SUFFIX.search(a).group(1): (ab.contig_id, ab.start(), ab.end(), ab.ori)
                for a, ab in blk.asm.items()}
        hits = [num for num, e in exp.items()
                if all(e[a][0] == mine[a][0] and e[a][1] <= mine[a][1] and mine[a][2] <= e[a][2]
                       and e[a][3] == mine[a][3] for a in mine)]
        assert len(hits) == 1, mine
