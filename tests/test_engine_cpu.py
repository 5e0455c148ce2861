"""Host logic of the HIP path (ntsynt_amd/synteny.py: rows C3-C12) against the oracle, without a GPU.

The engine gets its two device-side inputs from test doubles here -- the graph build from
tests/graph_ref.py (numpy) and the masked re-sketch from the oracle's indexlr restatement -- so that
what is compared is exactly the host-side rule set: simplification order, weight filter, path
orientation, block splitting, refinement bookkeeping, erosion, collinear merge, TSV bytes."""
import os

import numpy as np
import pytest

from ntsynt_amd import synth
from ntsynt_amd.graph import edge_degrees, walk_chains, walk_paths
from ntsynt_amd.synteny import SyntenyEngine
from oracle import nts_oracle as O
from oracle import synteny_oracle as SO
from tests.graph_ref import build_graph_numpy
from tests.helpers import oracle_flat


def run_both(tmp_path, paths, k, w, w_rounds, indel, merge, block_size, simplify=True, common=True, native=True):
    os.makedirs(tmp_path / "ora", exist_ok=True)
    os.makedirs(tmp_path / "eng", exist_ok=True)
    os.chdir(tmp_path / "ora")
    ora = SO.run_pipeline(paths, k=k, w=w, w_rounds=w_rounds, indel=indel, merge=merge, block_size=block_size,
                          prefix="p", simplify=simplify, common=common)
    os.chdir(tmp_path / "eng")
    genomes = [O.read_fasta(p) for p in paths]
    bf = ora.bf
    tsvs = [f"{os.path.basename(p)}.k{k}.w{w}.tsv" for p in paths]
    initial = [oracle_flat(O.minimize(g, k, w, bf)) for g in genomes]

    def sketch_fn(i, masks, new_w):
        g = genomes[i]
        seqs = []
        for r in range(len(g.names)):
            buf = bytearray(g.record(r))
            for mr, s, e in masks:
                if mr == r:
                    s, e = max(0, s), min(len(buf), e)
                    if e > s:
                        buf[s:e] = b"N" * (e - s)
            seqs.append(bytes(buf))
        return oracle_flat(O.minimize(O.Genome(g.names, seqs), k, new_w, bf))

    eng = SyntenyEngine(tsvs, [g.names for g in genomes], k, w, w_rounds, indel, merge, block_size, "p",
                        build_graph_numpy, sketch_fn, walk_paths if native else walk_chains, simplify=simplify,
                        degree_fn=edge_degrees if native else None)
    out = eng.run(initial)
    return ora.outputs, out


CASES = [
    # (n_genomes, total_bp, contigs, divergence, seed, k, w, w_rounds, indel, merge, block)
    (2, 600_000, 2, 0.01, 1, 24, 1000, [100, 10], 500, 3000, 500),
    (3, 900_000, 3, 0.01, 2, 24, 1000, [100, 10], 500, 3000, 500),
    (3, 900_000, 2, 0.02, 3, 20, 500, [100], 10000, "3w", 500),
    (4, 700_000, 2, 0.005, 4, 24, 400, [100, 10], 500, 1000, 300),
    (2, 1_200_000, 1, 0.03, 5, 24, 1000, [250, 100], 50000, 100000, 1000),
    (3, 600_000, 2, 0.01, 6, 32, 250, [50, 5], 200, "2w", 200),
    (2, 1_500_000, 2, 0.005, 9, 24, 200, [50, 10], 5000, 20000, 300),
    (3, 1_200_000, 2, 0.005, 12, 24, 150, [40, 10], 3000, "40w", 200),
    (2, 1_000_000, 1, 0.002, 15, 20, 100, [20, 5], 2000, 10000, 100),
]


@pytest.mark.parametrize("case", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_engine_matches_oracle(tmp_path, case):
    n, bp, ctg, div, seed, k, w, rounds, indel, merge, block = case
    cwd = os.getcwd()
    try:
        paths = synth.make_family(str(tmp_path), n, bp, ctg, div, seed=seed, n_runs=(seed % 2 == 0),
                                  micro=(12 if seed % 3 == 0 else 0))
        exp, got = run_both(tmp_path, paths, k, w, rounds, indel, merge, block)
    finally:
        os.chdir(cwd)
    assert set(got) == {"p.synteny_blocks.tsv", "p.pre-collinear-merge.synteny_blocks.tsv"}
    for name in got:
        assert got[name] == exp[name], name
    assert len(got["p.synteny_blocks.tsv"].splitlines()) >= 2 * n


@pytest.mark.parametrize("case", [CASES[0], CASES[2], CASES[5]], ids=["case0", "case2", "case5"])
def test_engine_with_plain_walk_and_numpy_degrees(tmp_path, case):
    """The engine's numpy orientation / degree code (used when walk_fn is nts_walk_chains and no degree_fn is given)
    against the oracle; the product configuration (nts_walk_paths + nts_edge_degrees) is the test above."""
    n, bp, ctg, div, seed, k, w, rounds, indel, merge, block = case
    cwd = os.getcwd()
    try:
        paths = synth.make_family(str(tmp_path), n, bp, ctg, div, seed=seed, n_runs=(seed % 2 == 0),
                                  micro=(12 if seed % 3 == 0 else 0))
        exp, got = run_both(tmp_path, paths, k, w, rounds, indel, merge, block, native=False)
    finally:
        os.chdir(cwd)
    for name in got:
        assert got[name] == exp[name], name


def test_walk_paths_mask_and_orientation():
    """nts_walk_paths (int64 edges, liveness mask, orientation key) == nts_walk_chains on the live edges followed by
    the flip rule; nts_edge_degrees == bincount."""
    rng = np.random.default_rng(5)
    nv = 150_000
    perm = rng.permutation(nv)
    brk = np.zeros(nv, bool)
    brk[rng.choice(np.arange(1, nv), size=nv // 9, replace=False)] = True
    eu, ev = perm[:-1][~brk[1:]], perm[1:][~brk[1:]]
    xu, xv = rng.integers(0, nv, eu.size // 4), rng.integers(0, nv, eu.size // 4)
    EU, EV = np.concatenate((eu, xu)).astype(np.int64), np.concatenate((ev, xv)).astype(np.int64)
    alive = np.concatenate((np.ones(eu.size, bool), np.zeros(xu.size, bool)))
    sh = rng.permutation(EU.size)
    EU, EV, alive = EU[sh], EV[sh], alive[sh]
    key = rng.permutation(nv).astype(np.int64)
    o1, v1 = walk_chains(nv, EU[alive], EV[alive])
    flip = key[v1[o1[1:] - 1]] < key[v1[o1[:-1]]]
    assert 0.3 < flip.mean() < 0.7
    seg = np.repeat(np.arange(o1.size - 1), np.diff(o1))
    j = np.arange(v1.size)
    want = v1[np.where(flip[seg], o1[seg] + o1[seg + 1] - 1 - j, j)]
    o2, v2 = walk_paths(nv, EU, EV, alive, key)
    assert np.array_equal(o1, o2) and np.array_equal(want, v2) and v2.dtype == np.int64
    o3, v3 = walk_paths(nv, EU[alive], EV[alive])
    assert np.array_equal(o3, o1) and np.array_equal(v3, v1)
    # dead edges that would branch a chain must not matter; live ones must
    o4, v4 = walk_paths(nv, EU, EV)
    assert o4.size < o1.size
    deg = edge_degrees(nv, EU, EV, alive)
    ref = np.bincount(EU[alive], minlength=nv) + np.bincount(EV[alive], minlength=nv)
    assert deg.dtype == np.uint8 and np.array_equal(deg, np.minimum(ref, 255))
    hub_u = np.concatenate((EU, np.zeros(400, np.int64)))
    hub_v = np.concatenate((EV, np.arange(1, 401, dtype=np.int64)))
    assert edge_degrees(nv, hub_u, hub_v)[0] == 255
    with pytest.raises(RuntimeError):
        edge_degrees(10, np.array([11], np.int64), np.array([0], np.int64))
    with pytest.raises(RuntimeError):
        walk_paths(10, np.array([11], np.int64), np.array([0], np.int64))


def test_engine_without_simplification(tmp_path):
    cwd = os.getcwd()
    try:
        paths = synth.make_family(str(tmp_path), 3, 600_000, 2, 0.02, seed=11)
        exp, got = run_both(tmp_path, paths, 24, 500, [100, 10], 500, 3000, 500, simplify=False)
    finally:
        os.chdir(cwd)
    for name in got:
        assert got[name] == exp[name], name


def test_walk_chains_shapes():
    # two paths, one cycle, one branching component, one isolated vertex
    eu = np.array([0, 1, 3, 5, 6, 7, 8, 9, 10], dtype=np.int64)
    ev = np.array([1, 2, 4, 6, 7, 5, 9, 10, 11], dtype=np.int64)
    eu = np.concatenate((eu, [9]))
    ev = np.concatenate((ev, [12]))       # vertex 9 gets degree 3
    off, verts = walk_chains(14, eu, ev)
    paths = [verts[off[i]:off[i + 1]].tolist() for i in range(off.size - 1)]
    assert paths == [[0, 1, 2], [3, 4]]


def test_walk_chains_threaded_matches_sequential_sweep():
    """Large enough for nts_walk_chains to spread the walks over host threads: random chains (shuffled vertex
    ids), cycles and branching components; expected = the sequential sweep over ascending vertex ids."""
    rng = np.random.default_rng(17)
    nv = 60000
    perm = rng.permutation(nv)
    eu, ev, expect = [], [], []
    at = 0
    while at < nv - 40:
        ln = int(rng.integers(1, 30))
        vs = perm[at:at + ln]
        at += ln
        kind = rng.random()
        for a, b in zip(vs[:-1], vs[1:]):
            eu.append(a if rng.random() < 0.5 else b)
            ev.append(b if eu[-1] == a else a)
        if ln >= 3 and kind < 0.1:                      # close a cycle
            eu.append(vs[-1])
            ev.append(vs[0])
        elif ln >= 3 and kind < 0.2:                    # branch: a third neighbour for a middle vertex
            eu.append(vs[ln // 2])
            ev.append(perm[at])
            at += 1
        elif ln >= 2:
            p = vs.tolist()
            expect.append(p if p[0] < p[-1] else p[::-1])
    order = rng.permutation(len(eu))
    eu, ev = np.array(eu, np.int64)[order], np.array(ev, np.int64)[order]
    off, verts = walk_chains(nv, eu, ev)
    paths = [verts[off[i]:off[i + 1]].tolist() for i in range(off.size - 1)]
    assert paths == sorted(expect, key=lambda p: p[0]) and len(paths) > 2000


def test_orientation_rule_matches_oracle_blocks():
    """synteny_block.py:48-65 on mixed position sequences (the synthetic families never produce them): segment-wise
    _orient_codes against the oracle's SynBlock.orient, with lengths and mixes chosen around the 90 % threshold."""
    rng = np.random.default_rng(23)
    eng = SyntenyEngine(["a.k24.w100.tsv", "b.k24.w100.tsv"], [["c"], ["c"]], 24, 100, [10], 500, 1000, 100, "p",
                        None, None, None)
    seqs = []
    for n in list(range(1, 25)) + [40, 41, 50, 100, 101]:
        for frac_down in (0.0, 0.05, 0.09, 0.1, 0.11, 0.5, 0.89, 0.9, 0.91, 1.0):
            steps = np.where(rng.random(max(n - 1, 0)) < frac_down, -1, 1) * rng.integers(1, 50, max(n - 1, 0))
            seqs.append(np.concatenate(([10_000], 10_000 + np.cumsum(steps))).astype(np.int64))
        if n >= 11:                                        # exactly 10 % and just over it
            for n_down in (max(1, (n - 1) // 10), (n - 1) // 10 + 1, n - 1 - (n - 1) // 10, n - 2 - (n - 1) // 10):
                sign = np.ones(n - 1, np.int64)
                sign[rng.permutation(n - 1)[:max(0, n_down)]] = -1
                seqs.append(np.concatenate(([10_000], 10_000 + np.cumsum(sign * 7))).astype(np.int64))
    pos = np.concatenate(seqs)
    end = np.cumsum([s.size for s in seqs])
    start = end - np.array([s.size for s in seqs])
    # rising steps per sequence, then the rule
    rise = np.concatenate(([0], np.cumsum(pos[1:] > pos[:-1])))
    n_up = rise[end - 1] - rise[start]
    got = ["+-?"[c] for c in eng._orient_codes(n_up, end - start - 1)]
    want = []
    for s in seqs:
        blk = SO.SynBlock(24, 90, ["x"])
        blk.asm["x"].minimizers = [("h", int(p)) for p in s]
        blk.orient()
        want.append(blk.asm["x"].ori)
    assert got == want and {"+", "-", "?"} <= set(want)


def test_path_scan_matches_numpy_statement():
    "nts_path_scan (threaded) against the same three rules written with numpy, on random paths over random tables"
    from ntsynt_amd.graph import scan_paths
    rng = np.random.default_rng(31)
    G, nv, n_paths = 3, 50000, 900
    v_rec = rng.integers(0, 3, (G, nv)).astype(np.int64)
    v_rec[:, : nv // 2] = 1                                    # long stretches without contig changes
    v_pos = rng.integers(0, 10**9, (G, nv)).astype(np.int64)
    base = np.sort(rng.integers(0, 10**9, nv))
    for a in range(G):                                         # mostly collinear positions with a few large jumps
        v_pos[a] = base + rng.integers(0, 50, nv) + (rng.random(nv) < 0.01) * 10**6
    lens = rng.integers(1, 120, n_paths)
    off = np.concatenate(([0], np.cumsum(lens)))
    verts = np.concatenate([np.sort(rng.choice(nv // 2 + (nv // 2) * (i % 7 == 0), ln, replace=False)) for i, ln in enumerate(lens)])
    for bp in (100, 10**5):
        start, n_up, over = scan_paths(v_rec, v_pos, off, verts, bp)
        for i in range(n_paths):
            p = verts[off[i]:off[i + 1]]
            change = np.zeros(max(p.size - 1, 0), bool)
            for a in range(G):
                change |= v_rec[a][p[1:]] != v_rec[a][p[:-1]]
            nz = np.flatnonzero(change)
            st = int(nz[-1]) + 1 if nz.size else 0
            assert start[i] == off[i] + st
            q = p[st:]
            gaps = np.stack([np.abs(np.diff(v_pos[a][q])) for a in range(G)])
            want_over = np.zeros(p.size, bool)
            want_over[st:p.size - 1] = (gaps.max(axis=0) - gaps.min(axis=0)) > bp
            assert np.array_equal(over[off[i]:off[i + 1]], want_over)
            for a in range(G):
                assert n_up[a, i] == int((np.diff(v_pos[a][q]) > 0).sum())
    assert over.any() and (start > off[:-1]).any()


def test_unoriented_block_is_deleted_in_situ(tmp_path):
    """The `all_oriented` == False branch (bin/ntsynt_synteny.py:84-86,96-104; synteny_block.py:48-70) inside a whole
    run, engine against oracle.  List-adjacent minimizers are monotone in every assembly, so no sequence-level family
    reaches it; it takes a refinement round that re-sights a block's terminal minimizer elsewhere in one assembly (a
    second copy of the k-mer that only becomes a minimizer at the smaller w) and overwrites its position
    (S:282-290).  Both sides are driven at list level: scripted minimizer lists instead of a re-sketch."""
    k, w = 24, 1000
    names = ["b.fa.k24.w1000.tsv", "a.fa.k24.w1000.tsv"]
    rng = np.random.default_rng(3)
    H = rng.integers(1 << 40, 1 << 62, size=40, dtype=np.uint64)
    X = rng.integers(1 << 40, 1 << 62, size=8, dtype=np.uint64)
    contigs = ["c1", "c2", "c3"]
    # initial round: c1 carries a 4-minimizer block, c2 and c3 longer ones; same order and positions in both assemblies
    layout = [(0, H[0:4]), (1, H[4:20]), (2, H[20:40])]

    def initial():
        h1 = np.concatenate([hs for _, hs in layout])
        rec = np.concatenate([np.full(hs.size, r, np.uint32) for r, hs in layout])
        pos = np.concatenate([(np.arange(hs.size, dtype=np.uint64) + 1) * 10000 for _, hs in layout])
        return h1, rec, pos
    # the scripted refinement list: new minimizers X0, X1 left of the c1 block, X2 right of it; in assembly b the
    # block's last minimizer H3 (position 40000) turns up at 3000
    script = {
        "a": ([X[0], X[1], H[0], H[3], X[2]], [2000, 5000, 10000, 40000, 45000]),
        "b": ([X[0], H[3], X[1], H[0], X[2]], [2000, 3000, 5000, 10000, 45000]),
    }

    def scripted(asm):
        hs, ps = script["a" if asm.startswith("a.") else "b"]
        return (np.array(hs, np.uint64), np.zeros(len(hs), np.uint32), np.array(ps, np.uint64))

    cwd = os.getcwd()
    try:
        os.makedirs(tmp_path / "ora")
        os.makedirs(tmp_path / "eng")
        os.chdir(tmp_path / "ora")

        class Scripted(SO.SyntenyOracle):
            def sketch_masked(self, asm, ctg_masks, new_w):
                h1, rec, pos = scripted(asm)
                return SO.mx_tables_from_tokens([("c1", [(str(h), int(p)) for h, p in zip(h1.tolist(), pos.tolist())])])
        ora = Scripted(names, {}, k, w, [100], 500, 3000, 500, "p")
        h1, rec, pos = initial()
        recs = [(contigs[r], [(str(h), int(p)) for h, p in zip(h1[rec == r].tolist(), pos[rec == r].tolist())]) for r in range(3)]
        ora.load({n: SO.mx_tables_from_tokens(recs) for n in names})
        exp = ora.main()
        os.chdir(tmp_path / "eng")
        eng = SyntenyEngine(names, [contigs, contigs], k, w, [100], 500, 3000, 500, "p", build_graph_numpy,
                            lambda i, masks, new_w: scripted(names[i]), walk_paths, degree_fn=edge_degrees)
        got = eng.run([initial(), initial()])
    finally:
        os.chdir(cwd)
    assert eng.stats["unoriented"] == 1
    for name in got:
        assert got[name] == exp[name], name
    final = got["p.synteny_blocks.tsv"]
    assert "\tc1\t" not in final and "\tc2\t" in final and "\tc3\t" in final     # the c1 block is gone, the others stay


def test_native_block_rules_match_the_python_rules():
    """nts_blocks_merge / nts_blocks_text / nts_bubble_rule (array passes the HBM-resident engine uses) against the
    Block-object rules of ntsynt_amd/synteny.py on random tables -- which the oracle tests pin in turn."""
    import ctypes

    from ntsynt_amd import _lib
    from ntsynt_amd.synteny import Block
    lib = _lib.load()
    rng = np.random.default_rng(12)
    G, k = 3, 24
    files = ["c.fa.k24.w100.tsv", "a.fa.k24.w100.tsv", "b.fa.k24.w100.tsv"]
    contigs = [[f"ctg{j}_{a}" for j in range(5)] for a in range(G)]
    total_merged = 0
    for trial in range(30):
        n = int(rng.integers(1, 60))
        eng = SyntenyEngine(files, contigs, k, 100, [], int(rng.choice([50, 500, 5000])), [300, 3000, "30w"][trial % 3],
                            int(rng.choice([100, 400])), "x", None, None, None)
        # sorted blocks on one or two contigs, mostly collinear with the odd break
        blocks, pos = [], np.zeros(G, np.int64)
        for i in range(n):
            rec = [int(rng.integers(0, 2))] * G if rng.random() < 0.9 else [int(rng.integers(0, 5)) for _ in range(G)]
            ori = ["+"] * G if rng.random() < 0.8 else [str(rng.choice(["+", "-"])) for _ in range(G)]
            length = rng.integers(50, 3000, size=G)
            gap = rng.integers(-30, 2500, size=G) if rng.random() < 0.3 else np.full(G, int(rng.integers(1, 400)))
            first = pos + gap
            last = first + length
            pos = last + k
            fp = [int(last[a]) if ori[a] == "-" else int(first[a]) for a in range(G)]
            lp = [int(first[a]) if ori[a] == "-" else int(last[a]) for a in range(G)]
            blocks.append(Block(None, rec, ori, None, fp, lp, int(rng.integers(4, 50))))
        ordered = eng._sorted(list(blocks))

        def table(bl):
            return (np.array([[b.rec[a] for b in bl] for a in range(G)], np.uint32), np.array([[b.first_pos[a] for b in bl] for a in range(G)], np.int64),
                    np.array([[b.last_pos[a] for b in bl] for a in range(G)], np.int64),
                    np.array([["+-".index(b.ori[a]) for b in bl] for a in range(G)], np.uint8), np.array([b.n_mx for b in bl], np.int64),
                    np.zeros(len(bl), np.uint8))
        # engine index order of the table rows = eng.files order (descending names)
        eo = [files.index(f) for f in eng.files]
        for b in ordered:
            b.rec, b.ori = [b.rec[i] for i in eo], [b.ori[i] for i in eo]
            b.first_pos, b.last_pos = [b.first_pos[i] for i in eo], [b.last_pos[i] for i in eo]
        rec, first, last, ori, n_mx, reason = [np.ascontiguousarray(x) for x in table(ordered)]
        names = [SO.MX_SUFFIX.search(f).group(1) for f in eng.files] + [nm for a in range(G) for nm in eng.contigs[a]]
        blob = ("\0".join(names) + "\0").encode()
        base = np.arange(G, dtype=np.uint64) * 5
        order = np.array(eng.out_order, np.uint32)

        def text(nn, verbose):
            buf, nb = ctypes.c_void_p(), ctypes.c_uint64()
            assert lib.nts_blocks_text(G, nn, k, eng.z, order.ctypes.data, blob, len(blob), base.ctypes.data, rec.ctypes.data, first.ctypes.data,
                                       last.ctypes.data, ori.ctypes.data, n_mx.ctypes.data, reason.ctypes.data if verbose else None,
                                       ctypes.byref(buf), ctypes.byref(nb)) == 0
            out = ctypes.string_at(buf, nb.value).decode()
            lib.nts_free(buf)
            return out
        cwd = os.getcwd()
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            os.chdir(td)
            try:
                eng._emit("pre.tsv", ordered)
                assert text(n, False) == eng.outputs["pre.tsv"]
                merged = eng._merge(ordered)
                n_out, n_merged = ctypes.c_uint64(), ctypes.c_uint64()
                assert lib.nts_blocks_merge(G, n, k, eng.bp, eng.collinear_merge, rec.ctypes.data, first.ctypes.data, last.ctypes.data,
                                            ori.ctypes.data, n_mx.ctypes.data, reason.ctypes.data, ctypes.byref(n_out), ctypes.byref(n_merged)) == 0
                assert n_out.value == len(merged) and n_merged.value == eng.stats["merged"]
                total_merged += n_merged.value
                # the in-place tables keep the stride n: compare through the verbose text of the first n_out blocks
                rec, first, last, ori = [np.ascontiguousarray(x[:, :n_out.value]) for x in (rec, first, last, ori)]
                n_mx, reason = np.ascontiguousarray(n_mx[:n_out.value]), np.ascontiguousarray(reason[:n_out.value])
                eng._emit("post.tsv", merged, verbose=True)
                assert text(n_out.value, True) == eng.outputs["post.tsv"]
            finally:
                os.chdir(cwd)
    assert total_merged > 50
    # bubble rule: random small graphs, against the dict-based rule of SyntenyEngine._simplify
    total_bubbles = 0
    for trial in range(400):
        nv = int(rng.integers(6, 40))
        weights = {}
        if trial % 2:
            # a backbone of full-weight edges with bubbles (s - t light, s - m - t) every few steps, adjacent ones included,
            # plus a little noise: what the minimizer graph looks like around small rearrangements
            spine, nxt = list(range(nv)), nv
            for i in range(nv - 1):
                if rng.random() < 0.35:
                    weights[(spine[i], spine[i + 1])] = int(rng.integers(1, 3))
                    weights[(spine[i], nxt)] = int(rng.integers(1, 4))
                    weights[(spine[i + 1], nxt)] = int(rng.integers(1, 4))
                    nxt += 1
                else:
                    weights[(spine[i], spine[i + 1])] = 3
            for _ in range(int(rng.integers(0, 4))):
                u, v = (int(x) for x in rng.integers(0, nxt, 2))
                if u != v:
                    weights.setdefault((min(u, v), max(u, v)), int(rng.integers(1, 4)))
            nv = nxt
        else:
            while len(weights) < nv * 2:
                u, v = (int(x) for x in rng.integers(0, nv, 2))
                if u != v:
                    weights[(min(u, v), max(u, v))] = int(rng.integers(1, 4))
        pairs = sorted(weights)
        rng.shuffle(pairs)
        eu = np.array([p[0] for p in pairs], np.int64)
        ev = np.array([p[1] for p in pairs], np.int64)
        ew = np.array([weights[tuple(p)] for p in pairs], np.int64)
        eng = SyntenyEngine(files, contigs, k, 100, [], 500, 3000, 100, "x", None, None, None)
        eng.v_hash = np.arange(nv, dtype=np.uint64)
        eng.v_alive = np.ones(nv, bool)
        eng.e_u, eng.e_v, eng.e_w, eng.e_alive = eu.copy(), ev.copy(), ew.copy(), np.ones(eu.size, bool)
        deg = np.bincount(eu, minlength=nv) + np.bincount(ev, minlength=nv)
        cand = np.flatnonzero((deg[eu] == 3) & (deg[ev] == 3)).astype(np.uint32)
        is_cv = np.zeros(nv, bool)
        is_cv[eu[cand]] = is_cv[ev[cand]] = True
        inc = np.flatnonzero(is_cv[eu] | is_cv[ev]).astype(np.uint32)
        eng._simplify(apply_deletions=True)
        doomed, promoted, n_out = np.zeros(max(cand.size, 1), np.uint32), np.zeros(max(cand.size, 1), np.uint32), ctypes.c_uint64()
        iu, iv, iw = eu[inc].astype(np.uint32), ev[inc].astype(np.uint32), ew[inc].astype(np.uint32)
        assert lib.nts_bubble_rule(cand.size, cand.ctypes.data, inc.size, inc.ctypes.data, iu.ctypes.data, iv.ctypes.data, iw.ctypes.data, G,
                                   doomed.ctypes.data, promoted.ctypes.data, ctypes.byref(n_out)) == 0
        assert n_out.value == eng.stats["bubbles"]
        total_bubbles += n_out.value
        want_w = ew.copy()
        want_w[promoted[:n_out.value]] = G
        assert np.array_equal(want_w, eng.e_w)
        dead = np.zeros(nv, bool)
        dead[doomed[:n_out.value]] = True
        assert np.array_equal(~dead, eng.v_alive)
    assert total_bubbles > 200


def test_dev_overlap_check_matches_the_oracle(capsys):
    "--dev: check_non_overlapping (bin/ntsynt_synteny.py:234-253) -- array sweep of the engines against the oracle's restatement"
    rng = np.random.default_rng(4)
    files = ["b.fa.k24.w100.tsv", "a.fa.k24.w100.tsv"]
    contigs = [["x1", "x2"], ["y1", "y2"]]
    total = 0
    for trial in range(25):
        n = int(rng.integers(2, 40))
        eng = SyntenyEngine(files, contigs, 24, 100, [], 500, 3000, 300, "x", None, None, None, dev=True)
        ora = SO.SyntenyOracle(files, {}, 24, 100, [], 500, 3000, 300, "x")
        rec = rng.integers(0, 2, size=(2, n))
        start = rng.integers(0, 20000, size=(2, n))
        end = start + rng.integers(300, 4000, size=(2, n))
        blocks = []
        for b in range(n):
            blk = SO.SynBlock(24, 90, ora.files)
            for a, f in enumerate(ora.files):
                ab = blk.asm[f]
                ab.contig_id, ab.ori = contigs[a][rec[a, b]], "+"
                ab.minimizers = [("h", int(start[a, b])), ("t", int(end[a, b]) - 24)]
            blocks.append(blk)
        ora.check_non_overlapping(blocks)
        want = capsys.readouterr().err
        eng._warn_overlaps(rec, start, end)
        got = capsys.readouterr().err
        assert got == want
        total += want.count("WARNING")
    assert total > 50


class _ShiftedOracle(SO.SyntenyOracle):
    """The oracle on genomes whose every record carries `shift` leading N: minimizer positions, block coordinates and
    hard-mask intervals all move by `shift`, nothing else changes (no k-mer spans an N)."""
    shift = 0

    def sketch_masked(self, asm, ctg_masks, new_w):
        back = {c: [(s - self.shift, e - self.shift) for s, e in ivs] for c, ivs in ctg_masks.items()}
        info, lists = super().sketch_masked(asm, back, new_w)
        return {h: (c, p + self.shift) for h, (c, p) in info.items()}, lists


def shifted_oracle_run(paths, shift, k, w, w_rounds, indel, merge, block_size):
    "oracle outputs for the family with `shift` N prepended to every record + what an engine under test needs"
    genomes = {p: O.read_fasta(p) for p in paths}
    bf = O.common_bf(genomes, k, 0.025, 1)
    tables, by_tsv, initial = {}, {}, {}
    for p in paths:
        tsv = f"{os.path.basename(p)}.k{k}.w{w}.tsv"
        mins = O.minimize(genomes[p], k, w, bf)
        info, lists = SO.mx_tables_from_tokens(SO.mx_records_from_arrays(genomes[p].names, mins))
        tables[tsv] = ({h: (c, pos + shift) for h, (c, pos) in info.items()}, lists)
        by_tsv[tsv] = genomes[p]
        h1, rec, pos = oracle_flat(mins)
        initial[tsv] = (h1, rec, pos.astype(np.uint64) + np.uint64(shift))
    ora = _ShiftedOracle(list(tables), by_tsv, k, w, w_rounds, indel, merge, block_size, "p", bf=bf)
    ora.shift = shift
    ora.load(tables)
    ora.main()
    return ora, genomes, bf, initial


@pytest.mark.parametrize("shift", [5_000_000_000, (1 << 40) - 1_500_000])
def test_positions_beyond_32_bits(tmp_path, shift):
    """Records longer than 2^32 bp (and up to the engine's 2^40 limit): every position-carrying step -- composite sort
    keys, gap arithmetic of the indel rule, mask intervals, span lookups of the refinement filter, erosion distances,
    the merge rule, the TSV text -- on 64-bit coordinates, against the oracle."""
    k, w, rounds, indel, merge, block = 24, 400, [100, 10], 500, 3000, 300
    paths = synth.make_family(str(tmp_path), 3, 900_000, 3, 0.01, seed=17, micro=6)
    cwd = os.getcwd()
    try:
        os.makedirs(tmp_path / "ora")
        os.chdir(tmp_path / "ora")
        ora, genomes, bf, initial = shifted_oracle_run(paths, shift, k, w, rounds, indel, merge, block)
        os.makedirs(tmp_path / "eng")
        os.chdir(tmp_path / "eng")
        tsvs = [f"{os.path.basename(p)}.k{k}.w{w}.tsv" for p in paths]
        glist = [genomes[p] for p in paths]

        def sketch_fn(i, masks, new_w):
            g = glist[i]
            seqs = []
            for r in range(len(g.names)):
                buf = bytearray(g.record(r))
                for mr, s, e in masks:
                    if mr == r:
                        s, e = max(0, int(s) - shift), min(len(buf), int(e) - shift)
                        if e > s:
                            buf[s:e] = b"N" * (e - s)
                seqs.append(bytes(buf))
            h1, rec, pos = oracle_flat(O.minimize(O.Genome(g.names, seqs), k, new_w, bf))
            return h1, rec, pos.astype(np.uint64) + np.uint64(shift)

        eng = SyntenyEngine(tsvs, [g.names for g in glist], k, w, rounds, indel, merge, block, "p", build_graph_numpy, sketch_fn,
                            walk_paths, degree_fn=edge_degrees)
        out = eng.run([initial[t] for t in tsvs])
    finally:
        os.chdir(cwd)
    for name, text in ora.outputs.items():
        assert out[name] == text, name
    rows = ora.outputs["p.synteny_blocks.tsv"].splitlines()
    assert len(rows) > 20 and all(int(r.split("\t")[3]) >= shift for r in rows)


def test_interarrivals_file_matches_the_oracle(tmp_path):
    """--interarrivals (ntsynt_run.py, S:557-564, S:626-627): the distances between neighbouring minimizers of the initial
    blocks.  Same lines as the oracle's; the order of the blocks is the engine's path order (the reference's is ntJoin's
    component order, unpinned), so the comparison is on the sorted lines, and on per-block runs being intact."""
    k, w, rounds, indel, merge, block = 24, 500, [100, 10], 500, 3000, 300
    paths = synth.make_family(str(tmp_path), 3, 900_000, 3, 0.01, seed=23, micro=6)
    cwd = os.getcwd()
    try:
        os.makedirs(tmp_path / "ora")
        os.chdir(tmp_path / "ora")
        ora = SO.run_pipeline(paths, k=k, w=w, w_rounds=rounds, indel=indel, merge=merge, block_size=block, prefix="p",
                              interarrivals=True)
        os.makedirs(tmp_path / "eng")
        os.chdir(tmp_path / "eng")
        genomes = [O.read_fasta(p) for p in paths]
        tsvs = [f"{os.path.basename(p)}.k{k}.w{w}.tsv" for p in paths]
        initial = [oracle_flat(O.minimize(g, k, w, ora.bf)) for g in genomes]

        def sketch_fn(i, masks, new_w):
            g = genomes[i]
            seqs = []
            for r in range(len(g.names)):
                buf = bytearray(g.record(r))
                for mr, s, e in masks:
                    if mr == r:
                        s, e = max(0, s), min(len(buf), e)
                        if e > s:
                            buf[s:e] = b"N" * (e - s)
                seqs.append(bytes(buf))
            return oracle_flat(O.minimize(O.Genome(g.names, seqs), k, new_w, ora.bf))

        eng = SyntenyEngine(tsvs, [g.names for g in genomes], k, w, rounds, indel, merge, block, "p", build_graph_numpy, sketch_fn,
                            walk_paths, degree_fn=edge_degrees, interarrivals=True)
        out = eng.run(initial)
        on_disk = open("p.interarrivals.tsv").read()
    finally:
        os.chdir(cwd)
    want = ora.outputs["p.interarrivals.tsv"]
    assert out["p.interarrivals.tsv"] == on_disk
    assert len(want.splitlines()) > 1000
    assert sorted(out["p.interarrivals.tsv"].splitlines()) == sorted(want.splitlines())
    assert out["p.synteny_blocks.tsv"] == ora.outputs["p.synteny_blocks.tsv"]


def test_a_last_round_without_blocks_ends_the_run_like_the_reference(tmp_path):
    """bin/ntsynt_synteny.py:505-510 calls merge_collinear_blocks twice without looking: when the last refinement round leaves no block of
    at least z bases the reference ends with `IndexError: list index out of range` at S:437 (found by tests/golden/refrun_stress.py
    running the reference's own code).  Oracle and engine stop the same way -- neither writes a final table of its own making."""
    cwd = os.getcwd()
    try:
        paths = synth.make_family(str(tmp_path), 2, 300_000, 2, 0.01, seed=3)
        with pytest.raises(IndexError):
            os.makedirs(tmp_path / "o")
            os.chdir(tmp_path / "o")
            SO.run_pipeline(paths, k=24, w=200, w_rounds=[50, 10], indel=500, merge=1000, block_size=10 ** 8, prefix="p")
        genomes = [O.read_fasta(p) for p in paths]
        bf = O.common_bf(dict(zip(paths, genomes)), 24, 0.025)
        tsvs = [f"{os.path.basename(p)}.k24.w200.tsv" for p in paths]

        def sketch_fn(i, masks, new_w):
            g = genomes[i]
            seqs = []
            for r in range(len(g.names)):
                buf = bytearray(g.record(r))
                for mr, s, e in masks:
                    if mr == r:
                        buf[max(0, s):min(len(buf), e)] = b"N" * (min(len(buf), e) - max(0, s))
                seqs.append(bytes(buf))
            return oracle_flat(O.minimize(O.Genome(g.names, seqs), 24, new_w, bf))
        os.makedirs(tmp_path / "e")
        os.chdir(tmp_path / "e")
        eng = SyntenyEngine(tsvs, [g.names for g in genomes], 24, 200, [50, 10], 500, 1000, 10 ** 8, "p", build_graph_numpy, sketch_fn, walk_paths,
                            degree_fn=edge_degrees)
        with pytest.raises(IndexError, match=r"ntsynt_synteny\.py:437"):
            eng.run([oracle_flat(O.minimize(g, 24, 200, bf)) for g in genomes])
    finally:
        os.chdir(cwd)


def test_bubble_rule_numbers_its_vertices_either_way():
    """nts_bubble_rule numbers the table's vertices through a direct id -> number table when the ids are about as dense as the table is long, through
    a sorted list otherwise (a handful of candidates in a graph of millions): the same rule either way -- here the same table with its vertex ids
    moved far up, which flips the choice."""
    import ctypes
    from ntsynt_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    n_inc, n_cand, nvg = 60000, 9000, 150000
    inc_edge = np.sort(rng.choice(np.arange(400000, dtype=np.uint32), size=n_inc, replace=False)).astype(np.uint32)
    u = rng.integers(0, nvg, size=n_inc, dtype=np.uint32)
    v = (u + rng.integers(1, 3, size=n_inc, dtype=np.uint32)).astype(np.uint32)
    w = rng.integers(1, 4, size=n_inc, dtype=np.uint32)
    cand = np.sort(rng.choice(inc_edge, size=n_cand, replace=False)).astype(np.uint32)
    outs = []
    for shift in (0, 3_000_000_000):
        uu, vv = (u + np.uint32(shift)).astype(np.uint32), (v + np.uint32(shift)).astype(np.uint32)
        doomed, promoted, n = np.empty(n_cand, np.uint32), np.empty(n_cand, np.uint32), ctypes.c_uint64()
        assert lib.nts_bubble_rule(n_cand, cand.ctypes.data, n_inc, inc_edge.ctypes.data, uu.ctypes.data, vv.ctypes.data, w.ctypes.data, 3,
                                   doomed.ctypes.data, promoted.ctypes.data, ctypes.byref(n)) == 0
        outs.append(((doomed[:n.value].astype(np.int64) - shift).tolist(), promoted[:n.value].tolist()))
    assert outs[0] == outs[1] and len(outs[0][0]) > 20
