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
