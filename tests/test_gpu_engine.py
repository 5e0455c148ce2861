"""The graph stage resident in HBM (ntsynt_amd/synteny_device.py over nts_engine_*, csrc/nts_dgraph.inc) against its
host-array twin (ntsynt_amd/synteny.py), state by state, on the same minimizer lists -- vertex tables, edge arrays in
the reference's order, liveness after every rule, the path order of the list ranking, block tables, the marks the next
refinement round filters by -- and both against the oracle pipeline's bytes."""
import os

import numpy as np
import pytest

from ntsynt_amd import synth
from oracle import nts_oracle as O
from oracle import synteny_oracle as SO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ntsynt_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _state_equal(host, dev, what):
    g = dev.graph
    nv, ne = g.size()
    assert nv == host.v_hash.size and ne == host.e_u.size, (what, nv, host.v_hash.size, ne, host.e_u.size)
    assert np.array_equal(g.read("v_hash"), host.v_hash), what
    assert np.array_equal(g.read("v_alive").astype(bool), host.v_alive), what
    # tables of dead vertices are never read again; compare the live ones
    live = host.v_alive
    assert np.array_equal(g.read("v_rec").astype(np.int64)[:, live], host.v_rec[:, live]), what
    assert np.array_equal(g.read("v_pos").astype(np.int64)[:, live], host.v_pos[:, live]), what
    assert np.array_equal(g.read("e_u").astype(np.int64), host.e_u), what
    assert np.array_equal(g.read("e_v").astype(np.int64), host.e_v), what
    assert np.array_equal(g.read("e_alive").astype(bool), host.e_alive), what
    alive = host.e_alive
    assert np.array_equal(g.read("e_w").astype(np.int64)[alive], host.e_w[alive]), what


def _blocks_equal(host, hb, dev, db, what):
    host._finish_all(hb)
    a = sorted((tuple(b.rec), tuple(b.ori), tuple(b.first_pos), tuple(b.last_pos), b.n_mx) for b in hb)
    b = type(dev).rows(db)                                        # the device engine keeps a round's blocks as a table
    assert a == b, what
    # the marks the next refinement round filters by
    term = np.zeros(host.v_hash.size, bool)
    inner = np.zeros(host.v_hash.size, bool)
    for blk in hb:
        term[blk.vids[0]] = term[blk.vids[-1]] = True
        inner[blk.vids[1:-1]] = True
    assert np.array_equal(dev.graph.read("terminal").astype(bool), term), what
    assert np.array_equal(dev.graph.read("internal").astype(bool), inner), what


def _round_blocks_both(host, dev, what):
    """One round's paths and blocks on both engines: the host walk and the device's list ranking see the same graph; the
    path order is compared as a set of oriented paths, then the block rules' outcome."""
    verts, off = host._paths()
    want = sorted(tuple(verts[off[i]:off[i + 1]].tolist()) for i in range(off.size - 1))
    db = dev._blocks()
    pv, po = dev.graph.read("path_verts"), dev.graph.read("path_off")
    got = sorted(tuple(pv[int(po[i]):int(po[i + 1])].tolist()) for i in range(po.size - 1))
    assert got == want, what + ": paths"
    hb = host._drop_small(host._blocks_of_paths((verts, off)), 4)
    _state_equal(host, dev, what)
    _blocks_equal(host, hb, dev, db, what)
    return hb, db


CASES = [
    # (n_genomes, total_bp, contigs, divergence, seed, k, w, w_rounds, indel, merge, block, micro, n_runs)
    (3, 3_000_000, 3, 0.01, 21, 24, 1000, [100, 10], 500, 3000, 500, 0, True),
    (2, 2_000_000, 2, 0.005, 9, 24, 200, [50, 10], 5000, 20000, 300, 12, False),
    (4, 1_600_000, 2, 0.02, 33, 20, 500, [250, 100], 50000, "100w", 1000, 6, True),
    (3, 2_400_000, 40, 0.01, 5, 24, 150, [40, 10], 300, "40w", 200, 20, False),
]


@pytest.mark.parametrize("case", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_device_engine_in_lockstep_with_host_engine(ctx, tmp_path, case):
    from ntsynt_amd import fasta as fa
    from ntsynt_amd.device import BloomFilter, Genome, Minimizers, bf_size_bytes, sketch
    from ntsynt_amd.graph import build_graph_device, edge_degrees, walk_paths
    from ntsynt_amd.synteny import SyntenyEngine
    from ntsynt_amd.synteny_device import DeviceSyntenyEngine
    n, bp_total, ctg, div, seed, k, w, rounds, indel, merge, block, micro, n_runs = case
    paths = synth.make_family(str(tmp_path), n, bp_total, ctg, div, seed=seed, micro=micro, n_runs=n_runs)
    recs = [fa.read_fasta(p) for p in paths]
    genomes = [Genome(ctx, r.names, r.seq, r.rec_off, r.rec_len) for r in recs]
    _, nbytes = bf_size_bytes(genomes[sorted(range(n), key=lambda i: paths[i])[0]].total_bp, 0.025)
    bf = BloomFilter(ctx, nbytes, k)
    tmp = BloomFilter(ctx, nbytes, k)
    for j, g in enumerate(genomes):
        if j == 0:
            bf.insert(g)
        else:
            tmp.clear()
            tmp.insert(g)
            bf.and_(tmp)
    tmp.free()
    tsvs = [f"{os.path.basename(p)}.k{k}.w{w}.tsv" for p in paths]
    names = [r.names for r in recs]

    def sketch_np(i, masks, new_w):
        mx = sketch(ctx, genomes[i], k, new_w, bf, masks)
        out = mx.to_numpy()
        mx.free()
        return out

    def sketch_dev(masks_by_asm, new_w):
        return {i: sketch(ctx, genomes[i], k, new_w, bf, m) for i, m in masks_by_asm.items()}

    cwd = os.getcwd()
    os.makedirs(tmp_path / "h")
    os.makedirs(tmp_path / "d")
    try:
        os.chdir(tmp_path / "h")
        host = SyntenyEngine(tsvs, names, k, w, rounds, indel, merge, block, "p", lambda ls, kp, li: build_graph_device(ctx, ls, kp, li),
                             sketch_np, walk_paths, degree_fn=edge_degrees)
        os.chdir(tmp_path / "d")
        dev = DeviceSyntenyEngine(ctx, tsvs, names, k, w, rounds, indel, merge, block, "p", sketch_dev)
        assert host.input_order == dev.input_order
        initial = [sketch_np(i, None, w) for i in range(n)]
        # ---- initial round
        host._add_graph(host.graph_fn([initial[i] for i in host.input_order], None, None))
        handles = [Minimizers.from_numpy(ctx, *initial[i]) for i in dev.input_order]
        dev._add(handles, None)
        for h in handles:
            h.free()
        _state_equal(host, dev, "initial add")
        host._simplify(apply_deletions=True)
        dev._simplify_dev(apply_deletions=True)
        assert host.stats["bubbles"] == dev.stats["bubbles"]
        _state_equal(host, dev, "initial simplify")
        host.e_alive &= host.e_w >= host.n
        dev._filter(flag=False)
        _state_equal(host, dev, "initial filter")
        hb, db = _round_blocks_both(host, dev, "initial blocks")
        prev_w = w
        for new_w in rounds:
            assert [sorted(m) for m in host._mask_intervals(hb, prev_w)] == \
                [sorted(tuple(int(x) for x in row) for row in m.tolist()) for m in dev._mask_intervals(db, prev_w)]
            host._new_round_graph(hb, new_w, prev_w)
            masks = dev._mask_intervals(db, prev_w)
            lists = dev._sketch_round(masks, new_w)
            dev._add(lists, dev._spans(db))
            for mx in lists:
                mx.free()
            _state_equal(host, dev, f"add w={new_w}")
            host._simplify(apply_deletions=False)
            dev._simplify_dev(apply_deletions=False)
            _state_equal(host, dev, f"simplify w={new_w}")
            last = new_w == rounds[-1]
            light = host.e_alive & (host.e_w < host.n)
            flagged = (host.e_u[light], host.e_v[light])
            host.e_alive &= ~light
            dev._filter(flag=last)
            _state_equal(host, dev, f"filter w={new_w}")
            if last:
                host._refine_graph(flagged)
                dev._erode()
                assert host.stats["eroded_edges"] == dev.stats["eroded_edges"]
                _state_equal(host, dev, "erosion")
            hb, db = _round_blocks_both(host, dev, f"blocks w={new_w}")
            prev_w = new_w
        for key in ("bubbles", "unoriented", "indel_cuts", "small_blocks", "eroded_edges"):
            assert host.stats[key] == dev.stats[key], key
    finally:
        os.chdir(cwd)
        for g in genomes:
            g.free()
        bf.free()


@pytest.mark.parametrize("engine", ["device", "host"])
def test_both_engines_through_the_pipeline_match_the_oracle(tmp_path, monkeypatch, engine):
    from ntsynt_amd import pipeline
    paths = synth.make_family(str(tmp_path), 3, 2_500_000, 30, 0.01, seed=12, micro=15, n_runs=True, soft_mask=True)
    kw = dict(k=24, w=300, w_rounds=[100, 20], indel=400, merge="10w", block_size=300)
    cwd = os.getcwd()
    try:
        os.makedirs(tmp_path / "hip")
        os.makedirs(tmp_path / "ora")
        os.chdir(tmp_path / "hip")
        eng = pipeline.run(paths, prefix="p", log=lambda *a: None, engine=engine, **kw)
        os.chdir(tmp_path / "ora")
        ora = SO.run_pipeline(paths, prefix="p", **kw)
    finally:
        os.chdir(cwd)
    assert type(eng).__name__ == ("DeviceSyntenyEngine" if engine == "device" else "SyntenyEngine")
    for name in ("p.synteny_blocks.tsv", "p.pre-collinear-merge.synteny_blocks.tsv"):
        assert eng.outputs[name] == ora.outputs[name], name
    assert len(eng.outputs["p.synteny_blocks.tsv"].splitlines()) > 30
    for p in paths:
        tsv = f"{os.path.basename(p)}.k24.w300.tsv"
        assert open(tmp_path / "hip" / tsv).read() == open(tmp_path / "ora" / tsv).read()


def test_mx_split_matches_host_split(ctx):
    from ntsynt_amd.device import Genome, sketch
    parts = [Genome.synth(ctx, 3_000_000 + 1000 * j, 3 + j, 5, 60 + j, 0.01) for j in range(3)]
    batch = Genome.concat(ctx, parts)
    mx = sketch(ctx, batch, 24, 200)
    want = batch.split_minimizers(*mx.to_numpy())
    got = mx.split(batch.rec_base)
    assert len(got) == 3
    for a, b in zip(got, want):
        for x, y in zip(a.to_numpy(), b):
            assert np.array_equal(x, y)
        a.free()
    mx.free()
    batch.free()
    for g in parts:
        g.free()


@pytest.mark.parametrize("shift", [5_000_000_000, (1 << 40) - 1_500_000])
def test_device_engine_on_positions_beyond_32_bits(ctx, tmp_path, shift):
    """Every record as if it began with `shift` N (records beyond 2^32 bp, up to the 2^40 limit): the device engine's 64-bit
    positions through graph build, list ranking, indel cuts, refinement filter, erosion, merge and text vs the oracle."""
    from ntsynt_amd.device import Genome, Minimizers, sketch, wrap_bloom  # noqa: F401
    from ntsynt_amd.device import BloomFilter
    from ntsynt_amd.synteny_device import DeviceSyntenyEngine
    from tests.test_engine_cpu import shifted_oracle_run
    k, w, rounds, indel, merge, block = 24, 400, [100, 10], 500, 3000, 300
    paths = synth.make_family(str(tmp_path), 3, 900_000, 3, 0.01, seed=17, micro=6)
    cwd = os.getcwd()
    genomes = []
    try:
        os.makedirs(tmp_path / "ora")
        os.chdir(tmp_path / "ora")
        ora, og, bits, initial = shifted_oracle_run(paths, shift, k, w, rounds, indel, merge, block)
        os.makedirs(tmp_path / "dev")
        os.chdir(tmp_path / "dev")
        tsvs = [f"{os.path.basename(p)}.k{k}.w{w}.tsv" for p in paths]
        for p in paths:
            g = og[p]
            seq = b"".join(g.record(r) for r in range(len(g.names)))
            lens = np.array([len(g.record(r)) for r in range(len(g.names))], np.uint64)
            off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
            genomes.append(Genome(ctx, g.names, np.frombuffer(seq, np.uint8), off, lens))
        bf = BloomFilter(ctx, bits.size, k)
        bf.from_numpy(bits)

        def moved(mx):
            h1, rec, pos = mx.to_numpy()
            mx.free()
            return Minimizers.from_numpy(ctx, h1, rec, pos + np.uint64(shift))

        def sketch_dev(masks_by_asm, new_w):
            out = {}
            for i, m in masks_by_asm.items():
                m = m.copy()
                m["start"] -= np.uint64(shift)
                m["end"] -= np.uint64(shift)
                out[i] = moved(sketch(ctx, genomes[i], k, new_w, bf, m))
            return out

        # the device's own initial sketch, moved, must be the oracle's moved list
        handles = []
        for i, t in enumerate(tsvs):
            mx = moved(sketch(ctx, genomes[i], k, w, bf))
            for a, b in zip(mx.to_numpy(), initial[t]):
                assert np.array_equal(a, b)
            handles.append(mx)
        dev = DeviceSyntenyEngine(ctx, tsvs, [og[p].names for p in paths], k, w, rounds, indel, merge, block, "p", sketch_dev)
        out = dev.run(handles)
    finally:
        os.chdir(cwd)
        for g in genomes:
            g.free()
    for name, text in ora.outputs.items():
        assert out[name] == text, name
    assert all(int(r.split("\t")[3]) >= shift for r in ora.outputs["p.synteny_blocks.tsv"].splitlines())


def test_interarrivals_from_the_device_engine(tmp_path):
    "--interarrivals through the product pipeline (device engine): the oracle's lines (block order aside)"
    from ntsynt_amd import pipeline
    paths = synth.make_family(str(tmp_path), 3, 1_500_000, 5, 0.01, seed=31, micro=8)
    kw = dict(k=24, w=400, w_rounds=[100, 20], indel=500, merge="10w", block_size=300)
    cwd = os.getcwd()
    try:
        os.makedirs(tmp_path / "hip")
        os.makedirs(tmp_path / "ora")
        os.chdir(tmp_path / "hip")
        eng = pipeline.run(paths, prefix="p", log=lambda *a: None, interarrivals=True, **kw)
        got = open("p.interarrivals.tsv").read()
        os.chdir(tmp_path / "ora")
        ora = SO.run_pipeline(paths, prefix="p", interarrivals=True, **kw)
    finally:
        os.chdir(cwd)
    assert type(eng).__name__ == "DeviceSyntenyEngine"
    want = ora.outputs["p.interarrivals.tsv"]
    assert len(want.splitlines()) > 1000
    assert sorted(got.splitlines()) == sorted(want.splitlines())
    assert eng.outputs["p.synteny_blocks.tsv"] == ora.outputs["p.synteny_blocks.tsv"]


def test_an_edge_that_exists_keeps_its_slot_and_takes_the_new_weight(ctx, tmp_path):
    """nts_engine_add on a graph that holds some of the new build's edges already (ntJoin's build_graph with graph=<the running graph>, as
    bin/ntsynt_synteny.py:476-485 calls it every refinement round): the existing edge keeps its slot and takes the new weight, the others
    are appended.  The join of new edges against live ones runs on the device (sorted pair keys + look-up; it was a host-side dictionary
    until dense sketches made it tens of millions of entries): here against the host-array twin, with thousands of such edges."""
    from ntsynt_amd.device import Minimizers
    from ntsynt_amd.graph import build_graph_device, edge_degrees, walk_paths
    from ntsynt_amd.synteny import SyntenyEngine
    from ntsynt_amd.synteny_device import DeviceSyntenyEngine
    rng = np.random.default_rng(8)
    n = 20000
    hashes = np.unique(rng.integers(1, 1 << 40, size=2 * n, dtype=np.uint64))[:n]         # (distinct; NOT a choice out of an arange of 2^40)
    rng.shuffle(hashes)
    assert hashes.size == n
    tsvs = ["a.fa.k24.w100.tsv", "b.fa.k24.w100.tsv", "c.fa.k24.w100.tsv"]
    names = [["c1", "c2"]] * 3

    def lists_of(idx, jitter):
        out = []
        for a in range(3):
            h = hashes[idx]
            pos = (np.arange(idx.size, dtype=np.uint64) * np.uint64(150) + np.uint64(1000 * a + jitter))
            rec = (np.arange(idx.size) >= idx.size // 2).astype(np.uint32)
            out.append((h, rec, pos))
        return out

    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        host = SyntenyEngine(tsvs, names, 24, 100, [20, 5], 500, 1000, 100, "p", lambda ls, kp, li: build_graph_device(ctx, ls, kp, li), None, walk_paths,
                             degree_fn=edge_degrees)
        dev = DeviceSyntenyEngine(ctx, tsvs, names, 24, 100, [20, 5], 500, 1000, 100, "p", None)
        first = lists_of(np.arange(0, n // 2), 0)
        # the second build: a third of the old chain again (its adjacent pairs are edges that exist), runs of new minimizers in between
        again = np.sort(rng.choice(np.arange(0, n // 2), size=n // 6, replace=False))
        take = np.sort(np.concatenate([np.arange(2000, 5000), again, np.arange(n // 2, n)]))
        take = np.unique(take)
        second = lists_of(take, 7)
        for step, lists in (("first build", first), ("second build", second)):
            ordered = [lists[i] for i in host.input_order]
            host._add_graph(host.graph_fn(ordered, None, None))
            handles = [Minimizers.from_numpy(ctx, *lists[i]) for i in dev.input_order]
            dev._add(handles, None)
            for mx in handles:
                mx.free()
            _state_equal(host, dev, step)
        # the case is what it claims: edges of the second build between vertices of the first that were edges already
        old = np.arange(n) < n // 2
        order = np.argsort(take)
        consecutive = np.abs(np.diff(take[order])) == 1
        assert int((consecutive & old[take[order]][:-1] & old[take[order]][1:]).sum()) > 1000
        dev.graph.free()
    finally:
        os.chdir(cwd)
