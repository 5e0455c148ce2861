"""The recorded run of the reference's own graph stage (tests/golden/refrun/, written by tests/golden/make_golden_refrun.py in the
build container: bin/ntsynt_synteny.py's main_synteny executed over stand-ins for ntJoin / igraph / ncls / intervaltree / bedtools) as a
checker: loads a scenario, and walks an engine's state against the trace, event by event.

An engine is looked at through a small "view" (EngineView below: the host-array twin ntsynt_amd.synteny.SyntenyEngine; the device engine
is compared with that twin state by state in tests/test_gpu_engine.py, and with the reference's bytes end to end)."""
import gzip
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFRUN = os.path.join(HERE, "golden", "refrun")


def scenario_names():
    return sorted(d for d in os.listdir(REFRUN) if os.path.isdir(os.path.join(REFRUN, d)))


class Scenario:
    def __init__(self, name):
        self.name = name
        self.dir = os.path.join(REFRUN, name)
        with open(os.path.join(self.dir, "meta.json")) as fh:
            self.meta = json.load(fh)
        self.prefix = self.meta["prefix"]
        self._trace = None

    @property
    def trace(self):
        if self._trace is None:
            with gzip.open(os.path.join(self.dir, "trace.json.gz")) as fh:
                self._trace = json.loads(fh.read())
        return self._trace

    def expected(self, suffix):
        with open(os.path.join(self.dir, f"{self.prefix}.{suffix}")) as fh:
            return fh.read()

    def unpack(self, dest):
        "the scenario's FASTA files (and .fai) into `dest`; returns the paths in the reference's argument order"
        out = []
        for f in self.meta["fastas"]:
            p = os.path.join(dest, f)
            with gzip.open(os.path.join(self.dir, f + ".gz")) as fi, open(p, "wb") as fo:
                fo.write(fi.read())
            out.append(p)
        return out

    def kwargs(self):
        "the run's parameters under the names oracle.synteny_oracle.run_pipeline / ntsynt_amd.pipeline.run share"
        m = self.meta
        return dict(k=m["k"], w=m["w"], w_rounds=m["w_rounds"], indel=m["indel"], merge=m["merge"], block_size=m["z"],
                    common=m.get("common", True), prefix=self.prefix, simplify=self.simplify)

    @property
    def simplify(self):
        "False: the run was made without --simplify-graph (ntsynt_run.py), i.e. ntSynt --no-simplify-graph"
        return self.meta.get("simplify", True)

    @property
    def filter_mode(self):
        "ntsynt_run.py --filter: None, 'Indexlr' (refinement sketches with -r <repeat filter>) or 'Filter' (lists read without the k-mers it holds)"
        return self.meta.get("filter")

    def repeat_filter(self, genomes):
        "the run's repeat filter, built again (oracle.nts_oracle.repeat_bf over the genomes in argument order) and checked against the record"
        from oracle import nts_oracle as O
        rep = O.repeat_bf(genomes, self.meta["k"], self.meta["repeat_bytes"])
        assert int(O.bf_popcount(rep)) == self.meta["repeat_popcount"]
        return rep

    @property
    def m(self):
        "ntsynt_run.py -m: per cent of position differences that must agree to orient a contig [90]"
        return self.meta.get("m", 90)

    @property
    def min_weight(self):
        return self.meta.get("min_weight", 0)

    @property
    def stopped(self):
        "the exception the reference's run ended in (a last round without blocks: IndexError at S:437), or None"
        return self.meta.get("stopped")

    def ends_like_the_reference(self):
        """Context for whatever runs the scenario to its end: a run the reference finished must finish, one it ended in an IndexError must
        end in one (what it had written by then is compared by the caller)."""
        import contextlib
        import pytest
        return pytest.raises(IndexError) if self.stopped else contextlib.nullcontext()


def edge_digest(edges):
    "the digest make_golden_refrun.py stores for a graph: sha1 over the sorted (smaller name, larger name, weight) lines"
    rows = sorted((min(a, b), max(a, b), int(w)) for a, b, w in edges)
    return {"ne": len(rows), "sha1": hashlib.sha1("\n".join(f"{a} {b} {w}" for a, b, w in rows).encode()).hexdigest()}


def lists_digest(lists_by_asm, info=None):
    "the digest make_golden_refrun.py stores for minimizer lists (optionally with contig and position per minimizer)"
    out = {}
    for a in sorted(lists_by_asm):
        rows = []
        for lst in lists_by_asm[a]:
            if len(lst):
                rows.append(" ".join(m if info is None else f"{m}:{info[a][m][0]}:{int(info[a][m][1])}" for m in lst))
        out[a] = {"lists": len(rows), "mx": sum(r.count(" ") + 1 for r in rows), "sha1": hashlib.sha1("\n".join(rows).encode()).hexdigest()}
    return out


class Cursor:
    "the trace's events in call order"

    def __init__(self, scenario):
        self.sc = scenario
        self.mx = scenario.trace["mx"]
        self.events = scenario.trace["events"]
        self.i = 0

    def peek(self):
        return self.events[self.i]["fn"] if self.i < len(self.events) else None

    def take(self, fn):
        assert self.i < len(self.events), f"trace exhausted, wanted {fn}"
        ev = self.events[self.i]
        assert ev["fn"] == fn, f"event {self.i} is {ev['fn']}, wanted {fn}"
        self.i += 1
        return ev

    def take_all(self, fn):
        out = []
        while self.peek() == fn:
            out.append(self.take(fn))
        return out

    def names(self, ids):
        return [self.mx[i] for i in ids]

    def pairs(self, rows):
        return {frozenset((self.mx[a], self.mx[b])) for a, b in rows}

    def done(self):
        return self.i == len(self.events)


def undelta(d):
    return np.cumsum(np.asarray(d, np.int64)).tolist()


def expected_blocks(cur, blocks):
    "trace blocks -> sorted [(minimizer names in path order, ((assembly, contig, orientation), ...))]"
    return sorted((tuple(cur.names(b["mx"])), tuple(sorted((a, d["contig"], d["ori"]) for a, d in b["asm"].items()))) for b in blocks)


def expected_positions(cur, blocks):
    "trace blocks -> {(first name, last name): {assembly: positions}}"
    return {(cur.mx[b["mx"][0]], cur.mx[b["mx"][-1]]): {a: undelta(d["dpos"]) for a, d in b["asm"].items()} for b in blocks}


# ------------------------------------------------------------------------------------------------ the host-array engine as a view
class EngineView:
    "ntsynt_amd.synteny.SyntenyEngine's state in the trace's terms (names = decimal hash strings, assemblies = TSV names)"

    def __init__(self, eng):
        self.e = eng

    def name(self, vid):
        return str(int(self.e.v_hash[vid]))

    def names(self, vids):
        return [str(int(h)) for h in self.e.v_hash[np.asarray(vids, np.int64)].tolist()]

    def live_edges(self):
        e = self.e
        idx = np.flatnonzero(e.e_alive)
        nu, nv = self.names(e.e_u[idx]), self.names(e.e_v[idx])
        return list(zip(nu, nv, e.e_w[idx].tolist()))

    def digest(self):
        return edge_digest(self.live_edges())

    def dead_vertices(self):
        "ids of the deleted vertices (a hash deleted in one round may come back as a new vertex in a later one: compare ids, then name them)"
        return set(np.flatnonzero(~self.e.v_alive).tolist())

    def newly_dead(self, before):
        return set(self.names(sorted(self.dead_vertices() - before)))

    def blocks(self, blocks):
        e = self.e
        out = []
        for b in blocks:
            asm = tuple(sorted((e.files[a], e.contigs[a][b.rec[a]], b.ori[a]) for a in range(e.G)))
            out.append((tuple(self.names(b.vids)), asm))
        return sorted(out)

    def positions(self, blocks):
        e = self.e
        return {(self.name(b.vids[0]), self.name(b.vids[-1])): {e.files[a]: e.v_pos[a][b.vids].tolist() for a in range(e.G)} for b in blocks}

    def masks(self, blocks, w):
        "per TSV name: sorted [contig, start, end] the next re-sketch hard-masks"
        e = self.e
        out = {}
        for a, lst in enumerate(e._mask_intervals(blocks, w)):
            out[e.files[a]] = sorted([e.contigs[a][int(r)], int(s), int(t)] for r, s, t in lst)
        return out

    def lookup(self, names):
        "vertex ids of the LIVE vertices carrying these names (-1: none)"
        hs, hid = self.e._live_index()
        q = np.array([int(n) for n in names], np.uint64)
        if hs.size == 0:
            return np.full(q.size, -1, np.int64)
        p = np.minimum(np.searchsorted(hs, q), hs.size - 1)
        return np.where(hs[p] == q, hid[p], -1)


class HostLockstep:
    """Drives a SyntenyEngine through SyntenyEngine.run's steps (ntsynt_amd/synteny.py:664-711) one at a time and holds each result
    against the reference's trace.  `after(step, host)` is called after every step (the GPU test compares the device engine there)."""

    def __init__(self, host, scenario, after=None):
        self.h, self.v, self.sc = host, EngineView(host), scenario
        self.cur = Cursor(scenario)
        self.after = after or (lambda step, host: None)
        self.checked = {}

    def _count(self, what, n=1):
        self.checked[what] = self.checked.get(what, 0) + n

    # -- C3: run_graph_simplification (S:566-590)
    def simplify(self, apply_deletions):
        h, cur = self.h, self.cur
        ev = cur.take("run_graph_simplification")
        assert self.v.digest() == ev["graph_before"], "graph before bubble removal"
        dead0 = self.v.dead_vertices()
        w0 = {frozenset((a, b)): w for a, b, w in self.v.live_edges()}
        h._simplify(apply_deletions=apply_deletions)
        # the promotions: judged on the edges that were alive before (a deleted vertex takes its edges along)
        promoted = set()
        e = h
        for i in range(e.e_u.size):
            key = frozenset((self.v.name(e.e_u[i]), self.v.name(e.e_v[i])))
            if key in w0 and int(e.e_w[i]) != w0[key]:
                promoted.add(key)
        assert promoted == cur.pairs(ev["promoted"]), "bubble rule: promoted edges"
        if apply_deletions:
            assert self.v.newly_dead(dead0) == set(cur.names(ev["removed"])), "bubble rule: removed vertices"
            assert self.v.digest() == ev["graph_after"], "graph after bubble removal"
        else:
            # S:483-491: the refinement rounds go on with the graph that kept its vertices and took the promotions
            assert self.v.dead_vertices() == dead0
            assert self.v.digest() == ev["input_after"], "graph after promotions"
        self._count("bubbles", len(ev["removed"]))
        self.after("simplify", h)

    def weight_filter(self, last):
        "ntJoin's filter_graph_global / S:292-303 on the last round; returns the flagged pairs for the erosion"
        h, cur = self.h, self.cur
        light = h.e_alive & (h.e_w < h.n)
        flagged = (h.e_u[light], h.e_v[light])
        if last:
            ev = cur.take("filter_graph_global_flag_overlaps")
            assert self.v.digest() == ev["graph_in"]
            got = {frozenset((self.v.name(a), self.v.name(b))) for a, b in zip(*flagged)}
            assert got == cur.pairs(ev["flagged"]), "flagged vertex pairs"
            self._count("flagged", len(ev["flagged"]))
        if last or h.n > 1:
            h.e_alive &= ~light
        if last:
            assert self.v.digest() == ev["graph_after"]
        self.after("filter", h)
        return flagged

    def erode(self, flagged):
        h, cur = self.h, self.cur
        ev = cur.take("refine_graph")
        before = {frozenset((a, b)) for a, b, _ in self.v.live_edges()}
        h._refine_graph(flagged)
        after = {frozenset((a, b)) for a, b, _ in self.v.live_edges()}
        assert before - after == cur.pairs(ev["eroded_edges"]), "eroded edges"
        assert self.v.digest() == ev["graph_after"]
        self._count("eroded_edges", len(ev["eroded_edges"]))
        self.after("erode", h)

    # -- C5-C9: paths -> blocks (S:66-106, S:391-426)
    def round_blocks(self):
        h, cur = self.h, self.cur
        finds = cur.take_all("find_synteny_blocks")
        verts, off = h._paths()
        got_paths = sorted(tuple(self.v.names(verts[off[i]:off[i + 1]])) for i in range(off.size - 1))
        assert got_paths == sorted(tuple(cur.names(ev["path"])) for ev in finds), "paths (vertex order included)"
        dead0 = self.v.dead_vertices()
        blocks = h._blocks_of_paths((verts, off))
        ev_i = cur.take("check_for_indels")
        assert ev_i["bp"] == h.bp
        want = expected_blocks(cur, ev_i["blocks_out"])
        assert self.v.blocks(blocks) == want, "blocks after orientation and indel split"
        assert self.v.positions(blocks) == expected_positions(cur, ev_i["blocks_out"]), "positions of the blocks' minimizers"
        unoriented = set().union(*[set(cur.names(ev["removed"])) for ev in finds]) if finds else set()
        assert self.v.newly_dead(dead0) == unoriented, "vertices of blocks without an orientation"
        dead1 = self.v.dead_vertices()
        blocks = h._drop_small(blocks, 4)
        ev_s = cur.take("filter_synteny_blocks")
        assert ev_s["threshold"] == 4 and ev_s["n_in"] == len(ev_i["blocks_out"])
        kept = [ev_i["blocks_out"][i] for i in ev_s["kept"]]
        assert self.v.blocks(blocks) == expected_blocks(cur, kept), "blocks of at least four minimizers"
        assert self.v.newly_dead(dead1) == set(cur.names(ev_s["removed"])), "vertices of the dropped blocks"
        assert self.v.digest() == ev_s["graph_after"], "graph after the round's block rules (indel edges, dropped vertices)"
        self._count("paths", len(finds))
        self._count("unoriented_vertices", len(unoriented))
        self._count("multi_run_paths", sum(len(ev["blocks"]) == 0 or len(ev["blocks"][0]["mx"]) < len(ev["path"]) for ev in finds))
        self._count("indel_edges", len(ev_i["removed_edges"]))
        self._count("small_block_vertices", len(ev_s["removed"]))
        self.after("blocks", h)
        return blocks

    # -- B5 + C11: the refinement round's inputs (S:118-290, S:532-541)
    def new_round_graph(self, blocks, new_w, prev_w):
        h, cur, v = self.h, self.cur, self.v
        k = h.k
        ev_b = cur.take("get_synteny_bed_lists")
        beds = {}
        for b in blocks:
            for a in range(h.G):
                beds.setdefault(h.files[a], {}).setdefault(h.contigs[a][b.rec[a]], []).append(
                    [min(int(h.v_pos[a][b.vids[0]]), int(h.v_pos[a][b.vids[-1]])), max(int(h.v_pos[a][b.vids[0]]), int(h.v_pos[a][b.vids[-1]])) + k])
        assert {a: {c: sorted(x) for c, x in d.items()} for a, d in beds.items()} == \
            {a: {c: sorted(x) for c, x in d.items()} for a, d in ev_b["beds"].items()}, "block extents (A:17-23)"
        ev_m = cur.take("mask_assemblies_with_synteny_extents")
        assert ev_m["w"] == prev_w
        tsv_of = {f: t for t, f in ((t, t[:t.rindex(".k")]) for t in h.files)}          # FASTA name -> TSV name
        want_masks = {t: [] for t in h.files}
        for pf in ev_m["per_fasta"]:
            want_masks[tsv_of[pf["fasta"]]] = sorted(pf["masked"])
        assert v.masks(blocks, prev_w) == want_masks, "hard-mask intervals"
        ev_g = cur.take("generate_new_minimizers")
        assert ev_g["w"] == new_w
        ev_x = cur.take("find_mx_in_blocks")
        ev_f = cur.take("filter_minimizers_synteny_blocks")
        ev_u = cur.take("update_list_mx_info")
        captured = {}
        graph_fn = h.graph_fn

        def spy(lists, keeps, list_ids):
            captured["args"] = (lists, keeps, list_ids)
            return graph_fn(lists, keeps, list_ids)
        h.graph_fn = spy
        try:
            terminal = h._new_round_graph(blocks, new_w, prev_w)
        finally:
            h.graph_fn = graph_fn
        if "args" not in captured:                             # a round without a block: the engine reads nothing and builds nothing
            assert not blocks and ev_g["sketch"] == lists_digest({}, {}), "only a round without blocks may skip the re-sketch"
            nothing = lists_digest({}, {})
            assert ev_f["lists_out"] == lists_digest({}) and ev_u["lists_common"] == lists_digest({}) and ev_u["n_valid"] == 0
            assert ev_u["info_after"] == nothing and not ev_x["terminal"] and not ev_x["internal"] and not terminal.any()
            self.after("add", h)
            return terminal
        lists, keeps, list_ids = captured["args"]
        sketch, sketch_info, filtered = {}, {}, {}
        for a in range(h.G):
            f = h.files[a]
            h1, rec, pos = (np.asarray(x) for x in lists[a])
            names = [str(int(x)) for x in h1.tolist()]
            _, inv, cnt = np.unique(h1, return_inverse=True, return_counts=True)
            once = cnt[inv] == 1 if h1.size else np.zeros(0, bool)
            # the re-sketch the engine was handed, after read_minimizers' duplicate removal: one list per record
            sketch_info[f] = {names[i]: (h.contigs[a][int(rec[i])], int(pos[i])) for i in np.flatnonzero(once).tolist()}
            per_rec, prev = [], None
            for i in np.flatnonzero(once).tolist():
                if prev is None or rec[i] != prev:
                    per_rec.append([])
                    prev = rec[i]
                per_rec[-1].append(names[i])
            sketch[f] = per_rec
            # S:256-280: what survives the blocks' interiors and the black list, and where the lists are cut
            kept = np.asarray(keeps[a], bool)
            lid = np.asarray(list_ids[a])
            got, prev = [], None
            for i in np.flatnonzero(kept).tolist():
                if prev is None or lid[i] != prev:
                    got.append([])
                    prev = lid[i]
                got[-1].append(names[i])
            filtered[f] = got
            self._count("filtered_lists", len(got))
        assert lists_digest(sketch, sketch_info) == ev_g["sketch"], "re-sketched minimizers"
        assert lists_digest(filtered) == ev_f["lists_out"], "filtered minimizer lists (S:256-280)"
        # S:205-226: terminal and internal minimizers of the blocks, the interiors' intervals
        assert set(v.names(np.flatnonzero(terminal))) == set(cur.names(ev_x["terminal"])), "terminal minimizers"
        inner = set()
        spans = {}
        for b in blocks:
            inner.update(v.names(b.vids[1:-1]))
        assert inner == set(cur.names(ev_x["internal"])), "internal minimizers"
        # S:282-290: positions of the minimizers that survived the intersection (= the vertices of this round's build)
        common = {f: [[n for n in lst if all(n in filtered_sets[g] for g in filtered)] for lst in filtered[f]] for f in filtered} \
            if (filtered_sets := {f: {n for lst in filtered[f] for n in lst} for f in filtered}) is not None else None
        assert lists_digest(common) == ev_u["lists_common"], "lists after the intersection across assemblies"
        valid = sorted({n for lst in common[h.files[0]] for n in lst})
        assert len(valid) == ev_u["n_valid"]
        vid = v.lookup(valid)
        assert (vid >= 0).all(), "a valid minimizer has no live vertex"
        table = {h.files[a]: {n: (h.contigs[a][int(r)], int(p)) for n, r, p in zip(valid, h.v_rec[a][vid].tolist(), h.v_pos[a][vid].tolist())}
                 for a in range(h.G)}
        assert lists_digest({f: [valid] for f in table}, table) == ev_u["info_after"], "position table after update_list_mx_info"
        self._count("valid_minimizers", len(valid))
        del spans
        self.after("add", h)
        return terminal

    def merges(self):
        return self.cur.take_all("merge_collinear_blocks")


def drive_host(lock, initial_lists, check_outputs=None):
    """SyntenyEngine.run (ntsynt_amd/synteny.py), step by step under a HostLockstep.  Returns the engine's outputs."""
    h = lock.h
    lists = [initial_lists[i] for i in h.input_order]
    h._add_graph(h.graph_fn(lists, None, None))
    lock.after("add", h)
    if h.simplify:
        lock.simplify(apply_deletions=True)
    if h.n > 1:
        h.e_alive &= h.e_w >= h.n
    lock.after("filter", h)
    blocks = lock.round_blocks()
    if h.interarrivals:
        h._write_interarrivals([b.vids for b in blocks])
    ordered = h._sorted(blocks)
    h._emit(f"{h.prefix}.synteny_blocks.tsv", ordered)
    prev_w = h.w
    for new_w in h.w_rounds:
        lock.new_round_graph(blocks, new_w, prev_w)
        if h.simplify:
            lock.simplify(apply_deletions=False)
        last = new_w == h.w_rounds[-1]
        flagged = lock.weight_filter(last)
        if last:
            lock.erode(flagged)
        blocks = lock.round_blocks()
        ordered = h._sorted(blocks)
        h._emit(f"{h.prefix}.pre-collinear-merge.synteny_blocks.tsv", ordered)
        if last:
            if not ordered:                                     # S:437: the reference stopped here, after its last recorded call
                assert lock.cur.done() and lock.sc.stopped, "the engine has no block where the reference's run had some"
            merged = h._merge(ordered)                          # (IndexError without a block, as S:437)
            merged = [b for b in merged if h._long_enough(b)]
            merged = h._merge(merged)
            if h.dev and merged:
                h._warn_overlaps([[b.rec[a] for b in merged] for a in range(h.G)],
                                 [[h._start(b, a) for b in merged] for a in range(h.G)],
                                 [[h._end(b, a) for b in merged] for a in range(h.G)])
            h._emit(f"{h.prefix}.synteny_blocks.tsv", merged, verbose=True)
            ev = lock.merges()
            assert len(ev) == 2 and ev[1]["n_out"] == len(merged)
        prev_w = new_w
    assert lock.cur.done(), "events of the reference's run left over"
    return h.outputs
