#!/usr/bin/env python3
"""Runs the reference's OWN graph stage in the development container and records what it did.

What runs is the reference's code, unmodified, imported from /root/reference/bin: NtSyntSynteny.main_synteny
(ntsynt_synteny.py:593-647) and everything it calls in ntsynt_synteny.py, synteny_block.py and assembly_block.py --
find_synteny_blocks, check_for_indels / break_synteny_block, filter_synteny_blocks, get_synteny_bed_lists,
mask_assemblies_with_synteny_extents, generate_new_minimizers, find_mx_in_blocks / update_intervals,
filter_minimizers_synteny_blocks, update_list_mx_info, run_graph_simplification, filter_graph_global_flag_overlaps,
has_overlap / erode_edges / refine_graph, merge_collinear_blocks, check_non_overlapping (--dev), print_interarrivals,
refine_block_coordinates, the SyntenyBlock / AssemblyBlock classes and their sort / text.

What is NOT the reference's code are the third-party modules the image lacks.  Each is replaced by a stand-in with the
semantics stated here; every statement is an ASSUMPTION about that library, listed in DESIGN.md section 2:

  ntjoin / ntjoin_utils (un-vendored submodule, .gitmodules:1-3): read_minimizers, filter_minimizers, build_graph,
      filter_graph_global, find_paths = the restatement in oracle/synteny_oracle.py (u6-u9); run_indexlr = the oracle's
      indexlr restatement (oracle/nts_oracle.c; ntHash and the window rule are pinned to the reference's own files).
  igraph: a graph whose vertices keep their insertion order and names, whose edges keep their relative order when
      others are deleted, es() iterates in edge-id order, Vertex.incident()/neighbors()/degree(),
      get_all_simple_paths(s, t, cutoff=2) = the direct edge and every two-edge path.
  ncls.NCLS(starts, ends, ids).has_overlap(s, e): half-open overlap, a < e and b > s (u11).
  intervaltree: tree[a:b] = data, tree[a:b] -> intervals with begin < b and end > a, slice(p) splits the intervals
      strictly containing p, iteration yields Interval(begin, end, data) tuples.
  pybedtools.BedTool(text).slop(g, l, r).sort().mask_fasta(fi, fo): start - l, end + r, clipped to [0, size]; an
      interval left empty masks nothing (u10); hard masking with N.
  seqtk seq / rm (subprocess): single-line FASTA copy / unlink.

Outputs (tests/golden/refrun/<scenario>/): the input FASTAs (gz) and .fai, the expected TSVs of the reference's run
(initial, pre-collinear-merge, final, interarrivals), its stderr warnings (--dev), and trace.json.gz: the inputs and
results of each reference function call, in call order.  tests/golden/unit_cases.json: small per-function vectors
(find_fa_name, update_intervals, AssemblyBlock accessors, SyntenyBlock.continue/start/extend/get_node).

Run ONLY where /root/reference exists:  python tests/golden/make_golden_refrun.py
"""
import collections
import contextlib
import gzip
import hashlib
import io
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "refrun")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import nts_oracle as O                      # noqa: E402
from oracle import synteny_oracle as SO                 # noqa: E402
from ntsynt_amd import synth                            # noqa: E402  (deterministic family generator; data only)

Minimizer = collections.namedtuple("Minimizer", ["mx", "position"])
Bed = collections.namedtuple("Bed", ["contig", "start", "end"])


# ------------------------------------------------------------------------------------------------ igraph stand-in
class _Edge:
    def __init__(self, g, rec):
        self.g, self.rec = g, rec

    index = property(lambda self: self.g._epos()[id(self.rec)])
    source = property(lambda self: self.rec[0])
    target = property(lambda self: self.rec[1])
    tuple = property(lambda self: (self.rec[0], self.rec[1]))

    def __getitem__(self, key):
        return self.rec[2][key]

    def __setitem__(self, key, value):
        self.rec[2][key] = value

    def __hash__(self):
        return id(self.rec)

    def __eq__(self, other):
        return isinstance(other, _Edge) and other.rec is self.rec


class _Vertex:
    def __init__(self, g, i):
        self.g, self.index = g, i

    def __getitem__(self, key):
        assert key == "name"
        return self.g.names[self.index]

    def degree(self):
        return len(self.g.inc[self.index])

    def incident(self):
        return [_Edge(self.g, r) for r in self.g.inc[self.index]]

    def neighbors(self):
        i = self.index
        return [_Vertex(self.g, r[1] if r[0] == i else r[0]) for r in self.g.inc[i]]


class _Seq:
    "vs() / es(): made on demand (the reference asks for graph.vs()[v] inside its loops)"

    def __init__(self, n, make):
        self.n, self.make = n, make

    def __iter__(self):
        return (self.make(i) for i in range(self.n))

    def __len__(self):
        return self.n

    def __getitem__(self, key):
        if isinstance(key, (list, tuple)):
            return [self.make(i) for i in key]
        return self.make(key)


class Graph:
    "igraph.Graph as far as the reference uses it; an edge record is [u, v, {weight, support}] with vertex INDICES"

    def __init__(self):
        self.names, self.idx, self.edges, self.inc = [], {}, [], []
        self._pos = None

    def _epos(self):
        if self._pos is None:
            self._pos = {id(r): i for i, r in enumerate(self.edges)}
        return self._pos

    # -- what the ntJoin stand-in (SO.build_graph) calls
    def add_vertex(self, name):
        if name not in self.idx:
            self.idx[name] = len(self.names)
            self.names.append(name)
            self.inc.append([])

    def find(self, a, b):
        ia, ib = self.idx.get(a), self.idx.get(b)
        if ia is None or ib is None:
            return None
        for r in self.inc[ia]:
            if r[0] == ib or r[1] == ib:
                return r
        return None

    def add_edge(self, s, t, weight, support):
        old = self.find(s, t)
        if old is not None:
            old[2]["weight"], old[2]["support"] = weight, support
            return
        r = [self.idx[s], self.idx[t], {"weight": weight, "support": support}]
        self.edges.append(r)
        self.inc[r[0]].append(r)
        self.inc[r[1]].append(r)
        self._pos = None

    # -- igraph API
    def vs(self):
        return _Seq(len(self.names), lambda i: _Vertex(self, i))

    def es(self):
        return _Seq(len(self.edges), lambda i: _Edge(self, self.edges[i]))

    def incident(self, vid):
        pos = self._epos()
        return [pos[id(r)] for r in self.inc[vid]]

    def copy(self):
        g = Graph()
        g.names, g.idx = list(self.names), dict(self.idx)
        g.inc = [[] for _ in g.names]
        for u, v, at in self.edges:
            r = [u, v, {"weight": at["weight"], "support": list(at["support"])}]
            g.edges.append(r)
            g.inc[u].append(r)
            g.inc[v].append(r)
        return g

    def delete_edges(self, which):
        "edge ids, or Edge objects -- of this graph or of the graph this one was copied from (same ids)"
        ids = {x if isinstance(x, int) else x.index for x in which}
        if not ids:
            return
        dead = {id(self.edges[i]) for i in ids}
        self.edges = [r for r in self.edges if id(r) not in dead]
        self.inc = [[r for r in lst if id(r) not in dead] for lst in self.inc]
        self._pos = None

    def delete_vertices(self, which):
        dead = set(int(i) for i in which)
        if not dead:
            return
        keep = [i for i in range(len(self.names)) if i not in dead]
        remap = {old: new for new, old in enumerate(keep)}
        edges = []
        for u, v, at in self.edges:
            if u in dead or v in dead:
                continue
            edges.append([remap[u], remap[v], at])
        self.names = [self.names[i] for i in keep]
        self.idx = {n: i for i, n in enumerate(self.names)}
        self.edges = edges
        self.inc = [[] for _ in self.names]
        for r in edges:
            self.inc[r[0]].append(r)
            self.inc[r[1]].append(r)
        self._pos = None

    def get_all_simple_paths(self, s, t, cutoff=-1):
        assert cutoff == 2
        out = []
        nb_s = [r[1] if r[0] == s else r[0] for r in self.inc[s]]
        nb_t = set(r[1] if r[0] == t else r[0] for r in self.inc[t])
        for x in nb_s:
            if x == t:
                out.append([s, t])
            elif x in nb_t:
                out.append([s, x, t])
        return out

    # -- for the recorder
    def edge_names(self):
        return [(self.names[u], self.names[v], at["weight"]) for u, v, at in self.edges]

    def to_mx(self):
        "the same graph as the oracle's MxGraph (vertex and edge order kept): input of SO.find_paths"
        g = SO.MxGraph()
        for n in self.names:
            g.add_vertex(n)
        for u, v, at in self.edges:
            g.add_edge(self.names[u], self.names[v], at["weight"], list(at["support"]))
        return g


# ------------------------------------------------------------------------------------------------ other stand-ins
class Interval(collections.namedtuple("Interval", ["begin", "end", "data"])):
    def __new__(cls, begin, end, data=None):
        return super().__new__(cls, begin, end, data)


class IntervalTree:
    def __init__(self):
        self.items = set()

    def __setitem__(self, sl, data):
        if sl.start >= sl.stop:
            raise ValueError("IntervalTree: Null Interval objects not allowed")
        self.items.add(Interval(sl.start, sl.stop, data))

    def __getitem__(self, sl):
        return {iv for iv in self.items if iv.begin < sl.stop and iv.end > sl.start}

    def slice(self, point):
        for iv in list(self.items):
            if iv.begin < point < iv.end:
                self.items.remove(iv)
                self.items.add(Interval(iv.begin, point, iv.data))
                self.items.add(Interval(point, iv.end, iv.data))

    def __iter__(self):
        return iter(self.items)


class NCLS:
    def __init__(self, starts, ends, ids):
        self.iv = sorted(zip(starts, ends))

    def has_overlap(self, s, e):
        return any(a < e and b > s for a, b in self.iv)


class BedTool:
    log = []            # (fasta name, intervals handed in by the reference's filter, intervals masked)

    def __init__(self, text, from_string=False):
        assert from_string
        self.iv = []
        for line in text.splitlines():
            if line.strip():
                c, a, b = line.split("\t")[:3]
                self.iv.append((c, int(a), int(b)))
        self.src = list(self.iv)

    def slop(self, g, l, r):
        sizes = {}
        with open(g) as fh:
            for line in fh:
                f = line.split("\t")
                sizes[f[0]] = int(f[1])
        out = []
        for c, a, b in self.iv:
            a2, b2 = max(0, a - l), min(sizes[c], b + r)
            if b2 > a2:
                out.append((c, a2, b2))
        self.iv = out
        return self

    def sort(self):
        self.iv.sort()
        return self

    def mask_fasta(self, fi, fo):
        g = O.read_fasta(fi)
        BedTool.log.append((os.path.basename(fi), list(self.src), list(self.iv)))
        with open(fo, "w") as fh:
            for i, name in enumerate(g.names):
                buf = bytearray(g.record(i))
                for c, a, b in self.iv:
                    if c == name:
                        buf[a:b] = b"N" * (b - a)
                fh.write(f">{name}\n")
                for j in range(0, len(buf), 60):
                    fh.write(buf[j:j + 60].decode() + "\n")


class _Subprocess:
    "seqtk seq <fa> -> single-line FASTA on stdout; rm <file>"

    @staticmethod
    def run(cmd, stdout=None, check=False, text=False):
        assert cmd[0] == "seqtk" and cmd[1] == "seq"
        g = O.read_fasta(cmd[2])
        for i, name in enumerate(g.names):
            stdout.write(f">{name}\n{g.record(i).decode()}\n")
        return types.SimpleNamespace(returncode=0)

    @staticmethod
    def call(cmd):
        assert cmd[0] == "rm"
        for f in cmd[1:]:
            os.remove(f)
        return 0


STATE = {"bf": {}}          # common.bf path -> bit array (what `indexlr -s <path>` would load)


def _run_indexlr(fasta, k, w, t, s=None, r=None):
    "ntjoin_utils.run_indexlr: `indexlr --long --pos --seq -k k -w w [-s <filter-in>] [-r <filter-out>]` into <fasta>.k<k>.w<w>.tsv"
    g = O.read_fasta(fasta)
    mins = O.minimize(g, k, w, STATE["bf"][s] if s else None, repeat=STATE["bf"][r] if r else None)
    out = f"{fasta}.k{k}.w{w}.tsv"
    O.write_indexlr_tsv(out, g, mins, k)
    return out


class _Ntjoin:
    "ntjoin.Ntjoin as far as NtSyntSynteny inherits from it (restated as in oracle/synteny_oracle.py, u6-u9)"

    def __init__(self, args):
        self.args = args
        self.list_mx_info, self.list_mxs, self.weights, self.graph = {}, {}, {}, None

    def load_minimizers(self, repeat_bf=None):
        for f, wt in zip(self.args.FILES, self.weights_list):
            self.list_mx_info[f], self.list_mxs[f] = SO.read_minimizers_tsv(f, repeat_bf)
            self.weights[f] = wt
        self.list_mxs = SO.filter_minimizers(self.list_mxs)

    def make_minimizer_graph(self):
        self.graph = SO.build_graph(self.list_mxs, self.weights, graph=Graph())

    def filter_graph_global(self, graph):
        if self.args.n <= min(self.weights.values()):
            return graph
        g = graph.copy()
        g.delete_edges([i for i, r in enumerate(g.edges) if r[2]["weight"] < self.args.n])
        return g

    def find_paths(self):
        ref = self.args.FILES[-1]
        return [[(p, None)] for p in SO.find_paths(self.graph.to_mx(), self.list_mx_info[ref])]

    ntjoin_find_paths = find_paths


def install():
    mods = {name: types.ModuleType(name) for name in ("intervaltree", "pybedtools", "btllib", "ncls", "ntjoin_utils", "ntjoin")}
    mods["intervaltree"].Interval, mods["intervaltree"].IntervalTree = Interval, IntervalTree
    mods["ncls"].NCLS = NCLS
    mods["pybedtools"].BedTool = BedTool
    mods["btllib"].KmerBloomFilter = lambda path: STATE["bf"][path]      # (S:606: what ntJoin's read_minimizers asks `contains` of; here the bits)
    u = mods["ntjoin_utils"]
    u.Minimizer, u.Bed = Minimizer, Bed
    u.read_minimizers = SO.read_minimizers_tsv
    u.filter_minimizers = SO.filter_minimizers
    u.build_graph = lambda list_mxs, weights, graph=None, black_list=None: \
        SO.build_graph(list_mxs, weights, graph=graph if graph is not None else Graph(), black_list=black_list)
    u.run_indexlr = _run_indexlr
    u.vertex_index = lambda g, name: g.idx[name]
    u.vertex_name = lambda g, i: g.names[i]
    u.edge_index = lambda g, a, b: g._epos()[id(g.find(a, b))]

    def remove_flagged_edges(g, ids):
        new = g.copy()
        new.delete_edges(sorted(set(ids)))
        return new
    u.remove_flagged_edges = remove_flagged_edges
    mods["ntjoin"].Ntjoin = _Ntjoin
    sys.modules.update(mods)
    sys.path.insert(0, os.path.join(REF, "bin"))
    import ntsynt_synteny
    import synteny_block
    import assembly_block
    ntsynt_synteny.subprocess = _Subprocess
    return ntsynt_synteny, synteny_block, assembly_block


# ------------------------------------------------------------------------------------------------ recorder
class MxTable:
    "minimizer names (decimal hash strings) <-> small integers, so that the trace stays small; trace['mx'][i] is the name"

    def __init__(self):
        self.names, self.ids = [], {}

    def __call__(self, name):
        i = self.ids.get(name)
        if i is None:
            i = self.ids[name] = len(self.names)
            self.names.append(name)
        return i


def delta(xs):
    "positions as first value + differences (decoded by tests/refrun.py: undelta)"
    return [int(xs[0])] + [int(b - a) for a, b in zip(xs, xs[1:])] if len(xs) else []


def blocks_json(blocks, mx=None, positions=True):
    """a block list as data.  With a table: the minimizers once per block (every assembly block of a synteny block lists the same
    minimizers, synteny_block.py:36-46), positions per assembly, delta coded."""
    if mx is None:
        return [{"reason": b.broken_reason,
                 "asm": {a: {"contig": ab.contig_id, "ori": ab.ori, "mx": [[m.mx, int(m.position)] for m in ab.minimizers]}
                         for a, ab in b.assembly_blocks.items()}} for b in blocks]
    out = []
    for b in blocks:
        lists = [[m.mx for m in ab.minimizers] for ab in b.assembly_blocks.values()]
        assert all(x == lists[0] for x in lists)
        out.append({"reason": b.broken_reason, "mx": [mx(m) for m in lists[0]],
                    "asm": {a: dict({"contig": ab.contig_id, "ori": ab.ori}, **({"dpos": delta([m.position for m in ab.minimizers])} if positions else {}))
                            for a, ab in b.assembly_blocks.items()}})
    return out


def edge_digest(edges):
    "sha1 over the sorted (smaller name, larger name, weight) lines of an edge set: how the tests compare whole graphs"
    rows = sorted((min(a, b), max(a, b), int(w)) for a, b, w in edges)
    return {"ne": len(rows), "sha1": hashlib.sha1("\n".join(f"{a} {b} {w}" for a, b, w in rows).encode()).hexdigest()}


def graph_digest(g):
    return edge_digest(g.edge_names())


def lists_digest(lists_by_asm, info=None):
    """minimizer lists per assembly (empty lists skipped) as counts + sha1 of their canonical text: one line per list, names separated
    by blanks -- with `info`, every name as name:contig:position.  (The lists of a refinement round hold every minimizer of the unmasked
    parts of the genomes: too large to store; the tests compare digests.)"""
    out = {}
    for a in sorted(lists_by_asm):
        rows = []
        for lst in lists_by_asm[a]:
            if lst:
                rows.append(" ".join(m if info is None else f"{m}:{info[a][m][0]}:{int(info[a][m][1])}" for m in lst))
        out[a] = {"lists": len(rows), "mx": sum(r.count(" ") + 1 for r in rows), "sha1": hashlib.sha1("\n".join(rows).encode()).hexdigest()}
    return out


def record(eng, trace, mx):
    "wrap the reference engine's methods on the instance; every call appends its inputs / results to `trace`"

    def wrap(name, before=None, after=None):
        fn = getattr(eng, name)

        def wrapped(*a, **k):
            ev = {"fn": name}
            if before:
                before(ev, *a, **k)
            out = fn(*a, **k)
            if after:
                after(ev, out, *a, **k)
            trace.append(ev)
            return out
        setattr(eng, name, wrapped)

    def names_of(g):
        return set(g.names)

    def edges_of(g):
        return {(min(a, b), max(a, b)) for a, b, _ in g.edge_names()}

    # C3 (S:566-590)
    def b_simpl(ev, graph):
        ev["graph_before"] = graph_digest(graph)
        ev["_w"] = {(min(a, b), max(a, b)): w for a, b, w in graph.edge_names()}

    def a_simpl(ev, out, graph):
        w0 = ev.pop("_w")
        ev["removed"] = sorted(mx(n) for n in names_of(graph) - names_of(out))
        ev["promoted"] = sorted([mx(a), mx(b)] for a, b, w in graph.edge_names() if w0[(min(a, b), max(a, b))] != w)
        ev["graph_after"] = graph_digest(out)
        ev["input_after"] = graph_digest(graph)         # the argument itself keeps its vertices and takes the promotions (S:586)
    wrap("run_graph_simplification", b_simpl, a_simpl)

    # C6 / C7 (S:66-106)
    def b_find(ev, path):
        ev["path"] = [mx(m) for m in path]
        ev["_v"] = names_of(eng.graph)

    def a_find(ev, out, path):
        ev["blocks"] = blocks_json(out, mx, positions=False)     # (their positions: in the check_for_indels event that follows)
        ev["removed"] = sorted(mx(n) for n in ev.pop("_v") - names_of(eng.graph))
    wrap("find_synteny_blocks", b_find, a_find)

    # C8 (S:391-409): input = the blocks of the find_synteny_blocks calls before it, in call order
    def b_indel(ev, paths):
        ev["bp"] = eng.args.bp
        ev["n_in"] = len(paths)
        ev["_e"] = edges_of(eng.graph)

    def a_indel(ev, out, paths):
        ev["blocks_out"] = blocks_json(out, mx)
        ev["removed_edges"] = sorted([mx(a), mx(b)] for a, b in ev.pop("_e") - edges_of(eng.graph))
    wrap("check_for_indels", b_indel, a_indel)

    # C9 (S:411-426): input = the check_for_indels result before it; the result is given by index into that list
    def b_small(ev, paths, mx_threshold=1):
        ev["threshold"] = mx_threshold
        ev["n_in"] = len(paths)
        ev["_ids"] = {id(b): i for i, b in enumerate(paths)}
        ev["_v"] = names_of(eng.graph)

    def a_small(ev, out, paths, mx_threshold=1):
        ids = ev.pop("_ids")
        ev["kept"] = [ids[id(b)] for b in out]
        ev["removed"] = sorted(mx(n) for n in ev.pop("_v") - names_of(eng.graph))
        ev["graph_after"] = graph_digest(eng.graph)
    wrap("filter_synteny_blocks", b_small, a_small)

    # C11 pieces
    def a_beds(ev, out, paths):
        ev["beds"] = {a: {c: [[b.start, b.end] for b in lst] for c, lst in d.items()} for a, d in out.items()}
    wrap("get_synteny_bed_lists", None, a_beds)

    def b_mask(ev, synteny_beds, w):
        ev["w"] = w
        BedTool.log.clear()

    def a_mask(ev, out, synteny_beds, w):
        ev["per_fasta"] = [{"fasta": f, "handed_to_slop": [list(x) for x in src], "masked": [list(x) for x in iv]}
                           for f, src, iv in BedTool.log]
        ev["tsv_to_masked_fasta"] = dict(out)
    wrap("mask_assemblies_with_synteny_extents", b_mask, a_mask)

    def a_gen(ev, out, tsv_to_fa, w, retain_files=False):
        list_mxs, info = out
        ev["w"] = w
        for a, v in list_mxs.items():                   # (every key of `info` is listed: read_minimizers drops repeated hashes from both)
            assert sum(len(x) for x in v) == len(info[a])
        ev["sketch"] = lists_digest(list_mxs, info)
    wrap("generate_new_minimizers", None, a_gen)

    def a_inblocks(ev, out, paths):
        terminal, internal, intervals = out
        ev["terminal"] = sorted(mx(m) for m in terminal)
        ev["internal"] = sorted(mx(m) for m in internal)
        ev["intervals"] = {a: {c: [list(x) for x in t.iv] for c, t in d.items()} for a, d in intervals.items()}
    wrap("find_mx_in_blocks", None, a_inblocks)

    def a_filt(ev, out, list_mxs, black_list, list_mx_info, intervals):
        ev["lists_out"] = lists_digest(out)
        ev["n_lists_in"] = {a: len(v) for a, v in list_mxs.items()}
        ev["n_lists_out"] = {a: len(v) for a, v in out.items()}             # (empty lists included: S:262-277 appends them)
    wrap("filter_minimizers_synteny_blocks", None, a_filt)

    def a_upd(ev, out, list_mxs, list_mx_info, new_info):
        ev["lists_common"] = lists_digest(list_mxs)
        valid = {m for v in list_mxs.values() for x in v for m in x}
        # the position table after the update, restricted to the minimizers the update may touch (S:287-290)
        ev["info_after"] = lists_digest({a: [sorted(m for m in valid if m in list_mx_info[a])] for a in new_info}, list_mx_info)
        ev["n_valid"] = len(valid)
    wrap("update_list_mx_info", None, a_upd)

    # C12: last-round filter + erosion (S:292-362)
    def a_flag(ev, out, graph):
        new, pairs = out
        ev["graph_in"] = graph_digest(graph)
        ev["flagged"] = [[mx(graph.names[s]), mx(graph.names[t])] for s, t in pairs]
        ev["graph_after"] = graph_digest(new)
    wrap("filter_graph_global_flag_overlaps", None, a_flag)

    def b_refine(ev, pairs):
        ev["_e"] = edges_of(eng.graph)

    def a_refine(ev, out, pairs):
        ev["eroded_edges"] = sorted([mx(a), mx(b)] for a, b in ev.pop("_e") - edges_of(out))
        ev["graph_after"] = graph_digest(out)
    wrap("refine_graph", b_refine, a_refine)

    def b_merge(ev, blocks):
        ev["n_in"] = len(blocks)

    def a_merge(ev, out, blocks):
        ev["n_out"] = len(out)
        ev["reasons"] = [b.broken_reason for b in out]
    wrap("merge_collinear_blocks", b_merge, a_merge)


# ------------------------------------------------------------------------------------------------ scenarios
SCENARIOS = [
    # name, genomes, total bp, contigs, divergence, seed, k, w, w_rounds, bp, collinear_merge, z, micro events, N runs
    dict(name="s1_three_genomes", n=3, bp=150_000, ctg=2, div=0.01, seed=101, k=24, w=60, w_rounds=[20, 5], indel=300, merge="3w", z=120, micro=10, n_runs=False),
    dict(name="s2_two_genomes_nruns", n=2, bp=200_000, ctg=3, div=0.02, seed=102, k=20, w=80, w_rounds=[25, 6], indel=500, merge=600, z=150, micro=14, n_runs=True),
    dict(name="s3_four_genomes", n=4, bp=120_000, ctg=2, div=0.005, seed=103, k=24, w=50, w_rounds=[15, 4], indel=200, merge="2w", z=100, micro=12, n_runs=False),
    dict(name="s4_one_round", n=3, bp=160_000, ctg=1, div=0.03, seed=104, k=16, w=70, w_rounds=[12], indel=1000, merge=2000, z=200, micro=8, n_runs=True),
    dict(name="s5_defaults_small", n=2, bp=400_000, ctg=2, div=0.004, seed=105, k=24, w=1000, w_rounds=[100, 10], indel=10000, merge=10000, z=500, micro=6, n_runs=False),
    # -n 2 with three genomes (ntsynt_run.py:17): edges two assemblies support stay, paths cross contigs and turn around
    # (S:71-77 contig change, S:80-86 / S:94-105 blocks without an orientation)
    dict(name="s6_min_weight_2_of_3", n=3, bp=150_000, ctg=2, div=0.005, seed=762111, k=16, w=80, w_rounds=[10, 6], indel=500, merge=800, z=50, micro=20, n_runs=True, min_weight=2),
    dict(name="s8_min_weight_3_of_4", n=4, bp=150_000, ctg=3, div=0.002, seed=774273, k=20, w=30, w_rounds=[10, 4], indel=500, merge="2w", z=50, micro=20, n_runs=False, min_weight=3),
    dict(name="s9_min_weight_3_of_4_b", n=4, bp=150_000, ctg=2, div=0.005, seed=298962, k=24, w=50, w_rounds=[15, 4], indel=500, merge="2w", z=100, micro=20, n_runs=False, min_weight=3),
    # a refinement round that ends without a block (found by refrun_stress.py): the next round's synteny_beds is empty, S:134-192 read no
    # assembly, and the last round's merge_collinear_blocks ends the reference in an IndexError at S:437 (kept: `stopped` in meta.json;
    # the final table on disk is the initial round's, the pre-merge table the empty one the last round wrote)
    dict(name="s10_round_without_blocks", n=4, bp=50_000, ctg=1, div=0.005, seed=655642, k=24, w=400, w_rounds=[10, 4], indel=150, merge="1w", z=100, micro=8, n_runs=True,
         min_weight=3, keep_stopped=True),
    # --no-simplify-graph (bin/ntSynt:75-76 -> ntsynt_run.py --simplify-graph absent: S:615-616, S:483-491 skipped) and -m 75 (the share of
    # increasing / decreasing position differences that orients a contig, ntsynt_run.py -m, synteny_block.py)
    dict(name="s11_no_simplify_m75", n=3, bp=150_000, ctg=2, div=0.008, seed=111, k=24, w=60, w_rounds=[20, 5], indel=300, merge="3w", z=120, micro=16, n_runs=True,
         simplify=False, m=75),
    # stage 3's experimental repeat filter (ntsynt_run.py --filter / --repeat; S:172-185, S:601-607): `Indexlr` = the refinement rounds' indexlr
    # runs get -r <repeat filter> next to -s <common>; `Filter` = ntJoin's read_minimizers(file, repeat_bf) leaves out the minimizers whose k-mer
    # the filter holds, for the initial files and every round's lists
    dict(name="s12_filter_indexlr", n=3, bp=180_000, ctg=2, div=0.006, seed=112, k=24, w=60, w_rounds=[20, 6], indel=400, merge="3w", z=120, micro=10, n_runs=False,
         filter="Indexlr"),
    dict(name="s13_filter_filter", n=3, bp=180_000, ctg=2, div=0.006, seed=113, k=24, w=60, w_rounds=[20, 6], indel=400, merge="3w", z=120, micro=10, n_runs=False,
         filter="Filter"),
    # no common filter (ntSynt --no-common: indexlr without -s, S:181)
    dict(name="s7_no_common_filter", n=2, bp=120_000, ctg=2, div=0.01, seed=107, k=24, w=64, w_rounds=[16, 5], indel=2000, merge=60, z=100, micro=10, n_runs=True, common=False),
]


def write_fai(fasta):
    g = O.read_fasta(fasta)
    with open(fasta + ".fai", "w") as fh:
        off = 0
        for i, name in enumerate(g.names):
            n = len(g.record(i))
            off += len(name) + 2
            fh.write(f"{name}\t{n}\t{off}\t{n}\t{n + 1}\n")
            off += n + 1


def run_scenario(ns, sc):
    out_dir = os.path.join(OUT, sc["name"])
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(out_dir)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            fastas = synth.make_family(tmp, sc["n"], sc["bp"], sc["ctg"], sc["div"], seed=sc["seed"], n_runs=sc["n_runs"], micro=sc["micro"], line_width=0)
            fastas = [os.path.basename(p) for p in fastas]            # the reference works in the CWD
            k, w = sc["k"], sc["w"]
            prefix = "ref"
            if sc.get("filter"):
                # something for a repeat filter to hold: in every genome a stretch of the first record once more further down
                for p in fastas:
                    g0 = O.read_fasta(p)
                    recs = [bytes(g0.record(i)) for i in range(len(g0.names))]
                    r0 = recs[0]
                    a, b, at = len(r0) // 10, len(r0) // 10 + max(2000, len(r0) // 8), len(r0) // 2
                    recs[0] = r0[:at] + r0[a:b] + r0[at:]
                    with open(p, "wb") as fh:
                        for nm, rec in zip(g0.names, recs):
                            fh.write(b">" + nm.encode() + b"\n" + rec + b"\n")
            genomes = {p: O.read_fasta(p) for p in fastas}
            use_common = sc.get("common", True)
            bf = O.common_bf(genomes, k, 0.025) if use_common else None
            STATE["bf"] = {f"{prefix}.common.bf": bf}
            rep = None
            if sc.get("filter"):
                # the repeat filter of rule make_repeat_bf (smk:65-72; restated: oracle/nts_oracle.py repeat_bf), sized like the common filter
                rep_bytes = int(bf.size) if use_common else O.bf_ctor_bytes(O.bf_approx_bytes(genomes[sorted(fastas)[0]].total_bp, 0.025))
                rep = O.repeat_bf([genomes[p] for p in fastas], k, rep_bytes)
                STATE["bf"][f"{prefix}.repeat.bf"] = rep
            tsvs = []
            for p in fastas:
                write_fai(p)
                tsv = f"{p}.k{k}.w{w}.tsv"
                O.write_indexlr_tsv(tsv, genomes[p], O.minimize(genomes[p], k, w, bf), k)
                tsvs.append(tsv)
            args = types.SimpleNamespace(FILES=list(tsvs), fastas=list(fastas), n=sc.get("min_weight", 0), p=prefix, k=k, w=w, z=sc["z"], filter=sc.get("filter"),
                                         common=f"{prefix}.common.bf" if use_common else None, repeat=f"{prefix}.repeat.bf" if rep is not None else None, btllib_t=1, w_rounds=list(sc["w_rounds"]),
                                         bp=sc["indel"], collinear_merge=str(sc["merge"]), simplify_graph=sc.get("simplify", True), m=sc.get("m", 90), dev=True,
                                         interarrivals=True, t=1)
            trace, mx = [], MxTable()
            so, se = io.StringIO(), io.StringIO()
            stopped = None
            with contextlib.redirect_stdout(so), contextlib.redirect_stderr(se):
                eng = ns.NtSyntSynteny(args)
                record(eng, trace, mx)
                try:
                    eng.main_synteny()
                except IndexError as e:
                    # (S:437: merge_collinear_blocks on an empty list -- the reference's own end of a run whose last round leaves no block).
                    # With sc["keep_stopped"] the run is kept as a scenario all the same: its trace up to there is what the engines are
                    # held against, its final table is the one the initial round wrote
                    if not sc.get("keep_stopped"):
                        raise
                    stopped = f"IndexError: {e}"
            # the initial table is overwritten by the final one (S:516-523): recover it from the trace's first C9 result
            outputs = {}
            for name in (f"{prefix}.synteny_blocks.tsv", f"{prefix}.pre-collinear-merge.synteny_blocks.tsv", f"{prefix}.interarrivals.tsv"):
                with open(name) as fh:
                    outputs[name] = fh.read()
            for name, text in outputs.items():
                with open(os.path.join(out_dir, name), "w") as fh:
                    fh.write(text)
            warnings = [ln for ln in se.getvalue().splitlines() if ln.startswith("WARNING")]
            stdout_not_oriented = [ln for ln in so.getvalue().splitlines() if ln.startswith("Not oriented")]
            for p in fastas:
                with open(p, "rb") as fi, gzip.GzipFile(os.path.join(out_dir, p + ".gz"), "wb", mtime=0) as fo:
                    fo.write(fi.read())
                shutil.copy(p + ".fai", os.path.join(out_dir, p + ".fai"))
            meta = dict(sc)
            meta.update(fastas=fastas, prefix=prefix, warnings=warnings, stopped=stopped, n_not_oriented=len(stdout_not_oriented),
                        bf_bytes=int(bf.size) if use_common else 0, bf_popcount=int(O.bf_popcount(bf)) if use_common else 0,
                        tsv_sha1={t: hashlib.sha1(open(t, "rb").read()).hexdigest() for t in tsvs})
            if rep is not None:
                meta.update(repeat_bytes=int(rep.size), repeat_popcount=int(O.bf_popcount(rep)))
            with open(os.path.join(out_dir, "meta.json"), "w") as fh:
                json.dump(meta, fh, indent=1)
            with gzip.GzipFile(os.path.join(out_dir, "trace.json.gz"), "wb", mtime=0) as fh:
                fh.write(json.dumps({"mx": mx.names, "events": trace}, separators=(",", ":")).encode())
            counts = collections.Counter(ev["fn"] for ev in trace)
            return outputs, counts, meta
        finally:
            os.chdir(cwd)


# ------------------------------------------------------------------------------------------------ small per-function vectors
def unit_cases(ns, sb, ab):
    import random
    rng = random.Random(11)
    out = {}
    # find_fa_name (S:108-116)
    names = ["a.fa.k24.w1000.tsv", "dir.with.dots/b.fasta.k20.w5.tsv", "x.k1.w2.tsv", "weird name.fa.k24.w10.tsv", "noext", "g.fa.k24.w1000.tsv.extra",
             "g.fa.gz.k32.w100.tsv", "a.fa.k.w1.tsv"]
    fa = []
    for n in names:
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                fa.append([n, ns.NtSyntSynteny.find_fa_name(n)])
        except SystemExit as e:
            fa.append([n, {"exit": e.code}])
    out["find_fa_name"] = fa
    # update_intervals (S:194-203)
    ui = []
    for _ in range(40):
        p1 = rng.randint(20, 70)
        p2 = p1 + rng.choice([0, 1, 2, 3, -1, -2, -3, 17, -17])
        intervals = collections.defaultdict(dict)
        pre = rng.random() < 0.5
        if pre:
            intervals["asm"]["c"] = [(1, 2, 1)]
        ns.NtSyntSynteny.update_intervals("asm", "c", Minimizer("a", p1), Minimizer("b", p2), intervals)
        ui.append({"p1": p1, "p2": p2, "pre": pre, "out": [list(x) for x in intervals.get("asm", {}).get("c", [])]})
    out["update_intervals"] = ui
    # AssemblyBlock accessors (A:17-39)
    acc = []
    for _ in range(30):
        k = rng.choice([16, 24, 32])
        blk = ab.AssemblyBlock(k)
        blk.contig_id = "c" + str(rng.randint(1, 3))
        n = rng.randint(1, 7)
        pos = [rng.randint(0, 10_000) for _ in range(n)]
        blk.minimizers = [Minimizer(str(1000 + i), p) for i, p in enumerate(pos)]
        ctg, m1, m2 = blk.get_block_terminal_mx()
        acc.append({"k": k, "contig": blk.contig_id, "mx": [[m.mx, m.position] for m in blk.minimizers],
                    "start": blk.get_block_start(), "end": blk.get_block_end(), "length": blk.get_block_length(),
                    "terminal": [ctg, list(m1), list(m2)], "contig_start_end": list(blk.get_block_contig_start_end()),
                    "internal": blk.get_block_internal_mx_hashes()})
    out["assembly_block"] = acc
    # SyntenyBlock.continue_block / start_block / extend_block / get_node / get_number_of_minimizers (B:31-46, 87-100)
    walk = []
    for _ in range(30):
        g = rng.randint(2, 4)
        asms = sorted((f"g{i}.fa.k1.w1.tsv" for i in range(g)), reverse=True)
        n = rng.randint(2, 9)
        mxs = [str(5000 + i) for i in range(n)]
        info = {}
        for a in asms:
            ctg = "c1"
            d = {}
            for m in mxs:
                if rng.random() < 0.15:
                    ctg = "c2" if ctg == "c1" else "c1"
                d[m] = (ctg, rng.randint(0, 100_000))
            info[a] = d
        blk = sb.SyntenyBlock(24, 90, *asms)
        steps = []
        for m in mxs:
            cont = blk.continue_block(m, info)
            if cont:
                blk.extend_block(m, info)
            else:
                blk = sb.SyntenyBlock(24, 90, *asms)
                blk.start_block(m, info)
            steps.append({"mx": m, "continued": bool(cont), "n": blk.get_number_of_minimizers(),
                          "last_node": [blk.get_node(blk.get_number_of_minimizers() - 1).mx,
                                        list(blk.get_node(blk.get_number_of_minimizers() - 1).positions)]})
        walk.append({"assemblies": asms, "info": {a: {m: list(v) for m, v in d.items()} for a, d in info.items()}, "steps": steps,
                     "final": blocks_json([blk])[0]})
    out["synteny_block_walk"] = walk
    # filter_minimizers_synteny_blocks (S:256-280) with the half-open interval stand-in (u11)
    fm = []
    for _ in range(40):
        n = rng.randint(3, 25)
        pos = sorted(rng.sample(range(2000), n))
        if rng.random() < 0.3:
            pos = pos[::-1]
        mxs = [str(9000 + i) for i in range(n)]
        cut = rng.randint(1, n)
        lists = [mxs[:cut], mxs[cut:]] if cut < n else [mxs]
        info = {"asm": {m: ("c1" if i < cut or rng.random() < 0.7 else "c2", p) for i, (m, p) in enumerate(zip(mxs, pos))}}
        ivs = []
        for _ in range(rng.randint(0, 3)):
            s = rng.randint(0, 1900)
            ivs.append((s, s + rng.randint(2, 300), 1))
        intervals = {"asm": {}}
        if ivs:
            s, e, d = zip(*ivs)
            intervals["asm"]["c1"] = NCLS(list(s), list(e), list(d))
        black = set(rng.sample(mxs, rng.randint(0, max(0, n // 4))))
        got = ns.NtSyntSynteny.filter_minimizers_synteny_blocks({"asm": lists}, black, info, intervals)
        fm.append({"lists": lists, "info": {m: list(v) for m, v in info["asm"].items()}, "intervals_c1": [list(x[:2]) for x in ivs],
                   "black": sorted(black), "out": got["asm"]})
    out["filter_minimizers_synteny_blocks"] = fm
    # update_list_mx_info (S:282-290)
    up = []
    for _ in range(20):
        old = {"a": {str(i): ("c1", i * 10) for i in range(10)}, "b": {str(i): ("c1", i * 11) for i in range(10)}}
        new = {"a": {str(i): ("c2", i * 7 + 1) for i in rng.sample(range(20), 8)}, "b": {str(i): ("c2", i * 5 + 2) for i in rng.sample(range(20), 8)}}
        lists = {"a": [[str(i) for i in rng.sample(range(20), 5)], []], "b": [[str(i) for i in rng.sample(range(20), 4)]]}
        before = {a: {m: list(v) for m, v in d.items()} for a, d in old.items()}
        ns.NtSyntSynteny.update_list_mx_info(lists, old, new)
        up.append({"lists": lists, "old": before, "new": {a: {m: list(v) for m, v in d.items()} for a, d in new.items()},
                   "after": {a: {m: list(v) for m, v in d.items()} for a, d in old.items()}})
    out["update_list_mx_info"] = up
    with open(os.path.join(HERE, "unit_cases.json"), "w") as fh:
        json.dump(out, fh)
    return {k: len(v) for k, v in out.items()}


def main():
    if os.environ.get("PYTHONHASHSEED") != "0":
        # (the numbering of the minimizers in a trace follows the iteration order of a few sets of strings: fixed, so that the fixtures
        #  reproduce byte for byte)
        os.execve(sys.executable, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], dict(os.environ, PYTHONHASHSEED="0"))
    ns, sb, ab = install()
    os.makedirs(OUT, exist_ok=True)
    print("unit vectors:", unit_cases(ns, sb, ab))
    for sc in SCENARIOS:
        outputs, counts, meta = run_scenario(ns, sc)
        n_final = len(outputs[f"{meta['prefix']}.synteny_blocks.tsv"].splitlines()) // sc["n"]
        print(sc["name"], "final blocks:", n_final, "warnings:", len(meta["warnings"]), "not oriented:", meta["n_not_oriented"], dict(counts))


if __name__ == "__main__":
    main()
