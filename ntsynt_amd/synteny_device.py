"""Synteny-block stage with the minimizer graph resident in HBM.

The graph, its path walk (list ranking by pointer jumping), the per-path scans and the refinement round's position
filters run in libntsynt_hip.so (csrc/nts_dgraph.inc, C ABI nts_engine_*); minimizer lists arrive as device handles
straight from nts_sketch / nts_mx_allgather.  What is left here is what the reference does per *block*, not per
minimizer: the bubble rule over its handful of candidate edges (bin/ntsynt_synteny.py:566-590, order dependent), block
order and text (synteny_block.py:72-109), collinear merging (S:428-472) and the mask intervals of the next round
(S:134-146) -- inherited from ntsynt_amd.synteny.SyntenyEngine, which stays as the host-array twin the GPU tests compare
this engine with, state by state.  It never touches the oracle."""
import ctypes
import os
import sys

import numpy as np

from . import _lib
from ._lib import c_vp, u64
from .synteny import EMPTY_FINAL, MX_SUFFIX, SyntenyEngine

IV_OFF = np.int64(1) << 40            # composite interval key: record * 2^40 + position (same as synteny.py)
# memory layout of nts_interval (include/ntsynt_hip.h): hard-mask intervals travel as one array
INTERVAL_DTYPE = np.dtype({"names": ["rec", "start", "end"], "formats": ["<u4", "<u8", "<u8"], "offsets": [0, 8, 16], "itemsize": 24})


class DeviceGraph:
    "thin handle over nts_engine_* (one per run)"

    def __init__(self, ctx, n_asm, ref_asm):
        self.ctx, self.G = ctx, int(n_asm)
        h = c_vp()
        ctx.check(ctx.lib.nts_engine_create(ctx.h, self.G, int(ref_asm), ctypes.byref(h)), "nts_engine_create")
        self.h = h

    def add(self, lists, spans=None):
        "lists: Minimizers handles in engine order; spans: per assembly (start, end_max) uint64 arrays or None"
        arr = (c_vp * self.G)(*[m.h for m in lists])
        sp, hold = None, []
        if spans is not None:
            sp = (_lib.Spans * self.G)()
            for a, (start, end_max) in enumerate(spans):
                start = np.ascontiguousarray(start, dtype=np.uint64)
                end_max = np.ascontiguousarray(end_max, dtype=np.uint64)
                hold += [start, end_max]
                sp[a].start, sp[a].end_max, sp[a].n = start.ctypes.data, end_max.ctypes.data, start.size
        nv, ne = u64(), u64()
        self.ctx.check(self.ctx.lib.nts_engine_add(self.ctx.h, self.h, arr, sp, ctypes.byref(nv), ctypes.byref(ne)), "nts_engine_add")
        return nv.value, ne.value

    def size(self):
        nv, ne = u64(), u64()
        self.ctx.lib.nts_engine_size(self.h, ctypes.byref(nv), ctypes.byref(ne))
        return nv.value, ne.value

    def bubbles(self):
        b = _lib.Bubbles()
        self.ctx.check(self.ctx.lib.nts_engine_bubbles(self.ctx.h, self.h, ctypes.byref(b)), "nts_engine_bubbles")

        def take(p, n):
            return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.uint32)
        out = (take(b.cand_edge, b.n_cand), take(b.inc_edge, b.n_inc), take(b.inc_u, b.n_inc), take(b.inc_v, b.n_inc),
               take(b.inc_w, b.n_inc))
        self.ctx.lib.nts_bubbles_free(ctypes.byref(b))
        return out

    def apply(self, dead, promote, weight):
        dead = np.ascontiguousarray(dead, dtype=np.uint32)
        promote = np.ascontiguousarray(promote, dtype=np.uint32)
        self.ctx.check(self.ctx.lib.nts_engine_apply(self.ctx.h, self.h, dead.ctypes.data, dead.size, promote.ctypes.data, promote.size,
                                                     int(weight)), "nts_engine_apply")

    def filter(self, min_weight, flag):
        n = u64()
        self.ctx.check(self.ctx.lib.nts_engine_filter(self.ctx.h, self.h, int(min_weight), int(bool(flag)), ctypes.byref(n)),
                       "nts_engine_filter")
        return n.value

    def erode(self, k):
        n = u64()
        self.ctx.check(self.ctx.lib.nts_engine_erode(self.ctx.h, self.h, int(k), ctypes.byref(n)), "nts_engine_erode")
        return n.value

    def blocks(self, bp, m, min_mx):
        b = _lib.Blocks()
        self.ctx.check(self.ctx.lib.nts_engine_blocks(self.ctx.h, self.h, int(bp), float(m), int(min_mx), ctypes.byref(b)),
                       "nts_engine_blocks")
        n, G = int(b.n_blocks), self.G

        def take(p, cnt, dtype):
            return np.ctypeslib.as_array(p, shape=(cnt,)).copy() if cnt else np.zeros(0, dtype)
        out = {"n": n, "first_vid": take(b.first_vid, n, np.uint32), "last_vid": take(b.last_vid, n, np.uint32),
               "n_mx": take(b.n_mx, n, np.uint32), "rec": take(b.rec, G * n, np.uint32).reshape(G, n),
               "first_pos": take(b.first_pos, G * n, np.uint64).reshape(G, n).astype(np.int64),
               "last_pos": take(b.last_pos, G * n, np.uint64).reshape(G, n).astype(np.int64),
               "ori": take(b.ori, G * n, np.uint8).reshape(G, n),
               "paths": int(b.stats_paths), "unoriented": int(b.stats_unoriented), "indel_cuts": int(b.stats_indel_cuts),
               "small": int(b.stats_small)}
        self.ctx.lib.nts_blocks_free(ctypes.byref(b))
        return out

    def read(self, field):
        "state read-back (tests): numpy array of the field"
        nv, ne = self.size()
        npaths, nverts = u64(), u64()
        self.ctx.lib.nts_engine_paths(self.h, ctypes.byref(npaths), ctypes.byref(nverts))
        shape = {"v_hash": (np.uint64, nv), "v_alive": (np.uint8, nv), "internal": (np.uint8, nv), "terminal": (np.uint8, nv),
                 "v_rec": (np.uint32, self.G * nv), "v_pos": (np.uint64, self.G * nv), "e_u": (np.uint32, ne), "e_v": (np.uint32, ne),
                 "e_w": (np.uint32, ne), "e_alive": (np.uint8, ne), "path_verts": (np.uint32, nverts.value),
                 "path_off": (np.uint64, npaths.value + 1)}[field]
        out = np.zeros(max(shape[1], 1), dtype=shape[0])[:shape[1]]
        self.ctx.check(self.ctx.lib.nts_engine_read(self.ctx.h, self.h, field.encode(), out.ctypes.data, out.nbytes), "nts_engine_read")
        return out.reshape(self.G, -1) if field in ("v_rec", "v_pos") else out

    def free(self):
        if self.h:
            self.ctx.lib.nts_engine_free(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceSyntenyEngine(SyntenyEngine):
    """files / contig_names / parameters as SyntenyEngine.  sketch_dev_fn({a: masks or None}, w) -> {a: Minimizers handle}
    for the caller's assembly indices a (device-resident lists; the engine frees them)."""

    def __init__(self, ctx, files, contig_names, k, w, w_rounds, bp, collinear_merge, z, prefix, sketch_dev_fn, simplify=True,
                 m=90, n=0, log=None, dev=False, interarrivals=False):
        super().__init__(files, contig_names, k, w, w_rounds, bp, collinear_merge, z, prefix, None, None, None, simplify=simplify,
                         m=m, n=n, log=log, scan_fn=False, dev=dev, interarrivals=interarrivals)
        self.ctx = ctx
        self.sketch_dev_fn = sketch_dev_fn
        self.graph = DeviceGraph(ctx, self.G, self.ref)
        self._ctg_rank = None
        self._blob = None

    def _instrument(self):
        import time

        def wrap(name, fn):
            def timed(*a, **k):
                t0 = time.perf_counter()
                try:
                    return fn(*a, **k)
                finally:
                    self.times[name] = self.times.get(name, 0.0) + time.perf_counter() - t0
            return timed
        for name in ("_add", "_simplify_dev", "_filter", "_erode", "_blocks", "_sorted", "_emit", "_merge", "_mask_intervals", "_spans",
                     "_sketch_round", "_long_mask"):
            setattr(self, name, wrap(name, getattr(self, name)))

    # ------------------------------------------------------------------ device steps
    def _add(self, lists, spans):
        self.graph.add(lists, spans)

    def _simplify_dev(self, apply_deletions):
        "run_graph_simplification (S:548-590) on the table of candidate edges and their neighbourhood (nts_bubble_rule)"
        cand, inc_e, inc_u, inc_v, inc_w = self.graph.bubbles()
        if os.environ.get("NTS_DEBUG_ENGINE"):
            print(f"[engine simplify] candidate edges {cand.size}, edges at their ends {inc_e.size}", file=sys.stderr, flush=True)
        if cand.size == 0:
            return
        wmax = self.G                                          # sum of the weights, all 1 (S:32, S:571)
        cand = np.ascontiguousarray(np.sort(cand), dtype=np.uint32)      # ascending edge index = reference edge order
        doomed = np.empty(cand.size, np.uint32)
        promoted = np.empty(cand.size, np.uint32)
        n = u64()
        rc = self.ctx.lib.nts_bubble_rule(cand.size, cand.ctypes.data, inc_e.size, inc_e.ctypes.data, inc_u.ctypes.data, inc_v.ctypes.data,
                                          inc_w.ctypes.data, wmax, doomed.ctypes.data, promoted.ctypes.data, ctypes.byref(n))
        if rc != 0:
            raise RuntimeError(f"nts_bubble_rule failed ({rc})")
        self.stats["bubbles"] += n.value
        self.graph.apply(doomed[:n.value] if apply_deletions else [], promoted[:n.value], wmax)

    def _filter(self, flag):
        return self.graph.filter(self.n, flag)

    def _erode(self):
        self.stats["eroded_edges"] += self.graph.erode(self.k)

    # ------------------------------------------------------------------ block tables (arrays, [assembly, block])
    # A round's blocks stay a table -- rec, first_pos, last_pos, ori (0 '+', 1 '-') per assembly, n_mx and reason per block --
    # from nts_engine_blocks to the TSV text; sort, length rule, mask intervals and interval keys are array passes, the
    # order-dependent merge and the text are native (nts_blocks_merge, nts_blocks_text).
    def _blocks(self):
        tb = self.graph.blocks(self.bp, self.m, 4)
        self.stats["unoriented"] += tb["unoriented"]
        self.stats["indel_cuts"] += tb["indel_cuts"]
        self.stats["small_blocks"] += tb["small"]
        n = tb["n"]
        self._last_vids = (tb["first_vid"], tb["last_vid"])        # (--interarrivals reads the blocks' vertices back)
        return {"n": n, "rec": tb["rec"], "first_pos": tb["first_pos"], "last_pos": tb["last_pos"], "ori": tb["ori"],
                "n_mx": tb["n_mx"].astype(np.int64), "reason": np.zeros(n, np.uint8)}

    def _interarrivals_dev(self):
        "a block is a stretch of its path: its vertices are path_verts between the positions of its first and last vertex"
        pv = self.graph.read("path_verts").astype(np.int64)
        v_pos = self.graph.read("v_pos").astype(np.int64)
        where = np.full(v_pos.shape[1], -1, np.int64)
        where[pv] = np.arange(pv.size)
        first, last = self._last_vids
        lists = []
        for a, b in zip(where[first.astype(np.int64)], where[last.astype(np.int64)]):
            lo, hi = (a, b) if a <= b else (b, a)
            seg = pv[lo:hi + 1]
            lists.append(seg if a <= b else seg[::-1])
        self._write_interarrivals(lists, v_pos)

    @staticmethod
    def _take(tb, idx):
        out = {"n": int(len(idx))}
        for key in ("rec", "first_pos", "last_pos", "ori"):
            out[key] = np.ascontiguousarray(tb[key][:, idx])
        for key in ("n_mx", "reason"):
            out[key] = np.ascontiguousarray(tb[key][idx])
        return out

    def _sorted(self, tb):
        "SyntenyBlock.__lt__ (synteny_block.py:102-109): by (contig name, start) in the lexicographically smallest assembly"
        if tb["n"] == 0:
            return tb
        if self._ctg_rank is None:
            names = self.contigs[self.ref]
            rank = {nm: i for i, nm in enumerate(sorted(set(names)))}
            self._ctg_rank = np.array([rank[nm] for nm in names], np.int64)
        ref = self.ref
        start = np.minimum(tb["first_pos"][ref], tb["last_pos"][ref])
        return self._take(tb, np.lexsort((start, self._ctg_rank[tb["rec"][ref]])))

    def _long_mask(self, tb):
        return (np.abs(tb["first_pos"] - tb["last_pos"]) >= self.z - self.k).all(axis=0)   # end - start = |first - last| + k

    def _names_blob(self):
        if self._blob is None:
            asm = []
            for f in self.files:
                mt = MX_SUFFIX.search(f)
                asm.append(mt.group(1) if mt else f)
            base = np.concatenate(([0], np.cumsum([len(c) for c in self.contigs])[:-1])).astype(np.uint64)
            text = "\0".join(asm + [nm for c in self.contigs for nm in c]) + "\0"
            self._blob = (text.encode(), base, np.array(self.out_order, np.uint32))
        return self._blob

    def _emit(self, name, tb, verbose=False):
        blob, base, order = self._names_blob()
        buf, nbytes = c_vp(), u64()
        n = tb["n"]
        arrs = [np.ascontiguousarray(tb[key]) for key in ("rec", "first_pos", "last_pos", "ori", "n_mx", "reason")]
        rc = self.ctx.lib.nts_blocks_text(self.G, n, self.k, self.z, order.ctypes.data, blob, len(blob), base.ctypes.data,
                                          arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data,
                                          arrs[4].ctypes.data, arrs[5].ctypes.data if verbose else None, ctypes.byref(buf),
                                          ctypes.byref(nbytes))
        if rc != 0:
            raise RuntimeError(f"nts_blocks_text failed ({rc})")
        raw = ctypes.string_at(buf, nbytes.value)
        self.ctx.lib.nts_free(buf)
        self.outputs[name] = raw.decode()
        with open(name, "wb") as fh:
            fh.write(raw)

    def _merge(self, tb):
        "merge_collinear_blocks (S:428-472) over a sorted table"
        if tb["n"] == 0:
            return tb
        out = {key: np.array(tb[key], copy=True, order="C") for key in ("rec", "first_pos", "last_pos", "ori", "n_mx", "reason")}
        n_out, n_merged = u64(), u64()
        rc = self.ctx.lib.nts_blocks_merge(self.G, tb["n"], self.k, self.bp, self.collinear_merge, out["rec"].ctypes.data,
                                           out["first_pos"].ctypes.data, out["last_pos"].ctypes.data, out["ori"].ctypes.data,
                                           out["n_mx"].ctypes.data, out["reason"].ctypes.data, ctypes.byref(n_out), ctypes.byref(n_merged))
        if rc != 0:
            raise RuntimeError(f"nts_blocks_merge failed ({rc})")
        self.stats["merged"] += n_merged.value
        out["n"] = tb["n"]
        return self._take(out, np.arange(n_out.value))

    def _mask_intervals(self, tb, w):
        "hard-mask intervals of the next re-sketch (S:134-146) per assembly, as Interval arrays (record, start, end)"
        lim = max(2 * w, w + self.k + 1)
        out = []
        for a in range(self.G):
            s = np.minimum(tb["first_pos"][a], tb["last_pos"][a])
            e = np.maximum(tb["first_pos"][a], tb["last_pos"][a]) + self.k
            s2, e2 = s + (w + self.k), e - (w + self.k)
            ok = (e - s > lim) & (e2 > s2)
            iv = np.zeros(int(ok.sum()), dtype=INTERVAL_DTYPE)
            iv["rec"], iv["start"], iv["end"] = tb["rec"][a][ok], s2[ok], e2[ok]
            out.append(iv)
        return out

    def _spans(self, tb):
        "block interiors [min+1, max) per assembly as sorted composite keys + running maximum of the ends (S:194-203)"
        out = []
        for a in range(self.G):
            p0, p1 = tb["first_pos"][a], tb["last_pos"][a]
            lo, hi = np.minimum(p0, p1), np.maximum(p0, p1)
            ok = hi - lo >= 2
            ir = tb["rec"][a][ok].astype(np.int64)
            iv_s, iv_e = lo[ok] + 1, hi[ok]
            order = np.argsort(ir * IV_OFF + iv_s, kind="stable")
            comp_s = (ir * IV_OFF + iv_s)[order]
            comp_mx = np.maximum.accumulate((ir * IV_OFF + iv_e)[order]) if order.size else comp_s
            out.append((comp_s.astype(np.uint64), comp_mx.astype(np.uint64)))
        return out

    @staticmethod
    def rows(tb):
        "a table as sorted tuples (tests compare it with the host-array engine's Block objects)"
        sym = "+-?"
        return sorted((tuple(tb["rec"][:, i].tolist()), tuple(sym[c] for c in tb["ori"][:, i].tolist()),
                       tuple(tb["first_pos"][:, i].tolist()), tuple(tb["last_pos"][:, i].tolist()), int(tb["n_mx"][i]))
                      for i in range(tb["n"]))

    def _sketch_round(self, masks, new_w):
        got = self.sketch_dev_fn({self.input_order[a]: masks[a] for a in range(self.G)}, new_w)
        return [got[self.input_order[a]] for a in range(self.G)]

    # ------------------------------------------------------------------ driver (S:476-530, S:593-647)
    def run(self, initial):
        """initial[i] = Minimizers handle (device) of assembly i in the caller's order; freed here."""
        if len(self.w_rounds) != len(set(self.w_rounds)):
            print("Error: duplicate values found in w_rounds!", file=sys.stderr, flush=True)
            sys.exit(1)
        lists = [initial[i] for i in self.input_order]
        self._add(lists, None)
        for mx in lists:
            mx.free()
        if self.simplify:
            self._simplify_dev(apply_deletions=True)
        if self.n > 1:
            self._filter(flag=False)
        blocks = self._blocks()
        if self.interarrivals:
            self._interarrivals_dev()
        ordered = self._sorted(blocks)
        if ordered["n"] == 0:
            print("Error - no paths found. Try adjusting the specified k/w parameters.")
            sys.exit(1)
        self._emit(f"{self.prefix}.synteny_blocks.tsv", ordered)
        if getattr(self, "initial_only", False):               # (ntsynt_run.py --initial-only of this build: the initial round's table,
            return                                              # which the reference overwrites at S:516-523 -- no FASTA needed)
        prev_w = self.w
        for new_w in self.w_rounds:
            self.log(f"Extending synteny blocks with w = {new_w}")
            if blocks["n"]:                                    # (no block: no assembly is sketched again -- SyntenyEngine._new_round_graph)
                masks = self._mask_intervals(blocks, prev_w)
                spans = self._spans(blocks)
                lists = self._sketch_round(masks, new_w)
                self._add(lists, spans)
                for mx in lists:
                    mx.free()
            if self.simplify:
                self._simplify_dev(apply_deletions=False)      # S:483-491: the deletions are lost, the promotions stay
            last = new_w == self.w_rounds[-1]
            if last or self.n > 1:
                self._filter(flag=last)
            if last:
                self._erode()
            blocks = self._blocks()
            ordered = self._sorted(blocks)
            self._emit(f"{self.prefix}.pre-collinear-merge.synteny_blocks.tsv", ordered)
            if last:
                if not ordered["n"]:                           # (S:437: the reference's IndexError -- see SyntenyEngine.run)
                    raise IndexError(EMPTY_FINAL.format(""))
                merged = self._merge(ordered)
                merged = self._take(merged, np.flatnonzero(self._long_mask(merged)))
                if not merged["n"]:
                    raise IndexError(EMPTY_FINAL.format(" of at least z bases"))
                merged = self._merge(merged)
                if self.dev and merged["n"]:
                    self._warn_overlaps(merged["rec"], np.minimum(merged["first_pos"], merged["last_pos"]),
                                        np.maximum(merged["first_pos"], merged["last_pos"]) + self.k)
                self._emit(f"{self.prefix}.synteny_blocks.tsv", merged, verbose=True)
            prev_w = new_w
        self.graph.free()
        return self.outputs
