"""Device-side minimizer-graph build (rows C1, C2) and the native chain walk (row C5) over the C ABI."""
import ctypes

import numpy as np

from . import _lib
from .synteny import GraphArrays


def build_graph_device(ctx, lists, keeps=None, list_ids=None):
    """lists[a] = (h1 uint64[], rec, pos) in the reference's assembly order; keeps[a]: bool mask or
    None; list_ids[a]: list id per element or None (= record).  Runs nts_graph_build on the GPU."""
    G = len(lists)
    arr = (_lib.MxList * G)()
    hold = []
    for a, (h1, rec, pos) in enumerate(lists):
        h1 = np.ascontiguousarray(h1, dtype=np.uint64)
        rec = np.ascontiguousarray(rec, dtype=np.uint32)
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        hold += [h1, rec, pos]
        arr[a].h1, arr[a].rec, arr[a].pos, arr[a].n = h1.ctypes.data, rec.ctypes.data, pos.ctypes.data, h1.size
        arr[a].keep = arr[a].list_id = None
        if keeps is not None and keeps[a] is not None:
            kp = np.ascontiguousarray(keeps[a], dtype=np.uint8)
            hold.append(kp)
            arr[a].keep = kp.ctypes.data
        if list_ids is not None and list_ids[a] is not None:
            li = np.ascontiguousarray(list_ids[a], dtype=np.uint32)
            hold.append(li)
            arr[a].list_id = li.ctypes.data
    g = _lib.Graph()
    ctx.check(ctx.lib.nts_graph_build(ctx.h, G, arr, ctypes.byref(g)), "nts_graph_build")
    nv, ne = int(g.nv), int(g.ne)

    def take(ptr, n, dtype):
        if n == 0:
            return np.zeros(0, dtype=dtype)
        out = np.empty(n, dtype=dtype)          # widened / copied by the library's host threads (nts_to_i64)
        src_bytes = ctypes.sizeof(ptr._type_)
        if ctx.lib.nts_to_i64(ctypes.cast(ptr, ctypes.c_void_p), src_bytes, n, out.ctypes.data) != 0:
            raise RuntimeError("nts_to_i64 failed")
        return out
    out = GraphArrays(
        v_hash=take(g.v_hash, nv, np.uint64),
        occ_rec=take(g.occ_rec, G * nv, np.int64).reshape(G, nv),
        occ_pos=take(g.occ_pos, G * nv, np.int64).reshape(G, nv),
        e_u=take(g.e_u, ne, np.int64), e_v=take(g.e_v, ne, np.int64), e_w=take(g.e_w, ne, np.int64),
        e_first=take(g.e_first, ne, np.int64), dict_ordered=True)
    ctx.lib.nts_graph_free(ctypes.byref(g))
    return out


def walk_chains(nv, e_u, e_v):
    """Components that are simple paths -> (offsets int64[n_paths+1], vertices int64[]).
    Native host helper nts_walk_chains (no GPU involved)."""
    lib = _lib.load()
    eu = np.ascontiguousarray(e_u, dtype=np.uint32)
    ev = np.ascontiguousarray(e_v, dtype=np.uint32)
    off, verts, n = _lib.c_u64p(), _lib.c_u32p(), _lib.u64()
    rc = lib.nts_walk_chains(int(nv), eu.size, eu.ctypes.data, ev.ctypes.data, ctypes.byref(off), ctypes.byref(verts),
                             ctypes.byref(n))
    if rc != 0:
        raise RuntimeError(f"nts_walk_chains failed ({rc})")
    o = np.ctypeslib.as_array(off, shape=(n.value + 1,)).astype(np.int64, copy=True)
    v = np.ctypeslib.as_array(verts, shape=(max(int(o[-1]), 1),))[:int(o[-1])].astype(np.int64, copy=True)
    lib.nts_free(off)
    lib.nts_free(verts)
    return o, v


def walk_paths(nv, e_u, e_v, e_alive=None, key=None):
    """walk_chains on the engine's own arrays (nts_walk_paths): int64 edge ends, optional liveness mask, optional
    per-vertex key that orients each path (smaller key first).  -> (offsets int64[n_paths+1], vertices int64[])."""
    lib = _lib.load()
    eu = np.ascontiguousarray(e_u, dtype=np.int64)
    ev = np.ascontiguousarray(e_v, dtype=np.int64)
    al = None if e_alive is None else np.ascontiguousarray(e_alive, dtype=np.uint8 if e_alive.dtype != np.bool_ else np.bool_)
    ky = None if key is None else np.ascontiguousarray(key, dtype=np.int64)
    if (al is not None and al.size != eu.size) or (ky is not None and ky.size != int(nv)) or ev.size != eu.size:
        raise ValueError("walk_paths: array sizes disagree")
    off, verts, n = _lib.c_u64p(), _lib.c_i64p(), _lib.u64()
    rc = lib.nts_walk_paths(int(nv), eu.size, eu.ctypes.data, ev.ctypes.data, al.ctypes.data if al is not None else None,
                            ky.ctypes.data if ky is not None else None, ctypes.byref(off), ctypes.byref(verts), ctypes.byref(n))
    if rc != 0:
        raise RuntimeError(f"nts_walk_paths failed ({rc})")
    o = np.ctypeslib.as_array(off, shape=(n.value + 1,)).astype(np.int64, copy=True)
    v = np.ctypeslib.as_array(verts, shape=(max(int(o[-1]), 1),))[:int(o[-1])].copy()
    lib.nts_free(off)
    lib.nts_free(verts)
    return o, v


walk_paths.oriented = True      # tells SyntenyEngine._paths to hand over the mask and the orientation key


def edge_degrees(nv, e_u, e_v, e_alive=None):
    """uint8[nv] live degree of every vertex, saturating at 255 (nts_edge_degrees, host threads)."""
    lib = _lib.load()
    eu = np.ascontiguousarray(e_u, dtype=np.int64)
    ev = np.ascontiguousarray(e_v, dtype=np.int64)
    al = None if e_alive is None else np.ascontiguousarray(e_alive, dtype=np.uint8 if e_alive.dtype != np.bool_ else np.bool_)
    if (al is not None and al.size != eu.size) or ev.size != eu.size:
        raise ValueError("edge_degrees: array sizes disagree")
    deg = np.empty(int(nv), np.uint8)
    rc = lib.nts_edge_degrees(int(nv), eu.size, eu.ctypes.data, ev.ctypes.data, al.ctypes.data if al is not None else None,
                              deg.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"nts_edge_degrees failed ({rc})")
    return deg


def scan_paths(v_rec, v_pos, off, verts, bp):
    """Per-path scan (nts_path_scan, host threads): (start int64[n_paths], n_up int64[G, n_paths], over bool[len(verts)])
    for paths verts[off[i]:off[i+1]] against the [G, nv] vertex tables."""
    lib = _lib.load()
    v_rec = np.ascontiguousarray(v_rec, dtype=np.int64)
    v_pos = np.ascontiguousarray(v_pos, dtype=np.int64)
    G, nv = v_rec.shape
    off = np.ascontiguousarray(off, dtype=np.uint64)
    verts = np.ascontiguousarray(verts, dtype=np.int64)
    n_paths = off.size - 1
    start = np.zeros(max(n_paths, 1), np.uint64)
    n_up = np.zeros((G, max(n_paths, 1)), np.uint64)
    over = np.zeros(max(verts.size, 1), np.uint8)
    if n_paths > 0:
        rc = lib.nts_path_scan(G, nv, v_rec.ctypes.data, v_pos.ctypes.data, n_paths, off.ctypes.data, verts.ctypes.data, int(bp),
                               start.ctypes.data, n_up.ctypes.data, over.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"nts_path_scan failed ({rc})")
    return start[:n_paths].astype(np.int64), n_up[:, :n_paths].astype(np.int64), over[:verts.size].astype(bool)
