#!/usr/bin/env python3
"""How this box's driver charges for device memory: hipMalloc / hipFree wall time by size and by count, first touch (a memset over
the fresh allocation), and whether a second allocation of memory just freed is cheaper.  Round 5's driver box took 0.84 s between
`bf_allocated` and `bf_first_insert` (24 GB of build workspaces) where builder boxes take 0.05 s; this separates per-call latency from
per-byte cost.  Talks to libamdhip64 directly (no product code)."""
import ctypes
import json
import sys
import time

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]


def timed(fn):
    t0 = time.perf_counter()
    r = fn()
    return r, (time.perf_counter() - t0) * 1e3


def malloc(n):
    p = ctypes.c_void_p()
    rc, ms = timed(lambda: hip.hipMalloc(ctypes.byref(p), n))
    assert rc == 0, rc
    return p, ms


def main():
    out = {"by_size": [], "many_small": None, "again": []}
    hip.hipSetDevice(0)
    hip.hipDeviceSynchronize()
    malloc(1 << 20)                                        # runtime start-up is not what is measured
    for gb in (0.25, 1, 4, 12, 24, 48):
        n = int(gb * (1 << 30))
        p, ms_a = malloc(n)
        _, ms_touch = timed(lambda: (hip.hipMemset(p, 0, n), hip.hipDeviceSynchronize()))
        _, ms_touch2 = timed(lambda: (hip.hipMemset(p, 0, n), hip.hipDeviceSynchronize()))
        _, ms_f = timed(lambda: hip.hipFree(p))
        out["by_size"].append({"GB": gb, "hipMalloc_ms": round(ms_a, 2), "first_memset_ms": round(ms_touch, 2),
                               "second_memset_ms": round(ms_touch2, 2), "hipFree_ms": round(ms_f, 2)})
    # 48 allocations of 0.5 GB against one of 24 GB
    ps, t = [], 0.0
    each = []
    for _ in range(48):
        p, ms = malloc(1 << 29)
        ps.append(p)
        t += ms
        each.append(round(ms, 2))
    tf = 0.0
    for p in ps:
        _, ms = timed(lambda: hip.hipFree(p))
        tf += ms
    out["many_small"] = {"count": 48, "GB_each": 0.5, "hipMalloc_ms_total": round(t, 2), "hipFree_ms_total": round(tf, 2), "max_ms": max(each),
                         "first_eight_ms": each[:8]}
    # the same 24 GB three times in a row: is memory just freed cheaper to get back?
    for _ in range(3):
        p, ms_a = malloc(24 << 30)
        _, ms_f = timed(lambda: hip.hipFree(p))
        out["again"].append({"hipMalloc_ms": round(ms_a, 2), "hipFree_ms": round(ms_f, 2)})
    # an allocation on a second thread while this one runs memsets: does hipMalloc stall the stream?
    import threading
    p0, _ = malloc(4 << 30)
    res = {}

    def bg():
        q, ms = malloc(24 << 30)
        res["ms"] = ms
        res["p"] = q
    th = threading.Thread(target=bg)
    t0 = time.perf_counter()
    th.start()
    n_sets = 0
    while th.is_alive():
        hip.hipMemset(p0, 0, 4 << 30)
        hip.hipDeviceSynchronize()
        n_sets += 1
    th.join()
    wall = (time.perf_counter() - t0) * 1e3
    _, one = timed(lambda: (hip.hipMemset(p0, 0, 4 << 30), hip.hipDeviceSynchronize()))
    out["concurrent"] = {"hipMalloc_24GB_on_second_thread_ms": round(res["ms"], 2), "memsets_of_4GB_finished_meanwhile": n_sets,
                         "wall_ms": round(wall, 2), "one_memset_alone_ms": round(one, 2)}
    hip.hipFree(res["p"])
    hip.hipFree(p0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
