#!/usr/bin/env python3
"""Page-cache write rate of the box's scratch file system: one file written by 1, 2, 4, 8 threads with pwrite (64 MiB
chunks), and through a shared mapping -- sizes the filter-file writer (nts_bf_save)."""
import json
import mmap
import os
import sys
import tempfile
import threading
import time

import numpy as np

GB = 1 << 30
CHUNK = 64 << 20


def run(n_threads, total, how, d):
    path = os.path.join(d, f"w_{how}_{n_threads}")
    fd = os.open(path, os.O_CREAT | os.O_TRUNC | os.O_RDWR, 0o644)
    os.ftruncate(fd, total)
    buf = np.full(CHUNK, 7, np.uint8)
    mm = mmap.mmap(fd, total) if how == "mmap" else None
    view = np.frombuffer(mm, dtype=np.uint8) if mm is not None else None
    nxt = [0]
    lock = threading.Lock()

    def work():
        while True:
            with lock:
                c = nxt[0]
                nxt[0] += 1
            off = c * CHUNK
            if off >= total:
                return
            if how == "pwrite":
                os.pwrite(fd, memoryview(buf), off)
            else:
                view[off:off + CHUNK] = buf
    t = time.time()
    th = [threading.Thread(target=work) for _ in range(n_threads)]
    [x.start() for x in th]
    [x.join() for x in th]
    dt = time.time() - t
    if mm is not None:
        del view
        mm.close()
    os.close(fd)
    os.remove(path)
    return round(total / dt / 1e9, 2)


def main():
    total = int(float(sys.argv[1]) * GB) if len(sys.argv) > 1 else 6 * GB
    d = tempfile.mkdtemp(prefix="fsw_", dir=os.environ.get("TMPDIR", "/tmp"))
    out = {"GB": total / 1e9, "dir": d, "fs": os.popen(f"df -T {d} | tail -1").read().split()[:2]}
    for how in ("pwrite", "mmap"):
        for n in (1, 2, 4, 8):
            out[f"{how}_{n}_threads_GBs"] = run(n, total, how, d)
    os.rmdir(d)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
