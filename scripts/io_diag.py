#!/usr/bin/env python3
"""Where does a slow box lose its FASTA uploads?  H2D copies from pinned memory, reads of a page-cached file into pinned memory
(1 and 8 threads), and plain host memcpy, each in GB/s."""
import os, sys, time, threading
import numpy as np
import torch

n = 2 << 30
dev = torch.device("cuda", 0)
pinned = torch.empty(n, dtype=torch.uint8).pin_memory()
dst = torch.empty(n, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for _ in range(2):
    t = time.time(); dst.copy_(pinned, non_blocking=True); torch.cuda.synchronize(); dt = time.time() - t
print(f"H2D from pinned: {n / dt / 1e9:.1f} GB/s")
path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "io_diag.bin")
buf = np.random.default_rng(1).integers(0, 255, size=1 << 26, dtype=np.uint8).tobytes()
with open(path, "wb") as fh:
    for _ in range(n // len(buf)):
        fh.write(buf)
pn = pinned.numpy()
def rd(lo, hi):
    fd = os.open(path, os.O_RDONLY)
    step = 8 << 20
    for off in range(lo, hi, step):
        m = memoryview(pn)[off:min(off + step, hi)]
        os.preadv(fd, [m], off)
    os.close(fd)
for threads in (1, 8, 16):
    for rep in range(2):
        t = time.time()
        th = [threading.Thread(target=rd, args=(i * n // threads, (i + 1) * n // threads)) for i in range(threads)]
        [x.start() for x in th]; [x.join() for x in th]
        dt = time.time() - t
    print(f"pread page cache -> pinned, {threads} threads: {n / dt / 1e9:.1f} GB/s")
a = np.empty(n, dtype=np.uint8); a[:] = 1
t = time.time(); pn[:] = a; dt = time.time() - t
print(f"host memcpy pageable -> pinned, 1 thread: {n / dt / 1e9:.1f} GB/s")
os.remove(path)
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print(open("/sys/fs/cgroup/cpu.max").read().strip())
except OSError:
    pass
