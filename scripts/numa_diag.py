#!/usr/bin/env python3
"H2D and page-cache -> pinned rates with the process bound to each NUMA node in turn (is the ingest's box-to-box spread a placement effect?)"
import glob, os, sys, time, threading, subprocess
import numpy as np

def cpus_of(node):
    txt = open(f"/sys/devices/system/node/node{node}/cpulist").read().strip()
    out = []
    for part in txt.split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out

for f in glob.glob("/sys/class/drm/card*/device/numa_node"):
    print(f, open(f).read().strip())
nodes = sorted(int(p.rsplit("node", 1)[1]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
print("nodes", nodes, "current cpu", os.sched_getcpu() if hasattr(os, "sched_getcpu") else "?")
if len(sys.argv) > 1:
    node = int(sys.argv[1])
    os.sched_setaffinity(0, cpus_of(node))
    import torch
    n = 2 << 30
    pinned = torch.empty(n, dtype=torch.uint8).pin_memory()
    dst = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    for _ in range(2):
        torch.cuda.synchronize(); t = time.time(); dst.copy_(pinned, non_blocking=True); torch.cuda.synchronize(); dt = time.time() - t
    h2d = n / dt / 1e9
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"numa_diag_{node}.bin")
    blk = np.random.default_rng(1).integers(0, 255, size=1 << 26, dtype=np.uint8).tobytes()
    with open(path, "wb") as fh:
        for _ in range(n // len(blk)):
            fh.write(blk)
    pn = pinned.numpy()
    def rd(lo, hi):
        fd = os.open(path, os.O_RDONLY)
        for off in range(lo, hi, 8 << 20):
            os.preadv(fd, [memoryview(pn)[off:min(off + (8 << 20), hi)]], off)
        os.close(fd)
    best = 0
    for rep in range(3):
        t = time.time()
        th = [threading.Thread(target=rd, args=(i * n // 8, (i + 1) * n // 8)) for i in range(8)]
        [x.start() for x in th]; [x.join() for x in th]
        best = max(best, n / (time.time() - t) / 1e9)
    os.remove(path)
    print(f"bound to node {node}: H2D {h2d:.1f} GB/s, pread x8 {best:.1f} GB/s (file written from this node)")
else:
    for node in nodes:
        subprocess.run([sys.executable, __file__, str(node)])
