"""Sketches of different genomes on separate contexts (streams) from separate host threads vs one after the other:
does the GPU overlap one genome's latency-bound kernels (compaction, windows, gather) with another's rolling?"""
import sys, time, os, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd.device import Context, Genome, BloomFilter, bf_size_bytes, sketch

n_g = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctxs = [Context(0) for _ in range(n_g)]
main = ctxs[0]
genomes = [Genome.synth(main, 3_000_000_000, 24, 20240207, 1000 + i, 0.005) for i in range(n_g)]
_, nb = bf_size_bytes(genomes[0].total_bp, 0.025)
bf = BloomFilter(main, nb, 24)
bf.insert(genomes[0])
tmp = BloomFilter(main, nb, 24)
for g in genomes[1:]:
    tmp.clear(); tmp.insert(g); bf.and_(tmp)
tmp.free()
main.sync()

def one(ctx, g, out, i):
    mx = sketch(ctx, g, 24, 1000, bf)
    out[i] = len(mx)
    mx.free()

def sequential(reps):
    out = [0] * n_g
    for _ in range(reps):
        for i, g in enumerate(genomes):
            one(main, g, out, i)
    main.sync()
    return out

def parallel(reps, n_ctx):
    out = [0] * n_g
    def worker(c):
        for _ in range(reps):
            for i in range(c, n_g, n_ctx):
                one(ctxs[c], genomes[i], out, i)
        ctxs[c].sync()
    th = [threading.Thread(target=worker, args=(c,)) for c in range(n_ctx)]
    for t in th: t.start()
    for t in th: t.join()
    return out

sequential(2); parallel(2, n_g)
for name, fn in (("sequential", lambda: sequential(5)), ("2 contexts", lambda: parallel(5, 2)), (f"{n_g} contexts", lambda: parallel(5, n_g)), ("sequential", lambda: sequential(5))):
    t = time.time(); out = fn(); dt = time.time() - t
    print(name, round(dt / 5 * 1e3, 2), "ms per step", round(n_g * 3.0 * 5 / dt, 1), "Gbases/s", out, flush=True)
