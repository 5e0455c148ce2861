#!/usr/bin/env python3
"""PCIe-inclusive sketch rate: host buffers -> nts_genome_upload -> batch -> sketch, against the
resident-input rate bench.py reports.  python scripts/pcie_rate.py [--mbp 100 --genomes 3]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbp", type=float, default=100.0)
    ap.add_argument("--genomes", type=int, default=3)
    args = ap.parse_args()
    import numpy as np
    from ntsynt_amd import synth
    from ntsynt_amd.device import BloomFilter, Context, Genome, bf_size_bytes, sketch
    ctx = Context(0)
    k, w = 24, 1000
    total = int(args.mbp * 1e6)
    anc = synth.make_ancestor(total, 4)
    host = [synth.derive_genome(anc, 0.01, j) for j in range(args.genomes)]

    def as_arrays(contigs):
        lens = np.array([c.size for c in contigs], dtype=np.uint64)
        off = np.concatenate(([0], np.cumsum(lens[:-1]))).astype(np.uint64)
        return [f"chr{i + 1}" for i in range(len(contigs))], np.concatenate(contigs), off, lens
    arrays = [as_arrays(g) for g in host]
    gs = [Genome(ctx, *a) for a in arrays]                 # warm-up + the filter
    _, nbytes = bf_size_bytes(total // 4 * 4, 0.025)
    common = BloomFilter(ctx, nbytes, k)
    common.insert(gs[0])
    tmp = BloomFilter(ctx, nbytes, k)
    for g in gs[1:]:
        tmp.clear()
        tmp.insert(g)
        common.and_(tmp)
    batch = Genome.concat(ctx, gs)
    sketch(ctx, batch, k, w, common).free()
    ctx.sync()
    bases = sum(g.total_bp for g in gs)
    reps = 5
    t_up = t_all = 0.0
    n = 0
    for _ in range(reps):
        t0 = time.perf_counter()
        fresh = [Genome(ctx, *a) for a in arrays]          # pageable host memory -> HBM, encode, valid stretches
        ctx.sync()
        t1 = time.perf_counter()
        b = Genome.concat(ctx, fresh)
        mx = sketch(ctx, b, k, w, common)
        n = len(mx)
        t2 = time.perf_counter()
        mx.free()
        b.free()
        for g in fresh:
            g.free()
        t_up += t1 - t0
        t_all += t2 - t0
    print(json.dumps({"workload": f"{args.genomes} x {args.mbp:g} Mbp from host buffers (pageable), k={k} w={w}",
                      "upload_ms": round(t_up / reps * 1e3, 2), "upload_GBs": round(bases / (t_up / reps) / 1e9, 2),
                      "upload_concat_sketch_ms": round(t_all / reps * 1e3, 2),
                      "pcie_inclusive_Gbases_s": round(bases / (t_all / reps) / 1e9, 2), "minimizers": n}))


if __name__ == "__main__":
    main()
