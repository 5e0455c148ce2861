#!/usr/bin/env python3
"""Human-scale run of the hot path on one MI355X (BASELINE.json configs[2]: 3 synthetic 3 Gbp genomes,
1 % divergence, k=24 w=1000).  Genomes are generated directly in HBM (nts_genome_synth), so the host never
holds them; parity is spot-checked against the CPU oracle on slices read back from the device, with the
device-built common Bloom filter, plus size-independent properties over the whole result.

  python scripts/scale_run.py --gbp 3 --genomes 3 [--no-dense] > profiles/<name>.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gbp", type=float, default=3.0)
    ap.add_argument("--genomes", type=int, default=3)
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--divergence", type=float, default=0.01)
    ap.add_argument("-k", type=int, default=24)
    ap.add_argument("-w", type=int, default=1000)
    ap.add_argument("--fpr", type=float, default=0.025)
    ap.add_argument("--no-dense", action="store_true")
    ap.add_argument("--slice-mbp", type=float, default=2.0)
    ap.add_argument("--repeats", type=int, default=3)
    args = ap.parse_args()
    from ntsynt_amd.device import BloomFilter, Context, Genome, bf_size_bytes, sketch
    from oracle import nts_oracle as O

    ctx = Context(0)
    k, w = args.k, args.w
    total = int(args.gbp * 1e9)
    out = {"workload": f"{args.genomes} x {args.gbp:g} Gbp synthetic genomes ({args.contigs} contigs), "
                       f"{args.divergence * 100:g}% divergence, k={k} w={w} fpr={args.fpr}"}
    t = time.time()
    genomes = [Genome.synth(ctx, total, args.contigs, 20240207, 1000 + j, args.divergence / 2) for j in range(args.genomes)]
    out["synth_s"] = round(time.time() - t, 3)
    bases = genomes[0].total_bp
    approx, nbytes = bf_size_bytes(bases, args.fpr)
    out["bloom_bytes"] = nbytes

    ctx.profile(True)
    t = time.time()
    common = BloomFilter(ctx, nbytes, k)
    common.insert(genomes[0])
    tmp = BloomFilter(ctx, nbytes, k)
    for g in genomes[1:]:
        tmp.clear()
        tmp.insert(g)
        common.and_(tmp)
    ctx.sync()
    out["bloom_build_s"] = round(time.time() - t, 4)
    tmp.free()
    ins_ms, ins_n = ctx.timing("bf_insert")
    and_ms, and_n = ctx.timing("bf_and")
    out["bf_insert_avg_ms"] = round(ins_ms / max(ins_n, 1), 3)
    out["bf_insert_Gbases_s"] = round(bases / (ins_ms / max(ins_n, 1) * 1e-3) / 1e9, 2)
    out["bf_and_avg_ms"] = round(and_ms / max(and_n, 1), 3)
    out["bloom_occupancy"] = round(common.get_fpr(), 6)

    def run(mode):
        ctx.sketch_mode(mode)
        res = None
        for g in genomes:                                      # warm-up: workspace allocation, per-genome run tables
            mx = sketch(ctx, g, k, w, common)
            if g is genomes[-1]:
                res = mx.to_numpy()                            # kept for the parity checks below (not timed)
            mx.free()
        ctx.profile(True)
        ctx.sync()
        t0 = time.time()
        for _ in range(args.repeats):
            for g in genomes:
                sketch(ctx, g, k, w, common).free()
        ctx.sync()
        dt = time.time() - t0
        names = ["hash_select", "cand_compact", "sparse_win", "gather_winners", "hash_probe", "window_min", "sort_minimizers", "merge_lists", "finalize"]
        tm = {n: ctx.timing(n) for n in names}
        return {"Gbases_s": round(bases * len(genomes) * args.repeats / dt / 1e9, 2),
                "ms_per_genome": round(dt / (len(genomes) * args.repeats) * 1e3, 3),
                "kernel_avg_ms": {n: round(v[0] / v[1], 3) for n, v in tm.items() if v[1]},
                "sketch_stats": ctx.sketch_stats()}, res

    out["pruned"], res_p = run("pruned")
    if not args.no_dense:
        out["dense"], res_d = run("dense")
        out["dense_equals_pruned"] = bool(all(np.array_equal(a, b) for a, b in zip(res_p, res_d)))
        hp = out["dense"]["kernel_avg_ms"].get("hash_probe")
        if hp:
            out["dense"]["hash_probe_GBs_at_65B_per_base"] = round(65.03 * bases / (hp * 1e-3) / 1e9, 1)
    ctx.sketch_mode("auto")

    # size-independent properties of the last genome's sketch
    h1, rec, pos = res_p
    out["minimizers_last_genome"] = int(h1.size)
    per = genomes[-1].rec_len[0]
    ok = True
    for r in range(args.contigs):
        p = pos[rec == r].astype(np.int64)
        ok &= bool((np.diff(p) > 0).all()) and int(np.diff(p).max()) <= 50 * w and int(p[0]) < 50 * w
    out["positions_sorted_per_record"] = ok
    # slice parity vs the oracle with the device-built filter
    t = time.time()
    bf_np = common.to_numpy()
    out["bloom_download_s"] = round(time.time() - t, 2)
    n_slice = int(args.slice_mbp * 1e6)
    checks = []
    g = genomes[-1]
    for r in (0, args.contigs // 2, args.contigs - 1):
        off = int(g.rec_off[r])
        seq = g.download(off, n_slice).tobytes()
        exp = O.minimize(O.Genome(["s"], [seq]), k, w, bf_np, threads=1)[0]
        m = (rec == r) & (pos < n_slice - k - w)
        n = int(m.sum())
        checks.append(bool(n > 100 and np.array_equal(pos[m], exp[1][:n]) and np.array_equal(h1[m], exp[0][:n])))
    out["oracle_slice_parity"] = checks
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
