#!/usr/bin/env python3
"""bench.py -- minimizer-sketch throughput of the HIP hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" is one pass of the sketch (canonical ntHash + common-Bloom probe + window-of-w argmin,
rows B1-B3) over the rank's batch of synthetic genomes, which are resident in HBM (together with
the common Bloom filter) before the timed region starts -- as one resident genome whose records are
those of genome 0, then 1, ... (nts_genome_concat), so that one sequence of launches sketches the batch
(--no-batch: one sequence per genome); for N>1 the step ends with the all-gather
of the minimizer lists (SURVEY.md 8(e) exchange 2).  Weak scaling: every rank holds its own
`--genomes` genomes of one family of N*genomes genomes; the common Bloom filter is the AND over the
whole family (exchange 1, timed separately and reported under "bloom").  The family's pairwise divergence
is divergence/N, which keeps the share of k-mers the common filter accepts -- and with it the candidates
and minimizers per base each GPU handles -- at its 1-GPU value (at a fixed 1 % the AND over 24 genomes
would accept ~6 % of the k-mers and the per-GPU work would not be the 1-GPU work any more).

N=1 workload = BASELINE.json configs[1]: 3 synthetic 100 Mbp genomes at 1 % divergence, k=24 w=1000.
The timed steps use the library's default policy (exact pruning of Bloom probes, nts_pruned.inc); the
same sketch with every k-mer probed ("dense", SURVEY.md 8(d)'s 65 B/base formulation) is run after the
timed region and reported under roofline.unpruned.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                                # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
SECTOR = 64.0                                        # bytes moved per Bloom probe


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genomes", type=int, default=3, help="genomes per GPU")
    ap.add_argument("--mbp", type=float, default=100.0, help="Mbp per genome")
    ap.add_argument("--contigs", type=int, default=4)
    ap.add_argument("--divergence", type=float, default=0.01)
    ap.add_argument("-k", type=int, default=24)
    ap.add_argument("-w", type=int, default=1000)
    ap.add_argument("--fpr", type=float, default=0.025)
    ap.add_argument("--mode", choices=["auto", "dense", "pruned"], default="auto")
    ap.add_argument("--prune-c", type=int, default=0, help="0 = adaptive (library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense-leg", action="store_true")
    ap.add_argument("--no-batch", action="store_true", help="one launch sequence per genome instead of one per step")
    return ap.parse_args()


def upload(ctx, contigs):
    from ntsynt_amd.device import Genome
    lens = np.array([c.size for c in contigs], dtype=np.uint64)
    off = np.concatenate(([0], np.cumsum(lens[:-1]))).astype(np.uint64)
    return Genome(ctx, [f"chr{i + 1}" for i in range(len(contigs))], np.concatenate(contigs), off, lens)


def effective_cores():
    "host threads this process may really use: CPU count, affinity mask and the cgroup CPU quota"
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(args, contigs, bf_np):
    """The CPU oracle (a port of btllib's algorithm class: rolling ntHash, ring-buffer window
    minimum, Bloom probe per k-mer) on the box's host cores, on a bounded sample of the workload."""
    from oracle import nts_oracle as O
    try:
        O.build(native=True)
        native = True
    except Exception:
        native = False
    cores = effective_cores()
    # genome 0 cut into four records per thread (windows do not cross records, so this is the same
    # algorithm on independent pieces); repeated until >= ~8 s of wall time have elapsed
    whole = np.concatenate(contigs)
    per = max(whole.size // (4 * cores), 4 * args.w)
    seqs = [whole[i:i + per].tobytes() for i in range(0, whole.size - per + 1, per)]
    g = O.Genome([f"s{i}" for i in range(len(seqs))], seqs)
    O.minimize(g, args.k, args.w, bf_np, threads=cores, native=native)   # warm-up (page in, spawn threads)
    done, t = 0, time.time()
    while time.time() - t < 8.0:
        O.minimize(g, args.k, args.w, bf_np, threads=cores, native=native)
        done += g.total_bp
    dt = time.time() - t
    return {"value": round(done / dt / 1e9, 4), "unit": "Gbases/s", "cores": cores, "kind": "port",
            "sample": f"genome 0 ({g.total_bp / 1e6:.0f} Mbp) as {len(seqs)} records over {cores} OpenMP threads "
                      f"(the cgroup CPU quota of the box; {os.cpu_count()} logical CPUs visible), "
                      f"sketch with the same common Bloom filter, {done // g.total_bp} passes in {dt:.1f} s"}


def pmc_traffic(args, pruned_run):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r01_pmc_traffic.json:
    rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of this same command).  FETCH_SIZE is
    corrected by +1/2 of the sequence stream where the kernel reads it with wide loads (the guide's gfx950 factor)."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    default = (args.mbp, args.genomes, args.divergence, args.k, args.w, args.fpr, args.no_batch) == (100.0, 3, 0.01, 24, 1000, 0.025, False)
    if not (default and os.path.exists(path)):
        return None
    k = json.load(open(path))["kernels"].get("k_hash_select" if pruned_run else "k_hash<0>")
    if not k:
        return None
    raw = (k["fetch_MB_per_launch_max"] + k["write_MB_per_launch_max"]) * 1024 * 1024
    # the dense kernel streams the bases with 16-byte loads (FETCH_SIZE counts half of such a stream on gfx950); the
    # pruned kernel reads the 2-bit image with dword loads, to which that correction does not apply
    corr = 0.0 if pruned_run else 0.5 * args.mbp * 1e6 * args.genomes
    return {"bytes_per_launch": int(raw + corr), "raw_fetch_plus_write_bytes": int(raw),
            "source": "profiles/r01_pmc_traffic.json"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    from ntsynt_amd import dist as ndist
    from ntsynt_amd import synth
    from ntsynt_amd.device import (BloomFilter, Context, and_raw, bf_size_bytes, export_minimizers, sketch,
                                   wrap_bloom)
    # NTS_BENCH_BACKEND=gloo is a verification mode for boxes with fewer GPUs than ranks (tests/test_gpu_multirank.py):
    # ranks share the visible GPUs and the collectives run on host copies; the measured configuration is nccl (= RCCL).
    backend = os.environ.get("NTS_BENCH_BACKEND", "nccl")
    host_comm = backend != "nccl"
    if host_comm and torch.cuda.is_available():
        local_rank = local_rank % torch.cuda.device_count()
    comm_dev = "cpu" if host_comm else f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if host_comm:
            dist.init_process_group(backend)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    ctx = Context(local_rank)
    ctx.sketch_mode(args.mode, args.prune_c)
    k, w = args.k, args.w
    total_bp = int(args.mbp * 1e6)

    # ---- synthetic family: rank r owns genomes r*G .. r*G+G-1 ------------------------------------
    anc = synth.make_ancestor(total_bp, args.contigs)
    mine = list(range(rank * args.genomes, (rank + 1) * args.genomes))
    # Weak scaling keeps the work per GPU fixed: the family grows to world x G genomes, all reduced into one common
    # filter, so the pairwise divergence is divided by `world` -- the share of k-mers the common filter accepts
    # ((1 - d/2)^(k x genomes)), hence candidates and minimizers per base, stays what it is on one GPU.
    div = args.divergence / world
    host = [synth.derive_genome(anc, div, j) for j in mine]
    genomes = [upload(ctx, g) for g in host]
    bases = sum(g.total_bp for g in genomes)
    # the rank's batch as one resident genome (records of genome 0, then 1, ...): one sequence of launches sketches all
    # of it, as ntsynt_amd/pipeline.py does for assemblies of this size (GpuBackend.sketch_batch)
    from ntsynt_amd.device import Genome
    units = [Genome.concat(ctx, genomes)] if (not args.no_batch and len(genomes) > 1) else genomes

    # ---- common Bloom filter: per-genome filters, local AND, AND-all-reduce over ranks --------------
    # sized from genome 0 of the family (the lexicographically first file, cpp:105-118): same on all ranks
    _, nbytes = bf_size_bytes(total_bp // args.contigs * args.contigs, args.fpr)
    ctx.profile(True)
    t0 = time.time()
    if world > 1:
        buf = torch.zeros(ndist.padded_len(nbytes, world), dtype=torch.uint8, device=f"cuda:{local_rank}")
        torch.cuda.synchronize()
        common = wrap_bloom(ctx, buf, nbytes, k)
    else:
        common = BloomFilter(ctx, nbytes, k)
    common.insert(genomes[0])
    tmp = BloomFilter(ctx, nbytes, k)
    for g in genomes[1:]:
        tmp.clear()
        tmp.insert(g)
        common.and_(tmp)
    ctx.sync()
    t_build = time.time() - t0
    t_allreduce = 0.0
    if world > 1:
        t1 = time.time()

        def and_into(a, b):
            and_raw(ctx, a.data_ptr(), b.data_ptr(), a.numel())
            ctx.sync()
        if host_comm:
            staged = buf.cpu()
            ndist.allreduce_and(staged, lambda a, b: a.bitwise_and_(b))
            buf.copy_(staged)
        else:
            ndist.allreduce_and(buf, and_into)
        torch.cuda.synchronize()
        t_allreduce = time.time() - t1
    tmp.free()
    ins_ms, ins_n = ctx.timing("bf_insert")
    fpr_final = common.get_fpr()

    # exchange 2: the rank's lists of a step go out in one all-gather, which runs behind the next
    # step's kernels (ntsynt_amd/dist.py PackedListGather); slots sized from the minimizer density 2/(w+1)
    gatherer = None
    if world > 1:
        # density 2/(w+1) of the nominal genome size + 25 %: identical on every rank, and little padding to ship
        cap = int(2.5 * total_bp * 1.05 / (w + 1)) + 4096
        gatherer = ndist.PackedListGather(len(units), cap * (len(genomes) // len(units)), f"cuda:{local_rank}", comm_dev)

    def step():
        if gatherer is not None:
            gatherer.begin()
        n = 0
        held = []
        for i, g in enumerate(units):
            mx = sketch(ctx, g, k, w, common)
            n += len(mx)
            if gatherer is not None:
                h1p, recp, posp = gatherer.slot_ptrs(i)
                export_minimizers(ctx, mx, h1p, recp, posp, wait=False)
                gatherer.set_count(i, len(mx), mine[i])
                held.append(mx)
            else:
                mx.free()
        if gatherer is not None:
            ctx.sync()                  # the three queued copies: one wait
            for mx in held:
                mx.free()
            gatherer.post()
        return n

    def fence():
        ctx.sync()
        if gatherer is not None:
            gatherer.drain()
        if world > 1:
            dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def timed(n_warm, n_steps, level=2):
        for _ in range(n_warm):
            step()
        ctx.profile(level)              # 2: HIP events around the dominant kernel only (every pair is a bubble in the stream)
        fence()
        t_start = time.time()
        n_mx = 0
        for _ in range(n_steps):
            n_mx = step()
        fence()
        el = time.time() - t_start
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, n_mx

    dt, n_mx = timed(args.warmup, args.steps)
    names = ["hash_select", "cand_compact", "sparse_win", "gather_winners", "hash_probe", "window_min", "sort_minimizers", "merge_lists",
             "finalize"]
    tm = {n: ctx.timing(n) for n in names}
    # the other kernels of the call: one more pass, untimed, with every kernel group bracketed by events
    ctx.profile(1)
    step()
    fence()
    detail = {n: ctx.timing(n) for n in names}
    ctx.profile(2)
    cand, gaps, gap_kmers = ctx.sketch_stats()
    c_used = getattr(ctx, "last_prune_c", 0)
    per_launch_bases = bases / len(units)
    per_genome_bases = bases / len(genomes)

    def avg(n):
        return tm[n][0] / max(tm[n][1], 1)

    dense = None
    if args.mode != "dense" and not args.no_dense_leg and world == 1:
        ctx.sketch_mode("dense")
        d_dt, _ = timed(1, max(2, args.steps // 2), level=1)
        hp_ms, hp_n = ctx.timing("hash_probe")
        wm_ms, wm_n = ctx.timing("window_min")
        a_ms = hp_ms / max(hp_n, 1)
        bpb = 1.0 + SECTOR + 32.0 / (w + 1)
        ach = bpb * per_launch_bases / (a_ms * 1e-3) / 1e9
        dense = {"kernel": "k_hash<MODE_KEYS> (every k-mer probed)", "algorithmic_bytes_per_base": round(bpb, 3),
                 "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4), "avg_launch_ms": round(a_ms, 4),
                 "window_min_avg_ms": round(wm_ms / max(wm_n, 1), 4),
                 "value_Gbases_s": round(bases * max(2, args.steps // 2) / d_dt / 1e9, 3)}
        ctx.sketch_mode(args.mode, args.prune_c)

    if rank == 0:
        value = bases * world * args.steps / dt / 1e9
        pruned_run = tm["hash_select"][1] > 0
        if pruned_run:
            kern, a_ms = "k_hash_select (hash every k-mer, probe candidates only)", avg("hash_select")
            probe_frac = min(1.0, c_used / w)
            # bytes the pruned kernel has to move per base: the base (2 bits, from the packed image of the genome),
            # one sector per probed candidate, 16 B per accepted candidate written
            bpb = 0.25 + SECTOR * probe_frac + 16.0 * cand / per_launch_bases
        else:
            kern, a_ms = "k_hash<MODE_KEYS> (every k-mer probed)", avg("hash_probe")
            bpb = 1.0 + SECTOR + 32.0 / (w + 1)
        achieved = bpb * per_launch_bases / (a_ms * 1e-3) / 1e9 if a_ms > 0 else 0.0
        out = {
            "metric": "minimizer-sketch Gbases/s (sketch with common Bloom filter, inputs resident in HBM)",
            "value": round(value, 3), "unit": "Gbases/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{args.genomes} synthetic {args.mbp:g} Mbp genomes per GPU at "
                                   f"{args.divergence * 100:g}% divergence, k={k} w={w} fpr={args.fpr}"
                                   + (f" (one family of {world * args.genomes} genomes, pairwise divergence {div * 100:g}%: "
                                      f"common-filter acceptance held at the 1-GPU value)" if world > 1 else ""),
                       "sketch_mode": args.mode, "prune_c": c_used,
                       "genomes_per_gpu": args.genomes, "sketch_launch_sequences_per_step": len(units),
                       "bases_per_step_per_gpu": bases,
                       "minimizers_per_step_per_gpu": n_mx, "parallelism": f"genomes sharded over {world} GPU(s)"},
            "roofline": {"bound": "hbm", "kernel": kern,
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(args, pruned_run),
                         "algorithmic_bytes_per_base": round(bpb, 3),
                         "avg_launch_ms": round(a_ms, 4), "launches": tm["hash_select" if pruned_run else "hash_probe"][1],
                         "other_kernels_avg_ms": {n: round(detail[n][0] / detail[n][1], 4) for n in names if detail[n][1]},
                         "candidates_per_launch": cand, "uncovered_ranges": gaps, "uncovered_kmers": gap_kmers,
                         # the formulation with one sector read per k-mer moves 65.03 B/base, i.e. at
                         # most 8 TB/s / 65.03 B = 123 Gbases/s; the timed path, expressed in those bytes:
                         "one_probe_per_kmer_equivalent": {
                             "bytes_per_base": round(1.0 + SECTOR + 32.0 / (w + 1), 3),
                             "GBs": round(value * (1.0 + SECTOR + 32.0 / (w + 1)) / world, 1),
                             "frac_of_peak": round(value * (1.0 + SECTOR + 32.0 / (w + 1)) / world / HBM_PEAK_GBS, 3)},
                         "unpruned": dense},
            "bloom": {"bytes": nbytes, "build_s": round(t_build, 4), "allreduce_and_s": round(t_allreduce, 4),
                      "bf_insert_avg_ms": round(ins_ms / max(ins_n, 1), 4),
                      "bf_insert_Gbases_s": round(per_genome_bases / (ins_ms / max(ins_n, 1) * 1e-3) / 1e9, 3)
                      if ins_ms > 0 else None,
                      # SURVEY.md 8(d) prices the build at 129 B/base (sector read + write-back per k-mer)
                      "bf_insert_GBs_at_129B_per_base": round(129.0 * per_genome_bases / (ins_ms / max(ins_n, 1) * 1e-3) / 1e9, 1)
                      if ins_ms > 0 else None,
                      "occupancy": round(fpr_final, 6)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, host[0], common.to_numpy())
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
