#!/usr/bin/env python3
"""bench.py -- minimizer-sketch throughput of the HIP hot path on MI355X, on BASELINE.json's configurations.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the sketch (canonical ntHash + common-Bloom probe + window-of-w argmin, rows B1-B3) over the
rank's synthetic genomes, which are resident in HBM together with the common Bloom filter before the timed region
starts (generated there: nts_genome_synth_plan).  `value` = bases sketched by all ranks per second.

The families are SURVEY.md 8(d)'s: an i.i.d. ancestor in contigs, genome j = the ancestor with substitutions at p/2 plus a fixed
set of structural events of its own (5 inversions of 1-5 Mbp, 2 inter-contig translocations, 20 indels of 1-60 kbp, 60 small
rearrangements, 200 indels of 1-50 bp: ntsynt_amd/synth.py structural_plan), so that the graph stage of the end-to-end leg has indel cuts, erosions and merges to make
(--substitutions-only gives round 2's family).  `nruns` is the sketch on the variant with 0.5 % of the bases in N runs.

`python bench.py --gpus N` on its own starts the N ranks itself (torch.distributed.run on 127.0.0.1); under a launcher WORLD_SIZE must equal
--gpus or the run refuses.  ONE workload along the scaling curve: c3 is `value` at every N, c4 is the `c4` leg of the same line at every N.

Workloads (--workload; default c3):
  c3   BASELINE configs[2], the configuration the metric is quoted on: 3 synthetic 3 Gbp genomes (24 contigs) at 1 %
       divergence, k=24 w=1000 fpr=0.025.  Fits one GPU (9 GB of bases + 2 x 14.8 GB of filters).  At N > 1 the family's 72 records
       are shared out over the ranks by bases, across genome boundaries (ntsynt_amd.pipeline.partition_plan: 8 x ~1.125 Gbp at N = 8);
       a rank builds a filter per genome its range touches, exchange 1 is nts_bf_allreduce_parts (OR over a genome's parts, AND across
       genomes), every step ends with the all-gather of the part lists (nts_mx_allgather_ex) strung together per genome.
  c4   BASELINE configs[3]: 8 synthetic 3 Gbp genomes at 10 % divergence, genome g on rank g mod N (one per GPU at
       N=8), common filter = AND over all eight (exchange 1: nts_bf_allreduce_and over RCCL), every step ends with the
       all-gather of the minimizer lists (exchange 2: nts_mx_allgather).  The family, the filter and therefore the work
       per genome are the same at every N: strong scaling.  At N=1 all eight genomes live on one GPU.
  c2   BASELINE configs[1]: 3 x 100 Mbp at 1 %, sketched as one batch genome (round 1's bench line).
Extra legs at N=1 (rank 0), all inside the one JSON line: `cold` (first sketch of a fresh genome, three times, split into allocation
and the rest), `roofline.unpruned` (every k-mer probed), `valley` (filters that accept a few per cent of the k-mers: 3 x 3 Gbp at
10 %, 8 x 3 Gbp at 4 %), `e2e` (FASTA files on disk -> final synteny TSV through the product
pipeline, per stage), `cpu_baseline` (the CPU oracle on the host cores, per stage, at the reference's default
parallelism and on all cores).
"""
import argparse
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                                # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
SECTOR = 64.0                                        # bytes per Bloom probe in SURVEY.md 8(d)'s accounting
L2_LINE = 128.0                                      # what one missing probe moves if the L2 fetches whole lines
ANCESTOR_SEED = 20240207
WORKLOADS = {
    # name: (genomes in the family, Mbp per genome, contigs, pairwise divergence, scaling)
    "c3": (3, 3000.0, 24, 0.01, "strong"),
    "c4": (8, 3000.0, 24, 0.10, "strong"),
    "c2": (3, 100.0, 4, 0.01, "strong"),
}
SKETCH_KERNELS = ["hash_select", "cand_compact", "sparse_win", "gather_winners", "hash_probe", "window_min", "sort_minimizers",
                  "merge_lists", "finalize"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None)
    ap.add_argument("--genomes", type=int, default=None, help="override: genomes in the family")
    ap.add_argument("--mbp", type=float, default=None, help="override: Mbp per genome")
    ap.add_argument("--contigs", type=int, default=None)
    ap.add_argument("--divergence", type=float, default=None)
    ap.add_argument("-k", type=int, default=24)
    ap.add_argument("-w", type=int, default=1000)
    ap.add_argument("--fpr", type=float, default=0.025)
    ap.add_argument("--mode", choices=["auto", "dense", "pruned"], default="auto")
    ap.add_argument("--prune-c", type=int, default=0, help="0 = adaptive (library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense-leg", action="store_true")
    ap.add_argument("--no-cold-leg", action="store_true")
    ap.add_argument("--no-c4-leg", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-batch", action="store_true", help="c2: one launch sequence per genome instead of one per step")
    ap.add_argument("--no-nruns-leg", action="store_true")
    ap.add_argument("--at-once", action="store_true", help="the timed steps sketch the genomes of a GPU at once (device.SketchPool) "
                                                           "instead of one after the other (an experiment: DESIGN.md section 8)")
    ap.add_argument("--substitutions-only", action="store_true", help="genomes differ by substitutions only (round 2's family)")
    ap.add_argument("--family", choices=["structural", "assembly-like"], default="structural",
                    help="assembly-like: ntsynt_amd.synth.realistic_plan + REPEATS (interspersed repeat families, satellite arrays, segmental "
                         "duplications, thousands of scaffolds with a tail of short ones, N gaps) -- the family of the c5_like leg, for every leg")
    ap.add_argument("--no-c5-leg", action="store_true")
    ap.add_argument("--no-valley-leg", action="store_true")
    ap.add_argument("--e2e-dir", default=None, help="where the e2e leg writes its FASTA files [a temp dir]")
    return ap.parse_args()


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_e2e_sample(k, w, fpr, slices, params, threads):
    """The oracle pipeline end to end on a bounded sample (size pass, Bloom build, sketch, graph stage, refinement
    rounds, each timed, inputs in memory-backed files): the first bases of every genome of the family, one FASTA record each, with
    the parameters the product's e2e leg runs with.  Sketch and Bloom build on `threads` threads, the graph stage single-threaded
    Python like the reference's."""
    from oracle import nts_oracle as O
    from oracle import synteny_oracle as SO
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix="nts_cpu_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    t = {}
    try:
        paths = []
        for j, a in enumerate(slices):
            p = os.path.join(tmp, f"syn{j}.fa")
            with open(p, "wb") as fh:              # six records: the reference parallelises over records (cpp:128,145; indexlr -t)
                per = (a.size + 5) // 6
                for r in range(6):
                    fh.write(b">chr%d\n" % (r + 1))
                    fh.write(a[r * per:(r + 1) * per].tobytes())
                    fh.write(b"\n")
            paths.append(p)
        os.chdir(tmp)
        t0 = time.time()
        first_bp = O.read_fasta(sorted(paths)[0]).total_bp                 # approximate_bf_size's own pass over the first file
        t["size_pass"] = time.time() - t0
        t0 = time.time()
        genomes = {p: O.read_fasta(p) for p in paths}
        t["read_fasta"] = time.time() - t0
        t0 = time.time()
        bf = O.common_bf(genomes, k, fpr, threads)
        t["make_common_bf"] = time.time() - t0
        t0 = time.time()
        tables, by_tsv = {}, {}
        n_mx = 0
        for p in paths:
            tsv = f"{os.path.basename(p)}.k{k}.w{w}.tsv"
            mins = O.minimize(genomes[p], k, w, bf, threads)
            n_mx += sum(len(m[0]) for m in mins)
            tables[tsv] = SO.mx_tables_from_tokens(SO.mx_records_from_arrays(genomes[p].names, mins))
            by_tsv[tsv] = genomes[p]
        t["indexlr"] = time.time() - t0
        eng = SO.SyntenyOracle(list(tables), by_tsv, k, w, params["w_rounds"], params["indel"], params["merge"], params["block_size"], "cpu",
                               bf=bf, threads=threads)
        eng.load(tables)
        inner = eng.refine
        spent = {}

        def timed_refine(blocks):
            t1 = time.time()
            out = inner(blocks)
            spent["refine"] = time.time() - t1
            return out
        eng.refine = timed_refine
        t0 = time.time()
        eng.main()
        t["ntsynt_synteny"] = time.time() - t0
        t["of_which_refinement_rounds"] = spent.get("refine", 0.0)
        blocks = len(eng.outputs["cpu.synteny_blocks.tsv"].splitlines()) // len(paths)
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)
    total = sum(v for n, v in t.items() if n != "of_which_refinement_rounds")
    bases = sum(int(a.size) for a in slices)
    return {"what": f"oracle pipeline on the first {slices[0].size / 1e6:.0f} Mbp of each of the {len(slices)} genomes (cut into six records each), "
                    f"ntSynt's parameters of the e2e leg, {threads} threads for Bloom build and sketch, graph stage single-threaded",
            "first_file_bp": int(first_bp), "seconds": round(total, 2), "stages_s": {n: round(v, 3) for n, v in t.items()},
            "Gbases_s_end_to_end": round(bases / total / 1e9, 4), "minimizers": n_mx, "blocks": blocks,
            # ru_maxrss is a high-water mark of the whole process: what stands here includes the earlier legs' host buffers
            "process_peak_host_rss_bytes_after_this_leg": __import__("resource").getrusage(0).ru_maxrss * 1024}


def cpu_baseline(k, w, fpr, sample, bf_np, budget_s=20.0, e2e_slices=None, e2e_par=None):
    """The CPU oracle (a port of btllib's algorithm class: rolling ntHash, ring-buffer window minimum, byte-atomic Bloom
    insert, one probe per k-mer) on this box's host cores, per stage: Bloom build and sketch, each at the
    reference's default parallelism -- make_common_bf with 12 threads (bin/ntSynt:59), indexlr 5 threads x 2 genomes at
    once (bin/ntsynt_run_pipeline.smk:79-80, bin/ntSynt:154) -- and on all cores the cgroup grants.  `sample` = the
    first bases of genome 0 read back from HBM (uint8 ASCII); the filter is the real common filter (so the probes miss
    the caches like they do in a full run); the Bloom build of the sample goes into a filter sized for the sample by the
    reference's rule.  Bounded to ~budget_s seconds of CPU work."""
    from oracle import nts_oracle as O
    try:
        O.build(native=True)
        native = True
    except Exception:
        native = False
    cores = effective_cores()

    def as_records(n_threads):
        per = max(sample.size // (4 * n_threads), 4 * w)
        seqs = [sample[i:i + per].tobytes() for i in range(0, sample.size - per + 1, per)]
        return O.Genome([f"s{i}" for i in range(len(seqs))], seqs)

    def rate(fn, g, slot_s):
        fn()                                                         # warm-up: page in, spawn threads
        done, t = 0, time.time()
        while True:
            fn()
            done += g.total_bp
            if time.time() - t >= slot_s:
                break
        return done / (time.time() - t) / 1e9

    stages = {}
    slot = budget_s / 8.0
    build_bytes = O.bf_ctor_bytes(O.bf_approx_bytes(sample.size, fpr))
    for label, n_thr in (("reference_default", None), ("all_cores", cores)):
        thr_bf = min(12, cores) if n_thr is None else n_thr
        thr_sk = min(10, cores) if n_thr is None else n_thr
        g_bf, g_sk = as_records(thr_bf), as_records(thr_sk)
        stages[label] = {
            "bf_build_Gbases_s": round(rate(lambda: O.bf_build(g_bf, k, build_bytes, None, thr_bf, native), g_bf, slot), 4),
            "bf_build_threads": thr_bf,
            "sketch_Gbases_s": round(rate(lambda: O.minimize(g_sk, k, w, bf_np, threads=thr_sk, native=native), g_sk, slot), 4),
            "sketch_threads": thr_sk,
        }
    if e2e_slices:
        stages["end_to_end_sample"] = cpu_e2e_sample(k, w, fpr, e2e_slices, e2e_par, cores)
        for path in ORACLE_RECORDS:            # and the full-size runs on record (scripts/e2e_oracle_check.py)
            try:
                rec = json.load(open(path))
                stages.setdefault("end_to_end_full_size_on_record", []).append(
                    {"seconds": rec["oracle_seconds"], "threads": rec["oracle_threads"], "key": rec["key"], "peak_host_rss_bytes": rec.get("oracle_peak_rss_bytes"),
                     "source": os.path.relpath(path, ROOT)})
            except (OSError, ValueError, KeyError):
                pass
    return {"value": stages["all_cores"]["sketch_Gbases_s"], "unit": "Gbases/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(), "logical_cpus_visible": os.cpu_count(), "stages": stages,
            "sample": f"first {sample.size / 1e6:.0f} Mbp of genome 0 read back from HBM, cut into 4 records per thread "
                      f"(windows do not cross records), sketched against the real {bf_np.size / 1e9:.1f} GB common filter; "
                      f"Bloom build of the same sample into a filter sized for it ({build_bytes / 1e6:.0f} MB); "
                      f"~{budget_s:.0f} s of CPU work in all; "
                      f"`value` = sketch on all {cores} cores the cgroup grants"}


def family_genome(ctx, args, total_bp, contigs, j, rate, n_runs=False):
    "genome j of the bench family, generated in HBM"
    from ntsynt_amd import synth
    from ntsynt_amd.device import Genome
    if getattr(args, "family", "structural") == "assembly-like":
        # NTS_FAM_VARIANT (experiments: which trait of the family costs what): "nosat" satellite arrays of 3 kbp, "noscaf" chromosomes
        # in one piece without gaps, "norep" no interspersed repeat families
        var = os.environ.get("NTS_FAM_VARIANT", "").split(",")
        kw = {}
        if "nosat" in var:
            kw["sat_scale"] = 1e-9
        if "noscaf" in var:
            kw.update(n_scaffolds=contigs, n_tail=0, n_gaps=0)
        if "notail" in var:
            kw.update(n_tail=0)
        if "nogaps" in var:
            kw.update(n_gaps=0)
        if "nocuts" in var:
            kw.update(n_scaffolds=contigs)
        plan = synth.realistic_plan(contigs, int(total_bp) // contigs, j, ANCESTOR_SEED, **kw)
        rep = dict(synth.REPEATS)
        if "norep" in var:
            rep["sine_prob_256"] = rep["line_prob_256"] = 0
        return Genome.synth_plan(ctx, plan, ANCESTOR_SEED, 1000 + j, rate, rep=rep, names=plan[2])
    if args.substitutions_only and not n_runs:
        return Genome.synth(ctx, total_bp, contigs, ANCESTOR_SEED, 1000 + j, rate)
    plan = synth.structural_plan(contigs, int(total_bp) // contigs, j, ANCESTOR_SEED, n_runs=n_runs,
                                 **({"inversions": 0, "translocations": 0, "indels": 0, "micro": 0, "small_indels": 0}
                                    if args.substitutions_only else {}))
    return Genome.synth_plan(ctx, plan, ANCESTOR_SEED, 1000 + j, rate)


def family_bases(args, n_fam, total_bp, contigs):
    "bases of every genome of the family (the plans are cheap: no device work)"
    from ntsynt_amd import synth
    if getattr(args, "family", "structural") == "assembly-like":
        return [int(synth.realistic_plan(contigs, int(total_bp) // contigs, j, ANCESTOR_SEED)[0].sum()) for j in range(n_fam)]
    if args.substitutions_only:
        return [int(total_bp) // contigs * contigs] * n_fam
    return [int(synth.structural_plan(contigs, int(total_bp) // contigs, j, ANCESTOR_SEED)[0].sum()) for j in range(n_fam)]


def write_fasta_from_device(g, path, chunk=1 << 28, soft_mask_seed=None, half_lower=False, line_width=0):
    """resident genome -> FASTA file (inputs of the e2e leg; not timed), single-line unless line_width.  soft_mask_seed: lower-case
    stretches (10-20,000 bases, about one per 250 kbp) as in SURVEY.md 8(d)'s soft-masked variant -- the parse has to fold them;
    half_lower: stretches of 50-3,000 bases alternately lower and upper case, about half of the bases each (a RepeatMasker-style
    soft-masked assembly; needs soft_mask_seed)."""
    rng = np.random.default_rng(soft_mask_seed) if soft_mask_seed is not None else None
    with open(path, "wb") as fh:
        for r, name in enumerate(g.names):
            fh.write(b">" + name.encode() + b"\n")
            off, ln = int(g.rec_off[r]), int(g.rec_len[r])
            col = 0
            for s in range(0, ln, chunk):
                part = g.download(off + s, min(chunk, ln - s))
                if rng is not None and half_lower:
                    ends = np.cumsum(rng.integers(50, 3001, size=part.size // 1500 + 2))
                    ends = ends[ends < part.size]
                    flips = np.zeros(part.size + 1, dtype=np.int8)
                    flips[ends] = 1
                    lower = (np.cumsum(flips[:-1]) + int(rng.integers(0, 2))) & 1
                    part[lower.astype(bool)] |= 0x20                            # ('N' becomes 'n': still invalid)
                elif rng is not None and part.size > 40000:
                    for st in rng.integers(0, part.size - 20000, size=max(1, part.size // 250000)):
                        part[st:st + int(rng.integers(10, 20000))] |= 0x20      # ('N' would become 'n': still invalid)
                if line_width:
                    at = 0
                    while at < part.size:
                        take = min(line_width - col, part.size - at)
                        fh.write(part[at:at + take].tobytes())
                        at += take
                        col += take
                        if col == line_width:
                            fh.write(b"\n")
                            col = 0
                else:
                    fh.write(part.tobytes())
            if not line_width or col:
                fh.write(b"\n")


ORACLE_RECORDS = [os.path.join(ROOT, "profiles", n) for n in ("r04_e2e_oracle_c3.json", "r04_e2e_oracle_c5_like.json", "r04_e2e_oracle_2x3Gbp_d0.1.json", "r04_e2e_oracle_c4.json", "r03_e2e_oracle.json")]   # scripts/e2e_oracle_check.py


def oracle_record(key):
    "the full-size oracle run on record for this family + parameter set (None: none)"
    for path in ORACLE_RECORDS:
        try:
            rec = json.load(open(path))
            if rec.get("key") == key:
                return rec, os.path.relpath(path, ROOT)
        except (OSError, ValueError):
            continue
    return None, None


def mx_digest(h1, rec, pos):
    """Order-independent digest of a minimizer list (count, XOR of the printed hashes, a multiply-mixed sum of hash, position and
    record modulo 2^64): what the full-size oracle runs leave on record (scripts/e2e_oracle_check.py) and tests/test_gpu_scale.py
    recomputes from the HIP path's lists -- the whole output at headline size, not slices of it."""
    h1 = np.ascontiguousarray(h1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        mix = h1 * (np.asarray(pos, dtype=np.uint64) * np.uint64(2) + np.uint64(1)) + np.asarray(rec, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        return {"n": int(h1.size), "xor_h1": int(np.bitwise_xor.reduce(h1)) if h1.size else 0, "sum_mix": int(mix.sum(dtype=np.uint64)) if h1.size else 0}


def e2e_key(args, n_fam, total_bp, contigs, div):
    "identity of an e2e family + parameter set (the oracle's recorded md5 applies to exactly this)"
    fam = "substitutions-only" if args.substitutions_only else "structural+softmask"
    if getattr(args, "family", "structural") == "assembly-like":
        fam = "assembly-like(repeat families+satellites+segdups+scaffold tail+N gaps)+half-lower"
    return f"{n_fam}x{total_bp}bp/{contigs}contigs/div{div:g}/k{args.k}/w{args.w}/fpr{args.fpr}/seed{ANCESTOR_SEED}/{fam}"


def e2e_inputs(args, device, n_fam, total_bp, contigs, div, workdir):
    "the e2e leg's FASTA files, written from genomes generated in HBM (same family as the sketch legs)"
    from ntsynt_amd.device import Context
    paths = []
    ctx = Context(device)
    for j in range(n_fam):
        g = family_genome(ctx, args, total_bp, contigs, j, div / 2.0)
        p = os.path.join(workdir, f"syn{j}.fa")
        if getattr(args, "family", "structural") == "assembly-like":
            write_fasta_from_device(g, p, soft_mask_seed=4000 + j, half_lower=True)
        else:
            write_fasta_from_device(g, p, soft_mask_seed=None if args.substitutions_only else 4000 + j)
        g.free()
        paths.append(p)
    ctx.close()
    return paths


def e2e_params(args, paths, div):
    "ntSynt's own defaults for the divergence (bin/ntSynt:89-99), resolved by the product's CLI code"
    from ntsynt_amd import cli
    divergence_pct = max(div * 100, 0.01)
    parser = cli.build_parser()
    a = parser.parse_args(paths + ["-d", f"{divergence_pct:g}", "-p", "e2e", "-k", str(args.k), "-w", str(args.w), "--fpr", str(args.fpr)])
    cli.resolve(parser, a)
    return a, divergence_pct


def e2e_leg(args, device, n_fam, total_bp, contigs, div, workdir, paths=None):
    """FASTA files on disk -> {prefix}.synteny_blocks.tsv through ntsynt_amd.pipeline.run with ntSynt's defaults for the
    divergence (the second half of BASELINE's metric).  The files are written from genomes generated in HBM first (same
    family as the sketch legs; not timed)."""
    from ntsynt_amd import pipeline
    t = time.time()
    if paths is None:
        paths = e2e_inputs(args, device, n_fam, total_bp, contigs, div, workdir)
    t_write = time.time() - t
    if os.environ.get("BENCH_E2E_INPROCESS") != "1" and not getattr(args, "e2e_is_child", False):
        # A user's run is a process of its own: so is this one (python bench.py --e2e-child <spec>: the timed region is pipeline.run
        # inside it, as before -- with the code object loaded and every allocation made there, nothing inherited from the legs before.
        # Round 5's first form ran it here after emptying the library's allocation cache; on boxes whose driver clears freed memory
        # lazily the run then waited for 60-100 GB of that: 3 s in front of the first insert, which no user's run sees.)
        import subprocess
        child_args = {k_: v_ for k_, v_ in vars(args).items() if isinstance(v_, (int, float, str, bool, type(None)))}
        spec = {"args": child_args, "device": device,
                "n_fam": n_fam, "total_bp": total_bp, "contigs": contigs, "div": div, "workdir": workdir, "paths": paths}
        spec_path = os.path.join(workdir, "e2e_child.json")
        with open(spec_path, "w") as fh:
            json.dump(spec, fh)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--e2e-child", spec_path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800)
        lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("E2E_CHILD_RESULT ")]
        if r.returncode != 0 or not lines:
            raise RuntimeError("the end-to-end run's process failed: " + r.stderr.decode()[-2000:])
        out = json.loads(lines[-1][len("E2E_CHILD_RESULT "):])
        out["write_inputs_s"] = round(t_write, 1)
        out["process"] = "a process of its own (python bench.py --e2e-child): nothing inherited from the legs before it"
        return out
    a, divergence_pct = e2e_params(args, paths, div)
    cache_released = 0
    from ntsynt_amd import _lib
    from ntsynt_amd.device import mem_events, mem_events_since
    ev0 = mem_events(_lib.load())
    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        t = time.time()
        eng = pipeline.run(paths, k=a.k, w=a.w, fpr=a.fpr, prefix=a.prefix, w_rounds=a.w_rounds, indel=a.indel, merge=a.merge,
                           block_size=a.block_size, device=device, log=lambda *x: None)
        wall = time.time() - t
    finally:
        os.chdir(cwd)
    tsv = eng.outputs["e2e.synteny_blocks.tsv"]
    extra = {}
    if getattr(eng, "times", None):                                  # NTS_ENGINE_TIMES=1: wall clock per engine step
        extra["engine_times_s"] = {n: round(v, 3) for n, v in sorted(eng.times.items(), key=lambda kv: -kv[1])}
    if getattr(eng, "stage_marks", None):
        extra["time_line_s"] = dict(eng.stage_marks)
    md5 = hashlib.md5(tsv.encode()).hexdigest()
    stages = {n: round(s, 3) for n, s in eng.stage_times}
    # the oracle pipeline's output for this very family, recorded once on the GPU box's host cores (tens of minutes of CPU):
    # scripts/e2e_oracle_check.py -> profiles/r03_e2e_oracle.json
    oracle = {"oracle_md5": None, "oracle_checked": "no record for this family (scripts/e2e_oracle_check.py makes one)"}
    rec, rec_path = oracle_record(e2e_key(args, n_fam, total_bp, contigs, div))
    if rec is not None:
        oracle = {"oracle_md5": rec["oracle_md5"], "oracle_checked": "identical" if rec["oracle_md5"] == md5 else "DIFFERENT",
                  "oracle_record": rec_path, "oracle_seconds_on_record": rec.get("oracle_seconds"), "oracle_threads_on_record": rec.get("oracle_threads")}
    return {**extra, "graph_stage": type(eng).__name__, "engine_stats": eng.stats,
            "what": f"{len(paths)} FASTA files on disk -> final synteny TSV (ntSynt -d {divergence_pct:g}: w_rounds {a.w_rounds}, "
                    f"indel {a.indel}, merge {a.merge}, block {a.block_size}), one GPU, files in the page cache",
            "seconds": round(wall, 3), "stages_s": stages,
            # the run ends when the 14.8 GB filter file is on disk: that artefact's share of the wall clock
            "share_waiting_for_the_bf_file": round(stages.get("wait_for_files", 0.0) / wall, 3) if wall > 0 else None,
            "blocks": len(tsv.splitlines()) // len(paths), "tsv_md5": md5, **oracle,
            # the reference's other published figure (README.md:156-158: "34 GB / 32 GB" peak memory, host RAM there): the library's
            # HBM high-water mark over the run (nts_mem_stats: every device allocation of every context) and this process's peak host
            # RSS (ru_maxrss: since process start, so the sketch legs' host buffers are in it)
            "peak_hbm_bytes": (eng.memory or {}).get("peak_hbm_bytes"), "peak_host_rss_bytes": (eng.memory or {}).get("peak_host_rss_bytes"),
            "hbm_live_at_marks_GB": {n: round(v / 1e9, 2) for n, v in ((eng.memory or {}).get("hbm_live_at_marks") or {}).items()},
            "allocation_cache_released_before_the_run_GB": round(cache_released / 1e9, 2),
            # what the run asked of the driver (one reserve call sized from the files, ntsynt_amd.pipeline.reserve_for_run; anything
            # else is a request the plan did not cover) and what other processes held of the GPU's memory when it started
            "allocator": dict(mem_events_since(_lib.load(), ev0), reserved_GB=round(getattr(eng, "reserved_bytes", 0) / 1e9, 2)),
            "device_memory_in_use_by_others_at_start_GB": getattr(args, "device_used_by_parent_GB", None),
            "GB_the_bench_process_gave_back_before_this_run": getattr(args, "parent_freed_GB", None),
            "write_inputs_s": round(t_write, 1)}


def c5_like_leg(args, ctx, device, total_bp, contigs, workdir):
    """BASELINE configs[4] stands on three real mammalian assemblies (human / chimp / bonobo, -d 1.3; reference README.md:157), which are
    not in the container.  This leg runs its parameter set on the assembly-like synthetic family instead (ntsynt_amd/synth.py
    realistic_plan + REPEATS: interspersed repeat families in ~10^6 copies at 1-20 % from their consensus, satellite arrays, segmental
    duplications, 600 / 1500 / 4000 scaffolds per genome + as many short ones, N gaps, half of the bases lower case in the files):
    Bloom build, sketch, and FASTA files -> final TSV, with the paths an i.i.d. family never takes counted."""
    import copy
    ev5 = ctx.mem_events()
    from ntsynt_amd.device import BloomFilter, bf_size_bytes, sketch
    a5 = copy.copy(args)
    a5.family, a5.substitutions_only = "assembly-like", False
    div, k, w = 0.013, args.k, args.w
    t0 = time.time()
    gens = [family_genome(ctx, a5, total_bp, contigs, j, div / 2.0) for j in range(3)]
    t_synth = time.time() - t0
    bases = sum(g.total_bp for g in gens)
    _, nbytes = bf_size_bytes(gens[0].total_bp, args.fpr)
    ctx.profile(True)
    ctx.sync()
    t0 = time.time()
    common = BloomFilter(ctx, nbytes, k)
    common.insert(gens[0])
    direct = [ctx.path_stats()["bf_direct_indices"]]
    fallback = 0
    for g in gens[1:]:
        common.insert_and(g)
        st = ctx.path_stats()
        direct.append(st["bf_direct_indices"])
        fallback += st["bf_list_fallback"]
    ctx.sync()
    t_build = time.time() - t0
    ins = ctx.timing("bf_insert"), ctx.timing("bf_insert_and")
    occ = common.get_fpr()
    ctx.profile(2)
    for g in gens:
        sketch(ctx, g, k, w, common).free()
    ctx.sync()
    t0 = time.time()
    n_steps, n_mx, many, dup = 3, 0, 0, 0
    for it in range(n_steps):
        n_mx = many = 0
        for g in gens:
            mx = sketch(ctx, g, k, w, common)
            n_mx += len(mx)
            many += ctx.path_stats()["sketch_many_listed"]
            mx.free()
    ctx.sync()
    dt = time.time() - t0
    sel_ms, sel_n = ctx.timing("hash_select")
    cand, gaps, gap_kmers = ctx.sketch_stats()
    mx = sketch(ctx, gens[0], k, w, common)
    h1 = mx.to_numpy()[0]
    mx.free()
    _, cnt = np.unique(h1, return_counts=True)
    dup = int((cnt > 1).sum())
    out = {"what": "BASELINE configs[4]'s parameter set (-d 1.3, k=24 w=1000 + --w_rounds 250 100) on 3 assembly-like synthetic genomes "
                   f"({total_bp / 1e6:g} Mbp, {contigs} chromosomes; the real assemblies are not in the container)",
           "records_per_genome": [len(g.names) for g in gens], "bases": bases, "synth_s": round(t_synth, 2),
           "sketch_Gbases_s": round(bases * n_steps / dt / 1e9, 3), "ms_per_step": round(dt / n_steps * 1e3, 3), "minimizers_per_step": n_mx,
           "select_kernel_avg_ms": round(sel_ms / max(sel_n, 1), 4), "prune_c": getattr(ctx, "last_prune_c", 0),
           "candidates_last_launch": cand, "uncovered_ranges_last_launch": gaps, "uncovered_kmers_last_launch": gap_kmers,
           "bloom": {"build_s": round(t_build, 4), "bf_insert_ms": round(ins[0][0] / max(ins[0][1], 1), 3),
                     "bf_insert_and_avg_ms": round(ins[1][0] / max(ins[1][1], 1), 3) if ins[1][1] else None,
                     "build_Gbases_s": round(bases / t_build / 1e9, 2), "occupancy_common": occ},
           # where an i.i.d. family never goes
           "paths_reached": {"bloom_indices_bypassing_full_buckets_per_genome": direct, "bloom_list_fallbacks": fallback,
                             "select_candidates_from_tiles_listing_more_than_their_slots_per_step": many,
                             "minimizer_hashes_occurring_twice_within_genome_0": dup}}
    common.free()
    for g in gens:
        g.free()
    out["allocator"] = ctx.mem_events_since(ev5)
    return out, a5, div


class Rig:
    """One workload resident on this rank -- the family's genomes (whole, or this rank's ranges of records), the common filter built
    through exchange 1 -- and step(): one sketch pass over what the rank holds followed by exchange 2.  Genomes deal out whole when their
    number is a multiple of the ranks (c4: eight genomes on 1 / 2 / 4 / 8 GPUs); otherwise the family's records are shared out by bases
    across genome boundaries (ntsynt_amd.pipeline.partition_plan: c3's three genomes on 8 GPUs are eight ranges of ~1.125 Gbp)."""

    def __init__(self, name, args, ctx, comm, world, rank, use_overrides=True):
        from ntsynt_amd import pipeline, synth
        from ntsynt_amd.device import Genome
        self.name, self.args, self.ctx, self.comm, self.world, self.rank = name, args, ctx, comm, world, rank
        n_fam, mbp, contigs, div, self.scaling = WORKLOADS[name]
        if use_overrides:
            n_fam = args.genomes or n_fam
            div = args.divergence if args.divergence is not None else div
        mbp, contigs = args.mbp or mbp, args.contigs or contigs            # (a leg on another workload keeps its family, at the run's size)
        self.n_fam, self.mbp, self.contigs, self.div = n_fam, mbp, contigs, div
        self.total_bp = total_bp = int(mbp * 1e6)
        self.fam_bases = family_bases(args, n_fam, total_bp, contigs)
        self.plan = None
        t0 = time.time()
        if world > 1 and n_fam % world != 0:
            if args.substitutions_only:
                rec_lens = [np.full(contigs, total_bp // contigs, dtype=np.int64)] * n_fam
            elif args.family == "assembly-like":
                rec_lens = [synth.realistic_plan(contigs, total_bp // contigs, j, ANCESTOR_SEED)[0] for j in range(n_fam)]
            else:
                rec_lens = [synth.structural_plan(contigs, total_bp // contigs, j, ANCESTOR_SEED)[0] for j in range(n_fam)]
            self.plan = pipeline.partition_plan(rec_lens, world)
            self.n_rec = [len(x) for x in rec_lens]
            self.filters_of, self.slot_group, self.n_groups = pipeline.partition_groups(self.plan, self.n_rec)
            self.parts = self.plan[rank]
            self.genomes = []
            for g, a, b in self.parts:
                whole = family_genome(ctx, args, total_bp, contigs, g, div / 2.0)
                if a == 0 and b == self.n_rec[g]:
                    self.genomes.append(whole)
                else:
                    self.genomes.append(whole.slice(a, b))
                    whole.free()
            self.part_base = sum(len(ps) for ps in self.plan[:rank])
            self.n_parts = sum(len(ps) for ps in self.plan)
            self.list_slots = max(1, max(len(ps) for ps in self.plan))
            self.mine = [self.part_base + i for i in range(len(self.parts))]        # (exchange 2 numbers the parts in family order)
            all_bases = [sum(int(rec_lens[g][a:b].sum()) for g, a, b in ps) for ps in self.plan]
            self.balance = {"bases_per_rank": all_bases, "max_over_mean": round(max(all_bases) / (sum(all_bases) / world), 4)}
        else:
            self.mine = [g for g in range(n_fam) if g % world == rank]
            self.genomes = [family_genome(ctx, args, total_bp, contigs, g, div / 2.0) for g in self.mine]
            per_rank = [sum(self.fam_bases[g] for g in range(n_fam) if g % world == r) for r in range(world)]
            self.balance = {"bases_per_rank": per_rank, "max_over_mean": round(max(per_rank) / (sum(per_rank) / world), 4)}
        ctx.sync()                                   # (the generators are kernels: what follows must not be charged for them)
        self.t_synth = time.time() - t0
        self.bases = sum(g.total_bp for g in self.genomes)
        batch = name == "c2" and not args.no_batch and len(self.genomes) > 1 and world == 1
        self.units = [Genome.concat(ctx, self.genomes)] if batch else self.genomes
        self.common = None
        self.n_all = None
        self.t_sketch = 0.0                          # seconds spent in the sketch calls of step() (reset by the caller around a timed region)

    def build_filter(self, again=False, levels=False, trim=False):
        "per-genome filters, local cascade, exchange 1; returns the timings"
        self.levels = []
        from ntsynt_amd.device import BloomFilter, bf_size_bytes
        ctx, comm, world, k = self.ctx, self.comm, self.world, self.args.k
        _, self.nbytes = bf_size_bytes(self.fam_bases[0], self.args.fpr)       # sized by the first file of the family, on every rank (A1)
        t0 = time.time()
        common = BloomFilter(ctx, self.nbytes, k, world=world if comm is not None else 1)
        extra = []
        occ_single = None
        if self.plan is not None:
            # one filter for the genomes this rank holds whole (their cascade is local), one per genome it holds a part of
            for fi, (_, idxs) in enumerate(self.filters_of[self.rank]):
                f_ = common if fi == 0 else BloomFilter(ctx, self.nbytes, k, world=world)
                if fi:
                    extra.append(f_)
                f_.insert(self.genomes[idxs[0]])
                for i_ in idxs[1:]:
                    f_.insert_and(self.genomes[i_])
        else:
            common.insert(self.genomes[0])
            occ_single = common.get_fpr()
            for g in self.genomes[1:]:
                tl = time.time()
                common.insert_and(g)                 # one cascade level inside the build's last pass (nts_bf_insert_and), as pipeline.run does
                if levels:
                    common.get_fpr()                 # (the pipeline prints the occupancy after every level; the library then knows when the filter is sparse)
                    # which way the level went: the build with the AND in its last pass, or -- the running filter all but empty -- the
                    # literal look-up of every k-mer (nts_bf_level_stats)
                    self.levels.append({"ms": round((time.time() - tl) * 1e3, 2), "literal": bool(ctx.bf_level_stats()["sparse_level"])})
        ctx.sync()
        t_build = time.time() - t0
        t_warm = None
        if world == 1 and again:
            # the same filter once more, into the same allocation, with the build's workspaces (two bucket arrays and the bypass list:
            # ~25 GB of hipMalloc at 3 Gbp) in place: what a level costs once a run is under way
            t1 = time.time()
            common.clear()
            common.insert(self.genomes[0])
            for g in self.genomes[1:]:
                common.insert_and(g)
            ctx.sync()
            t_warm = time.time() - t1
        t_allreduce = 0.0
        if world > 1:
            t1 = time.time()
            if self.plan is not None:
                comm.allreduce_parts([common] + extra, self.slot_group, max(1, max(len(x) for x in self.slot_group)), self.n_groups)
            else:
                comm.allreduce_and(common)
            ctx.sync()
            t_allreduce = time.time() - t1
        for f_ in extra:
            f_.free()
        self.common = common
        if trim:
            ctx.trim_bf_build()                      # as the pipeline does once its common filter stands (pipeline.run): the buckets go to the allocation cache
        return {"build_s": t_build, "build_again_s": t_warm, "allreduce_s": t_allreduce, "occ_single": occ_single}

    def step(self, pool=None):
        from ntsynt_amd.device import Minimizers, sketch
        ctx, comm, world, k, w = self.ctx, self.comm, self.world, self.args.k, self.args.w
        held, n = [], 0
        t0 = time.time()
        if pool is not None:
            held = pool.sketch(self.units, k, w, self.common)
            n = sum(len(mx) for mx in held)
        else:
            for g in self.units:
                mx = sketch(ctx, g, k, w, self.common)
                n += len(mx)
                held.append(mx)
        self.t_sketch += time.time() - t0                           # (a sketch call returns when its list's length is known: the kernels are through)
        if world > 1:                                               # exchange 2: every rank receives every list
            if self.plan is not None:
                parts = comm.allgather_minimizers(held, self.mine, self.n_parts, self.list_slots)
                by_genome, at = {}, 0
                for ps in self.plan:
                    for g_, a, b in ps:
                        by_genome.setdefault(g_, []).append((parts[at], a))
                        at += 1
                n_all = 0
                for g_ in range(self.n_fam):                        # the genome's list: its parts in record order
                    whole_list = Minimizers.concat(ctx, [m for m, _ in by_genome[g_]], [a for _, a in by_genome[g_]])
                    n_all += len(whole_list)
                    whole_list.free()
                self.n_all = n_all
                for mx in parts:
                    mx.free()
            else:
                everything = comm.allgather_minimizers(held, self.mine, self.n_fam)
                self.n_all = sum(len(mx) for mx in everything)
                for mx in everything:
                    mx.free()
        for mx in held:
            mx.free()
        return n

    def free(self):
        for g in self.genomes:
            g.free()
        if len(self.units) == 1 and self.units[0] not in self.genomes:
            self.units[0].free()
        self.genomes, self.units = [], []
        if self.common is not None:
            self.common.free()
            self.common = None


def valley_leg(args, ctx, total_bp, contigs, n_fam, div, reps=2):
    """Filters that accept a few per cent of a genome's k-mers -- between the pruned path's regime (one threshold, ~11 / p probes per
    window) and the sparse-filter one: three genomes at ~10 %, eight at ~4 %, the reference's eleven-genome row (README.md:158).  The
    tiered selection (k_hash_tiers, csrc/nts_tiers.inc) against every k-mer probed, which is what these inputs got until round 4."""
    from ntsynt_amd.device import BloomFilter, bf_size_bytes, sketch
    k, w = args.k, args.w
    g0 = family_genome(ctx, args, total_bp, contigs, 0, div / 2.0)
    _, nbytes = bf_size_bytes(g0.total_bp, args.fpr)
    bf = BloomFilter(ctx, nbytes, k)
    bf.insert(g0)
    g0.free()
    for j in range(1, n_fam):
        g = family_genome(ctx, args, total_bp, contigs, j, div / 2.0)
        bf.insert_and(g)
        g.free()
    occ = bf.get_fpr()
    t_all, bases, n_mx, probes, kmers, path = 0.0, 0, 0, 0, 0, None
    dense = None
    for j in range(n_fam):
        g = family_genome(ctx, args, total_bp, contigs, j, div / 2.0)
        sketch(ctx, g, k, w, bf).free()
        ctx.sync()
        t1 = time.time()
        for _ in range(reps):
            mx = sketch(ctx, g, k, w, bf)
            n = len(mx)
            mx.free()
        ctx.sync()
        t_all += (time.time() - t1) / reps
        bases += g.total_bp
        n_mx += n
        pr, rounds, tiers = ctx.sketch_tiers()
        probes += pr
        kmers += g.valid_kmers(k)
        path = "k_hash_tiers (%d tiers)" % tiers if tiers else ("one threshold, c = %d" % ctx.last_prune_c if ctx.sketch_stats() and ctx.last_prune_c else "every k-mer probed")
        if j == 0:                                                  # the same genome with every k-mer probed: what it replaces
            ctx.sketch_tiers("never")
            ctx.sketch_mode("dense")
            sketch(ctx, g, k, w, bf).free()
            ctx.sync()
            t1 = time.time()
            mx = sketch(ctx, g, k, w, bf)
            n_d = len(mx)
            mx.free()
            ctx.sync()
            dense = {"Gbases_s": round(g.total_bp / (time.time() - t1) / 1e9, 2), "same_list_length": n_d == n}
            ctx.sketch_mode(args.mode, args.prune_c)
            ctx.sketch_tiers("auto")
        g.free()
    bf.free()
    return {"workload": f"{n_fam} synthetic {total_bp / 1e6:g} Mbp genomes at {div * 100:g}% divergence, k={k} w={w}", "common_filter_occupancy": occ,
            "value_Gbases_s": round(bases / t_all / 1e9, 2), "ms_per_genome": round(t_all / n_fam * 1e3, 2), "path": path,
            "probes_per_kmer": round(probes / max(kmers, 1), 4) if probes else None, "minimizers_per_step": n_mx,
            "every_kmer_probed_genome0": dense}


def cold_leg(args, ctx, rig):
    """The first sketch of a genome nothing has been derived from yet (rule indexlr runs once per genome, bin/ntsynt_run_pipeline.smk:74-85:
    in a pipeline run cold is the real case) -- three times, on three fresh genomes, each split into what the call spends in
    hipMalloc / hipFree (nts_alloc_stats around the call), what the library's workspaces grew by, and the rest (derived structures --
    2-bit image, run table, first-k-mer tables -- plus the sketch itself); then one more fresh genome with every kernel group between
    events.  The first of the three meets a context whose sketch workspaces do not exist yet (the Bloom build's do)."""
    from ntsynt_amd.device import sketch
    k, w = args.k, args.w
    runs = []
    n_cold = 0
    for j in range(3):
        fresh = family_genome(ctx, args, rig.total_bp, rig.contigs, rig.mine[0] if rig.plan is None else 0, rig.div / 2.0)
        ctx.sync()
        c0, a0 = ctx.alloc_stats()
        live0 = ctx.mem_stats()["live"]
        t1 = time.time()
        mx = sketch(ctx, fresh, k, w, rig.common)
        n_cold = len(mx)
        dt = time.time() - t1
        c1, a1 = ctx.alloc_stats()
        live1 = ctx.mem_stats()["live"]
        runs.append({"ms": round(dt * 1e3, 2), "hipMalloc_hipFree_calls": c1 - c0, "ms_in_hipMalloc_hipFree": round(a1 - a0, 2),
                     "workspace_growth_MB": round((live1 - live0) / 1048576.0, 1)})
        mx.free()
        fresh.free()
    fresh = family_genome(ctx, args, rig.total_bp, rig.contigs, rig.mine[0] if rig.plan is None else 0, rig.div / 2.0)
    ctx.sync()
    ctx.profile(1)
    sketch(ctx, fresh, k, w, rig.common).free()
    ctx.sync()
    groups = {n: ctx.timing(n) for n in SKETCH_KERNELS + ["pack_image"]}
    ctx.profile(True)
    fresh.free()
    t1 = time.time()
    sketch(ctx, rig.genomes[-1], k, w, rig.common).free()           # after its Bloom insert (2-bit image exists): the
    t_first = time.time() - t1                                      # pipeline's situation
    ms = [r["ms"] for r in runs]
    return {"first_sketch_of_a_fresh_genome_ms": ms[0], "min_ms": min(ms), "max_ms": max(ms), "three_fresh_genomes": runs,
            "Gbases_s": round(rig.genomes[0].total_bp / (ms[0] * 1e-3) / 1e9, 2),
            "kernel_groups_ms_of_a_fourth": {n: round(v[0], 3) for n, v in groups.items() if v[1]},
            "first_sketch_after_its_bloom_insert_ms": round(t_first * 1e3, 2), "minimizers": n_cold,
            "includes": "2-bit image, run table of valid k-mers, first-k-mer tables, workspace allocation (the first of the three)"}


def vs_n1(which, now):
    "this run's figure over the committed one-GPU line's (profiles/r06_bench_n1.json): `value`, or the c4 leg's value_Gbases_s"
    try:
        base = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_n1.json")))
        ref = base["value"] if which == "value" else base["c4"]["value_Gbases_s"]
        return round(now / ref, 3)
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (torch.distributed.run, one per GPU, rendezvous on
    127.0.0.1) and hand their exit status on; the line comes from rank 0 of that job."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def e2e_child(spec_path):
    "the end-to-end leg's own process: runs the pipeline on the files the parent wrote, prints the leg's record"
    spec = json.load(open(spec_path))
    a = argparse.Namespace(**spec["args"])
    a.e2e_is_child = True
    out = e2e_leg(a, spec["device"], spec["n_fam"], spec["total_bp"], spec["contigs"], spec["div"], spec["workdir"], paths=spec["paths"])
    print("E2E_CHILD_RESULT " + json.dumps(out), flush=True)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--e2e-child":
        return e2e_child(sys.argv[2])
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))                     # (plain `python bench.py --gpus N`: the ranks are ours to start)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): refusing to report one as the other")
    # ONE workload along the scaling curve: BASELINE configs[2] (c3, the configuration the metric is quoted on) is `value` at every N;
    # configs[3] (c4) is the `c4` leg of the same line at every N
    name = args.workload or "c3"
    import torch
    import torch.distributed as dist
    from ntsynt_amd.device import BloomFilter, Comm, Context, Genome, SketchPool, allgather_minimizers, bf_size_bytes, sketch
    # The process group carries the communicator id, the barriers and the max-over-ranks of the clock; the two exchanges run
    # inside libntsynt_hip.so.  NTS_BENCH_BACKEND=gloo + NTS_RCCL_LIB=<tests/rccl_standin>: ranks sharing the visible GPUs (boxes
    # with fewer GPUs than ranks, tests/test_gpu_multirank.py) -- same code path, not a measurement; the measured one is nccl.
    backend = os.environ.get("NTS_BENCH_BACKEND", "nccl")
    shared_gpus = backend != "nccl"
    if shared_gpus and torch.cuda.is_available():
        local_rank = local_rank % torch.cuda.device_count()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if shared_gpus:
            dist.init_process_group(backend)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    ctx = Context(local_rank)
    ctx.sketch_mode(args.mode, args.prune_c)
    # The line's legs build and free families of up to eight 3 Gbp genomes one after the other.  Round 5's driver box took 1.39 s for a
    # filter build whose seven levels sum to 0.15 s: memory the leg before had given back to the driver was not ready when the next leg
    # asked for it.  Now ONE driver allocation sized for the largest leg is made here (nts_mem_reserve) and every genome, filter and
    # workspace of every leg is cut from it; each leg reports what it asked of the driver all the same (`allocator` in c4 / valley /
    # c5_like / cold, `allocator` + `device_memory_in_use_by_others_at_start_GB` in the end-to-end runs, which are processes of their own
    # and reserve their own plan: ntsynt_amd.pipeline.reserve_for_run).
    from ntsynt_amd.pipeline import plan_bytes
    arena_plan, arena_got, t_arena = 0, 0, 0.0
    if os.environ.get("NTS_BENCH_ARENA", "1") != "0":
        n_res = max(3, -(-8 // world)) if not args.workload or args.workload == "c3" else 3
        per_genome = int((args.mbp or {"c2": 100, "c3": 3000, "c4": 3000}.get(args.workload or "c3", 3000)) * 1e6)
        # (the largest leg's family + 0.7 of the headline family's plan: its filter and workspaces stay live next to the leg's)
        arena_plan = plan_bytes([per_genome] * n_res, args.fpr, True) + int(0.7 * plan_bytes([per_genome] * 3, args.fpr, True))
        t_ar = time.time()
        arena_got = ctx.mem_reserve(arena_plan)
        t_arena = time.time() - t_ar
    ev_start = ctx.mem_events()
    # The two exchanges run inside libntsynt_hip.so over RCCL (nts_bf_allreduce_and / _parts, nts_mx_allgather) and nowhere else:
    # a communicator that does not come up ends the run with its error (round 3 fell back to torch.distributed here)
    comm, exchanges, rccl_ranks, served_by = None, "none (one GPU)", 1, None
    pg_dev = "cpu" if shared_gpus else f"cuda:{local_rank}"
    if world > 1:
        comm = Comm.from_torch(ctx)
        served_by = ctx.lib.nts_comm_library().decode()
        rccl_ranks = comm.rccl_ranks()
        if rccl_ranks != world:
            sys.exit(f"bench.py: the communicator holds {rccl_ranks} ranks, the launcher started {world}")
        exchanges = f"libntsynt_hip.so (nts_bf_allreduce_and / nts_bf_allreduce_parts, nts_mx_allgather) over {'RCCL' if served_by == 'librccl' else served_by}"
    k, w = args.k, args.w

    # ---- the family and its common filter (Rig) ------------------------------------------------------------------------------
    rig = Rig(name, args, ctx, comm, world, rank)
    n_fam, mbp, contigs, div, scaling, total_bp = rig.n_fam, rig.mbp, rig.contigs, rig.div, rig.scaling, rig.total_bp
    genomes, units, mine, fam_bases, bases, t_synth = rig.genomes, rig.units, rig.mine, rig.fam_bases, rig.bases, rig.t_synth
    shard = rig.plan
    ctx.profile(True)
    bt = rig.build_filter(again=True, trim=True)    # (the headline rig does what a run does; the later legs' rigs keep their buckets for one another)
    common, nbytes = rig.common, rig.nbytes
    t_build, t_build_warm, t_allreduce, occ_single = bt["build_s"], bt["build_again_s"], bt["allreduce_s"], bt["occ_single"]
    ins_ms, ins_n = ctx.timing("bf_insert")
    insa_ms, insa_n = ctx.timing("bf_insert_and")
    occ_common = common.get_fpr()

    # ---- cold leg: the first sketch of a genome nothing has been derived from yet ---------------------------------
    cold = None
    if world == 1 and not args.no_cold_leg:
        cold = cold_leg(args, ctx, rig)

    # --at-once: the genomes of a step are sketched at once, each on a context of its own (device.SketchPool; NTS_SKETCH_POOL=3
    # in the pipeline): one genome's latency-bound tail and the host's round trips overlap another's rolling.  Between +6 %
    # and -18 % in throughput from run to run, and the kernels' own durations are then measured under each
    # other's load -- the timed steps run one genome after the other.
    pool = None
    if len(units) > 1 and name != "c4" and args.at_once:
        pool = SketchPool(ctx, min(len(units), 3))
        pool.configure(lambda c: c.sketch_mode(args.mode, args.prune_c))

    def step():
        return rig.step(pool)

    def fence():
        ctx.sync()
        if world > 1:
            dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def all_ctx(fn):
        (pool.configure if pool is not None else (lambda f: f(ctx)))(fn)

    def timing_of(n):
        return pool.timing(n) if pool is not None else ctx.timing(n)

    def timed(n_warm, n_steps, level=2, step=step):
        for _ in range(n_warm):
            step()
        all_ctx(lambda c: c.profile(level))   # 2: HIP events around the dominant kernel only (every pair is a bubble in the stream)
        fence()
        t_start = time.time()
        n_mx = 0
        for _ in range(n_steps):
            n_mx = step()
        fence()
        el = time.time() - t_start
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=pg_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, n_mx

    for _ in range(args.warmup):
        step()
    rig.t_sketch = 0.0
    dt, n_mx = timed(0, args.steps)
    x2_headline = comm.last_exchange2() if comm is not None else None     # (of the last timed step: the later legs exchange other lists)
    t_sk = rig.t_sketch
    if world > 1:                                                   # the slowest rank's sketch time
        tt = torch.tensor([t_sk], dtype=torch.float64, device=pg_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_sk = float(tt.item())
    tm = {n: timing_of(n) for n in SKETCH_KERNELS}
    # the other kernels of the call: one more pass, untimed, with every kernel group bracketed by events
    all_ctx(lambda c: c.profile(1))
    step()
    fence()
    detail = {n: timing_of(n) for n in SKETCH_KERNELS}
    # and the dominant kernel alone on the GPU: one genome after the other on one stream (with the genomes sketched at once its
    # launches share the chip with another genome's tail kernels, so the duration above is the one under that load)
    alone_ms = None
    if pool is not None:
        ctx.profile(2)
        for g in units:
            sketch(ctx, g, k, w, common).free()
        ctx.sync()
        a_ms_, a_n_ = ctx.timing("hash_select")
        alone_ms = a_ms_ / max(a_n_, 1)
    all_ctx(lambda c: c.profile(2))
    cand, gaps, gap_kmers = ctx.sketch_stats()
    c_used = getattr(ctx, "last_prune_c", 0)
    per_launch_bases = bases / len(units)
    dense_bpb = 1.0 + SECTOR + 32.0 / (w + 1)

    def avg(n):
        return tm[n][0] / max(tm[n][1], 1)

    nruns = None
    if world == 1 and name == "c3" and not args.no_nruns_leg:
        # SURVEY.md 8(d)'s N-run variant: 0.5 % of the bases in runs of 100-50,000 N -- the valid-k-mer path (run table, tiles that
        # span runs) at full size, against the same common filter
        g_n = family_genome(ctx, args, total_bp, contigs, mine[0], div / 2.0, n_runs=True)
        for _ in range(2):
            sketch(ctx, g_n, k, w, common).free()
        ctx.sync()
        t1 = time.time()
        n_steps_n, n_mx_n = max(3, args.steps // 2), 0
        for _ in range(n_steps_n):
            mx = sketch(ctx, g_n, k, w, common)
            n_mx_n = len(mx)
            mx.free()
        ctx.sync()
        d_n = time.time() - t1
        nruns = {"what": "genome 0 of the family with 0.5 % of its bases in N runs of 100-50,000 (SURVEY.md 8(d) variant), same filter",
                 "value_Gbases_s": round(g_n.total_bp * n_steps_n / d_n / 1e9, 3), "ms_per_sketch": round(d_n / n_steps_n * 1e3, 3),
                 "minimizers": n_mx_n, "bases": g_n.total_bp}
        g_n.free()

    dense = None
    n_at_once = len(pool.ctxs) if pool is not None else 1
    if pool is not None:                  # the remaining legs run one sketch at a time on the main context
        pool.close()
        pool = None
    if args.mode != "dense" and not args.no_dense_leg and world == 1:
        ctx.sketch_mode("dense")
        n_d = max(2, args.steps // 2)
        d_dt, _ = timed(1, n_d, level=1)
        hp_ms, hp_n = ctx.timing("hash_probe")
        wm_ms, wm_n = ctx.timing("window_min")
        a_ms = hp_ms / max(hp_n, 1)
        ach = dense_bpb * per_launch_bases / (a_ms * 1e-3) / 1e9
        line = (1.0 + L2_LINE + 32.0 / (w + 1)) * per_launch_bases / (a_ms * 1e-3) / 1e9
        dense = {"kernel": "k_hash<MODE_KEYS> (every k-mer probed)", "algorithmic_bytes_per_base": round(dense_bpb, 3),
                 "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4), "avg_launch_ms": round(a_ms, 4),
                 "G_probes_s": round(per_launch_bases / (a_ms * 1e-3) / 1e9, 2),
                 # the L2 fetches 128-byte lines: what a missing 4-byte probe really moves (profiles/r02_probe_granularity.md)
                 "at_128B_line_granularity": {"GBs": round(line, 1), "frac": round(line / HBM_PEAK_GBS, 4)},
                 "window_min_avg_ms": round(wm_ms / max(wm_n, 1), 4),
                 "value_Gbases_s": round(bases * n_d / d_dt / 1e9, 3)}
        ctx.sketch_mode(args.mode, args.prune_c)

    valu = None
    if world == 1:
        # VALU roof of the issue-bound kernel: measured issue rate of its instruction mix's slowest member (nts_bench_valu)
        rows = {kind: ctx.bench_valu(kind, 8, 20000) for kind in ("v_xor_b32", "v_alignbit_b32", "v_lshl_add_u64", "v_add3_u32",
                                                                  "v_cmp_ge_u32+v_addc_co_u32", "roll31 step (9 instructions)")}
        valu = {"wave_instr_per_s_per_cu": {n: round(r["wave_instr_per_s_per_cu"] / 1e9, 3) for n, r in rows.items()},
                "unit": "G wave-instructions/s/CU", "how": "nts_bench_valu: 8 waves per SIMD, eight independent chains per lane, wall clock"}

    c4 = valley = rows = None
    if name == "c3" and not args.no_c4_leg:
        # BASELINE configs[3] at this N, in the same line: eight genomes at 10 %, genome g on rank g mod N, the filter's cascade local,
        # then the AND all-reduce; every step ends with the all-gather of the lists.  (At N = 1: all eight on this GPU.)
        for g in rig.genomes:
            g.free()
        rig.genomes, rig.units = [], []
        ev4 = ctx.mem_events()
        r4 = Rig("c4", args, ctx, comm, world, rank, use_overrides=False)
        b4 = r4.build_filter(levels=(world == 1))
        alloc4_build = ctx.mem_events_since(ev4)
        d4, n4 = timed(1, 2, level=0, step=r4.step)
        x2_c4 = comm.last_exchange2() if comm is not None else None
        c4 = {"workload": f"c4: 8 synthetic {r4.mbp:g} Mbp genomes at 10% divergence, genome g on GPU g mod {world}",
              "n_gpus": world, "value_Gbases_s": round(sum(r4.fam_bases) * 2 / d4 / 1e9, 3), "ms_per_step": round(d4 / 2 * 1e3, 2),
              "common_filter_occupancy": r4.common.get_fpr(), "minimizers_per_step_rank0": n4, "minimizers_per_step_all_genomes": r4.n_all if world > 1 else n4,
              "common_filter_build_s": round(b4["build_s"], 4), "allreduce_and_s": round(b4["allreduce_s"], 4),
              "all_reduce_gathered_set_bit_indices": bool(comm.last_sparse()) if comm is not None else None,
              "common_filter_levels": r4.levels or None, "balance": r4.balance,
              "speedup_vs_n1": vs_n1("c4", round(sum(r4.fam_bases) * 2 / d4 / 1e9, 3)) if world > 1 else None,
              "exchange2_bytes_per_step": x2_c4,
              # what the leg asked of the driver while it generated its genomes and built its filter, and over the whole leg
              "allocator": {"genomes_and_filter_build": alloc4_build, "whole_leg": ctx.mem_events_since(ev4)}}
        r4.free()
        if world == 1 and not args.no_valley_leg:
            evv = ctx.mem_events()
            valley = {"three_genomes_at_10pct": valley_leg(args, ctx, total_bp, contigs, 3, 0.10),
                      "eight_genomes_at_4pct": valley_leg(args, ctx, total_bp, contigs, 8, 0.04)}
            valley["allocator"] = ctx.mem_events_since(evv)
            # the reference's third published row as a shape (README.md:158: eleven bee genomes of 0.44 Gbp): eleven synthetic genomes of 16
            # chromosomes at 4 %; its end-to-end run against the CPU restatement is on record (profiles/r06_e2e_oracle_11x440Mbp_4pct.json)
            if not args.mbp:
                rows = {"eleven_genomes_of_440Mbp_at_4pct": valley_leg(args, ctx, 440_000_000, 16, 11, 0.04)}
        if world == 1:                                                  # the later legs of the N = 1 line work on the headline family again
            rig.genomes = [family_genome(ctx, args, total_bp, contigs, g, div / 2.0) for g in mine]
            rig.units = rig.genomes
            genomes = units = rig.genomes

    out = None
    if rank == 0:
        value = sum(fam_bases) * args.steps / dt / 1e9
        pruned_run = tm["hash_select"][1] > 0
        if pruned_run:
            # (the library's rule: upper-halves kernel for k <= 32 while a 4096-index tile lists at most ~220 k-mers)
            hi_kernel = k <= 32 and 4096.0 * c_used / w * 1.15 <= 256.0
            kern = ("k_hash_select_hi (upper halves of both strand hashes rolled for every k-mer; the ~c/w listed k-mers hashed in full and probed, "
                    "probes overlapped with the next tile)") if hi_kernel else "k_hash_select (hash every k-mer, probe candidates only)"
            a_ms = avg("hash_select")
            probe_frac = min(1.0, c_used / w)
            # bytes the pruned kernel has to move per base: the base (2 bits, from the packed image of the genome),
            # one sector per probed candidate, 16 B per accepted candidate written
            bpb = 0.25 + SECTOR * probe_frac + 16.0 * cand / per_launch_bases
        else:
            kern, a_ms = "k_hash<MODE_KEYS> (every k-mer probed)", avg("hash_probe")
            bpb = dense_bpb
            hi_kernel = False
        achieved = bpb * per_launch_bases / (a_ms * 1e-3) / 1e9 if a_ms > 0 else 0.0
        if valu is not None and pruned_run and a_ms > 0:
            # VALU wave-instructions per 64 k-mers from the PMC pass (profiles/r02_sq_counters.json); the roof: the issue rates
            # the microbenchmark measures for the classes of the kernel's instruction mix
            key = "k_hash_select_hi" if hi_kernel else "k_hash_select"
            per_64 = 17.2 if hi_kernel else 29.8
            sq_source = None
            for rnd in (6, 5, 4, 3, 2):
                try:
                    sq = json.load(open(os.path.join(ROOT, "profiles", f"r0{rnd}_sq_counters.json")))
                    per_64 = float(sq["kernels"][key]["valu_wave_instructions_per_64_kmers"])
                    sq_source = f"profiles/r0{rnd}_sq_counters.json"
                    break
                except (OSError, KeyError, ValueError):
                    continue
            n_cu = 256
            ach_v = per_64 * per_launch_bases / 64.0 / (a_ms * 1e-3) / n_cu / 1e9
            rate = valu["wave_instr_per_s_per_cu"]
            fast, slow = rate["v_xor_b32"], min(rate["v_alignbit_b32"], rate["v_lshl_add_u64"])
            mix_file = os.path.join(ROOT, "profiles", "r03_valu_mix.json")
            if hi_kernel and os.path.exists(mix_file):
                # the mix counted from the kernel's ISA (profiles/valu_mix.py: one turn of the tile loop, straight-line code), each
                # class priced at its measured issue rate; what the PMC count holds beyond that range at the mix's average
                mx = json.load(open(mix_file))["classes"]
                n2, n3, nc = mx["two-operand 32-bit"], mx["three-operand or 64-bit"], mx["compare / carry"]
                pair = rate["v_cmp_ge_u32+v_addc_co_u32"]
                peak_v = (n2 + n3 + nc) / (n2 / fast + n3 / min(slow, rate["v_add3_u32"]) + nc / pair)
                mix = (f"counted from the ISA (profiles/r03_valu_mix.json): {n2} two-operand 32-bit, {n3} three-operand or 64-bit, "
                       f"{nc} compare / carry instructions per tile of 4096 k-mers, each class at its measured rate")
            elif hi_kernel:
                # per k-mer: the 9-instruction rolling step (measured as a unit) + 2 two-operand instructions for the table offset;
                # the rest (listing, full hashes of the listed k-mers, probes, output: per_64 - 11 per k-mer) priced half and half
                rest = max(per_64 - 11.0, 0.0)
                peak_v = per_64 / (9.0 / rate["roll31 step (9 instructions)"] + 2.0 / fast + rest * 0.5 / fast + rest * 0.5 / slow)
                mix = "assumed: 9 (rolling step, measured as a unit) + 2 two-operand + the rest half two-operand, half three-operand"
            else:
                # the rolling step is ~12 two-operand 32-bit operations (xor / and / shift class) and ~17 three-operand or 64-bit ones
                # (alignbit, bfi, lshl_add_u64 class): the roof of that mix
                peak_v = 29.0 / (12.0 / fast + 17.0 / slow)
                mix = "12 two-operand + 17 three-operand or 64-bit per k-mer"
            valu.update({"kernel": key, "valu_wave_instr_per_64_kmers": per_64, "sq_source": sq_source, "achieved": round(ach_v, 3),
                         "peak": round(peak_v, 3), "frac": round(ach_v / peak_v, 3), "mix": mix,
                         "peak_if_all_slow_class": slow, "peak_if_all_fast_class": fast})
        out = {
            "metric": "minimizer-sketch Gbases/s (sketch with common Bloom filter, inputs resident in HBM)",
            "value": round(value, 3), "unit": "Gbases/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{name}: {n_fam} synthetic {mbp:g} Mbp genomes ({contigs} contigs) at {div * 100:g}% divergence, "
                                   f"k={k} w={w} fpr={args.fpr}, genome g on GPU g mod {world}",
                       "sketch_mode": args.mode, "prune_c": c_used, "genomes_on_rank0": len(genomes), "exchanges": exchanges,
                       "sketch_launch_sequences_per_step_rank0": len(units), "bases_per_step": sum(fam_bases),
                       "family": "substitutions only" if args.substitutions_only else
                                 "substitutions + per genome 5 inversions (1-5 Mbp), 2 inter-contig translocations, 20 indels (1-60 kbp), "
                                 "60 small rearrangements (2-20 kbp moved / copied / inverted within 80 kbp), 200 indels of 1-50 bp",
                       "all_reduce_gathered_set_bit_indices": bool(comm.last_sparse()) if comm is not None else None,
                       "minimizers_per_step_rank0": n_mx, "minimizers_per_step_all_genomes": rig.n_all if world > 1 else n_mx,
                       "parallelism": ((f"{n_fam} genomes over {world} GPUs: the family's records shared out by bases across genome boundaries "
                                        f"(rank 0: {', '.join(f'records {a}..{b} of genome {g}' for g, a, b in rig.parts) or 'none'})") if shard is not None else
                                       f"genomes dealt out whole over {world} GPU(s)") + (f"; the {n_at_once} genomes of a GPU sketched at once, a stream each" if n_at_once > 1 else ""),
                       "balance": rig.balance,
                       # `value` times the whole step (sketch + exchange 2: the all-gather of the lists the replicated graph stage reads); the
                       # sketch calls alone, slowest rank:
                       "sketch_only_Gbases_s": round(sum(fam_bases) * args.steps / t_sk / 1e9, 3) if t_sk > 0 else None,
                       "exchange2_share_of_step": round(max(0.0, 1.0 - t_sk / dt), 4) if world > 1 else 0.0,
                       # exchange 2's payload (nts_comm_last_exchange2): every rank's lists packed (12 B per minimizer + a record table; the
                       # lists themselves hold 20 B per minimizer) and what rank 0 sent per step (its slot to each of the others)
                       "exchange2_bytes_per_step": ({**x2_headline, "packed_over_unpacked": round(x2_headline["packed_bytes"] / max(1, x2_headline["unpacked_bytes"]), 3)}
                                                    if x2_headline is not None else None),
                       # against the N = 1 line of this tree (profiles/r06_bench_n1.json, the builder's last one-GPU run: a different box than the
                       # driver's -- the driver computes scaling from its own runs)
                       "value_speedup_vs_n1": vs_n1("value", value) if world > 1 else None,
                       # did RCCL see N ranks, and which library served the exchanges
                       "rccl_ranks": rccl_ranks, "rccl_library": served_by, "library": ctx.lib._name,
                       "scaling_base": "c3 (BASELINE configs[2], the metric's configuration) is `value` at every N; c4 (configs[3]) is the `c4` leg of the same line at every N",
                       "synth_s": round(t_synth, 3),
                       },
            # `bound`: what the counters say limits the dominant kernel -- the pruned select kernel issues VALU instructions in 0.9 of
            # its cycles (profiles/r0N_sq_counters.json; its roof: `valu` below), the every-k-mer kernel waits for HBM.  achieved / peak /
            # unit / frac stay the HBM figures the contract names (algorithmic bytes per launch / launch time against 8 TB/s)
            "roofline": {"bound": "valu" if pruned_run else "hbm", "achieved_peak_frac_are": "HBM bytes per second against the 8 TB/s peak", "kernel": kern,
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "algorithmic_bytes_per_base": round(bpb, 3),
                         "avg_launch_ms": round(a_ms, 4), "launches": tm["hash_select" if pruned_run else "hash_probe"][1],
                         "genomes_sketched_at_once": n_at_once,
                         "avg_launch_ms_alone_on_the_gpu": round(alone_ms, 4) if alone_ms else None,
                         "frac_alone_on_the_gpu": round(bpb * per_launch_bases / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if alone_ms else None,
                         "other_kernels_avg_ms": {n: round(detail[n][0] / detail[n][1], 4) for n in SKETCH_KERNELS if detail[n][1]},
                         "candidates_per_launch": cand, "uncovered_ranges": gaps, "uncovered_kmers": gap_kmers,
                         # the formulation with one sector read per k-mer moves 65.03 B/base, i.e. at
                         # most 8 TB/s / 65.03 B = 123 Gbases/s; the timed path, expressed in those bytes:
                         "one_probe_per_kmer_equivalent": {
                             "bytes_per_base": round(dense_bpb, 3),
                             "GBs": round(value * dense_bpb / world, 1),
                             "frac_of_peak": round(value * dense_bpb / world / HBM_PEAK_GBS, 3)},
                         "valu": valu, "unpruned": dense},
            "bloom": {"bytes": nbytes, "build_s": round(t_build, 4), "build_again_s": round(t_build_warm, 4) if t_build_warm else None,
                      "allreduce_and_s": round(t_allreduce, 4),
                      "bf_insert_avg_ms": round(ins_ms / max(ins_n, 1), 4),
                      "bf_insert_Gbases_s": round(total_bp / (ins_ms / max(ins_n, 1) * 1e-3) / 1e9, 3) if ins_ms > 0 else None,
                      # a cascade level = the same build with the running filter AND-ed in its last pass (nts_bf_insert_and)
                      "bf_insert_and_avg_ms": round(insa_ms / max(insa_n, 1), 4) if insa_n else None,
                      "bf_insert_and_Gbases_s": round(total_bp / (insa_ms / max(insa_n, 1) * 1e-3) / 1e9, 3) if insa_ms > 0 else None,
                      "roofline": bloom_roofline(total_bp, ins_ms / max(ins_n, 1)) if ins_ms > 0 else None,
                      "occupancy_one_genome": round(occ_single, 6) if occ_single is not None else None, "occupancy_common": occ_common},
        }
        pm = pmc_traffic(name, pruned_run, bpb * per_launch_bases,
                         (0.25 + L2_LINE * probe_frac + 16.0 * cand / per_launch_bases) * per_launch_bases if pruned_run else None)
        if pm:
            out["roofline"]["traffic"] = pm
            # the same launch priced at what it really moves (a probe = one 128-byte line)
            out["roofline"]["achieved_at_128B_per_probe"] = {
                "GBs": round(pm["bytes_per_launch"] / (a_ms * 1e-3) / 1e9, 1) if a_ms > 0 else None,
                "frac": round(pm["bytes_per_launch"] / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if a_ms > 0 else None,
                "what": "corrected PMC bytes per launch (from the committed counter pass) / this run's average launch duration"}
        # Which figures of the block were measured by THIS run (HIP events, wall clock, the library's counters) and which are
        # constants read from the committed counter passes (rocprofv3 --pmc runs of this same command on the builder's box; they
        # cannot be collected inside a timed run)
        rf = out["roofline"]
        rf["provenance"] = {
            "measured_this_run": ["achieved", "frac", "avg_launch_ms", "launches", "other_kernels_avg_ms", "candidates_per_launch", "uncovered_ranges",
                                  "uncovered_kmers", "one_probe_per_kmer_equivalent", "unpruned", "valu.wave_instr_per_s_per_cu", "valu.achieved (its time)",
                                  "achieved_at_128B_per_probe (its time)"],
            "from_committed_profiles": {
                "traffic": (pm or {}).get("source"),
                "achieved_at_128B_per_probe (its bytes)": (pm or {}).get("source"),
                "valu.valu_wave_instr_per_64_kmers": (valu.get("sq_source") or "built-in constant") + " (SQ_INSTS_VALU pass)" if valu and "kernel" in valu else None,
                "valu.mix, valu.peak": "profiles/r03_valu_mix.json (instruction classes counted from the ISA) priced at this run's measured issue rates"
                if valu and "kernel" in valu else None},
            "constants": {"peak": "MI355X_MICROARCH.md: HBM3E 8 TB/s", "algorithmic_bytes_per_base": "SURVEY.md 8(d): 64 B per probe; DESIGN.md 4.1"}}
        out["allocator"] = {"what": "one nts_mem_reserve at start, sized for the largest leg; every leg's genomes, filters and workspaces are cut from it",
                            "arena_planned_GB": round(arena_plan / 1e9, 2), "arena_reserved_GB": round(arena_got / 1e9, 2), "reserve_s": round(t_arena, 4),
                            # what this box charges for memory it has not handed out before (0.004 on a box whose memory had been used, 28 on one fresh from its driver)
                            "reserve_ms_per_GB": round(t_arena * 1e3 / max(arena_got / 1e9, 1e-9), 3) if arena_got else None,
                            "sketch_legs_after_the_reserve": ctx.mem_events_since(ev_start)}
        if cold:
            out["cold"] = cold
        if nruns:
            out["nruns"] = nruns
        if valley:
            out["valley"] = valley
        if rows:
            out["rows"] = rows
        if c4:
            out["c4"] = c4
            if world == 1:
                out["c4_n1"] = c4                                       # (the name rounds 2-4 gave the N = 1 point)
    if world == 1:
        sample = bf_np = None
        e2e_slices = e2e_par = None
        if not args.no_cpu_baseline:
            sample = genomes[0].download(0, min(genomes[0].total_bp, 100_000_000))
            bf_np = common.to_numpy()
            # the first 30 Mbp of every genome: the oracle pipeline's end-to-end sample
            e2e_slices = [g.download(0, min(int(g.rec_len[0]), 30_000_000)) for g in genomes]
            pa, _ = e2e_params(args, [f"syn{j}.fa" for j in range(len(genomes))], div)
            e2e_par = {"w_rounds": pa.w_rounds, "indel": pa.indel, "merge": pa.merge, "block_size": pa.block_size}
        workdir = args.e2e_dir or tempfile.mkdtemp(prefix="nts_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
        os.makedirs(workdir, exist_ok=True)
        try:
            c5 = None
            if name == "c3" and not args.no_c5_leg and args.family == "structural":
                for g in genomes:
                    g.free()
                common.free()
                genomes, common = [], None
                out["c5_like"], a5, div5 = c5 = c5_like_leg(args, ctx, local_rank, total_bp, contigs, workdir)
            if not args.no_e2e:
                for g in genomes:                                    # everything of the sketch legs goes (workspaces included: the
                    g.free()                                         # e2e legs' peak-memory figures are the pipeline's own), the
                if common is not None:                               # pipeline starts from files like a user's run
                    common.free()
                # This process gives its memory back before the runs below start in processes of their own.  Memory a GPU has not handed
                # out since its driver came up costs 7-28 ms per GB to allocate on the boxes this build ran on (`allocator.reserve_s`
                # above is this process paying that for its arena; scripts/malloc_probe.py), memory another process has just freed next
                # to nothing: with the arena kept, the first of the two runs below took its 66 GB from untouched memory (1.8 s in
                # hipMalloc on one box, 0.48 s on another) and the second from what the first had freed (0.002-0.12 s).  Freed first,
                # both see the same; what an untouched GPU adds to a run is the reserve rate above times the run's `reserved_GB`.
                ctx.sync()
                ev_close = ctx.mem_events()
                lib_ = ctx.lib
                ctx.close()                                          # (the process's last context: its workspaces go, then everything kept -- nts_destroy)
                from ntsynt_amd.device import mem_events_since
                args.parent_freed_GB = mem_events_since(lib_, ev_close)["GB_to_driver"]
                args.device_used_by_parent_GB = None
                if c5 is not None:
                    a5.parent_freed_GB, a5.device_used_by_parent_GB = args.parent_freed_GB, args.device_used_by_parent_GB
                    sub = os.path.join(workdir, "c5_like")
                    os.makedirs(sub, exist_ok=True)
                    try:
                        out["c5_like"]["e2e"] = e2e_leg(a5, local_rank, 3, total_bp, contigs, div5, sub)
                    finally:
                        shutil.rmtree(sub, ignore_errors=True)
                out["e2e"] = e2e_leg(args, local_rank, n_fam, total_bp, contigs, div, workdir)
        finally:
            if not args.e2e_dir:
                shutil.rmtree(workdir, ignore_errors=True)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(k, w, args.fpr, sample, bf_np, e2e_slices=e2e_slices, e2e_par=e2e_par)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


def bloom_roofline(total_bp, insert_ms):
    """The partitioned Bloom build (k_bin1 -> k_bin2 -> k_bin3) against HBM: the formulation's own bytes -- per k-mer
    0.25 B of bases in, a 4-byte level-1 residue out and in, a packed level-2 residue (2.9 B) out and in, 4.9 B of filter out: 19 B --
    over this run's average insert time, and next to it the bytes the counters saw per genome (rocprofv3 --pmc passes of this command,
    2 x FETCH_SIZE + WRITE_SIZE summed over the three kernels; committed, not collected in a timed run).  SURVEY.md 8(d)'s 129 B/base
    prices a sector read-modify-write per k-mer, which this build does not do: it is not this formulation's roofline."""
    alg = 19.0 * total_bp
    out = {"bound": "hbm", "formulation_bytes_per_kmer": 19.0, "achieved": round(alg / (insert_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(alg / (insert_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
           "limiter": "k_bin1 (half of the build's time) is co-limited by VALU issue and the LDS pipe, k_bin2 / k_bin3 stream at 4.4-5.5 TB/s (DESIGN.md 4.3)"}
    for path in (os.path.join(ROOT, "profiles", n_) for n_ in ("r05_bloom_pmc_traffic.json", "r04_pmc_traffic.json")):
        try:
            kern = json.load(open(path))["kernels"]
            tot = 0
            for name in ("k_bin1", "k_bin2", "k_bin3<true, false>"):       # (k_bin3: the store-only finish of an insert into an empty filter)
                e = next(v for n_, v in kern.items() if n_.startswith(name))
                tot += 2 * e["fetch_MB_per_launch_max"] * 1048576 + e["write_MB_per_launch_max"] * 1048576
            out["traffic"] = {"bytes_per_genome": int(tot), "GBs_at_this_runs_time": round(tot / (insert_ms * 1e-3) / 1e9, 1),
                              "frac": round(tot / (insert_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "ratio_to_formulation_bytes": round(tot / alg, 2),
                              "source": os.path.relpath(path, ROOT), "correction": "2 x FETCH_SIZE + WRITE_SIZE"}
            break
        except (OSError, KeyError, ValueError, StopIteration):
            continue
    return out


def pmc_traffic(name, pruned_run, algorithmic_bytes=None, line_bytes=None):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/rNN_pmc_traffic.json: rocprofv3
    --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of this same command), corrected as MI355X_MICROARCH.md's HBM
    section prescribes for gfx950: FETCH_SIZE tallies every 128-byte request at 64 B, so it is doubled (the probes of this
    kernel are 128-byte requests one and all: profiles/r02_probe_granularity.md; so are its 16-byte-per-lane LDS-DMA reads of
    the base words); WRITE_SIZE is taken as it is (uncalibrated in the guide)."""
    path = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_pmc_traffic.json") for r in (6, 5, 4, 3, 2)) if os.path.exists(q)), None)
    if name != "c3" or path is None:
        return None
    kernels = json.load(open(path))["kernels"]
    k = (kernels.get("k_hash_select_hi") or kernels.get("k_hash_select")) if pruned_run else kernels.get("k_hash<0>")
    if not k:
        return None
    fetch = k["fetch_MB_per_launch_max"] * 1024 * 1024
    write = k["write_MB_per_launch_max"] * 1024 * 1024
    out = {"bytes_per_launch": int(2 * fetch + write), "raw_fetch_size_bytes": int(fetch), "raw_write_size_bytes": int(write),
           "correction": "2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts 128-byte requests at 64 B)",
           "source": os.path.relpath(path, ROOT)}
    if algorithmic_bytes:
        out["ratio_to_algorithmic_bytes_at_64B_per_probe"] = round(out["bytes_per_launch"] / algorithmic_bytes, 2)
    if line_bytes:
        out["algorithmic_bytes_at_128B_per_probe"] = int(line_bytes)
        out["ratio_to_algorithmic_bytes_at_128B_per_probe"] = round(out["bytes_per_launch"] / line_bytes, 2)
    return out


if __name__ == "__main__":
    main()
