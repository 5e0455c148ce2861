"""In-process equivalent of ntSynt's Snakemake workflow (bin/ntsynt_run_pipeline.smk) on the GPU:

    faidx (smk:48-53) -> make_common_bf (smk:55-62) -> indexlr per genome (smk:74-85)
        -> ntsynt_run.py graph stage (smk:87-103)

with the same artefact names in the CWD: {basename}.fai, {prefix}.common.bf,
{basename}.k{k}.w{w}.tsv, {prefix}.synteny_blocks.tsv, {prefix}.pre-collinear-merge.synteny_blocks.tsv.
All sequence-scale compute runs in libntsynt_hip.so; there is no CPU fallback.

Multi-GPU (one process per GPU, torch.distributed over RCCL): genome g lives on rank
g mod N.  Exchange 1: every rank ANDs the filters of its own genomes, then a bitwise-AND all-reduce gives
the common filter on every rank.  Exchange 2: the owner of a genome sketches it and broadcasts the
minimizer list; the graph stage then runs replicated (identical, deterministic) on every rank and rank 0
writes the files.  The orchestration below is written against a small `backend` object so that the
schedule can be exercised on CPU (gloo) with test doubles; the product backend is GpuBackend."""
import math
import os
import time

import numpy as np

from . import fasta as fa
from .graph import edge_degrees
from .synteny import SyntenyEngine


BF_SIGNATURE = "[BTLKmerBloomFilter_v5]"


def bf_header(nbytes, k, hash_num=1, signature=BF_SIGNATURE):
    return (f"{signature}\nbytes = {nbytes}\nhash_fn = \"ntHash_v2\"\n"
            f"hash_num = {hash_num}\nk = {k}\n\n[HeaderEnd]\n").encode()


def write_bf(path, bits, k, hash_num=1, signature=BF_SIGNATURE):
    """btllib KmerBloomFilter file: TOML-style header + raw bit array.  The layout (table name `signature`, keys
    bytes / hash_fn / hash_num / k, terminator [HeaderEnd]) is recalled from btllib's BloomFilter::save, which is not in
    the reference tree: the file is NOT guaranteed to load in a stock btllib (`indexlr -s`).
    The table name is a parameter (`--bf-signature`) so that a maintainer holding a real btllib file can match it."""
    with open(path, "wb") as fh:
        fh.write(bf_header(bits.size, k, hash_num, signature))
        bits.tofile(fh)


def read_bf(path):
    "(bit array, k) of a filter file in write_bf's layout (the bits are read straight into one array: a 3 Gbp filter is 14.8 GB)"
    with open(path, "rb") as fh:
        head = fh.read(4096)
    if b"[HeaderEnd]\n" not in head:
        raise ValueError(f"{path}: no [HeaderEnd] line -- not a Bloom filter file in btllib's layout")
    end = head.index(b"[HeaderEnd]\n") + len(b"[HeaderEnd]\n")
    meta = {}
    for line in head[:end].decode().splitlines():
        if "=" in line:
            key, val = [x.strip() for x in line.split("=", 1)]
            meta[key] = val.strip('"')
    bits = np.fromfile(path, dtype=np.uint8, offset=end)
    if "bytes" in meta and int(meta["bytes"]) != bits.size:
        raise ValueError(f"{path}: header says {meta['bytes']} bytes, the file holds {bits.size}")
    if int(meta.get("hash_num", 1)) != 1:
        raise ValueError(f"{path}: {meta['hash_num']} hash functions; ntSynt's common filter has one (src/ntsynt_make_common_bf.cpp:18-19)")
    return bits, int(meta["k"])


write_indexlr_tsv = fa.write_indexlr_tsv      # native writer (csrc/nts_hostio.cpp)

MAX_W = 12000                  # nts_sketch: NTS_ERANGE beyond (csrc/ntsynt_hip.hip)
MAX_RECORDS = 1 << 22          # ntsynt_amd/synteny.py _new_round_graph: record * 2^40 + position keys
MAX_RECORD_BP = (1 << 40) - 1


class Stages:
    "wall-clock per stage (stands in for `--benchmark`'s /usr/bin/time wrappers, smk:26-35)"

    def __init__(self):
        self.rows = []
        self._t = None

    def start(self, name):
        self._t = (name, time.time())

    def stop(self):
        name, t0 = self._t
        self.rows.append((name, time.time() - t0))

    def mark(self, label):
        "a point on the run's time line (seconds since the first mark), for e2e breakdowns; with `mem` set, the HBM bytes live there"
        now = time.time()
        if not hasattr(self, "t0"):
            self.t0 = now
            self.marks = []
            self.hbm_marks = []
        self.marks.append((label, round(now - self.t0, 4)))
        if self.mem is not None:
            try:
                self.hbm_marks.append((label, self.mem()["live"]))
            except Exception:                                   # noqa: BLE001 -- a closed context: the time line still stands
                pass

    mem = None                  # callable -> dict(live=...) (device.Context.mem_stats), set by run() on the GPU backend

    def memory(self):
        """Peak memory of the run, the second figure the reference publishes per run (README.md:156-158; `--benchmark` records
        the RSS of every rule, smk:26-35): the library's HBM high-water mark (nts_mem_stats, all contexts of the process, since
        the run began) and the process's peak host RSS (ru_maxrss: since process start -- a monotone figure)."""
        import resource
        out = {"peak_host_rss_bytes": int(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss) * 1024}
        if self.mem is not None:
            try:
                m = self.mem()
                out["peak_hbm_bytes"] = int(m["peak"])
                out["hbm_live_at_marks"] = dict(self.hbm_marks)
            except Exception:                                   # noqa: BLE001
                pass
        return out

    def write(self, path, memory=None):
        with open(path, "w", encoding="utf-8") as fh:
            for name, dt in self.rows:
                fh.write(f"{name}\t{dt:.6f}\n")
            for name in ("peak_hbm_bytes", "peak_host_rss_bytes"):
                if memory and name in memory:
                    fh.write(f"{name}\t{memory[name]}\n")


class GpuBackend:
    """The product backend: every call lands in libntsynt_hip.so on this rank's GPU."""

    def __init__(self, device=0, ctx=None):
        from .device import Context
        self.own_ctx = ctx is None
        self.ctx = ctx or Context(device)
        self.device = self.ctx.device
        self.comm = None
        self._pool = None                  # device.SketchPool: large assemblies sketched at once
        self._batch = None                 # (key, Genome): the run's assemblies as one resident batch genome

    # genomes
    def read_host(self, path):
        "host half of load_genome: parse the FASTA (native reader, releases the GIL: safe to run in a thread)"
        return fa.read_fasta(path)

    def upload_host(self, recs):
        "device half of load_genome (one context, one stream: called from the main thread only)"
        from .device import Genome
        g = Genome(self.ctx, recs.names, recs.seq, recs.rec_off, recs.rec_len)
        g.recs = recs
        return g

    def load_genome(self, path):
        "FASTA file -> resident genome with the parse on the GPU (fasta.read_fasta_device); NTS_FASTA=host: the host reader"
        if os.environ.get("NTS_FASTA", "device") == "host":
            return self.upload_host(self.read_host(path))
        g, _ = fa.read_fasta_device(self.ctx, path)
        return g

    # multi-GPU: the two exchanges run in the library over RCCL (nts_bf_allreduce_and, nts_mx_allgather); the id of the
    # communicator travels through torch.distributed's default group (any backend: it carries 128 bytes and barriers).
    def init_comm(self):
        if self.comm is None:
            from .device import Comm
            self.comm = Comm.from_torch(self.ctx)

    def bf_new(self, nbytes, k, world=1, ones=False):
        from .device import BloomFilter
        return BloomFilter(self.ctx, nbytes, k, world=world, ones=ones)

    def bf_insert(self, bf, genome):
        bf.insert(genome)

    def bf_and(self, acc, other):
        acc.and_(other)

    def bf_insert_and(self, acc, genome):
        acc.insert_and(genome)

    def bf_clear(self, bf):
        bf.clear()

    def bf_fpr(self, bf):
        return bf.get_fpr()

    def bf_bits(self, bf):
        return bf.to_numpy()

    def and_into(self, a, b):
        from .device import and_raw
        and_raw(self.ctx, a.data_ptr(), b.data_ptr(), a.numel())
        self.ctx.sync()

    def sync(self):
        self.ctx.sync()

    # sketch + graph
    def sketch(self, genome, k, w, bf, masks=None):
        from .device import sketch
        mx = sketch(self.ctx, genome, k, w, bf, masks)
        out = mx.to_numpy()
        mx.free()
        return out

    # below this many bases per assembly the fixed cost of a launch sequence shows: sketch the assemblies as one batch
    BATCH_BELOW_BP = int(os.environ.get("NTS_BATCH_BELOW_BP", 1 << 30))     # (0: always one launch sequence per genome -- the path of 1 Gbp+ assemblies)

    def sketch_batch(self, genomes, k, w, bf, masks=None):
        """Sketches of several resident genomes with one sequence of launches (Genome.concat): the same lists as sketch()
        per genome.  masks[i]: hard-mask intervals (record, start, end) of genome i (a refinement round, row B5); they
        address the batch through its record numbering.  The batch genome stays resident for the run: the refinement
        rounds sketch it again."""
        from .device import Genome, sketch
        if len(genomes) < 2 or max(g.total_bp for g in genomes) >= self.BATCH_BELOW_BP:
            return [self.sketch(g, k, w, bf, masks[i] if masks else None) for i, g in enumerate(genomes)]
        key = tuple(id(g) for g in genomes)
        if self._batch is None or self._batch[0] != key:
            if self._batch is not None:
                self._batch[1].free()
            self._batch = (key, Genome.concat(self.ctx, genomes))
        batch = self._batch[1]
        joined = None
        if masks:
            joined = [(int(r) + int(batch.rec_base[i]), s, e) for i, m in enumerate(masks) for r, s, e in (m or [])]
        mx = sketch(self.ctx, batch, k, w, bf, joined)
        out = mx.to_numpy()
        mx.free()
        return batch.split_minimizers(*out)

    def sketch_dev(self, genomes, k, w, bf, masks=None, repeat=None):
        """sketch_batch with the lists left in HBM: [Minimizers] (one per genome) for the device-resident graph stage; a
        batch genome's list is taken apart on the device (nts_mx_split).  repeat: filter-out filter (indexlr -r), per genome."""
        from .device import Genome, SketchPool, sketch
        if repeat is not None or len(genomes) < 2 or max(g.total_bp for g in genomes) >= self.BATCH_BELOW_BP:
            # large assemblies: one launch sequence per genome; NTS_SKETCH_POOL=3: up to three genomes at once on contexts of their
            # own (device.SketchPool: +6 % sketch throughput at 3 x 3 Gbp for ~4 GB of workspaces per context; off by default)
            n_pool = min(len(genomes), int(os.environ.get("NTS_SKETCH_POOL", "1")))
            if n_pool > 1 and repeat is None:
                if self._pool is None:
                    self._pool = SketchPool(self.ctx, n_pool)
                    mode, c = getattr(self.ctx, "_sketch_mode", ("auto", 0))     # the pool's contexts follow the main one's policy
                    self._pool.configure(lambda cx: cx is self.ctx or cx.sketch_mode(mode, c))
                return self._pool.sketch(genomes, k, w, bf, masks)
            return [sketch(self.ctx, g, k, w, bf, masks[i] if masks else None, repeat=repeat) for i, g in enumerate(genomes)]
        key = tuple(id(g) for g in genomes)
        if self._batch is None or self._batch[0] != key:
            if self._batch is not None:
                self._batch[1].free()
            self._batch = (key, Genome.concat(self.ctx, genomes))
        batch = self._batch[1]
        joined = None
        if masks:
            if all(isinstance(m, np.ndarray) for m in masks):   # Interval arrays: shift the record numbers, concatenate
                parts_iv = []
                for i, m in enumerate(masks):
                    m = m.copy()
                    m["rec"] += np.uint32(batch.rec_base[i])
                    parts_iv.append(m)
                joined = np.concatenate(parts_iv) if parts_iv else None
            else:
                joined = [(int(r) + int(batch.rec_base[i]), s, e) for i, m in enumerate(masks) for r, s, e in (m if m is not None else [])]
        mx = sketch(self.ctx, batch, k, w, bf, joined)
        parts = mx.split(batch.rec_base)
        mx.free()
        return parts

    def exchange_dev(self, local, n_total):
        "exchange 2 on device handles: {genome index: Minimizers} of this rank -> [Minimizers of every genome], all in HBM"
        ids = sorted(local)
        return self.comm.allgather_minimizers([local[i] for i in ids], ids, n_total)

    def graph(self, lists, keeps, list_ids):
        from .graph import build_graph_device
        return build_graph_device(self.ctx, lists, keeps, list_ids)

    def walk(self, nv, eu, ev, e_alive=None, key=None):
        from .graph import walk_paths
        return walk_paths(nv, eu, ev, e_alive, key)
    walk.oriented = True

    def allreduce_and(self, bf):
        "exchange 1: common filter = AND over the ranks' filters, in place"
        self.ctx.sync()
        self.comm.allreduce_and(bf)

    def exchange_lists(self, local, n_total):
        """exchange 2: {genome index: (h1, rec, pos)} of this rank -> the lists of all genomes on every rank, as ONE
        device all-gather (nts_mx_allgather)."""
        from .device import Minimizers
        ids = sorted(local)
        mine = [Minimizers.from_numpy(self.ctx, *local[i]) for i in ids]
        everything = self.comm.allgather_minimizers(mine, ids, n_total)
        out = [m.to_numpy() for m in everything]
        for m in mine + everything:
            m.free()
        return out

    def close(self):
        if self._batch is not None:
            self._batch[1].free()
            self._batch = None
        if self.comm is not None:
            self.comm.close()
            self.comm = None
        if self._pool is not None:
            self._pool.close()
            self._pool = None
        if self.own_ctx:
            self.ctx.close()


def shard_plan(rec_lens, world):
    """Fewer genomes than GPUs (last paragraph): genome g is worked on by the ranks r with r mod G == g (its group),
    each taking a range of the genome's records -- windows never cross records, and records are what the reference parallelises
    over (src/ntsynt_make_common_bf.cpp:128-131,145-153) -- balanced by bases.  rec_lens[g]: record lengths of genome g.
    Returns (group_of[rank], ranges[rank] = (genome, shard number, rec0, rec1)); a genome with fewer records than ranks leaves
    ranks with an empty range (they contribute nothing to the OR and an empty list)."""
    G = len(rec_lens)
    assert 0 < G < world
    group_of = [r % G for r in range(world)]
    ranges = [None] * world
    for g in range(G):
        ranks = [r for r in range(world) if r % G == g]
        lens = np.asarray(rec_lens[g], dtype=np.float64)
        cum = np.concatenate(([0.0], np.cumsum(lens)))
        cuts = [0]
        for s_ in range(1, len(ranks)):
            at = int(np.searchsorted(cum, cum[-1] * s_ / len(ranks), side="left"))
            cuts.append(min(max(at, cuts[-1]), lens.size))
        cuts.append(lens.size)
        for s_, r in enumerate(ranks):
            ranges[r] = (g, s_, cuts[s_], cuts[s_ + 1])
    return group_of, ranges


def partition_plan(rec_lens, world):
    """A family's records shared out over the ranks by BASES, across genome boundaries: the records of all genomes in
    family order are cut into `world` consecutive ranges at the record boundaries nearest to the multiples of total / world -- three
    3 Gbp genomes on eight GPUs are eight ranges of ~1.125 Gbp, not 3 / 3 / 2 ranks per genome.  Records are the unit (windows never cross
    them, and records are what the reference parallelises over, src/ntsynt_make_common_bf.cpp:128-131,145-153).  rec_lens[g]: record
    lengths of genome g.  Returns parts[rank] = [(genome, rec0, rec1), ...] in family order: every record in exactly one part, a
    rank's parts belong to different genomes; a rank may be left without any (more ranks than records)."""
    n_rec = [len(x) for x in rec_lens]
    flat = np.concatenate([np.asarray(x, dtype=np.float64) for x in rec_lens]) if sum(n_rec) else np.zeros(0)
    cum = np.concatenate(([0.0], np.cumsum(flat)))
    cuts = [0]
    for r in range(1, world):
        target = cum[-1] * r / world
        at = int(np.searchsorted(cum, target, side="left"))
        if at > 0 and (at >= cum.size or target - cum[at - 1] <= cum[at] - target):
            at -= 1                                              # the boundary before the target is the nearer one
        cuts.append(min(max(at, cuts[-1]), flat.size))
    cuts.append(flat.size)
    g_start = np.concatenate(([0], np.cumsum(n_rec))).astype(np.int64)
    parts = []
    for r in range(world):
        mine = []
        for g in range(len(rec_lens)):
            lo, hi = max(cuts[r], int(g_start[g])), min(cuts[r + 1], int(g_start[g + 1]))
            if lo < hi:
                mine.append((g, lo - int(g_start[g]), hi - int(g_start[g])))
        parts.append(mine)
    return parts


def partition_groups(parts, n_rec):
    """The filters of exchange 1 under partition_plan.  A rank builds ONE filter for the genomes it holds whole (their cascade is local:
    insert, then insert_and) and one per genome it holds a part of.  Returns (filters_of[rank] = [(group label, [part indices])], slot_group[rank] =
    [dense group numbers], n_groups): a genome in parts is a group (OR over its parts' filters), a rank's whole genomes together are a
    group of their own; the common filter is the AND over the groups (nts_bf_allreduce_parts)."""
    filters_of, labels = [], []
    for r, mine in enumerate(parts):
        whole = [i for i, (g, a, b) in enumerate(mine) if a == 0 and b == n_rec[g]]
        fs = []
        if whole:
            fs.append((("whole", r), whole))
        for i, (g, a, b) in enumerate(mine):
            if i not in whole:
                fs.append((("genome", g), [i]))
        filters_of.append(fs)
        for lab, _ in fs:
            if lab not in labels:
                labels.append(lab)
    dense = {lab: i for i, lab in enumerate(labels)}
    return filters_of, [[dense[lab] for lab, _ in fs] for fs in filters_of], len(labels)


def shard_masks(masks, rec0, rec1):
    "the hard-mask intervals of a genome that fall into records [rec0, rec1), renumbered for the slice (nts_interval arrays or (rec, start, end) tuples)"
    if masks is None:
        return None
    if isinstance(masks, np.ndarray):
        m = masks[(masks["rec"] >= rec0) & (masks["rec"] < rec1)].copy()
        m["rec"] -= np.uint32(rec0)
        return m
    return [(int(r) - rec0, s_, e) for r, s_, e in masks if rec0 <= int(r) < rec1]


def load_genomes(backend, paths, max_threads=8):
    """FASTA files -> resident genomes.  The files are parsed concurrently on host threads (the reference runs one
    indexlr/faidx process per file under Snakemake); uploads happen in order on the caller's thread while later
    files are still being read."""
    if len(paths) < 2 or not hasattr(backend, "read_host") or \
            (isinstance(backend, GpuBackend) and os.environ.get("NTS_FASTA", "device") != "host"):
        return {p: backend.load_genome(p) for p in paths}      # (the device parse spreads one file over host threads itself)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(max_threads, len(paths))) as pool:
        pending = [pool.submit(backend.read_host, p) for p in paths]
        return {p: backend.upload_host(f.result()) for p, f in zip(paths, pending)}


class _Arriving(dict):
    """{path: genome} whose values arrive from a loader thread: a lookup waits for that file.  On one GPU the common
    filter is built while the later files are still on their way up (the inserts run on the compute stream, the uploads on
    the loader's context: copy engines and host threads) -- the filter file, which is what the run waits for in the end,
    gets started that much earlier."""

    def __init__(self, paths, loaders, on_arrival):
        "loaders: one callable per loader thread (each with a context of its own); file i goes to loader i mod len(loaders)"
        super().__init__()
        from concurrent.futures import ThreadPoolExecutor
        self._pools = [ThreadPoolExecutor(max_workers=1) for _ in loaders]
        self._order = list(paths)

        def task(load, p):
            g = load(p)
            on_arrival(p, g)
            return g
        self._fut = {p: self._pools[i % len(loaders)].submit(task, loaders[i % len(loaders)], p) for i, p in enumerate(paths)}

    def __getitem__(self, p):
        if not dict.__contains__(self, p):
            dict.__setitem__(self, p, self._fut[p].result())
        return dict.__getitem__(self, p)

    def wait_all(self):
        for p in self._order:
            self[p]
        for pool in self._pools:
            pool.shutdown(wait=True)


def _exchange_lists(backend, local, n_total):
    """Exchange 2: every rank contributes the minimizer lists of its own genomes ({genome index: (h1,
    rec, pos)}) and receives those of all genomes, in one all-gather -- on the GPUs through the backend
    (nts_mx_allgather over RCCL), otherwise (test doubles, verification mode) as host objects."""
    import torch.distributed as dist
    if hasattr(backend, "exchange_lists"):
        out = backend.exchange_lists(local, n_total)
        if out is not None:
            return out
    box = [None] * dist.get_world_size()
    dist.all_gather_object(box, {i: tuple(np.ascontiguousarray(x) for x in v) for i, v in local.items()})
    merged = {i: v for part in box for i, v in part.items()}
    return [merged[i] for i in range(n_total)]


def plan_bytes(file_bytes, fpr, build_filter, w=1000):
    """Device bytes a run over files of these sizes will have live at its peak (an upper estimate from the files alone: a base per byte,
    four per byte of a .gz): the filter (39.5 bits per base of the first file in sorted order at fpr 0.025: A1), the Bloom build's two
    bucket arrays and bypass list (12 bytes per k-mer of the largest genome), the resident genomes with their 2-bit images and tables
    (1.3 bytes per base), sketch and graph workspaces (growing with the sketch density 2 / (w + 1)).  3 x 3 Gbp at w = 1000: 66.5 GB planned, 60.2 GB
    measured (bench.py e2e)."""
    import math
    if not file_bytes:
        return 0
    n0, n_max, total = file_bytes[0], max(file_bytes), sum(file_bytes)
    plan = 1.3 * total + (2 << 30)
    # the lists and the graph build grow with the sketch's density: ~2 / (w + 1) minimizers per base, ~100 bytes each through nts_engine_add's
    # input columns, sort buffers and vertex tables (nothing to speak of at the default w = 1000; 9 GB for 3 x 1 Gbp at w = 64)
    plan += 100.0 * total * 2.0 / (max(int(w), 1) + 1)
    if build_filter:
        plan += math.ceil(-n0 / math.log(1 - fpr)) / 8 + 12 * n_max
    return int(plan)


def reserve_for_run(ctx, fastas, fpr, build_filter, log=None, w=1000):
    """nts_mem_reserve of what plan_bytes says, less what the process's allocation cache already holds, started on a thread of its own;
    returns an object whose join() gives the bytes reserved (0: nothing asked, or the driver refused -- the run then allocates as it goes)"""
    class _Nothing:
        def join(self):
            return 0
    if os.environ.get("NTS_RESERVE", "1") == "0":
        return _Nothing()
    sizes = []
    for p in sorted(fastas):
        try:
            sz = os.path.getsize(p)
        except OSError:
            return _Nothing()
        sizes.append(sz * 4 if p.endswith(".gz") else sz)
    want = plan_bytes(sizes, fpr, build_filter, w) - ctx.mem_cache_stats()[0]
    if want < (256 << 20):                                      # small runs: the driver's latency is not what they wait for
        return _Nothing()
    return ctx.mem_reserve_async(want)


def run(fastas, k=24, w=1000, fpr=0.025, prefix=None, w_rounds=(100, 10), indel=10000, merge=10000,
        block_size=500, common=True, simplify=True, device=0, write_mx_tsv=True, mx_with_seq=True,
        benchmark=False, log=print, ctx=None, backend=None, bf_rounding="up", bf_signature=BF_SIGNATURE, dev=False, interarrivals=False, repeat=False,
        mx_tsvs=None, common_file=None, m=90, n=0, initial_only=False, write_fai=True, engine="device", refine_repeat_file=None, screen_repeat_file=None):
    """FASTA paths -> engine (outputs in .outputs and in the CWD).  Mirrors oracle.synteny_oracle.run_pipeline's
    signature so the parity tests read alike.  Under torch.distributed (WORLD_SIZE > 1, process group already
    initialised by the caller) genomes are sharded over the ranks.

    The reference's stage 3 on its own (bin/ntsynt_run.py, rule ntsynt_synteny smk:87-103): mx_tsvs = the minimizer TSV of
    every assembly (aligned with `fastas`: read instead of sketched, ntJoin's read_minimizers) and common_file = the
    `--common` filter file (uploaded instead of built; None with common=False: the refinement rounds sketch unfiltered);
    m / n: ntsynt_run.py's -m / -n."""
    prefix = prefix or f"ntSynt.k{k}.w{w}"
    if mx_tsvs is not None and len(mx_tsvs) != len(fastas):
        raise ValueError("one minimizer TSV per FASTA file")
    world, rank = 1, 0
    dist = None
    import sys
    if "torch" in sys.modules:       # a process group can only exist if the caller imported torch (saves ~1 s otherwise)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            world, rank = dist.get_world_size(), dist.get_rank()
    own_backend = backend is None
    backend = backend or GpuBackend(device, ctx)
    st = Stages()
    if isinstance(backend, GpuBackend):
        lib = backend.ctx.lib                                  # (ctx = NULL: live / peak of the process, readable after the contexts closed)
        lib.nts_mem_reset_peak()

        def _mem():
            import ctypes
            a, b = ctypes.c_uint64(), ctypes.c_uint64()
            lib.nts_mem_stats(None, ctypes.byref(a), ctypes.byref(b), None, None)
            return {"live": a.value, "peak": b.value}
        st.mem = _mem
    st.mark("start")
    if world > 1 and hasattr(backend, "init_comm"):
        backend.init_comm()
    owner = {p: i % world for i, p in enumerate(fastas)}
    mine = [p for p in fastas if owner[p] == rank]
    # genomes that do not deal out evenly over the GPUs (three on eight, three on two): the family's records are shared out by bases,
    # across genome boundaries (partition_plan); the device engine only
    shard_mode = (world > 1 and len(fastas) % world != 0 and isinstance(backend, GpuBackend) and engine != "host"
                  and mx_tsvs is None and not repeat and os.environ.get("NTS_SHARD_RECORDS", "1") != "0")
    if shard_mode:
        # before the plan exists (it needs every genome's record lengths) a rank loads the genome its range most likely starts in;
        # with more genomes than ranks, a consecutive block of them -- every genome is loaded somewhere
        G_ = len(fastas)
        mine = [fastas[(rank * G_) // world]] if world >= G_ else [p for i, p in enumerate(fastas) if (i * world) // G_ == rank]

    reserving = []

    def join_reserve():
        if reserving:
            st.reserved_bytes = reserving.pop().join()
            st.mark("memory_reserved")
    if isinstance(backend, GpuBackend) and not (mx_tsvs is not None and initial_only):
        # the run's device memory in ONE driver allocation, before its first file is opened (nts_mem_reserve): the filter, the Bloom
        # build's workspaces and the resident genomes are cut from it, and nothing goes back to the driver while the run lasts
        # (the reference allocates its two filters once: src/ntsynt_make_common_bf.cpp:121-137)
        # On a thread of its own: memory the GPU has not handed out since the driver came up costs 7-20 ms per GB on the boxes this
        # build ran on (0.5 s for a 3 x 3 Gbp run's 65 GB; scripts/malloc_probe.py, bench.py `allocator`), time the first file's ingest
        # can share.  Joined before the filter is allocated.
        reserving.append(reserve_for_run(backend.ctx, fastas if world == 1 else mine, fpr, common and common_file is None, log, w=w))
    # limits of this implementation, checked before anything is written (the reference has none of them)
    for ww in [w] + list(w_rounds):
        if not 1 <= int(ww) <= MAX_W:
            raise ValueError(f"window size {ww} outside 1..{MAX_W} (limit of the window kernels)")
    st.start("read_fasta+upload")

    refused = []                                                # (several ranks: said by every rank after the first exchange, so that none waits for one that left)

    def refuse(msg):
        if world > 1:
            refused.append(msg)
        else:
            raise ValueError(msg)

    def arrived(p, g):
        if len(g.names) == 0 or g.total_bp == 0:
            return refuse(f"{p}: no sequence records")
        rl = getattr(g, "rec_len", None)
        if len(g.names) >= MAX_RECORDS or (rl is not None and len(rl) and int(np.max(rl)) >= MAX_RECORD_BP):
            return refuse(f"{p}: more than 2^22 records or a record of 2^40 bases or more "
                          "(limits of the refinement rounds' composite interval keys)")
        if write_fai and not shard_mode:                                    # (shard mode: the first rank that holds records of a genome writes for it, below)
            fa.write_fai(f"{fa.basename(p)}.fai", g.recs)

    # stage 3's initial round alone needs no sequence: record ids come from the minimizer files
    tsv_data = [fa.read_indexlr_tsv(t) for t in mx_tsvs] if mx_tsvs is not None else None
    skip_genomes = mx_tsvs is not None and initial_only
    overlap_load = (world == 1 and common and common_file is None and len(mine) > 1 and isinstance(backend, GpuBackend) and not skip_genomes
                    and os.environ.get("NTS_FASTA", "device") != "host" and os.environ.get("NTS_LOAD_OVERLAP", "1") != "0")
    if skip_genomes:
        genomes = {}
        common, common_file = False, None
    elif overlap_load:
        # files in the order the filter takes them (sorted: src/ntsynt_make_common_bf.cpp:105-107), on a context of their own
        from .device import Context
        n_loaders = max(1, min(int(os.environ.get("NTS_LOADERS", "1")), len(mine)))
        load_ctxs = [Context(backend.device, backend.ctx.variant) for _ in range(n_loaders)]      # (the build of the library the run's context is from: one allocation cache)
        genomes = _Arriving(sorted(mine), [(lambda p, c=c: fa.read_fasta_device(c, p)[0]) for c in load_ctxs], arrived)
        try:
            genomes[sorted(mine)[0]]                           # the first one sizes the filter
        except BaseException:
            for pool_ in genomes._pools:                       # (a loader failed: nothing of its context may outlive the call)
                pool_.shutdown(wait=True, cancel_futures=True)
            for c in load_ctxs:
                c.close()
            raise
    else:
        genomes = load_genomes(backend, mine)
        for p in mine:
            arrived(p, genomes[p])
    st.mark("first_genome_resident" if overlap_load else "genomes_resident")
    # record names and sizes are needed everywhere (output text, filter sizing)
    if skip_genomes:
        meta = {p: (tsv_data[i][0], 0) for i, p in enumerate(fastas)}
    else:
        meta = {p: (genomes[p].names, genomes[p].total_bp) for p in mine} if not overlap_load else None
    if world > 1:
        gathered = [None] * world
        mine_meta = {p: v + (np.asarray(genomes[p].rec_len, dtype=np.uint64),) for p, v in meta.items()} if shard_mode else dict(meta)
        mine_meta["\0refused"] = list(refused)
        dist.all_gather_object(gathered, mine_meta)
        all_refused = sorted({msg for part in gathered for msg in part.pop("\0refused")})
        if all_refused:                                         # an input no rank can work with: every rank says so and stops (a genome
            raise ValueError("; ".join(all_refused))            # without records would otherwise be missing from the plan: ADVICE r5)
        meta = {p: v for part in gathered for p, v in part.items()}
    shard = None
    if shard_mode:
        n_rec = [len(meta[p][2]) for p in fastas]
        plan = partition_plan([meta[p][2] for p in fastas], world)
        filters_of, slot_group, n_groups = partition_groups(plan, n_rec)
        my_parts = plan[rank]
        leader_of = {}                                          # genome -> the first rank that holds records of it: keeps the whole genome
        for r_, ps in enumerate(plan):                          # (k-mer text of the minimizer TSV) and writes the genome's files
            for g_, _, _ in ps:
                leader_of.setdefault(g_, r_)
        need = {g_ for g_, _, _ in my_parts}
        for g_ in sorted(need):                                 # what the first guess did not bring
            if fastas[g_] not in genomes:
                genomes.update(load_genomes(backend, [fastas[g_]]))
                arrived(fastas[g_], genomes[fastas[g_]])
        subs = []
        for g_, a, b in my_parts:
            whole_g = genomes[fastas[g_]]
            subs.append(whole_g if (a == 0 and b == n_rec[g_]) else whole_g.slice(a, b, backend.ctx))
        for p in list(genomes):                                 # a genome stays whole only with its leader or where a part is the whole of it
            g_ = fastas.index(p)
            keep = leader_of.get(g_) == rank or any(sub is genomes[p] for sub in subs)
            if not keep:
                genomes.pop(p).free()
        if write_fai:
            for g_, r_ in leader_of.items():
                if r_ == rank:
                    fa.write_fai(f"{fa.basename(fastas[g_])}.fai", genomes[fastas[g_]].recs)
        # exchange 2 numbers the parts of all ranks in family order
        part_base = int(sum(len(ps) for ps in plan[:rank]))
        shard = {"parts": my_parts, "subs": subs, "plan": plan, "filters_of": filters_of, "slot_group": slot_group, "n_groups": n_groups,
                 "n_slots": max(1, max(len(fs) for fs in filters_of)), "list_slots": max(1, max(len(ps) for ps in plan)),
                 "part_base": part_base, "n_parts": int(sum(len(ps) for ps in plan)), "leader_of": leader_of}
        st.mark("shard_resident")
    st.stop()

    bf = None
    # artefacts nobody downstream of us reads (the filter file, the minimizer TSVs) are written behind the next stages
    from concurrent.futures import ThreadPoolExecutor
    writers = ThreadPoolExecutor(max_workers=4)
    pending_files = []
    file_of = {}                         # writer -> the file it leaves behind

    def submit_file(name, fn, *a):
        fut = writers.submit(fn, *a)
        pending_files.append(fut)
        file_of[fut] = name
    filter_save_started = []

    def start_filter_save():
        """the filter file, behind everything that follows: straight out of HBM on the library's own copy threads where the filter has a
        save() (the GPU backend), else device -> host copy + write.  Called as soon as the common filter stands (the file is 0.7 of a
        run's wall clock: the loaders' clean-up, 50 ms of pinned-memory frees, used to come first) and again, to no effect, further down."""
        if filter_save_started or bf is None or rank != 0 or common_file is not None:
            return
        filter_save_started.append(True)
        if hasattr(bf, "save"):
            def save_filter():
                st.mark("bf_save_begin")
                bf.save(f"{prefix}.common.bf", bf_header(bf.bytes, k, signature=bf_signature))
                st.mark("bf_save_end")
            submit_file(f"{prefix}.common.bf", save_filter)
        else:
            submit_file(f"{prefix}.common.bf", lambda: write_bf(f"{prefix}.common.bf", backend.bf_bits(bf), k, signature=bf_signature))

    if common_file is not None:
        # stage 3 on its own: `--common <file>` (ntsynt_run.py:23; consumed by the refinement rounds' indexlr -s, S:175-177)
        if world > 1 or not isinstance(backend, GpuBackend):
            raise ValueError("a filter file is read on one GPU")
        st.start("load_common_bf")
        bits, k_file = read_bf(common_file)
        if k_file != k:
            raise ValueError(f"{common_file}: built for k = {k_file}, this run uses k = {k}")
        bf = backend.bf_new(bits.size, k)
        bf.from_numpy(bits)
        del bits
        st.stop()
        common = False
    if refine_repeat_file is not None and screen_repeat_file is not None:
        raise ValueError("the repeat filter serves either the refinement sketches (--filter Indexlr) or the reading of the lists (--filter Filter)")
    screen_lists = screen_repeat_file is not None
    if screen_lists:
        # stage 3's `--filter Filter --repeat <file>` (S:183-184,601-604): ntJoin's read_minimizers leaves out the minimizers whose k-mer the
        # filter holds -- of the initial files and of every refinement round's lists (nts_mx_screen); the sketches themselves are unscreened
        if mx_tsvs is None or initial_only:
            raise ValueError("lists are screened in stage 3 on given minimizer files, with the FASTA files read")
        refine_repeat_file = screen_repeat_file
    refine_rep = None
    if refine_repeat_file is not None:
        # stage 3's `--filter Indexlr --repeat <file>` (bin/ntsynt_synteny.py:172-180): the refinement rounds' indexlr runs get `-r <file>`
        # (next to `-s <common>` when there is one); the initial lists come from the files as they are
        if world > 1 or not isinstance(backend, GpuBackend):
            raise ValueError("a filter file is read on one GPU")
        st.start("load_repeat_bf")
        bits, k_file = read_bf(refine_repeat_file)
        if k_file != k:
            raise ValueError(f"{refine_repeat_file}: built for k = {k_file}, this run uses k = {k}")
        refine_rep = backend.bf_new(bits.size, k)
        refine_rep.from_numpy(bits)
        del bits
        st.stop()
    if common:
        st.start("make_common_bf")
        ordered = sorted(fastas)                               # src/ntsynt_make_common_bf.cpp:105-107
        from .device import bf_size_bytes
        first_bp = meta[ordered[0]][1] if meta is not None else genomes[ordered[0]].total_bp
        approx, nbytes = bf_size_bytes(first_bp, fpr, bf_rounding)
        log(f"Genome size (bp): {first_bp}")
        log(f"BF size (bytes): {approx}")
        my_sorted = [p for p in ordered if owner[p] == rank] if not shard_mode else []
        join_reserve()                                        # (the filter and the build's workspaces are cut from the reserved memory)
        bf = backend.bf_new(nbytes, k, world, ones=(world > 1 and not my_sorted and not shard_mode))
        st.mark("bf_allocated")
        part_filters = [bf]
        if shard_mode:
            # one filter for the genomes this rank holds whole (their cascade is local), one per genome it holds a part of: a part's
            # k-mers are OR-ed with the genome's other parts in the exchange, the genomes AND-ed (nts_bf_allreduce_parts)
            for fi, (_, idxs) in enumerate(shard["filters_of"][rank]):
                f_ = bf if fi == 0 else backend.bf_new(nbytes, k, world)
                if fi:
                    part_filters.append(f_)
                backend.bf_insert(f_, shard["subs"][idxs[0]])
                for i_ in idxs[1:]:
                    backend.bf_insert_and(f_, shard["subs"][i_])
        if my_sorted:
            backend.bf_insert(bf, genomes[my_sorted[0]])
            st.mark("bf_first_insert")
            if world == 1:
                log(f"Bloom filter FPR: {backend.bf_fpr(bf)}")
            if len(my_sorted) > 1 and hasattr(backend, "bf_insert_and"):
                for p in my_sorted[1:]:
                    backend.bf_insert_and(bf, genomes[p])       # bf &= G_i inside the build's last pass: AND == cascade level
                    if world == 1:
                        log(f"Bloom filter FPR: {backend.bf_fpr(bf)}")
            elif len(my_sorted) > 1:                            # (test doubles)
                tmp = backend.bf_new(nbytes, k)
                for p in my_sorted[1:]:
                    backend.bf_clear(tmp)
                    backend.bf_insert(tmp, genomes[p])
                    backend.bf_and(bf, tmp)
                    if world == 1:
                        log(f"Bloom filter FPR: {backend.bf_fpr(bf)}")
                if hasattr(tmp, "free"):
                    tmp.free()
        if world > 1:
            if shard_mode:
                backend.ctx.sync()
                backend.comm.allreduce_parts(part_filters, shard["slot_group"], shard["n_slots"], shard["n_groups"])
                for f_ in part_filters[1:]:
                    f_.free()
            else:
                backend.allreduce_and(bf)                      # GpuBackend: nts_bf_allreduce_and; test doubles bring their own
        log(f"Final Bloom filter FPR: {backend.bf_fpr(bf)}")
        if isinstance(backend, GpuBackend):
            backend.ctx.trim_bf_build()                       # (the run builds no further filter: the buckets serve what is allocated next)
        st.stop()
        st.mark("common_filter_done")
        start_filter_save()
    if overlap_load:
        try:
            genomes.wait_all()
        finally:
            # the loaders are done (or failed: wait_all re-raises a loader's exception): their contexts (the raw image of the
            # largest file in HBM, pinned staging) go either way
            for pool_ in genomes._pools:
                pool_.shutdown(wait=True)
            for p in mine:
                if dict.__contains__(genomes, p):
                    dict.__getitem__(genomes, p).ctx = backend.ctx      # the genomes belong to the run's context from here on
            for c in load_ctxs:
                c.close()
        meta = {p: (genomes[p].names, genomes[p].total_bp) for p in mine}
        st.mark("genomes_resident")
    elif isinstance(backend, GpuBackend) and not skip_genomes:
        backend.ctx.trim_ingest()

    # The reference's experimental repeat filter (config "repeat": rules make_repeat_bf and indexlr -r, smk:65-85): k-mers seen
    # twice within a genome, excluded from the whole-genome sketches; the refinement rounds do not use it (ntsynt_run.py gets
    # --repeat without --filter: S:172-180).  One GPU, device engine.
    rep_bf = None
    if repeat:
        if world > 1 or not isinstance(backend, GpuBackend):
            raise ValueError("the repeat filter is served on one GPU only")
        from .device import BloomFilter
        size_bits = math.ceil((-1 * genomes[fastas[0]].total_bp) / (math.log(1 - fpr)))       # ntsynt_make_repeat_bfs.py:25-34
        rep_bytes = (int(size_bits / 8) + 7) // 8 * 8                                         # + btllib's constructor rounding (u1)
        rep_bf = BloomFilter(backend.ctx, rep_bytes, k)
        own = BloomFilter(backend.ctx, rep_bytes, k)
        for p in fastas:                                                                      # :53-67, genomes in the order given
            own.clear()
            rep_bf.insert_repeats_of(genomes[p], own)
        own.free()
        rep_bf.save(f"{prefix}.repeat.bf", bf_header(rep_bytes, k, signature=bf_signature))

    # Graph stage: resident in HBM (ntsynt_amd/synteny_device.py) on the GPU backend.  engine="host": its host-array twin
    # (ntsynt_amd/synteny.py, the class the device engine derives from) for a whole run -- an argument of this function that the
    # lockstep tests and the CPU test doubles use, not a switch a user of the command line has.
    device_engine = isinstance(backend, GpuBackend) and engine != "host"
    if not device_engine and (mx_tsvs is not None or initial_only):
        raise ValueError("minimizer TSVs as input (stage 3 on its own) and initial_only are served by the device engine only")
    if rep_bf is not None and not device_engine:
        raise ValueError("the repeat filter is served by the device engine only")
    tsv_names = [f"{fa.basename(p)}.k{k}.w{w}.tsv" for p in fastas]
    # stage 3 on given minimizer files: the engine orders the assemblies -- and names them in the output -- by the paths the caller
    # passed, as the reference does (sorted(self.args.FILES, reverse=True), bin/ntsynt_synteny.py:34; synteny_block.py:14,76-77)
    engine_files = list(mx_tsvs) if mx_tsvs is not None else tsv_names
    if rank != 0:                       # replicas compute, only rank 0 leaves files behind
        scratch = os.path.join(os.getcwd(), f".ntsynt_rank{rank}")
        os.makedirs(scratch, exist_ok=True)
        out_prefix = os.path.join(scratch, os.path.basename(prefix))
    else:
        out_prefix = prefix

    start_filter_save()

    join_reserve()                                            # (runs without a filter build get here with the reservation still under way)
    if device_engine:
        from .synteny_device import DeviceSyntenyEngine
        mine_idx = [i for i, p in enumerate(fastas) if owner[p] == rank]
        if shard_mode:
            mine_idx = sorted(g_ for g_, r_ in shard["leader_of"].items() if r_ == rank)      # (the genomes whose minimizer TSV this rank writes)

        def sketch_dev_round(masks_by_asm, new_w):
            "device lists of all assemblies: every rank sketches its own genomes (one batch when they are small), one all-gather"
            if shard_mode:
                # this rank's parts; the all-gather hands every rank every part's list (numbered in family order), the parts of a
                # genome are strung together in record order
                from .device import Minimizers
                ms = None
                if masks_by_asm is not None:
                    ms = [shard_masks(masks_by_asm[g_], a, b) for g_, a, b in shard["parts"]]
                mine_l = backend.sketch_dev(shard["subs"], k, new_w, bf, ms) if shard["subs"] else []
                ids = [shard["part_base"] + i for i in range(len(mine_l))]
                parts = backend.comm.allgather_minimizers(mine_l, ids, shard["n_parts"], shard["list_slots"])
                for m_ in mine_l:
                    m_.free()
                out, at = {}, 0
                by_genome = {}
                for ps in shard["plan"]:
                    for g_, a, b in ps:
                        by_genome.setdefault(g_, []).append((parts[at], a))
                        at += 1
                for g_ in range(len(fastas)):
                    out[g_] = Minimizers.concat(backend.ctx, [m_ for m_, _ in by_genome[g_]], [a for _, a in by_genome[g_]])
                for m_ in parts:
                    m_.free()
                return out
            ml = [masks_by_asm[i] for i in mine_idx] if masks_by_asm is not None else None
            if rep_bf is not None and masks_by_asm is None:
                got = backend.sketch_dev([genomes[fastas[i]] for i in mine_idx], k, new_w, bf, ml, repeat=rep_bf)
            elif refine_rep is not None and masks_by_asm is not None and not screen_lists:
                got = backend.sketch_dev([genomes[fastas[i]] for i in mine_idx], k, new_w, bf, ml, repeat=refine_rep)
            else:
                got = backend.sketch_dev([genomes[fastas[i]] for i in mine_idx], k, new_w, bf, ml)
            if screen_lists:
                seen = [m_.screened(genomes[fastas[i]], k, refine_rep) for i, m_ in zip(mine_idx, got)]
                for m_ in got:
                    m_.free()
                got = seen
            local = dict(zip(mine_idx, got))
            if world == 1:
                return local
            everything = backend.exchange_dev(local, len(fastas))
            for m in got:
                m.free()
            return dict(enumerate(everything))

        st.start("indexlr" if mx_tsvs is None else "read_minimizers")
        if mx_tsvs is not None:
            # ntJoin's read_minimizers (S:607-609): the lists come from the files; records are matched to the FASTA's by id
            from .device import Minimizers
            if world > 1:
                raise ValueError("minimizer TSVs are read on one GPU")
            initial_dev = {}
            for i, (p, tsv) in enumerate(zip(fastas, mx_tsvs)):
                ids, h1, pos, line = tsv_data[i]
                index = {name: r for r, name in enumerate(meta[p][0])}
                try:
                    rec_of_line = np.array([index[x] for x in ids], dtype=np.uint32)
                except KeyError as exc:
                    raise ValueError(f"{tsv}: record {exc.args[0]!r} is not in {p}") from None
                rec = rec_of_line[line] if line.size else np.zeros(0, dtype=np.uint32)
                if rec.size and np.any(rec[1:] < rec[:-1]):              # (lines in another order than the FASTA's records)
                    order = np.lexsort((pos, rec))
                    h1, rec, pos = h1[order], rec[order], pos[order]
                initial_dev[i] = Minimizers.from_numpy(backend.ctx, h1, rec, pos)
                if screen_lists:
                    whole = initial_dev[i]
                    initial_dev[i] = whole.screened(genomes[p], k, refine_rep)
                    whole.free()
            write_mx_tsv = False
        else:
            initial_dev = sketch_dev_round(None, w)
        if write_mx_tsv:
            for i in mine_idx:
                out = initial_dev[i].to_numpy()
                recs = genomes[fastas[i]].recs
                if recs.seq is None:                           # bases never left HBM: the k-mer text is gathered there
                    km = initial_dev[i].kmers(genomes[fastas[i]], k) if mx_with_seq else None
                    submit_file(tsv_names[i], fa.write_indexlr_tsv_kmers, tsv_names[i], recs, out[0], out[1], out[2], k, km)
                else:
                    submit_file(tsv_names[i], write_indexlr_tsv, tsv_names[i], recs, out[0], out[1], out[2], k, mx_with_seq)
        st.stop()
        st.mark("sketches_done")
        st.start("ntsynt_synteny")
        eng = DeviceSyntenyEngine(backend.ctx, engine_files, [meta[p][0] for p in fastas], k, w, w_rounds, indel, merge, block_size,
                                  out_prefix, sketch_dev_round, simplify=simplify, log=log, dev=dev, interarrivals=interarrivals, m=m, n=n)
        if initial_only:
            eng.initial_only = True
        first = [initial_dev[i] for i in range(len(fastas))]
    else:
        st.start("indexlr")
        if world == 1 and hasattr(backend, "sketch_batch"):
            initial = backend.sketch_batch([genomes[p] for p in fastas], k, w, bf)
        else:
            local = {i: backend.sketch(genomes[p], k, w, bf) for i, p in enumerate(fastas) if owner[p] == rank}
            initial = _exchange_lists(backend, local, len(fastas)) if world > 1 else [local[i] for i in range(len(fastas))]
        if write_mx_tsv:
            for i, p in enumerate(fastas):
                if owner[p] == rank:
                    out = initial[i]
                    recs = genomes[p].recs
                    if getattr(recs, "seq", 0) is None:
                        from .device import Minimizers
                        km = None
                        if mx_with_seq:
                            tmp_mx = Minimizers.from_numpy(backend.ctx, *out)
                            km = tmp_mx.kmers(genomes[p], k)
                            tmp_mx.free()
                        submit_file(tsv_names[i], fa.write_indexlr_tsv_kmers, tsv_names[i], recs, out[0], out[1], out[2], k, km)
                    else:
                        submit_file(tsv_names[i], write_indexlr_tsv, tsv_names[i], recs, out[0], out[1], out[2], k, mx_with_seq)
        st.stop()
        st.start("ntsynt_synteny")

        def sketch_fn(i, masks, new_w):
            return sketch_round({i: masks}, new_w)[i]

        def sketch_round(masks_by_asm, new_w):
            "re-sketch of a refinement round for the assemblies given: each owner sketches its own, one exchange hands them round"
            if world == 1 and hasattr(backend, "sketch_batch") and len(masks_by_asm) == len(fastas):
                got = backend.sketch_batch([genomes[p] for p in fastas], k, new_w, bf, [masks_by_asm[i] for i in range(len(fastas))])
                return dict(enumerate(got))
            local = {i: backend.sketch(genomes[fastas[i]], k, new_w, bf, m) for i, m in masks_by_asm.items() if owner[fastas[i]] == rank}
            if world == 1:
                return local
            ids = sorted(masks_by_asm)
            # (the exchange numbers lists 0..n-1: positions in `ids`)
            got = _exchange_lists(backend, {ids.index(i): v for i, v in local.items()}, len(ids))
            return {i: got[j] for j, i in enumerate(ids)}
        sketch_fn.all_at_once = sketch_round

        eng = SyntenyEngine(tsv_names, [meta[p][0] for p in fastas], k, w, w_rounds, indel, merge, block_size, out_prefix,
                            backend.graph, sketch_fn, backend.walk, simplify=simplify, log=log, degree_fn=edge_degrees, dev=dev, interarrivals=interarrivals,
                            m=m, n=n)
        first = initial
    try:
        eng.run(first)
    except BaseException:
        # a run that dies after its first round must not leave a plausible-looking block table behind.  The outputs of the stages
        # before it stay, as under the reference's Snakemake (a failed rule loses its own outputs only: "no paths found", S:630-632,
        # leaves <prefix>.common.bf and the minimizer TSVs of the rules that had finished) -- once their writers are through: those
        # still stream from `bf` and the lists, so they are waited for before the unwinding frees what they read, and a file whose
        # writer failed goes
        doomed = [f"{out_prefix}.synteny_blocks.tsv", f"{out_prefix}.pre-collinear-merge.synteny_blocks.tsv"]
        for f in pending_files:
            try:
                f.result()
            except Exception:                                   # noqa: BLE001 -- the run is failing already
                doomed.append(file_of[f])
        writers.shutdown()
        for name in doomed:
            if os.path.exists(name):
                os.remove(name)
        if rank != 0:
            import shutil
            shutil.rmtree(scratch, ignore_errors=True)
        raise
    if rank != 0:
        eng.outputs = {os.path.basename(n): t for n, t in eng.outputs.items()}
        import shutil
        shutil.rmtree(scratch, ignore_errors=True)
    st.stop()
    st.mark("synteny_done")
    st.start("wait_for_files")
    for f in pending_files:
        f.result()                      # re-raises a writer's exception
    writers.shutdown()
    st.stop()
    memory = st.memory()
    if benchmark and rank == 0:
        st.write(f"{prefix}.stage_times.tsv", memory)
    if getattr(backend, "_batch", None) is not None:
        backend._batch[1].free()
        backend._batch = None
    for g in genomes.values():
        if hasattr(g, "free"):
            g.free()
    if shard is not None:
        for sub in shard["subs"]:
            if not any(sub is g for g in genomes.values()):
                sub.free()
    if bf is not None and hasattr(bf, "free"):
        bf.free()
    if refine_rep is not None:
        refine_rep.free()
    if own_backend:
        backend.close()
    st.mark("end")
    eng.stage_times = st.rows
    eng.stage_marks = st.marks
    eng.reserved_bytes = getattr(st, "reserved_bytes", 0)
    eng.memory = memory
    return eng
