"""Host-side objects over the C ABI (include/ntsynt_hip.h): Context, Genome, BloomFilter, sketch.

Names follow the interfaces of the reference they stand in for:
  KmerBloomFilter.insert / contains-cascade / get_fpr / save   (btllib, used by
      src/ntsynt_make_common_bf.cpp:121-164)
  Indexlr-style minimize() -> minimizers per record             (btllib `indexlr`, smk:81-85)
All compute happens in libntsynt_hip.so on the GPU; there is no CPU path here."""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import Interval, c_vp, u64


class NtsError(RuntimeError):
    pass


MEM_EVENT_NAMES = ("driver_calls", "ns_in_driver_calls", "served_from_cache", "retried_after_emptying_the_cache", "bytes_from_driver",
                   "bytes_to_driver", "ns_waiting_for_device_before_keeping_a_block", "reserve_calls")


def mem_events(lib):
    "nts_mem_events of a loaded library (no context needed: the counters are the process's)"
    out = (ctypes.c_uint64 * 8)()
    lib.nts_mem_events(out)
    return dict(zip(MEM_EVENT_NAMES, [int(x) for x in out]))


def mem_events_since(lib, before):
    "what a stage did to the allocator, in the units a bench line prints"
    now = mem_events(lib)
    d = {k: now[k] - before[k] for k in now}
    return {"hipMalloc_hipFree_calls": d["driver_calls"], "ms_in_hipMalloc_hipFree": round(d["ns_in_driver_calls"] * 1e-6, 2),
            "allocations_served_from_kept_memory": d["served_from_cache"], "allocations_retried_after_emptying_the_cache": d["retried_after_emptying_the_cache"],
            "GB_from_driver": round(d["bytes_from_driver"] / 1e9, 3), "GB_to_driver": round(d["bytes_to_driver"] / 1e9, 3),
            "ms_waiting_for_device_before_keeping_a_block": round(d["ns_waiting_for_device_before_keeping_a_block"] * 1e-6, 2),
            "reserve_calls": d["reserve_calls"]}


class Context:
    """One per GPU (owns a HIP stream)."""

    def __init__(self, device=0, variant=None):
        "variant: None = the product build of the library; 'experiments' = the build with the experiment switches (tests, measurements)"
        self.lib = _lib.load(variant)
        self.variant = variant                 # (contexts that share device memory come from ONE build: each build has its own allocation cache)
        h = c_vp()
        rc = self.lib.nts_init(int(device), ctypes.byref(h))
        if rc != 0:
            raise NtsError(f"nts_init({device}) failed: {self.lib.nts_last_error(None).decode()} "
                           "(a real MI355X is required; there is no CPU fallback)")
        self.h = h
        self.device = int(device)

    def check(self, rc, what=""):
        if rc != 0:
            raise NtsError(f"{what}: {self.lib.nts_last_error(self.h).decode()} (code {rc})")

    def sync(self):
        self.check(self.lib.nts_sync(self.h), "nts_sync")

    def profile(self, enable=True):
        "True/1: time every kernel group; 2: only the dominant kernels (cheaper for short calls); False/0: off"
        level = int(enable) if not isinstance(enable, bool) else (1 if enable else 0)
        self.check(self.lib.nts_profile(self.h, level), "nts_profile")

    def timing(self, name):
        ms, n = ctypes.c_double(), u64()
        self.check(self.lib.nts_timing(self.h, name.encode(), ctypes.byref(ms), ctypes.byref(n)), "nts_timing")
        return ms.value, n.value

    def sketch_mode(self, mode="auto", prune_c=0):
        """'auto' (pruned when w >= 200; tiers below that where they pay), 'dense' (probe the filter for every k-mer) or 'pruned'
        (probe only k-mers whose hash is <= prune_c/w of the hash range; identical output)."""
        code = {"auto": 0, "dense": 1, "pruned": 2}[mode]
        self.check(self.lib.nts_sketch_mode(self.h, code, int(prune_c)), "nts_sketch_mode")
        self._sketch_mode = (mode, int(prune_c))

    def sketch_summary(self, mode=None):
        """Summary-first probing of sparse filters in the dense sketch (nts_sketch_summary): mode 'auto' / 'never' / 'no-lds'
        (summary, but no folded copy of the filter in LDS) / None (leave as is).  Returns the granule shift the last sketch call used (0: no summary)."""
        last = ctypes.c_uint32()
        code = {None: -1, "auto": 0, "never": 1, "no-lds": 2}[mode]
        self.check(self.lib.nts_sketch_summary(self.h, code, ctypes.byref(last)), "nts_sketch_summary")
        return last.value

    def sketch_select(self, impl="auto"):
        """candidate selection kernel of the pruned sketch (nts_sketch_select): 'auto', 'full' (full-width rolling) or 'hi' (upper
        halves rolled, also for assemblies in many pieces); same result"""
        self.check(self.lib.nts_sketch_select(self.h, {"auto": 0, "full": 1, "hi": 2}[impl]), "nts_sketch_select")

    def sketch_tiers(self, mode=None, x0=0.0, half_steps=False):
        """Tiered selection for filters that accept a few per cent of the k-mers (nts_sketch_tiers): mode 'auto' / 'never' / 'always'
        (wherever the kernel applies) / None (leave as is).  Returns (k-mers probed, rounds summed over the tiles, tiers planned) of
        the last sketch call; tiers == 0: it did not go that way.  Same result whichever way."""
        a, b, c = u64(), u64(), ctypes.c_uint32()
        code = {None: -1, "auto": 0, "never": 1, "always": 2}[mode]
        self.check(self.lib.nts_sketch_tiers(self.h, code, float(x0), int(bool(half_steps)), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)),
                   "nts_sketch_tiers")
        return a.value, b.value, c.value

    def bf_build_mode(self, mode="auto"):
        """How BloomFilter.insert sets the bits: 'auto' (partitioned streaming build for large genomes), 'atomic'
        (one atomic OR per k-mer) or 'binned' (partitioned whenever the filter layout allows); same filter."""
        self.check(self.lib.nts_bf_build_mode(self.h, {"auto": 0, "atomic": 1, "binned": 2}[mode]), "nts_bf_build_mode")

    def sketch_stats(self):
        "(accepted candidates, uncovered ranges, k-mers in them) of the last sketch call"
        a, b, c, d = u64(), u64(), u64(), ctypes.c_uint32()
        self.check(self.lib.nts_sketch_stats(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d)),
                   "nts_sketch_stats")
        self.last_prune_c = d.value
        return a.value, b.value, c.value

    def path_stats(self):
        """dict(sketch_many_listed, bf_direct_indices, bf_list_fallback): the repeat / fragmentation paths the last sketch and the
        last partitioned Bloom build took (nts_path_stats)"""
        a, b, c = u64(), u64(), ctypes.c_uint32()
        self.check(self.lib.nts_path_stats(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), "nts_path_stats")
        return {"sketch_many_listed": a.value, "bf_direct_indices": b.value, "bf_list_fallback": c.value}

    def bf_level_stats(self):
        """dict(sparse_level, accepted_kmers): whether the last BloomFilter.insert_and went the literal way over a running filter that
        is all but empty (every k-mer looked up, the bits that were hit kept) and how many k-mers it accepted (nts_bf_level_stats)"""
        a, b = ctypes.c_uint32(), u64()
        self.check(self.lib.nts_bf_level_stats(self.h, ctypes.byref(a), ctypes.byref(b)), "nts_bf_level_stats")
        return {"sparse_level": a.value, "accepted_kmers": b.value}

    VALU_KINDS = ["v_xor_b32", "v_alignbit_b32", "v_lshl_add_u64", "v_lshlrev_b64", "v_mul_lo_u32", "v_mad_u64_u32",
                  "v_add_co_u32+v_addc_co_u32", "v_xor_b32 (dependent chain)", "v_add3_u32", "v_cmp_ge_u32+v_addc_co_u32",
                  "roll31 step (9 instructions)"]

    def bench_valu(self, kind, waves_per_simd=4, iters=20000):
        """Issue-rate microbenchmark of one integer VALU instruction (nts_bench_valu): dict with the shader cycles a SIMD
        spends per wave-instruction and the wave-instructions per second per CU the wall time gives."""
        ms, cyc, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        k = self.VALU_KINDS.index(kind) if isinstance(kind, str) else int(kind)
        self.check(self.lib.nts_bench_valu(self.h, k, int(waves_per_simd), int(iters), ctypes.byref(ms), ctypes.byref(cyc),
                                           ctypes.byref(n)), "nts_bench_valu")
        per_cu = 4 * waves_per_simd * n.value / (ms.value * 1e-3)
        return {"instruction": self.VALU_KINDS[k], "waves_per_simd": int(waves_per_simd), "wall_ms": round(ms.value, 4),
                "cycles_per_wave_instr_per_simd": round(cyc.value, 3), "wave_instr_per_s_per_cu": per_cu,
                # 4 SIMDs per CU, each issuing one wave-instruction per `cycles`: the clock the two figures imply
                "implied_clock_GHz": round(per_cu * cyc.value / 4 / 1e9, 3)}

    def mem_stats(self):
        """Device memory of the library's allocations in this process: dict(live, peak, device_used, device_total) in bytes
        (nts_mem_stats; `peak` is the high-water mark since the last mem_reset_peak())."""
        a, b, c, d = u64(), u64(), u64(), u64()
        self.check(self.lib.nts_mem_stats(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d)), "nts_mem_stats")
        return {"live": a.value, "peak": b.value, "device_used": c.value, "device_total": d.value}

    def alloc_stats(self):
        "(hipMalloc / hipFree calls the library has made in this process, host milliseconds spent in them): nts_alloc_stats"
        n, ms = u64(), ctypes.c_double()
        self.lib.nts_alloc_stats(ctypes.byref(n), ctypes.byref(ms))
        return n.value, ms.value

    def mem_trim(self):
        "every block of the library's allocation cache back to the driver; bytes released (nts_mem_trim)"
        return int(self.lib.nts_mem_trim())

    def mem_cache_stats(self):
        "(bytes the allocation cache holds, allocations it has served): nts_mem_cache_stats"
        a, b = u64(), u64()
        self.lib.nts_mem_cache_stats(ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def mem_reset_peak(self):
        self.lib.nts_mem_reset_peak()

    def mem_reserve(self, nbytes):
        "one driver allocation of `nbytes` kept for the allocations to come (nts_mem_reserve); returns the bytes reserved"
        got = u64()
        self.check(self.lib.nts_mem_reserve(self.device, int(nbytes), ctypes.byref(got)), "nts_mem_reserve")
        return got.value

    def mem_reserve_async(self, nbytes):
        """the same on a thread of its own (the call spends its time in the driver, outside the interpreter lock): a run starts it
        before it opens its first file; .join() returns the bytes reserved (0 when the driver refused: the run then allocates as it goes)"""
        import threading
        box = {"got": 0}

        def work():
            try:
                box["got"] = self.mem_reserve(nbytes)
            except Exception:
                box["got"] = 0
        th = threading.Thread(target=work, name="nts_mem_reserve", daemon=True)
        th.start()

        class _Pending:
            def join(self_inner):
                th.join()
                return box["got"]
        return _Pending()

    def mem_events(self):
        "process-wide allocation counters (nts_mem_events) as a dict; take differences around a stage (mem_events_since)"
        return mem_events(self.lib)

    def mem_events_since(self, before):
        return mem_events_since(self.lib, before)

    def trim_ingest(self):
        "give back the FASTA ingest's workspaces (the raw image of the largest file, pinned staging): nts_ingest_trim"
        self.check(self.lib.nts_ingest_trim(self.h), "nts_ingest_trim")

    def trim_bf_build(self):
        "the Bloom build's bucket arrays back to the allocation cache (the run's last filter is made): nts_bf_build_trim"
        self.check(self.lib.nts_bf_build_trim(self.h), "nts_bf_build_trim")

    def close(self):
        if self.h:
            self.lib.nts_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


BF_ROUNDING = {"up": 0, "down": 1, "none": 2}


def bf_size_bytes(genome_bp, fpr, rounding="up"):
    """(approx_bytes, ctor_bytes): src/ntsynt_make_common_bf.cpp:28-40 + btllib ctor rounding.  `rounding` selects the
    constructor's rule ('up' = ceil(bytes / 8.0) * 8, recalled from btllib; 'down' and 'none' are the alternatives a
    reader of btllib's source can switch to: SURVEY.md 8(c) u1)."""
    a, c = u64(), u64()
    rc = _lib.load().nts_bf_size_bytes_ex(int(genome_bp), float(fpr), BF_ROUNDING[rounding], ctypes.byref(a), ctypes.byref(c))
    if rc != 0:
        raise NtsError("nts_bf_size_bytes: bad arguments")
    return a.value, c.value


class Genome:
    """A FASTA resident in HBM.  names: record ids; seq: uint8 array of the concatenated bases."""

    def __init__(self, ctx, names, seq, rec_off, rec_len):
        self.ctx = ctx
        self.names = list(names)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        self.rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        self.rec_len = np.ascontiguousarray(rec_len, dtype=np.uint64)
        self.n_bytes = int(seq.size)
        h = c_vp()
        ctx.check(ctx.lib.nts_genome_upload(ctx.h, seq.ctypes.data, seq.size,
                                            self.rec_off.ctypes.data_as(_lib.c_u64p),
                                            self.rec_len.ctypes.data_as(_lib.c_u64p),
                                            len(self.names), ctypes.byref(h)), "nts_genome_upload")
        self.h = h

    @classmethod
    def synth(cls, ctx, total_bp, n_contigs, seed_ancestor, seed_genome, substitution_rate):
        """Synthetic genome generated in HBM (bench / scale tests): relatives share seed_ancestor."""
        g = cls.__new__(cls)
        g.ctx = ctx
        per = int(total_bp) // int(n_contigs)
        g.names = [f"chr{i + 1}" for i in range(n_contigs)]
        g.rec_len = np.full(n_contigs, per, dtype=np.uint64)
        g.rec_off = (np.arange(n_contigs, dtype=np.uint64) * np.uint64(per)).astype(np.uint64)
        g.n_bytes = per * int(n_contigs)
        h = c_vp()
        ctx.check(ctx.lib.nts_genome_synth(ctx.h, int(total_bp), int(n_contigs), int(seed_ancestor), int(seed_genome),
                                           float(substitution_rate), ctypes.byref(h)), "nts_genome_synth")
        g.h = h
        return g

    @classmethod
    def synth_plan(cls, ctx, plan, seed_ancestor, seed_genome, substitution_rate, rep=None, names=None):
        """Synthetic genome with structural events generated in HBM (nts_genome_synth_plan[_ex]): `plan` = (record lengths,
        pieces) from ntsynt_amd.synth.structural_plan / realistic_plan; relatives share seed_ancestor and the ancestor's layout.
        rep: dict of nts_synth_repeats fields (synth.REPEATS) -- the assembly-like ancestor with interspersed repeats and
        satellite arrays."""
        rec_len, pieces = plan[0], plan[1]
        g = cls.__new__(cls)
        g.ctx = ctx
        rec_len = np.ascontiguousarray(rec_len, dtype=np.uint64)
        pieces = np.ascontiguousarray(pieces)
        assert pieces.dtype.itemsize == 32
        g.names = list(names) if names is not None else [f"chr{i + 1}" for i in range(rec_len.size)]
        g.rec_len = rec_len
        g.rec_off = np.concatenate(([0], np.cumsum(rec_len[:-1]))).astype(np.uint64)
        g.n_bytes = int(rec_len.sum())
        h = c_vp()
        rp = None
        if rep is not None:
            rp = _lib.SynthRepeats(**{k_: int(v) for k_, v in rep.items()})
        ctx.check(ctx.lib.nts_genome_synth_plan_ex(ctx.h, rec_len.size, rec_len.ctypes.data, pieces.size, pieces.ctypes.data,
                                                   int(seed_ancestor), int(seed_genome), float(substitution_rate),
                                                   ctypes.byref(rp) if rp is not None else None, ctypes.byref(h)),
                  "nts_genome_synth_plan")
        g.h = h
        return g

    @classmethod
    def concat(cls, ctx, parts):
        """The batch of resident genomes `parts` as one resident genome (nts_genome_concat, a device-to-device copy):
        record ids of part p start at rec_base[p].  One sketch of the batch replaces one sketch per part --
        split_minimizers() takes the list apart again."""
        g = cls.__new__(cls)
        g.ctx = ctx
        g.names = [n for p in parts for n in p.names]
        g.rec_base = np.concatenate(([0], np.cumsum([len(p.names) for p in parts]))).astype(np.int64)
        sizes = np.array([p.n_bytes for p in parts], dtype=np.uint64)
        g.n_bytes = int(sizes.sum())
        starts = np.concatenate(([0], np.cumsum(sizes)[:-1])).astype(np.uint64)
        g.rec_off = np.concatenate([p.rec_off + s for p, s in zip(parts, starts)]).astype(np.uint64)
        g.rec_len = np.concatenate([p.rec_len for p in parts]).astype(np.uint64)
        arr = (c_vp * len(parts))(*[p.h for p in parts])
        h = c_vp()
        ctx.check(ctx.lib.nts_genome_concat(ctx.h, len(parts), arr, ctypes.byref(h)), "nts_genome_concat")
        g.h = h
        return g

    def slice(self, rec0, rec1, ctx=None):
        """Records [rec0, rec1) as a resident genome of their own (nts_genome_slice): the shard one rank of this genome's group
        works on when there are fewer genomes than GPUs; record r of the slice is record rec0 + r here."""
        ctx = ctx or self.ctx
        g = Genome.__new__(Genome)
        g.ctx = ctx
        g.names = self.names[rec0:rec1]
        g.rec_len = np.ascontiguousarray(self.rec_len[rec0:rec1], dtype=np.uint64)
        g.rec_off = (np.concatenate(([0], np.cumsum(g.rec_len[:-1]))) if rec1 > rec0 else np.zeros(0)).astype(np.uint64)
        g.n_bytes = int(g.rec_len.sum())
        g.rec0 = int(rec0)
        h = c_vp()
        ctx.check(ctx.lib.nts_genome_slice(ctx.h, self.h, int(rec0), int(rec1), ctypes.byref(h)), "nts_genome_slice")
        g.h = h
        return g

    def split_minimizers(self, h1, rec, pos):
        """(h1, rec, pos) of a sketch of a concat() batch -> one (h1, rec, pos) per part, record ids local to the part
        (the list is in (record, position) order, so each part is a slice)."""
        cut = np.searchsorted(rec, self.rec_base.astype(rec.dtype))
        return [(h1[a:b], rec[a:b] - rec.dtype.type(base), pos[a:b])
                for a, b, base in zip(cut[:-1], cut[1:], self.rec_base[:-1])]

    def download(self, offset, length):
        "upper-case ASCII of bases [offset, offset+length) of the concatenated records"
        out = np.empty(int(length), dtype=np.uint8)
        self.ctx.check(self.ctx.lib.nts_genome_download(self.ctx.h, self.h, int(offset), int(length), out.ctypes.data),
                       "nts_genome_download")
        return out

    @property
    def total_bp(self):
        return int(self.ctx.lib.nts_genome_bases(self.h))

    def valid_kmers(self, k):
        n = u64()
        self.ctx.check(self.ctx.lib.nts_genome_valid_kmers(self.ctx.h, self.h, k, ctypes.byref(n)), "valid_kmers")
        return n.value

    def hash_all(self, k):
        "canonical h0 of every valid k-mer in (record, position) order (test hook, row B1)"
        p, n = _lib.c_u64p(), u64()
        self.ctx.check(self.ctx.lib.nts_hash_all(self.ctx.h, self.h, k, ctypes.byref(p), ctypes.byref(n)), "nts_hash_all")
        out = np.ctypeslib.as_array(p, shape=(max(n.value, 1),))[:n.value].copy()
        self.ctx.lib.nts_free(p)
        return out

    def free(self):
        if self.h:
            self.ctx.lib.nts_genome_free(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class BloomFilter:
    """btllib::KmerBloomFilter(bytes, 1, k) stand-in, bit array in HBM."""

    def __init__(self, ctx, nbytes, k, world=1, ones=False):
        """world > 1: allocation laid out for Comm.allreduce_and (nts_bf_create_sharded); ones: all bits set, the
        identity of AND for a rank that owns no genome."""
        self.ctx, self.k = ctx, k
        h = c_vp()
        if world > 1:
            ctx.check(ctx.lib.nts_bf_create_sharded(ctx.h, int(nbytes), int(world), ctypes.byref(h)), "nts_bf_create_sharded")
        else:
            ctx.check(ctx.lib.nts_bf_create(ctx.h, int(nbytes), ctypes.byref(h)), "nts_bf_create")
        self.h = h
        self.bytes = int(nbytes)
        if ones:
            ctx.check(ctx.lib.nts_bf_fill_ones(ctx.h, h), "nts_bf_fill_ones")

    def insert(self, genome):
        "bf->insert(record.seq) for every record (cpp:128-131)"
        self.ctx.check(self.ctx.lib.nts_bf_insert(self.ctx.h, self.h, genome.h, self.k), "nts_bf_insert")

    def insert_and(self, genome):
        """one cascade level (cpp:134-160) in place: self &= the filter of `genome`, fused into the build's last pass
        (nts_bf_insert_and) -- what clear + insert into a second filter + and_ give, without the second filter"""
        self.ctx.check(self.ctx.lib.nts_bf_insert_and(self.ctx.h, self.h, genome.h, self.k), "nts_bf_insert_and")

    def cascade_from(self, prev, genome):
        "`if prev.contains(h): self.insert(h)` over every k-mer of genome (cpp:145-153)"
        self.ctx.check(self.ctx.lib.nts_bf_cascade(self.ctx.h, prev.h, self.h, genome.h, self.k), "nts_bf_cascade")

    def insert_repeats_of(self, genome, genome_bf):
        """self = the repeat filter: every k-mer of `genome` whose bit is set in genome_bf already sets its bit here, the others set
        it in genome_bf (bin/ntsynt_make_repeat_bfs.py:56-67)"""
        self.ctx.check(self.ctx.lib.nts_bf_insert_repeats(self.ctx.h, genome_bf.h, self.h, genome.h, self.k), "nts_bf_insert_repeats")

    def and_(self, other):
        self.ctx.check(self.ctx.lib.nts_bf_and(self.ctx.h, self.h, other.h), "nts_bf_and")

    def clear(self):
        self.ctx.check(self.ctx.lib.nts_bf_clear(self.ctx.h, self.h), "nts_bf_clear")

    def popcount(self):
        n = u64()
        self.ctx.check(self.ctx.lib.nts_bf_popcount(self.ctx.h, self.h, ctypes.byref(n)), "nts_bf_popcount")
        return n.value

    def get_fpr(self):
        "occupancy (one hash function), what the reference prints at cpp:132,154,162"
        return self.popcount() / float(self.bytes * 8)

    def bench_random_probe(self, n_probes, repeats=3):
        "microbenchmark: average ms for n_probes random single-bit reads (no hashing)"
        ms, hits = ctypes.c_double(), u64()
        self.ctx.check(self.ctx.lib.nts_bench_random_probe(self.ctx.h, self.h, int(n_probes), int(repeats), ctypes.byref(ms),
                                                           ctypes.byref(hits)), "nts_bench_random_probe")
        return ms.value

    def device_ptr(self):
        return int(self.ctx.lib.nts_bf_device_ptr(self.h))

    def to_numpy(self):
        out = np.empty(self.bytes, dtype=np.uint8)
        self.ctx.check(self.ctx.lib.nts_bf_download(self.ctx.h, self.h, out.ctypes.data, self.bytes), "nts_bf_download")
        return out

    def save(self, path, header=b"", threads=0):
        "bf->save(path): header bytes + the bit array streamed out of HBM into the file (nts_bf_save), no host copy"
        self.ctx.check(self.ctx.lib.nts_bf_save(self.ctx.h, self.h, os.fsencode(path), header, len(header), int(threads)), "nts_bf_save")

    def from_numpy(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        self.ctx.check(self.ctx.lib.nts_bf_upload(self.ctx.h, self.h, arr.ctypes.data, arr.size), "nts_bf_upload")

    def free(self):
        if self.h:
            self.ctx.lib.nts_bf_free(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Comm:
    """RCCL communicator of the multi-GPU path, one rank per GPU (nts_comm_*).  `exchange_id(id_or_None) -> id` hands the
    128-byte id made on rank 0 to every rank (torch.distributed's store in this package; any launcher would do)."""

    def __init__(self, ctx, world, rank, exchange_id):
        self.ctx, self.world, self.rank = ctx, int(world), int(rank)
        ident = None
        if rank == 0:
            buf = (ctypes.c_uint8 * 128)()
            if ctx.lib.nts_comm_unique_id(buf) != 0:
                raise NtsError("nts_comm_unique_id failed: librccl could not be loaded")
            ident = bytes(buf)
        ident = exchange_id(ident)
        h = c_vp()
        raw = (ctypes.c_uint8 * 128).from_buffer_copy(ident)
        ctx.check(ctx.lib.nts_comm_init(ctx.h, raw, self.world, self.rank, ctypes.byref(h)), "nts_comm_init")
        self.h = h

    @classmethod
    def from_torch(cls, ctx):
        "communicator over the ranks of torch.distributed's default group (used only to pass the id around)"
        import torch.distributed as dist

        def exchange(ident):
            box = [ident]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        return cls(ctx, dist.get_world_size(), dist.get_rank(), exchange)

    def allreduce_and(self, bf):
        "exchange 1: bf &= every other rank's filter, in place (bf from BloomFilter(..., world=N))"
        self.ctx.check(self.ctx.lib.nts_bf_allreduce_and(self.ctx.h, bf.h, self.h), "nts_bf_allreduce_and")

    def allreduce_groups(self, bf, group_of):
        """exchange 1 with genomes sharded over groups of ranks: bf = AND over groups of (OR over the filters of the group's ranks),
        in place; group_of[r] = group (genome) of rank r (nts_bf_allreduce_groups)"""
        arr = (ctypes.c_int32 * self.world)(*[int(g) for g in group_of])
        self.ctx.check(self.ctx.lib.nts_bf_allreduce_groups(self.ctx.h, bf.h, self.h, arr, max(group_of) + 1), "nts_bf_allreduce_groups")

    def allreduce_parts(self, filters, slot_group, n_slots, n_groups):
        """exchange 1 with a family's records shared out by bases (pipeline.partition_plan): this rank's `filters` (one per genome its
        range touches, all from BloomFilter(..., world=N); filters[0] receives the result; a rank without records passes one filter),
        slot_group[r][s] = genome of rank r's s-th filter or -1 (nts_bf_allreduce_parts)"""
        flat = [int(g) for row in slot_group for g in (list(row) + [-1] * (n_slots - len(row)))]
        arr = (ctypes.c_int32 * len(flat))(*flat)
        hs = (c_vp * len(filters))(*[f.h for f in filters])
        self.ctx.check(self.ctx.lib.nts_bf_allreduce_parts(self.ctx.h, hs, len(filters), self.h, int(n_slots), arr, int(n_groups)),
                       "nts_bf_allreduce_parts")

    def rccl_ranks(self):
        "ranks of the communicator as the library sees it (nts_comm_world)"
        return int(self.ctx.lib.nts_comm_world(self.h))

    def last_sparse(self):
        "True if the last all-reduce gathered set-bit indices instead of the reduced chunks"
        return bool(self.ctx.lib.nts_comm_last_sparse(self.ctx.h))

    def last_exchange2(self):
        "the last exchange 2 of this context: dict(packed_bytes, unpacked_bytes, sent_bytes) -- nts_comm_last_exchange2"
        a, b, c = u64(), u64(), u64()
        self.ctx.lib.nts_comm_last_exchange2(self.ctx.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return {"packed_bytes": a.value, "unpacked_bytes": b.value, "sent_bytes": c.value}

    def allgather_minimizers(self, local, local_ids, n_total, slots=0):
        """exchange 2: the ranks' Minimizers -> [Minimizers of list g for g in range(n_total)], resident in HBM; slots: lists a rank may
        hold (0: ceil(n_total / world))"""
        return allgather_minimizers(self.ctx, self, local, local_ids, n_total, slots)

    def close(self):
        if self.h:
            self.ctx.lib.nts_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def allgather_minimizers(ctx, comm, local, local_ids, n_total, slots=0):
    "nts_mx_allgather_ex; comm None = one rank (the lists are copied into fresh handles)"
    n = len(local)
    arr = (c_vp * max(n, 1))(*[m.h for m in local])
    ids = (ctypes.c_uint32 * max(n, 1))(*[int(i) for i in local_ids])
    out = (c_vp * int(n_total))()
    ctx.check(ctx.lib.nts_mx_allgather_ex(ctx.h, comm.h if comm is not None else None, n, arr, ids, int(n_total), int(slots), out),
              "nts_mx_allgather")
    return [Minimizers(ctx, c_vp(out[g])) for g in range(int(n_total))]


class Minimizers:
    """Device-resident minimizer list of one genome, in (record, position) order."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    def __len__(self):
        return int(self.ctx.lib.nts_mx_count(self.h))

    def to_numpy(self):
        n = len(self)
        h1 = np.empty(n, dtype=np.uint64)
        rec = np.empty(n, dtype=np.uint32)
        pos = np.empty(n, dtype=np.uint64)
        if n:
            self.ctx.check(self.ctx.lib.nts_mx_download(self.ctx.h, self.h, h1.ctypes.data, rec.ctypes.data,
                                                        pos.ctypes.data), "nts_mx_download")
        return h1, rec, pos

    @classmethod
    def from_numpy(cls, ctx, h1, rec, pos):
        h1 = np.ascontiguousarray(h1, dtype=np.uint64)
        rec = np.ascontiguousarray(rec, dtype=np.uint32)
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        h = c_vp()
        ctx.check(ctx.lib.nts_mx_upload(ctx.h, h1.ctypes.data, rec.ctypes.data, pos.ctypes.data, h1.size,
                                        ctypes.byref(h)), "nts_mx_upload")
        return cls(ctx, h)

    def kmers(self, genome, k):
        "k-mer text of every minimizer (uint8 array of len(self) * k upper-case bytes) gathered from the resident genome"
        out = np.empty(len(self) * int(k), dtype=np.uint8)
        if len(self):
            self.ctx.check(self.ctx.lib.nts_mx_kmers(self.ctx.h, genome.h, self.h, int(k), out.ctypes.data), "nts_mx_kmers")
        return out

    def screened(self, genome, k, filter_out):
        """ntJoin's read_minimizers(file, repeat_bf) (stage 3's `--filter Filter`): a new list without the minimizers whose k-mer -- the
        k bases of `genome` at their record and position -- the filter holds (nts_mx_screen)"""
        h = c_vp()
        self.ctx.check(self.ctx.lib.nts_mx_screen(self.ctx.h, genome.h, self.h, int(k), filter_out.h, ctypes.byref(h)), "nts_mx_screen")
        return Minimizers(self.ctx, h)

    def split(self, rec_base):
        """The list of a Genome.concat() batch taken apart on the device (nts_mx_split): one Minimizers per part, record ids
        local to the part; rec_base = Genome.rec_base of the batch."""
        n = len(rec_base) - 1
        base = (ctypes.c_uint32 * (n + 1))(*[int(x) for x in rec_base])
        out = (c_vp * n)()
        self.ctx.check(self.ctx.lib.nts_mx_split(self.ctx.h, self.h, n, base, out), "nts_mx_split")
        return [Minimizers(self.ctx, c_vp(out[p])) for p in range(n)]

    @classmethod
    def concat(cls, ctx, parts, rec_offsets):
        "the lists of a genome's shards, in record order, as the genome's list: part p's record numbers raised by rec_offsets[p] (nts_mx_concat)"
        n = len(parts)
        arr = (c_vp * n)(*[m.h for m in parts])
        off = (ctypes.c_uint32 * n)(*[int(x) for x in rec_offsets])
        h = c_vp()
        ctx.check(ctx.lib.nts_mx_concat(ctx.h, n, arr, off, ctypes.byref(h)), "nts_mx_concat")
        return cls(ctx, h)

    def device_ptrs(self):
        a, b, c = c_vp(), c_vp(), c_vp()
        self.ctx.lib.nts_mx_device_ptrs(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return a.value, b.value, c.value

    def free(self):
        if self.h:
            self.ctx.lib.nts_mx_free(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sketch(ctx, genome, k, w, bf=None, masks=None, repeat=None):
    """`indexlr -k k -w w --long --pos [-s bf] [-r repeat]` on the resident genome.  masks: iterable of
    (record index, start, end) hard-mask intervals applied on the fly (refinement rounds).  repeat: filter-out
    Bloom filter (the reference's experimental repeat filter; served by the every-k-mer-probed kernels)."""
    n_mask, arr = 0, None
    if masks is not None and len(masks):
        n_mask = len(masks)
        if isinstance(masks, np.ndarray) and masks.dtype.itemsize == ctypes.sizeof(Interval) and masks.dtype.names:
            masks = np.ascontiguousarray(masks)                 # already in nts_interval's layout (synteny_device.INTERVAL_DTYPE)
            arr = ctypes.cast(masks.ctypes.data, ctypes.POINTER(Interval))
        else:
            arr = (Interval * n_mask)()
            for i, (r, s, e) in enumerate(masks):
                arr[i].rec, arr[i].start, arr[i].end = int(r), int(s), int(e)
    h = c_vp()
    if repeat is not None:
        ctx.check(ctx.lib.nts_sketch_ex(ctx.h, genome.h, int(k), int(w), bf.h if bf is not None else None, repeat.h,
                                        arr, n_mask, ctypes.byref(h)), "nts_sketch_ex")
    else:
        ctx.check(ctx.lib.nts_sketch(ctx.h, genome.h, int(k), int(w), bf.h if bf is not None else None,
                                     arr, n_mask, ctypes.byref(h)), "nts_sketch")
    return Minimizers(ctx, h)


class SketchPool:
    """Sketches of several resident genomes at once: every genome on a context (stream, workspaces) of its own, driven from a host
    thread of its own -- the GPU overlaps one genome's latency-bound kernels (list compaction, window decisions, gather,
    finalize, the uncovered ranges: a third of a sketch) with another's rolling.  Three 3 Gbp genomes: 5.8 ms against 6.5 one after
    the other (scripts/concurrent_sketch.py).  Same lists as sketch() per genome, in the order given; the lists belong to the
    pool's contexts (free them before close()).  The library serialises what the genomes share (the filter's summary)."""

    def __init__(self, ctx, n):
        self.main = ctx
        self.ctxs = [ctx] + [Context(ctx.device, ctx.variant) for _ in range(max(0, int(n) - 1))]
        self._pool = None

    def configure(self, fn):
        "apply fn(ctx) to every context of the pool (sketch_mode, profile, ...)"
        for c in self.ctxs:
            fn(c)

    def sketch(self, genomes, k, w, bf=None, masks=None, repeat=None):
        n = len(genomes)
        if n <= 1 or len(self.ctxs) == 1:
            return [sketch(self.main, g, k, w, bf, masks[i] if masks else None, repeat=repeat) for i, g in enumerate(genomes)]
        from concurrent.futures import ThreadPoolExecutor
        if self._pool is None:
            self._pool = ThreadPoolExecutor(max_workers=len(self.ctxs))
        out = [None] * n
        lanes = min(n, len(self.ctxs))

        def lane(c):
            for i in range(c, n, lanes):                  # (the C calls release the GIL)
                out[i] = sketch(self.ctxs[c], genomes[i], k, w, bf, masks[i] if masks else None, repeat=repeat)
        for f in [self._pool.submit(lane, c) for c in range(lanes)]:
            f.result()
        for mx in out:            # the lanes are done: the lists belong to the caller's context from here on (they outlive the pool)
            mx.ctx = self.main
        return out

    def timing(self, name):
        "(ms, launches) summed over the pool's contexts"
        ms = n = 0
        for c in self.ctxs:
            a, b = c.timing(name)
            ms, n = ms + a, n + b
        return ms, n

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        for c in self.ctxs[1:]:
            c.close()
        self.ctxs = self.ctxs[:1]


def wrap_bloom(ctx, tensor, nbytes, k):
    """BloomFilter view over a caller-owned device buffer (a torch uint8 tensor of at least
    ceil16(nbytes) bytes, zero-initialised): lets RCCL collectives run on the very same memory."""
    bf = BloomFilter.__new__(BloomFilter)
    bf.ctx, bf.k, bf.bytes = ctx, k, int(nbytes)
    h = c_vp()
    ctx.check(ctx.lib.nts_bf_wrap(ctx.h, ctypes.c_void_p(tensor.data_ptr()), int(nbytes), ctypes.byref(h)), "nts_bf_wrap")
    bf.h = h
    bf._keepalive = tensor
    return bf


def and_raw(ctx, acc_ptr, other_ptr, nbytes):
    "acc &= other on raw device pointers (local step of the AND all-reduce)"
    ctx.check(ctx.lib.nts_and_raw(ctx.h, ctypes.c_void_p(acc_ptr), ctypes.c_void_p(other_ptr), int(nbytes)), "nts_and_raw")


def export_minimizers(ctx, mx, h1_ptr, rec_ptr, pos_ptr, wait=True):
    "device-to-device copy of a list into caller buffers; wait=False: queued only, ctx.sync() before use / mx.free()"
    fn = ctx.lib.nts_mx_export if wait else ctx.lib.nts_mx_export_async
    ctx.check(fn(ctx.h, mx.h, ctypes.c_void_p(h1_ptr), ctypes.c_void_p(rec_ptr), ctypes.c_void_p(pos_ptr)), "nts_mx_export")
