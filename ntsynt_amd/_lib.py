"""ctypes binding of libntsynt_hip.so (C ABI: include/ntsynt_hip.h).

The product path has no CPU fallback: if the HIP library is missing or cannot be loaded this
module raises, loudly."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libntsynt_hip.so")
# The same sources built with -DNTS_EXPERIMENTS: the environment switches that pick kernel variants, force fallbacks or cut lists short
# (csrc/nts_knobs.h) exist only there.  Tests and measurement scripts that use them ask for it (Context(variant="experiments"), or
# NTS_LIB_VARIANT=experiments for a whole process); the product -- bin/, bench.py, the pipeline -- loads libntsynt_hip.so.
EXP_LIB_PATH = os.path.join(_HERE, "libntsynt_hip_exp.so")

c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_vp = ctypes.c_void_p
u32, u64 = ctypes.c_uint32, ctypes.c_uint64


class Interval(ctypes.Structure):
    _fields_ = [("rec", u32), ("start", u64), ("end", u64)]


class MxList(ctypes.Structure):
    _fields_ = [("h1", c_vp), ("rec", c_vp), ("pos", c_vp), ("keep", c_vp), ("list_id", c_vp), ("n", u64)]


class Fasta(ctypes.Structure):
    _fields_ = [("seq", c_u8p), ("n", u64), ("n_rec", u32), ("rec_off", c_u64p), ("rec_len", c_u64p),
                ("names", ctypes.POINTER(ctypes.c_char)), ("names_bytes", u64), ("fai_offset", c_u64p),
                ("fai_linebases", c_u32p), ("fai_linewidth", c_u32p)]


class SynthRepeats(ctypes.Structure):
    "nts_synth_repeats (include/ntsynt_hip.h): the repeat families of the assembly-like ancestor"
    _fields_ = [(n, u32) for n in ("sine_cell_log2", "sine_len", "sine_prob_256", "sine_families", "line_cell_log2", "line_len", "line_min_len",
                                   "line_prob_256", "line_families", "div_min_1024", "div_max_1024", "sat_unit", "sat_div_1024")]


class MxTsv(ctypes.Structure):
    _fields_ = [("n_lines", u64), ("names", ctypes.POINTER(ctypes.c_char)), ("names_bytes", u64), ("n", u64), ("h1", c_u64p), ("pos", c_u64p),
                ("line", c_u32p)]


class Graph(ctypes.Structure):
    _fields_ = [("nv", u64), ("v_hash", c_u64p), ("occ_rec", c_u32p), ("occ_pos", c_u64p),
                ("ne", u64), ("e_u", c_u32p), ("e_v", c_u32p), ("e_w", c_u32p), ("e_first", c_u64p)]


class Spans(ctypes.Structure):
    _fields_ = [("start", c_vp), ("end_max", c_vp), ("n", u64)]


class Bubbles(ctypes.Structure):
    _fields_ = [("n_cand", u64), ("cand_edge", c_u32p), ("n_inc", u64), ("inc_edge", c_u32p), ("inc_u", c_u32p),
                ("inc_v", c_u32p), ("inc_w", c_u32p)]


class Blocks(ctypes.Structure):
    _fields_ = [("n_blocks", u64), ("first_vid", c_u32p), ("last_vid", c_u32p), ("n_mx", c_u32p), ("rec", c_u32p),
                ("first_pos", c_u64p), ("last_pos", c_u64p), ("ori", c_u8p), ("stats_paths", u64), ("stats_unoriented", u64),
                ("stats_indel_cuts", u64), ("stats_small", u64)]


# every symbol include/ntsynt_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("nts_init", ctypes.c_int, [ctypes.c_int, ctypes.POINTER(c_vp)]),
    ("nts_destroy", None, [c_vp]),
    ("nts_last_error", ctypes.c_char_p, [c_vp]),
    ("nts_sync", ctypes.c_int, [c_vp]),
    ("nts_stream", c_vp, [c_vp]),
    ("nts_profile", ctypes.c_int, [c_vp, ctypes.c_int]),
    ("nts_timing", ctypes.c_int, [c_vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), c_u64p]),
    ("nts_mem_stats", ctypes.c_int, [c_vp, c_u64p, c_u64p, c_u64p, c_u64p]),
    ("nts_mem_reset_peak", None, []),
    ("nts_bf_size_bytes", ctypes.c_int, [u64, ctypes.c_double, c_u64p, c_u64p]),
    ("nts_bf_size_bytes_ex", ctypes.c_int, [u64, ctypes.c_double, ctypes.c_int, c_u64p, c_u64p]),
    ("nts_genome_upload", ctypes.c_int, [c_vp, c_vp, u64, c_u64p, c_u64p, u32, ctypes.POINTER(c_vp)]),
    ("nts_genome_synth", ctypes.c_int, [c_vp, u64, u32, u64, u64, ctypes.c_double, ctypes.POINTER(c_vp)]),
    ("nts_genome_synth_plan", ctypes.c_int, [c_vp, u32, c_vp, u32, c_vp, u64, u64, ctypes.c_double, ctypes.POINTER(c_vp)]),
    ("nts_genome_synth_plan_ex", ctypes.c_int, [c_vp, u32, c_vp, u32, c_vp, u64, u64, ctypes.c_double, c_vp, ctypes.POINTER(c_vp)]),
    ("nts_genome_download", ctypes.c_int, [c_vp, c_vp, u64, u64, c_vp]),
    ("nts_genome_concat", ctypes.c_int, [c_vp, u32, c_vp, ctypes.POINTER(c_vp)]),
    ("nts_genome_slice", ctypes.c_int, [c_vp, c_vp, u32, u32, ctypes.POINTER(c_vp)]),
    ("nts_genome_free", None, [c_vp, c_vp]),
    ("nts_genome_bases", u64, [c_vp]),
    ("nts_genome_valid_kmers", ctypes.c_int, [c_vp, c_vp, u32, c_u64p]),
    ("nts_bf_create", ctypes.c_int, [c_vp, u64, ctypes.POINTER(c_vp)]),
    ("nts_bf_free", None, [c_vp, c_vp]),
    ("nts_bf_bytes", u64, [c_vp]),
    ("nts_bf_device_ptr", c_vp, [c_vp]),
    ("nts_bf_clear", ctypes.c_int, [c_vp, c_vp]),
    ("nts_bf_insert", ctypes.c_int, [c_vp, c_vp, c_vp, u32]),
    ("nts_bf_insert_and", ctypes.c_int, [c_vp, c_vp, c_vp, u32]),
    ("nts_bf_build_mode", ctypes.c_int, [c_vp, ctypes.c_int]),
    ("nts_bf_cascade", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, u32]),
    ("nts_bf_insert_repeats", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, u32]),
    ("nts_bf_and", ctypes.c_int, [c_vp, c_vp, c_vp]),
    ("nts_bf_popcount", ctypes.c_int, [c_vp, c_vp, c_u64p]),
    ("nts_bf_download", ctypes.c_int, [c_vp, c_vp, c_vp, u64]),
    ("nts_bf_upload", ctypes.c_int, [c_vp, c_vp, c_vp, u64]),
    ("nts_bf_save", ctypes.c_int, [c_vp, c_vp, ctypes.c_char_p, c_vp, u64, u32]),
    ("nts_bench_random_probe", ctypes.c_int, [c_vp, c_vp, u64, u32, ctypes.POINTER(ctypes.c_double), c_u64p]),
    ("nts_bench_valu", ctypes.c_int, [c_vp, ctypes.c_int, u32, u32, ctypes.POINTER(ctypes.c_double),
                                      ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    ("nts_mod_indices", ctypes.c_int, [c_vp, u64, ctypes.c_int, c_vp, u64, c_vp]),
    ("nts_bf_wrap", ctypes.c_int, [c_vp, c_vp, u64, ctypes.POINTER(c_vp)]),
    ("nts_and_raw", ctypes.c_int, [c_vp, c_vp, c_vp, u64]),
    ("nts_comm_unique_id", ctypes.c_int, [c_vp]),
    ("nts_comm_init", ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(c_vp)]),
    ("nts_comm_wrap", ctypes.c_int, [c_vp, c_vp, ctypes.POINTER(c_vp)]),
    ("nts_comm_destroy", None, [c_vp]),
    ("nts_comm_world", ctypes.c_int, [c_vp]),
    ("nts_comm_rank", ctypes.c_int, [c_vp]),
    ("nts_comm_handle", c_vp, [c_vp]),
    ("nts_comm_library", ctypes.c_char_p, []),
    ("nts_bf_create_sharded", ctypes.c_int, [c_vp, u64, ctypes.c_int, ctypes.POINTER(c_vp)]),
    ("nts_bf_fill_ones", ctypes.c_int, [c_vp, c_vp]),
    ("nts_bf_allreduce_and", ctypes.c_int, [c_vp, c_vp, c_vp]),
    ("nts_bf_allreduce_groups", ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.POINTER(ctypes.c_int32), u32]),
    ("nts_comm_last_sparse", ctypes.c_int, [c_vp]),
    ("nts_comm_last_exchange2", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp]),
    ("nts_mx_allgather", ctypes.c_int, [c_vp, c_vp, u32, ctypes.POINTER(c_vp), c_u32p, u32, ctypes.POINTER(c_vp)]),
    ("nts_alloc_stats", ctypes.c_int, [c_vp, c_vp]),
    ("nts_mem_trim", ctypes.c_uint64, []),
    ("nts_mem_cache_stats", ctypes.c_int, [c_vp, c_vp]),
    ("nts_mem_reserve", ctypes.c_int, [ctypes.c_int, ctypes.c_uint64, c_vp]),
    ("nts_mem_events", ctypes.c_int, [c_vp]),
    ("nts_mx_allgather_ex", ctypes.c_int, [c_vp, c_vp, u32, ctypes.POINTER(c_vp), c_u32p, u32, u32, ctypes.POINTER(c_vp)]),
    ("nts_bf_allreduce_parts", ctypes.c_int, [c_vp, ctypes.POINTER(c_vp), u32, c_vp, u32, ctypes.POINTER(ctypes.c_int32), u32]),
    ("nts_mx_export", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    ("nts_mx_export_async", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    ("nts_sketch", ctypes.c_int, [c_vp, c_vp, u32, u32, c_vp, ctypes.POINTER(Interval), u64,
                                  ctypes.POINTER(c_vp)]),
    ("nts_sketch_ex", ctypes.c_int, [c_vp, c_vp, u32, u32, c_vp, c_vp, ctypes.POINTER(Interval), u64,
                                     ctypes.POINTER(c_vp)]),
    ("nts_sketch_mode", ctypes.c_int, [c_vp, ctypes.c_int, u32]),
    ("nts_sketch_summary", ctypes.c_int, [c_vp, ctypes.c_int, c_u32p]),
    ("nts_sketch_select", ctypes.c_int, [c_vp, ctypes.c_int]),
    ("nts_sketch_tiers", ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    ("nts_sketch_stats", ctypes.c_int, [c_vp, c_u64p, c_u64p, c_u64p, c_u32p]),
    ("nts_path_stats", ctypes.c_int, [c_vp, c_u64p, c_u64p, c_u32p]),
    ("nts_bf_level_stats", ctypes.c_int, [c_vp, c_u32p, c_u64p]),
    ("nts_mx_count", u64, [c_vp]),
    ("nts_mx_free", None, [c_vp, c_vp]),
    ("nts_mx_download", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    ("nts_mx_device_ptrs", ctypes.c_int, [c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                                          ctypes.POINTER(c_vp)]),
    ("nts_mx_split", ctypes.c_int, [c_vp, c_vp, u32, c_u32p, ctypes.POINTER(c_vp)]),
    ("nts_mx_concat", ctypes.c_int, [c_vp, u32, ctypes.POINTER(c_vp), c_u32p, ctypes.POINTER(c_vp)]),
    ("nts_mx_screen", ctypes.c_int, [c_vp, c_vp, c_vp, u32, c_vp, ctypes.POINTER(c_vp)]),
    ("nts_mx_upload", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, u64, ctypes.POINTER(c_vp)]),
    ("nts_hash_all", ctypes.c_int, [c_vp, c_vp, u32, ctypes.POINTER(c_u64p), c_u64p]),
    ("nts_graph_build", ctypes.c_int, [c_vp, u32, ctypes.POINTER(MxList), ctypes.POINTER(Graph)]),
    ("nts_engine_create", ctypes.c_int, [c_vp, u32, u32, ctypes.POINTER(c_vp)]),
    ("nts_engine_free", None, [c_vp, c_vp]),
    ("nts_engine_size", ctypes.c_int, [c_vp, c_u64p, c_u64p]),
    ("nts_engine_add", ctypes.c_int, [c_vp, c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(Spans), c_u64p, c_u64p]),
    ("nts_engine_bubbles", ctypes.c_int, [c_vp, c_vp, ctypes.POINTER(Bubbles)]),
    ("nts_bubbles_free", None, [ctypes.POINTER(Bubbles)]),
    ("nts_engine_apply", ctypes.c_int, [c_vp, c_vp, c_vp, u64, c_vp, u64, u32]),
    ("nts_engine_filter", ctypes.c_int, [c_vp, c_vp, u32, ctypes.c_int, c_u64p]),
    ("nts_engine_erode", ctypes.c_int, [c_vp, c_vp, u32, c_u64p]),
    ("nts_engine_blocks", ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, ctypes.c_double, u32, ctypes.POINTER(Blocks)]),
    ("nts_blocks_free", None, [ctypes.POINTER(Blocks)]),
    ("nts_engine_paths", ctypes.c_int, [c_vp, c_u64p, c_u64p]),
    ("nts_engine_read", ctypes.c_int, [c_vp, c_vp, ctypes.c_char_p, c_vp, u64]),
    ("nts_bubble_rule", ctypes.c_int, [u64, c_vp, u64, c_vp, c_vp, c_vp, c_vp, u32, c_vp, c_vp, c_u64p]),
    ("nts_blocks_merge", ctypes.c_int, [u32, u64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                        c_u64p, c_u64p]),
    ("nts_blocks_text", ctypes.c_int, [u32, u64, ctypes.c_int64, ctypes.c_int64, c_vp, ctypes.c_char_p, u64, c_vp, c_vp, c_vp, c_vp,
                                       c_vp, c_vp, c_vp, ctypes.POINTER(c_vp), c_u64p]),
    ("nts_walk_chains", ctypes.c_int, [u64, u64, c_vp, c_vp, ctypes.POINTER(c_u64p), ctypes.POINTER(c_u32p), c_u64p]),
    ("nts_walk_paths", ctypes.c_int, [u64, u64, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(c_u64p), ctypes.POINTER(c_i64p), c_u64p]),
    ("nts_edge_degrees", ctypes.c_int, [u64, u64, c_vp, c_vp, c_vp, c_vp]),
    ("nts_to_i64", ctypes.c_int, [c_vp, u32, u64, c_vp]),
    ("nts_path_scan", ctypes.c_int, [u32, u64, c_vp, c_vp, u64, c_vp, c_vp, ctypes.c_int64, c_vp, c_vp, c_vp]),
    ("nts_graph_free", None, [ctypes.POINTER(Graph)]),
    ("nts_fasta_read", ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(Fasta)]),
    ("nts_fasta_free", None, [ctypes.POINTER(Fasta)]),
    ("nts_read_indexlr_tsv", ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(MxTsv)]),
    ("nts_mx_tsv_free", None, [ctypes.POINTER(MxTsv)]),
    ("nts_genome_from_fasta", ctypes.c_int, [c_vp, ctypes.c_char_p, ctypes.POINTER(c_vp), ctypes.POINTER(Fasta)]),
    ("nts_ingest_trim", ctypes.c_int, [c_vp]),
    ("nts_bf_build_trim", ctypes.c_int, [c_vp]),
    ("nts_mx_kmers", ctypes.c_int, [c_vp, c_vp, c_vp, u32, c_vp]),
    ("nts_write_indexlr_tsv_kmers", ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(Fasta), c_vp, c_vp, c_vp, u64, u32, c_vp]),
    ("nts_write_indexlr_tsv", ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(Fasta), c_vp, c_vp, c_vp, u64, u32, ctypes.c_int]),
    ("nts_free", None, [c_vp]),
]


def build(force=False):
    """Compile libntsynt_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", os.path.join(_HERE, "csrc")]
    if force:
        args.append("-B")
    subprocess.run(args, check=True)
    return LIB_PATH


_libs = {}


def load(variant=None):
    """variant None: the product build (or what NTS_LIB_VARIANT names); 'experiments': libntsynt_hip_exp.so"""
    variant = variant or os.environ.get("NTS_LIB_VARIANT") or "product"
    if variant not in ("product", "experiments"):
        raise ValueError(f"unknown library variant {variant!r}")
    if variant in _libs:
        return _libs[variant]
    path = LIB_PATH if variant == "product" else EXP_LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the HIP extension is required (there is no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C ntsynt_amd/csrc`.")
    lib = ctypes.CDLL(path)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)       # AttributeError if the ABI and the header drift apart
        fn.restype = restype
        fn.argtypes = argtypes
    _libs[variant] = lib
    return lib
