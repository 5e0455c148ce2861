"""The stage executables of the reference's workflow (bin/ntsynt_run_pipeline.smk), same command lines, on the GPU:

    ntsynt_make_common_bf --genome FILE... -k K [--fpr F] [-p PREFIX] [--bf BYTES] [-t T]     rule make_common_bf, smk:55-62
                                                                                              (src/ntsynt_make_common_bf.cpp:46-81)
    indexlr -k K -w W --long --seq --pos [-t T] [-s common.bf] [-r repeat.bf] FILE > out.tsv  rule indexlr, smk:74-85 (btllib's binary:
                                                                                              the options ntSynt passes)
    ntsynt_run.py FILES... --fastas FILE... -k K -w W [-n N] [-p P] [-z Z] [--common BF] [--btllib_t T] [--w-rounds W...]
                  [--bp BP] [--collinear-merge M] [--simplify-graph] [-m M] [--dev] [--interarrivals]    rule ntsynt_synteny,
                                                                                              smk:87-103 (bin/ntsynt_run.py:10-44)

With them the reference's Snakefile drives the GPU stages unchanged, a maintainer holding btllib can
diff the pipeline stage by stage, and files made by the reference's own tools (minimizer TSVs, a `.bf`) can be fed to this build
from a shell.  Each is a thin front of ntsynt_amd.pipeline; all sequence-scale work runs in libntsynt_hip.so."""
import argparse
import os
import re
import sys

NTSYNT_VERSION = "ntSynt v1.0.4"


# ---- ntsynt_make_common_bf ------------------------------------------------------------------------------------------------------
def make_common_bf_parser():
    p = argparse.ArgumentParser(prog="ntsynt_make_common_bf")
    p.add_argument("--genome", nargs="+", help="Input genome file(s)", required=True)
    p.add_argument("-k", help="k-mer size (bp)", required=True, type=int)
    p.add_argument("--fpr", help="False positive rate for Bloom filter", default=0.025, type=float)
    p.add_argument("-p", help="Prefix for output Bloom filter", default="common_bf")
    p.add_argument("--bf", help="Bloom filter size in bytes (optional)", type=int)
    p.add_argument("-t", help="Number of threads (accepted for compatibility: the GPU does the work)", default=12, type=int)
    p.add_argument("--device", help="GPU index [0]", type=int, default=0)
    p.add_argument("--bf-rounding", help=argparse.SUPPRESS, choices=["up", "down", "none"], default="up")
    p.add_argument("--bf-signature", help=argparse.SUPPRESS, default=None)
    return p


def make_common_bf(argv=None):
    "src/ntsynt_make_common_bf.cpp main(): parameter echo (:90-99), sort (:105-107), size (:109-118), level 1 (:121-132), cascade (:134-160), save (:162-164)"
    args = make_common_bf_parser().parse_args(argv)
    from . import fasta as fa
    from . import pipeline
    from .device import BloomFilter, Context, bf_size_bytes
    print("Parameters:")
    print("\t\t--genome " + "".join(g + " " for g in args.genome))
    print(f"\t\t-t {args.t}")
    print(f"\t\t-k {args.k}")
    print(f"\t\t--fpr {args.fpr:g}")
    print(f"\t\t-p {args.p}", flush=True)
    files = sorted(args.genome)
    for f in files:
        if not os.path.isfile(f):
            raise FileNotFoundError(f"Input file {f} not found.")
    ctx = Context(args.device)
    first, _ = fa.read_fasta_device(ctx, files[0])
    if args.bf is not None:
        approx = int(args.bf)
        print(f"\t\t--bf {approx}")
        nbytes = {"up": (approx + 7) // 8 * 8, "down": approx // 8 * 8, "none": approx}[args.bf_rounding]      # btllib's constructor (u1)
    else:
        print("Calculating BF size based on input genome size")
        print(f"Genome size (bp): {first.total_bp}")              # approximate_bf_size, src/ntsynt_make_common_bf.cpp:37
        approx, nbytes = bf_size_bytes(first.total_bp, args.fpr, args.bf_rounding)
    print(f"BF size (bytes): {approx}", flush=True)
    bf = BloomFilter(ctx, nbytes, args.k)
    bf.insert(first)
    first.free()
    print(f"Bloom filter FPR: {bf.get_fpr():g}", flush=True)       # (a double through operator<<: six significant digits)
    for f in files[1:]:
        g, _ = fa.read_fasta_device(ctx, f)
        bf.insert_and(g)                                       # one cascade level (cpp:134-160)
        g.free()
        print(f"Bloom filter FPR: {bf.get_fpr():g}", flush=True)       # (a double through operator<<: six significant digits)
    print(f"Final Bloom filter FPR: {bf.get_fpr():g}", flush=True)
    bf.save(f"{args.p}.bf", pipeline.bf_header(nbytes, args.k, signature=args.bf_signature or pipeline.BF_SIGNATURE))
    bf.free()
    ctx.close()
    return 0


# ---- indexlr (the options ntSynt's workflow and ntJoin's run_indexlr pass) --------------------------------------------------------
def indexlr_parser():
    p = argparse.ArgumentParser(prog="indexlr", description="minimizers of every record of a FASTA file, btllib indexlr's text on stdout")
    p.add_argument("-k", required=True, type=int, help="k-mer size")
    p.add_argument("-w", required=True, type=int, help="window size")
    p.add_argument("--long", action="store_true", help="accepted (long-sequence mode is the only one)")
    p.add_argument("--seq", action="store_true", help="print the k-mer text of every minimizer")
    p.add_argument("--pos", action="store_true", help="print positions (always on: ntJoin needs them)")
    p.add_argument("-t", type=int, default=5, help="threads (accepted for compatibility)")
    p.add_argument("-s", metavar="BF", help="filter-in Bloom filter: only k-mers in it can be minimizers")
    p.add_argument("-r", metavar="BF", help="filter-out Bloom filter (repeat filter)")
    p.add_argument("-o", metavar="FILE", default="/dev/stdout", help="output file [stdout]")
    p.add_argument("--device", type=int, default=0)
    p.add_argument("fasta")
    return p


def indexlr(argv=None):
    args = indexlr_parser().parse_args(argv)
    from . import fasta as fa
    from . import pipeline
    from .device import BloomFilter, Context, sketch
    ctx = Context(args.device)
    filters = []
    for path in (args.s, args.r):
        bf = None
        if path:
            bits, k_file = pipeline.read_bf(path)
            if k_file != args.k:
                raise ValueError(f"{path}: built for k = {k_file}, -k is {args.k}")
            bf = BloomFilter(ctx, bits.size, args.k)
            bf.from_numpy(bits)
            del bits
        filters.append(bf)
    g, _ = fa.read_fasta_device(ctx, args.fasta)
    mx = sketch(ctx, g, args.k, args.w, filters[0], repeat=filters[1])
    h1, rec, pos = mx.to_numpy()
    km = mx.kmers(g, args.k) if args.seq else None
    fa.write_indexlr_tsv_kmers(args.o, g.recs, h1, rec, pos, args.k, km)
    return 0


# ---- ntsynt_run.py ----------------------------------------------------------------------------------------------------------------
def run_parser():
    "bin/ntsynt_run.py:10-44, flag for flag"
    p = argparse.ArgumentParser(prog="ntsynt_run.py", description="Run the dynamic minimizer graph stage of ntSynt")
    p.add_argument("FILES", nargs="+", help="Minimizer TSV files of input assemblies")
    p.add_argument("--fastas", nargs="+", help="Assembly fasta files", required=True, type=str)
    p.add_argument("-n", help="Minimum edge weight [Number of input assemblies]", default=0, type=int)
    p.add_argument("-p", help="Output prefix [out]", default="out", type=str, required=False)
    p.add_argument("-k", help="k-mer size used for minimizer step", required=True, type=int)
    p.add_argument("-w", help="Window size used for minimizers", required=True, type=int)
    p.add_argument("-z", help="Minimum synteny block size (bp) [500]", type=int, default=500)
    p.add_argument("--filter", help="Type of repeat filtering (experimental in the reference; Indexlr: the refinement rounds sketch with -r <repeat filter>; Filter: minimizers whose k-mer the repeat filter holds are not read)", choices=["Filter", "Indexlr"], type=str)
    p.add_argument("--common", help="Input common BF for minimizer selection", type=str)
    p.add_argument("--repeat", help="Repeat BF (must be included if --filter is specified)", type=str)
    p.add_argument("--btllib_t", help="accepted for compatibility: threads of the reference's btllib wrappers [4]", type=int, default=4)
    p.add_argument("--w-rounds", help="decreasing list of 'w' values to use for refining ends", default=[100, 10], nargs="+", type=int)
    p.add_argument("--bp", help="Maximum tolerated indel size [500]", default=500, type=int)
    p.add_argument("--collinear-merge", help="Maximum distance between collinear blocks for merging "
                   "(length in bp or string in the form '<num>w' to indicate multiples of w) [1w]", default="1w", type=str, required=False)
    p.add_argument("--simplify-graph", help="Run minimizer graph simplification", action="store_true")
    p.add_argument("-m", help="Require at least m %% of minimizer positions to be increasing/decreasing to assign contig orientation [90]",
                   default=90, type=int)
    p.add_argument("--dev", action="store_true", help="Developer mode - more verbose logging, overlap self-check")
    p.add_argument("--interarrivals", action="store_true", help="Output interarrival distances in initial graph")
    p.add_argument("--initial-only", action="store_true", help=argparse.SUPPRESS)   # this build's: stop after the initial round's table (no FASTA read)
    p.add_argument("--device", help="GPU index [0]", type=int, default=0)
    p.add_argument("-v", "--version", action="version", version=NTSYNT_VERSION)
    return p


def collinear_merge_bp(text, w):
    "bin/ntsynt_synteny.py:37-42: '<n>w' -> n * w, else int; ValueError otherwise"
    m = re.search(r"^(\d+)w$", text)
    if m:
        return int(m.group(1)) * w
    if text.isdigit():
        return int(text)
    raise ValueError("--collinear-merge must be provided with an integer value or string in the form '<num>w'")


FAI_RE = re.compile(r'^(\S+).k\d+.w\d+.tsv')          # bin/ntsynt_synteny.py:25, as written there (unescaped dots, no end anchor)


def fasta_name_of(tsv_name):
    """bin/ntsynt_synteny.py:108-116 (find_fa_name): the FASTA a minimizer TSV belongs to is its name up to '.k<k>.w<w>.tsv'; a name
    that does not follow the convention ends the run with the reference's message and exit status 1"""
    m = FAI_RE.search(tsv_name)
    if m:
        return m.group(1)
    print("ERROR: Target assembly minimizer TSV file must follow the naming convention:")
    print("\ttarget_assembly.fa.k<k>.w<w>.tsv, where <k> and <w> are parameters used for minimizering")
    sys.exit(1)


def pair_files(tsvs, fastas):
    """the FASTA of every minimizer TSV: '<basename of the FASTA>.k<k>.w<w>.tsv' (rule indexlr's output name, smk:78; the reference
    matches them the same way, by the name with the suffix stripped: synteny_block.py:14,76-77, S:137-144)"""
    by_base = {os.path.basename(f): f for f in fastas}
    out = []
    for t in tsvs:
        base = fasta_name_of(os.path.basename(t))
        if base not in by_base:
            raise ValueError(f"{t}: no FASTA named {base} among --fastas")
        out.append(by_base[base])
    if len(set(out)) != len(out):
        raise ValueError("two minimizer TSVs name the same FASTA")
    return out


def run(argv=None):
    print(f"Running {NTSYNT_VERSION}", flush=True)
    args = run_parser().parse_args(argv)
    if args.filter and not args.repeat:
        raise ValueError("If --filter is specified, must supply repeat Bloom filter with --repeat")     # bin/ntsynt_synteny.py:598-599
    if args.filter == "Filter" and args.initial_only:
        print("--filter Filter reads the FASTA files (the lists are screened against their k-mers): not with --initial-only", file=sys.stderr)
        return 2
    merge = collinear_merge_bp(args.collinear_merge, args.w)
    fastas = pair_files(args.FILES, args.fastas)
    if not args.initial_only:
        for f in fastas:
            if not os.path.isfile(f):
                raise FileNotFoundError(f"Input file {f} not found.")
    # parameter echo of bin/ntsynt_synteny.py:44-63 (the files in the order the reference holds them: descending, S:34; the
    # collinear-merge value as given -- the reference prints it before resolving '<n>w', S:33-42)
    print("Parameters:")
    print("\tMinimizer TSV files: ", sorted(args.FILES, reverse=True))
    print("\t--fastas", args.fastas)
    print("\t-n", args.n or len(args.FILES))
    print("\t-p", args.p)
    print("\t-k", args.k)
    print("\t-w", args.w)
    print("\t--btllib_t", args.btllib_t)
    print("\t--w-rounds", args.w_rounds)
    print("\t-m", args.m)
    print("\t-z", args.z)
    print("\t--collinear-merge", args.collinear_merge, flush=True)
    if args.common:
        print("\t--common", args.common, flush=True)
    if args.repeat:
        print("\t--repeat", args.repeat, flush=True)
    from . import pipeline
    pipeline.run(fastas, k=args.k, w=args.w, prefix=args.p, w_rounds=args.w_rounds, indel=args.bp, merge=merge, block_size=args.z,
                 common=False, common_file=args.common, simplify=args.simplify_graph, device=args.device, mx_tsvs=list(args.FILES),
                 m=args.m, n=args.n, dev=args.dev, interarrivals=args.interarrivals, initial_only=args.initial_only, write_fai=False,
                 refine_repeat_file=args.repeat if args.filter == "Indexlr" else None,
                 screen_repeat_file=args.repeat if args.filter == "Filter" else None,
                 log=print if args.dev else (lambda *a, **k: None))
    print("Done ntSynt synteny stage", flush=True)
    return 0
