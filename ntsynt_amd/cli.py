"""`ntSynt` command line of the HIP path: same flags, defaults, validation and messages as the
reference driver (bin/ntSynt:43-170), with the Snakemake hop replaced by an in-process GPU pipeline.

Differences a user can see: `-t` is accepted but unused (the GPU does the work), `-n/--dry-run` prints
the stage plan instead of Snakemake's, `--benchmark` writes {prefix}.stage_times.tsv, `-f/--force` is
accepted (every run recomputes everything)."""
import argparse
import os
import sys

NTSYNT_VERSION = "ntSynt v1.0.4 (ntsynt_amd / MI355X)"

NTSYNT_BANNER = "ntSynt on MI355X -- minimizer-graph macrosynteny, sketch / Bloom filter / graph stage in HBM"


def read_fasta_files(filename):
    "one FASTA path per line (bin/ntSynt:25-31)"
    with open(filename, "r", encoding="utf-8") as fin:
        return [line.strip() for line in fin]


def build_parser():
    epilog = "\n".join([
        "Parameters derived from -d unless given explicitly (the reference's table, bin/ntSynt:89-99):",
        "  -d below 1      block_size 500    indel 10000    merge 10000     w_rounds 100 10",
        "  -d 1 to 10      block_size 1000   indel 50000    merge 100000    w_rounds 250 100",
        "  -d above 10     block_size 10000  indel 100000   merge 1000000   w_rounds 500 250",
    ])
    p = argparse.ArgumentParser(prog="ntSynt", description="Macrosynteny blocks of two or more genome assemblies from a minimizer graph "
                                "(ntSynt's method and command line; all sequence-scale work on the GPU)",
                                formatter_class=argparse.RawTextHelpFormatter, epilog=epilog)
    p.add_argument("fastas", help="genome assemblies (FASTA, plain or .gz), two or more", nargs="*")
    p.add_argument("--fastas_list", help="text file naming the assemblies, one path per line (instead of positional arguments)",
                   required=False, type=str)
    p.add_argument("-d", "--divergence", help="upper estimate of the sequence divergence between the assemblies, in percent (-d 1 = 1%%);\n"
                   "selects --indel, --merge, --w_rounds and --block_size (table below)", required=True, type=float)
    p.add_argument("-p", "--prefix", help="prefix of the output files [ntSynt.k<k>.w<w>]", required=False)
    p.add_argument("-k", help="k-mer size of the minimizers [24]", type=int, required=False, default=24)
    p.add_argument("-w", help="window size of the minimizers [1000]", type=int, required=False, default=1000)
    p.add_argument("-t", help="threads [12]: accepted for compatibility with the reference, the GPU path does not use it", type=int, default=12)
    p.add_argument("--fpr", help="false positive rate the common Bloom filter is sized for [0.025]", default=0.025, type=float)
    p.add_argument("-b", "--block_size", help="shortest synteny block reported (bp)", type=int, required=False)
    p.add_argument("--merge", help="collinear blocks closer than this are merged (bp, or a multiple of the window size such as 3w)", type=str)
    p.add_argument("--w_rounds", help="window sizes of the refinement rounds, decreasing", nargs="+", type=int)
    p.add_argument("--indel", help="largest difference between assemblies in the distance of neighbouring minimizers before a block is split (bp)",
                   type=int)
    p.add_argument("--no-common", help=argparse.SUPPRESS, action="store_true")
    p.add_argument("--no-simplify-graph", help=argparse.SUPPRESS, action="store_true")
    p.add_argument("-n", "--dry-run", help="list the stages that would run, then stop", action="store_true")
    p.add_argument("--benchmark", help="write the wall-clock time of every stage to <prefix>.stage_times.tsv", action="store_true")
    p.add_argument("-f", "--force", help="accepted for compatibility (every run recomputes everything)", action="store_true")
    p.add_argument("--dev", help="developer mode: verbose log, overlap self-check of the final blocks", action="store_true")
    p.add_argument("--repeat", help=argparse.SUPPRESS, action="store_true")   # the Snakefile's experimental config "repeat": <prefix>.repeat.bf + indexlr -r
    p.add_argument("--interarrivals", help=argparse.SUPPRESS, action="store_true")   # ntsynt_run.py --interarrivals: <prefix>.interarrivals.tsv
    p.add_argument("--device", help="GPU index [0]", type=int, default=0)
    # switches for the two btllib details this implementation recalls rather than reads
    p.add_argument("--bf-rounding", help=argparse.SUPPRESS, choices=["up", "down", "none"], default="up")
    p.add_argument("--bf-signature", help=argparse.SUPPRESS, default=None)
    p.add_argument("-v", "--version", action="version", version=NTSYNT_VERSION)
    return p


def resolve(parser, args):
    "divergence -> defaults and input validation (bin/ntSynt:86-120,141-143)"
    if not args.prefix:
        args.prefix = f"ntSynt.k{args.k}.w{args.w}"
    if args.divergence < 1:
        args.indel, args.merge, args.w_rounds, args.block_size = \
            args.indel or 10000, args.merge or 10000, args.w_rounds or [100, 10], args.block_size or 500
    elif 1 <= args.divergence <= 10:
        args.indel, args.merge, args.w_rounds, args.block_size = \
            args.indel or 50000, args.merge or 100000, args.w_rounds or [250, 100], args.block_size or 1000
    elif 10 < args.divergence <= 100:
        args.indel, args.merge, args.w_rounds, args.block_size = \
            args.indel or 100000, args.merge or 1000000, args.w_rounds or [500, 250], args.block_size or 10000
    else:
        parser.error("--divergence must be a value between 0 and 100")
    for w in args.w_rounds:
        if w > args.w:
            parser.error("All values specified for --w_rounds must be smaller than -w")
    if not args.fastas and not args.fastas_list:
        parser.error("Please supply the input genome fasta files as positional arguments, "
                     "or specify a file listing the files (one fasta per line) with --fastas_list")
    if args.fastas and args.fastas_list:
        parser.error("Please supply the input genome fasta files as positional arguments, "
                     "or specify a single file (one fasta per line) with --fastas_list, NOT both.")
    fastas = read_fasta_files(args.fastas_list) if args.fastas_list else args.fastas
    if len(fastas) < 2:
        parser.error("Must supply at least two reference genomes to compare")
    return fastas


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    fastas = resolve(parser, args)
    rank0 = int(os.environ.get("RANK", "0")) == 0                # under torchrun every rank runs this; one of them talks
    say = print if rank0 else (lambda *a, **k: None)
    say(NTSYNT_BANNER)
    say("\n".join(["Running ntSynt...",
                     f"Specified percent divergence: {args.divergence}",
                     "Parameter settings:",
                     f"\tfastas {fastas}",
                     f"\t--divergence {args.divergence}",
                     f"\t--block_size {args.block_size}",
                     f"\t--merge {args.merge}",
                     f"\t--w_rounds {args.w_rounds}",
                     f"\t--indel {args.indel}",
                     f"\t-p {args.prefix}",
                     f"\t-k {args.k}",
                     f"\t-w {args.w}",
                     f"\t-t {args.t}",
                     f"\t--fpr {args.fpr}"]), flush=True)
    if not args.no_common:
        say(f"Note: {args.prefix}.common.bf is written in btllib's Bloom filter layout as recalled from its source "
            "(header table name: --bf-signature); a stock btllib is not guaranteed to load it.")
    for fasta in fastas:
        if not os.path.isfile(fasta):
            raise FileNotFoundError(f"Input file {fasta} not found.")
    plan = ["faidx x%d" % len(fastas)] + ([] if args.no_common else ["make_common_bf"]) + \
           ["indexlr x%d" % len(fastas), "ntsynt_synteny"]
    if args.dry_run:
        say("Stages (GPU, in process):", " -> ".join(plan))
        return 0
    import subprocess

    def stage_failed(cause=None):
        "bin/ntSynt:166-170: a stage that stops (its message is on the terminal already) ends the run with this error, exit status 1"
        raise subprocess.SubprocessError("ntSynt failed - check the logs for the error.") from cause
    if len(args.w_rounds) != len(set(args.w_rounds)):          # stage 3's own check (bin/ntsynt_synteny.py:597-599): not under -n
        print("Error: duplicate values found in w_rounds!", file=sys.stderr, flush=True)
        stage_failed()
    from . import pipeline
    # one process per GPU under `python -m torch.distributed.run --nproc-per-node N bin/ntSynt ...`:
    # genomes are sharded over the ranks (ntsynt_amd/pipeline.py), rank 0 writes the outputs
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = args.device
    if world > 1:
        import torch
        import torch.distributed as dist
        device = int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # The process group only hands the communicator id round and synchronises the ranks; the two exchanges run inside
        # libntsynt_hip.so (nts_bf_allreduce_and, nts_mx_allgather).  NTS_DIST_BACKEND=gloo + NTS_RCCL_LIB=<stand-in>: ranks that
        # share GPUs (a box with fewer GPUs than ranks; tests/test_gpu_multirank.py); production is nccl (= RCCL) throughout
        backend = os.environ.get("NTS_DIST_BACKEND", "nccl")
        if backend != "nccl":
            device %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(device)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
    quiet = (lambda *a, **k: None)
    try:
        _run(pipeline, fastas, args, device, quiet)
    except SystemExit as exc:                                  # a stage's own exit ("no paths found", S:630-632)
        if exc.code in (0, None):
            raise
        stage_failed()
    except Exception as exc:                                   # noqa: BLE001 -- whatever stopped a stage: its traceback is the log
        stage_failed(exc)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if int(os.environ.get("RANK", "0")) == 0:
        print("Done ntSynt!")
    return 0


def _run(pipeline, fastas, args, device, quiet):
    pipeline.run(fastas, k=args.k, w=args.w, fpr=args.fpr, prefix=args.prefix, w_rounds=args.w_rounds,
                 indel=args.indel, merge=args.merge, block_size=args.block_size, common=not args.no_common,
                 simplify=not args.no_simplify_graph, device=device, benchmark=args.benchmark,
                 dev=args.dev, interarrivals=args.interarrivals, repeat=args.repeat, bf_rounding=args.bf_rounding, bf_signature=args.bf_signature or pipeline.BF_SIGNATURE,
                 log=print if (args.dev and int(os.environ.get("RANK", "0")) == 0) else quiet)


if __name__ == "__main__":
    sys.exit(main())
