"""`ntSynt` command line of the HIP path: same flags, defaults, validation and messages as the
reference driver (bin/ntSynt:43-170), with the Snakemake hop replaced by an in-process GPU pipeline.

Differences a user can see: `-t` is accepted but unused (the GPU does the work), `-n/--dry-run` prints
the stage plan instead of Snakemake's, `--benchmark` writes {prefix}.stage_times.tsv, `-f/--force` is
accepted (every run recomputes everything)."""
import argparse
import os
import sys

NTSYNT_VERSION = "ntSynt v1.0.4 (ntsynt_amd / MI355X)"

NTSYNT_ASCII = r"""
        _    ____                 _
 _ __  | |_ / ___|  _   _  _ __  | |_
| '_ \ | __|\___ \ | | | || '_ \ | __|
| | | || |_  ___) || |_| || | | || |_
|_| |_| \__||____/  \__, ||_| |_| \__|
                    |___/
"""


def read_fasta_files(filename):
    "one FASTA path per line (bin/ntSynt:25-31)"
    with open(filename, "r", encoding="utf-8") as fin:
        return [line.strip() for line in fin]


def build_parser():
    epilog = "\n".join([
        "Default parameter settings for divergence values:",
        "< 1% divergence:\t--block_size 500 --indel 10000 --merge 10000 --w_rounds 100 10",
        "1% - 10% divergence:\t--block_size 1000 --indel 50000 --merge 100000 --w_rounds 250 100",
        "> 10% divergence:\t--block_size 10000 --indel 100000 --merge 1000000 --w_rounds 500 250",
        "If any of these parameters are set manually, those values will override the above.",
    ])
    p = argparse.ArgumentParser(prog="ntSynt",
                                description="ntSynt: Multi-genome synteny detection using minimizer graphs",
                                formatter_class=argparse.RawTextHelpFormatter, epilog=epilog)
    p.add_argument("fastas", help="Input genome fasta files", nargs="*")
    p.add_argument("--fastas_list", help="File listing input genome fasta files, one per line", required=False, type=str)
    p.add_argument("-d", "--divergence",
                   help="Approx. maximum percent sequence divergence between input genomes (Ex. -d 1 for 1%% divergence).\n"
                        "This will be used to set --indel, --merge, --w_rounds, --block_size",
                   required=True, type=float)
    p.add_argument("-p", "--prefix", help="Prefix for ntSynt output files [ntSynt.k<k>.w<w>]", required=False)
    p.add_argument("-k", help="Minimizer k-mer size [24]", type=int, required=False, default=24)
    p.add_argument("-w", help="Minimizer window size [1000]", type=int, required=False, default=1000)
    p.add_argument("-t", help="Number of threads [12] (accepted for compatibility; the GPU path ignores it)", type=int, default=12)
    p.add_argument("--fpr", help="False positive rate for Bloom filter creation [0.025]", default=0.025, type=float)
    p.add_argument("-b", "--block_size", help="Minimum synteny block size (bp)", type=int, required=False)
    p.add_argument("--merge", help="Maximum distance between collinear synteny blocks for merging (bp). \n"
                                   "Can also specify a multiple of the window size (ex. 3w)", type=str)
    p.add_argument("--w_rounds", help="List of decreasing window sizes for synteny block refinement", nargs="+", type=int)
    p.add_argument("--indel", help="Threshold for indel detection (bp)", type=int)
    p.add_argument("--no-common", help=argparse.SUPPRESS, action="store_true")
    p.add_argument("--no-simplify-graph", help=argparse.SUPPRESS, action="store_true")
    p.add_argument("-n", "--dry-run", help="Print out the stages that will be executed", action="store_true")
    p.add_argument("--benchmark", help="Store wall-clock times for each step of the ntSynt pipeline", action="store_true")
    p.add_argument("-f", "--force", help="Run all ntSynt steps, regardless of existing output files", action="store_true")
    p.add_argument("--dev", help="Run in developer mode: more verbose logging", action="store_true")
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
    say(NTSYNT_ASCII)
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
    for fasta in fastas:
        if not os.path.isfile(fasta):
            raise FileNotFoundError(f"Input file {fasta} not found.")
    if len(args.w_rounds) != len(set(args.w_rounds)):          # bin/ntsynt_synteny.py:597-599
        print("Error: duplicate values found in w_rounds!", file=sys.stderr, flush=True)
        sys.exit(1)
    plan = ["faidx x%d" % len(fastas)] + ([] if args.no_common else ["make_common_bf"]) + \
           ["indexlr x%d" % len(fastas), "ntsynt_synteny"]
    if args.dry_run:
        say("Stages (GPU, in process):", " -> ".join(plan))
        return 0
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
        # NTS_DIST_BACKEND=gloo: verification mode for boxes with fewer GPUs than ranks (ranks share the GPUs, the
        # collectives run on host copies: ntsynt_amd/pipeline.py GpuBackend.host_comm); production is nccl (= RCCL)
        backend = os.environ.get("NTS_DIST_BACKEND", "nccl")
        if backend != "nccl":
            device %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(device)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
    quiet = (lambda *a, **k: None)
    pipeline.run(fastas, k=args.k, w=args.w, fpr=args.fpr, prefix=args.prefix, w_rounds=args.w_rounds,
                 indel=args.indel, merge=args.merge, block_size=args.block_size, common=not args.no_common,
                 simplify=not args.no_simplify_graph, device=device, benchmark=args.benchmark,
                 dev=args.dev, bf_rounding=args.bf_rounding, bf_signature=args.bf_signature or pipeline.BF_SIGNATURE,
                 log=print if (args.dev and int(os.environ.get("RANK", "0")) == 0) else quiet)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if int(os.environ.get("RANK", "0")) == 0:
        print("Done ntSynt!")
    return 0


if __name__ == "__main__":
    sys.exit(main())
