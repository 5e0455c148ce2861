"""In-process equivalent of ntSynt's Snakemake workflow (bin/ntsynt_run_pipeline.smk) on the GPU:

    faidx (smk:48-53) -> make_common_bf (smk:55-62) -> indexlr per genome (smk:74-85)
        -> ntsynt_run.py graph stage (smk:87-103)

with the same artefact names in the CWD: {basename}.fai, {prefix}.common.bf,
{basename}.k{k}.w{w}.tsv, {prefix}.synteny_blocks.tsv, {prefix}.pre-collinear-merge.synteny_blocks.tsv.
All sequence-scale compute runs in libntsynt_hip.so; there is no CPU fallback."""
import os
import time

import numpy as np

from . import fasta as fa
from .device import BloomFilter, Context, Genome, bf_size_bytes, sketch
from .graph import build_graph_device, walk_chains
from .synteny import SyntenyEngine


def write_bf(path, bits, k, hash_num=1):
    """btllib KmerBloomFilter file: TOML-style header + raw bit array (layout recalled from btllib's
    BloomFilter::save, not verifiable here: SURVEY.md 8(f) rank 3)."""
    header = (f"[BTLKmerBloomFilter_v5]\nbytes = {bits.size}\nhash_fn = \"ntHash_v2\"\n"
              f"hash_num = {hash_num}\nk = {k}\n\n[HeaderEnd]\n")
    with open(path, "wb") as fh:
        fh.write(header.encode())
        fh.write(bits.tobytes())


def read_bf(path):
    with open(path, "rb") as fh:
        data = fh.read()
    end = data.index(b"[HeaderEnd]\n") + len(b"[HeaderEnd]\n")
    meta = {}
    for line in data[:end].decode().splitlines():
        if "=" in line:
            key, val = [x.strip() for x in line.split("=", 1)]
            meta[key] = val.strip('"')
    return np.frombuffer(data[end:], dtype=np.uint8).copy(), int(meta["k"])


def write_indexlr_tsv(path, recs, h1, rec, pos, k, with_seq=True):
    """`indexlr --long --pos [--seq]` text: one line per FASTA record."""
    n_rec = len(recs.names)
    bounds = np.searchsorted(rec, np.arange(n_rec + 1))
    with open(path, "w", encoding="utf-8") as out:
        for r, name in enumerate(recs.names):
            lo, hi = int(bounds[r]), int(bounds[r + 1])
            hs, ps = h1[lo:hi].tolist(), pos[lo:hi].tolist()
            if with_seq:
                seq = recs.record_bytes(r)
                toks = [f"{h}:{p}:{seq[p:p + k].tobytes().decode().upper()}" for h, p in zip(hs, ps)]
            else:
                toks = [f"{h}:{p}" for h, p in zip(hs, ps)]
            out.write(f"{name}\t{' '.join(toks)}\n")


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

    def write(self, path):
        with open(path, "w", encoding="utf-8") as fh:
            for name, dt in self.rows:
                fh.write(f"{name}\t{dt:.6f}\n")


def run(fastas, k=24, w=1000, fpr=0.025, prefix=None, w_rounds=(100, 10), indel=10000, merge=10000,
        block_size=500, common=True, simplify=True, device=0, write_mx_tsv=True, mx_with_seq=True,
        benchmark=False, log=print, ctx=None):
    """FASTA paths -> {output file name: text}.  Mirrors oracle.synteny_oracle.run_pipeline's signature so the
    parity tests read alike; every stage here runs on the GPU."""
    prefix = prefix or f"ntSynt.k{k}.w{w}"
    st = Stages()
    own_ctx = ctx is None
    ctx = ctx or Context(device)
    st.start("read_fasta+upload")
    recs = {p: fa.read_fasta(p) for p in fastas}
    genomes = {}
    for p in fastas:
        r = recs[p]
        genomes[p] = Genome(ctx, r.names, r.seq, r.rec_off, r.rec_len)
        fa.write_fai(f"{fa.basename(p)}.fai", r)
    st.stop()

    bf = None
    if common:
        st.start("make_common_bf")
        ordered = sorted(fastas)                               # src/ntsynt_make_common_bf.cpp:105-107
        approx, nbytes = bf_size_bytes(genomes[ordered[0]].total_bp, fpr)
        log(f"Genome size (bp): {genomes[ordered[0]].total_bp}")
        log(f"BF size (bytes): {approx}")
        bf = BloomFilter(ctx, nbytes, k)
        bf.insert(genomes[ordered[0]])
        log(f"Bloom filter FPR: {bf.get_fpr()}")
        if len(ordered) > 1:
            tmp = BloomFilter(ctx, nbytes, k)
            for p in ordered[1:]:
                tmp.clear()
                tmp.insert(genomes[p])                          # G_i; AND == cascade level
                bf.and_(tmp)
                log(f"Bloom filter FPR: {bf.get_fpr()}")
            tmp.free()
        log(f"Final Bloom filter FPR: {bf.get_fpr()}")
        write_bf(f"{prefix}.common.bf", bf.to_numpy(), k)
        st.stop()

    st.start("indexlr")
    tsv_names, initial = [], []
    for p in fastas:
        mx = sketch(ctx, genomes[p], k, w, bf)
        h1, rec, pos = mx.to_numpy()
        mx.free()
        tsv = f"{fa.basename(p)}.k{k}.w{w}.tsv"
        if write_mx_tsv:
            write_indexlr_tsv(tsv, recs[p], h1, rec, pos, k, mx_with_seq)
        tsv_names.append(tsv)
        initial.append((h1, rec, pos))
    st.stop()

    st.start("ntsynt_synteny")

    def graph_fn(lists, keeps, list_ids):
        return build_graph_device(ctx, lists, keeps, list_ids)

    def sketch_fn(i, masks, new_w):
        mx = sketch(ctx, genomes[fastas[i]], k, new_w, bf, masks)
        out = mx.to_numpy()
        mx.free()
        return out

    eng = SyntenyEngine(tsv_names, [recs[p].names for p in fastas], k, w, w_rounds, indel, merge, block_size, prefix,
                        graph_fn, sketch_fn, walk_chains, simplify=simplify, log=log)
    outputs = eng.run(initial)
    st.stop()
    if benchmark:
        st.write(f"{prefix}.stage_times.tsv")
    for g in genomes.values():
        g.free()
    if bf is not None:
        bf.free()
    if own_ctx:
        ctx.close()
    eng.stage_times = st.rows
    return eng
