#!/usr/bin/env python3
"""Instruction mix of k_hash_select_hi's tile loop, counted from the ISA of the built library (not assumed).

Reads the gfx950 code object out of ntsynt_amd/libntsynt_hip.so, disassembles the <HAS_BF = true, PER = 2> instantiation and
walks the straight-line code of one turn of the tile loop: from the loop's first stage load to the wait for the probes, i.e.
geometry, stage loads, probe addresses, the 64-step roll, listing and full hashes -- the part every tile executes (branches
into the rare paths -- tiles that span runs, lists that overflow -- are not followed: their targets lie outside the range).
Classes: VOP1/VOP2-style two-operand 32-bit operations (the v_xor_b32 issue rate of nts_bench_valu), three-operand and
64-bit operations (the v_alignbit_b32 / v_add3_u32 / v_lshl_add_u64 rate), compares + carry adds, LDS, scalar.

  python profiles/valu_mix.py > profiles/r03_valu_mix.json"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
SYMBOL = "_ZN12_GLOBAL__N_116k_hash_select_hiILb1ELj2EEEvNS_9SelParamsEjjm"
THREE_OPERAND = ("v_alignbit", "v_add3", "v_lshl_add", "v_lshl_or", "v_and_or", "v_or3", "v_xad", "v_bfe", "v_bfi", "v_perm", "v_mad", "v_mul_hi",
                 "v_mul_lo", "v_add_lshl", "v_xor3", "v_cndmask", "v_lshlrev_b64", "v_lshrrev_b64", "v_readlane", "v_writelane", "v_mbcnt",
                 "v_bcnt", "v_min3", "v_max3", "v_med3", "v_fma", "v_cvt", "v_ashrrev_i64")


def classify(op):
    if op.startswith("v_cmp") or op.startswith("v_addc") or op.startswith("v_subb") or op.startswith("v_add_co") or op.startswith("v_sub_co"):
        return "compare / carry"
    if op.startswith("v_"):
        return "three-operand or 64-bit" if op.startswith(THREE_OPERAND) or op.endswith("_e64") else "two-operand 32-bit"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("s_"):
        return "scalar"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vector memory"
    return "other"


def main():
    lib = os.path.join(ROOT, "ntsynt_amd", "libntsynt_hip.so")
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(d, "unused.so")], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--output={co}"], check=True)
        txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", f"--disassemble-symbols={SYMBOL}", co], check=True,
                             capture_output=True, text=True).stdout
    ins = [ln.split("//")[0].strip() for ln in txt.splitlines() if ln.startswith("\t")]
    stage = [n for n, i in enumerate(ins) if i.startswith("global_load_lds_dwordx4")]
    first = stage[2]                                                    # the loop's first stage load
    wait_roll = next(n for n in range(first, len(ins)) if ins[n] == "s_waitcnt vmcnt(4)")
    wait_probes = next(n for n in range(wait_roll, len(ins)) if ins[n] == "s_waitcnt vmcnt(0)")
    body = ins[first:wait_probes]
    ops = collections.Counter(i.split()[0] for i in body)
    classes = collections.Counter()
    for op, c in ops.items():
        classes[classify(op)] += c
    valu = sum(c for k, c in classes.items() if k in ("two-operand 32-bit", "three-operand or 64-bit", "compare / carry"))
    out = {"kernel": "k_hash_select_hi<true, 2>", "range": "one turn of the tile loop: first stage load .. wait for the probes (static, rare branches not followed)",
           "instructions": len(body), "valu_instructions": valu, "valu_per_64_kmers_static": round(valu / 64.0, 2),
           "classes": dict(classes), "top_opcodes": dict(ops.most_common(24))}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
