"""Build-time guard for the hand-scheduled kernel (csrc/nts_pruned.inc k_hash_select_hi).

Its software pipeline issues its loads with explicit instructions and waits with hand-counted `s_waitcnt vmcnt(N)`
(N = the LDS-DMA loads of the next tile + the Bloom probes of the previous one).  That arithmetic holds only while the
compiler puts no vector-memory operation of its own between them -- a spilled register or a struct copied through
scratch has done that before (nts_pruned.inc, comments at the geometry variables).  This test reads the gfx950 code object
out of the built library and checks the four instantiations, so that a toolchain or source change that breaks the
assumption fails the CPU suite here instead of a stress run on a GPU box."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ntsynt_amd", "libntsynt_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
VMEM = re.compile(r"^\s*(global_|flat_|buffer_|scratch_|tbuffer_)\w+")


@pytest.fixture(scope="module")
def code_object(tmp_path_factory):
    assert os.path.exists(LIB), "libntsynt_hip.so is not built (__graft_entry__.build())"
    for tool in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump", "llvm-readelf"):
        if not os.path.exists(os.path.join(LLVM, tool)):
            pytest.skip(f"{tool} not in {LLVM}")
    d = tmp_path_factory.mktemp("co")
    fat, co = str(d / "fat.bin"), str(d / "dev.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", LIB, str(d / "unused.so")], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    return co, notes


def _kernel_meta(notes, symbol):
    "the metadata block of one kernel in the code object's note record"
    at = notes.index(f".name:           {symbol}\n")
    start = notes.rfind("  - .agpr_count", 0, at)
    if start < 0:
        start = notes.rfind("  - .args", 0, at)
    end = notes.find("\n  - ", at)
    block = notes[start:end if end > 0 else None]
    return {m.group(1): m.group(2) for m in re.finditer(r"\.(\w+):\s+(\S+)", block)}


def _disassemble(co, symbol):
    out = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", f"--disassemble-symbols={symbol}", co], check=True,
                         capture_output=True, text=True).stdout
    return [ln.split("//")[0].strip() for ln in out.splitlines() if ln.startswith("\t")]


@pytest.mark.parametrize("has_bf,per", [(True, 2), (True, 4), (False, 2), (False, 4)])
def test_select_hi_pipeline_has_only_its_own_vector_memory_operations(code_object, has_bf, per):
    co, notes = code_object
    symbol = f"_ZN12_GLOBAL__N_116k_hash_select_hiILb{int(has_bf)}ELj{per}EEEvNS_9SelParamsEjjm"
    assert symbol in notes, "k_hash_select_hi instantiation not found (renamed? update this guard with it)"
    meta = _kernel_meta(notes, symbol)
    # nothing of the kernel lives in scratch memory: a spill would travel through the queue the waits count
    assert meta["private_segment_fixed_size"] == "0", meta
    assert meta.get("vgpr_spill_count", "0") == "0", meta          # (scalar spills go to lanes of a vector register, not to memory)
    ins = _disassemble(co, symbol)
    assert len(ins) > 1000
    assert not [i for i in ins if i.startswith(("scratch_", "buffer_", "tbuffer_"))]
    # the explicit loads: the stage (two 16-byte LDS-DMA loads, in the prologue and in the loop) and `per` probes in the loop
    assert sum(i.startswith("global_load_lds_dwordx4") for i in ins) == 4
    assert sum(i.startswith("global_load_lds_dword ") for i in ins) == (2 * per if has_bf else 0)     # loop + the last tile behind it
    # the wait that lets the rolling start: everything but the loads issued in this turn has landed.  Walk back from it: the
    # vector-memory operations since the loop's first stage load are exactly [stage, stage, probe x per]
    n_wait = per + 2 if has_bf else 2
    stage = [n for n, i in enumerate(ins) if i.startswith("global_load_lds_dwordx4")]
    loop_first_stage = stage[2]
    waits = [n for n, i in enumerate(ins) if i == f"s_waitcnt vmcnt({n_wait})" and n > loop_first_stage]
    assert waits, f"no s_waitcnt vmcnt({n_wait}) behind the loop's stage loads"
    wait_at = waits[0]
    between = [i.split()[0] for i in ins[loop_first_stage:wait_at] if VMEM.match(i)]
    assert between == ["global_load_lds_dwordx4"] * 2 + ["global_load_lds_dword"] * (per if has_bf else 0), between
    # and the compiler did not slip a wait of its own for vector memory in between (it would show a load it tracks there)
    assert not [i for i in ins[loop_first_stage:wait_at] if i.startswith("s_waitcnt") and "vmcnt" in i]
    if has_bf:
        # the probes are read from LDS only behind a full wait
        after = ins[wait_at + 1:]
        assert any(i == "s_waitcnt vmcnt(0)" for i in after)


@pytest.mark.parametrize("has_bf,per", [(True, 2), (True, 4), (False, 2), (False, 4)])
def test_select_hi_m0_is_only_ever_used_next_to_its_own_write(code_object, has_bf, per):
    """The LDS-DMA loads take their LDS address from m0, which the kernel's inline asm writes itself (`s_mov_b32 m0, sN;
    s_nop 0; global_load_lds_*` in ONE asm statement each, csrc/nts_pruned.inc stage_dma / issue_probes).  m0 is a reserved
    register: a clobber of it is ignored by the compiler (hence none is written; with one the code object is the same byte for byte),
    which uses m0 as a scratch scalar of its own -- today for the lane
    index of `v_writelane_b32 v, s, m0`, written by an `s_mov_b32 m0` right in front.  Both uses are safe only while every reader
    of m0 sits directly behind the write that serves it, with nothing of the other party in between.  This walks the ISA of the
    four instantiations and checks exactly that."""
    co, notes = code_object
    symbol = f"_ZN12_GLOBAL__N_116k_hash_select_hiILb{int(has_bf)}ELj{per}EEEvNS_9SelParamsEjjm"
    ins = _disassemble(co, symbol)
    writes = [n for n, i in enumerate(ins) if re.match(r"s_\w+ m0\b", i)]
    assert all(ins[n].startswith("s_mov_b32 m0, s") for n in writes), [ins[n] for n in writes]
    dma = [n for n, i in enumerate(ins) if i.startswith("global_load_lds_")]
    assert len(dma) == 4 + (2 * per if has_bf else 0)
    ours = set()
    for n in dma:            # every LDS-DMA load: its own write of m0 two instructions ahead, the s_nop between them, nothing else
        assert ins[n - 1] == "s_nop 0" and ins[n - 2].startswith("s_mov_b32 m0, s"), ins[n - 3:n + 1]
        ours.add(n - 2)
    readers = [n for n, i in enumerate(ins) if re.search(r"\bm0\b", i) and n not in writes]
    for n in readers:        # every other reader of m0 (the compiler's): served by a write of the compiler's own, straight behind it
        back = n - 1
        while back not in writes:
            # between the write and its reader: straight-line scalar / vector ALU work, none of our asm, no branch
            assert not ins[back].startswith(("global_load_lds_", "s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")), \
                (ins[back], "between a write of m0 and", ins[n])
            back -= 1
        assert back not in ours, ("a compiler-made reader of m0 behind the kernel's own write", ins[back:n + 1])
        assert n - back <= 8, ins[back:n + 1]
    # and the compiler's writes are all consumed where they stand (no value of m0 is expected to survive one of our asm blocks)
    for n in writes:
        if n not in ours:
            assert any(m in readers for m in range(n + 1, n + 4)), ins[n:n + 4]


def test_bloom_build_kernels_keep_their_occupancy_and_batched_loads(code_object):
    """k_bin1 (persistent, two workgroups of 1024 lanes per CU) and k_bin2 (four of 512) are tuned to 64 registers; what made
    them fast was found in the ISA (csrc/nts_bloom_bin.inc): no scratch traffic on the tile paths -- in k_bin1 a
    spill reload would queue behind the stores of the tile before --, and k_bin2's four 16-byte loads of a whole tile issued
    together (the optimiser sinks each load to its first use unless a statement takes all sixteen values)."""
    co, notes = code_object
    k2 = "_ZN12_GLOBAL__N_16k_bin2ENS_10Bin2ParamsE"
    k1s = [f"_ZN12_GLOBAL__N_16k_bin1ILi{form}EEEvNS_9BinParamsE" for form in (0, 1, 2)]      # one instantiation per form of `h mod bits`
    for sym in k1s + [k2]:
        assert sym in notes, f"{sym} not found (renamed? update this guard with it)"
    m2 = _kernel_meta(notes, k2)
    assert int(m2["vgpr_count"]) <= 64, m2
    assert int(m2["group_segment_fixed_size"]) * 4 <= 160 * 1024, m2      # four of k_bin2
    assert m2["private_segment_fixed_size"] == "0", m2
    for k1 in k1s:
        m1 = _kernel_meta(notes, k1)
        assert int(m1["vgpr_count"]) <= 64, m1
        assert int(m1["group_segment_fixed_size"]) * 2 <= 160 * 1024, m1  # two workgroups of k_bin1 per CU
        i1 = _disassemble(co, k1)
        # k_bin1: at most the one reload of the rare path (a lane with a run boundary among its k-mers)
        assert sum(i.startswith("scratch_load") for i in i1) <= 1, [i for i in i1 if i.startswith("scratch_")]
    # the form every filter beyond 512 MiB takes (2^32 < bits < 2^38): its unrolled loop over a whole tile inside one run is free of
    # the scalar-register reloads (v_readlane) that a run-time `form` put there -- a dozen per k-mer, with all three forms and their
    # branches in the loop: 90 instructions per k-mer instead of 46
    i1 = _disassemble(co, k1s[2])
    adds = [n for n, i in enumerate(i1) if i.startswith("ds_add_rtn_u32")]
    gaps = sorted(b - a for a, b in zip(adds, adds[1:]))
    assert len(adds) >= 16 and gaps[len(gaps) // 4] <= 55, gaps          # (the fast loop's eight k-mers: <= 55 instructions each)
    fast = [a for a, b in zip(adds, adds[1:]) if b - a <= 55]
    assert sum(i.startswith("v_readlane") for i in i1[fast[0]:fast[-1]]) <= 8      # (a constant fetched once, not a dozen per k-mer)
    i2 = _disassemble(co, k2)
    # k_bin2, whole-tile path: four global_load_dwordx4 with no wait for vector memory between them
    loads = [n for n, i in enumerate(i2) if i.startswith("global_load_dwordx4")]
    assert len(loads) == 4, loads
    assert not [i for i in i2[loads[0]:loads[-1]] if i.startswith("s_waitcnt") and "vmcnt" in i]


def test_sparse_filter_kernels_fit_their_launch_shapes(code_object):
    """The accept kernels of the sparse-filter sketch and of the literal cascade level (csrc/nts_pruned.inc, nts_bf_sparse.inc): a
    workgroup of k_hash_accept4 / k_hash_accept4r is 1024 lanes -- four waves per SIMD, which is a launch failure above 128 vector
    registers -- and keeps its accumulators in registers (they sat in scratch memory once: an array indexed by a run-time count,
    DESIGN.md 4.2); none of them may spill vector registers."""
    co, notes = code_object
    acc4r = [f"_ZN12_GLOBAL__N_115k_hash_accept4rILi8ELi{form}EEEvNS_12AcceptParamsEPKjS3_m" for form in (0, 1, 2)]
    others = ["_ZN12_GLOBAL__N_114k_hash_accept4ENS_12AcceptParamsEPKjm", "_ZN12_GLOBAL__N_113k_hash_acceptENS_12AcceptParamsE"]
    for sym in acc4r + others:
        assert sym in notes, f"{sym} not found (renamed? update this guard with it)"
        m = _kernel_meta(notes, sym)
        assert m["private_segment_fixed_size"] == "0" and m["vgpr_spill_count"] == "0", (sym, m)
        assert int(m["vgpr_count"]) <= 128, (sym, m)
    small = re.findall(r"\.name:\s+(\S*k_bf_sparse_\S*)\n", notes)                # the level's two small kernels: plain streaming shapes
    assert len(small) == 2, small
    for sym in small:
        m = _kernel_meta(notes, sym)
        assert m["private_segment_fixed_size"] == "0" and int(m["vgpr_count"]) <= 32, (sym, m)


def test_tiered_selection_keeps_five_workgroups_per_cu(code_object):
    """k_hash_tiers (csrc/nts_tiers.inc) is bound by chains of dependent trips (list -> hash -> probe) and what hides them is occupancy:
    five workgroups of 256 lanes per CU need at most 96 vector registers (512 / 5, in granules of 8) AND at most 32 KB of LDS each --
    measured against four: -4 ... -9 %.  A change that costs a register or a kilobyte would lose it silently."""
    co, notes = code_object
    for form in (0, 1, 2):
        sym = f"_ZN12_GLOBAL__N_112k_hash_tiersILi{form}EEEvNS_10TierParamsE"
        assert sym in notes, f"{sym} not found (renamed? update this guard with it)"
        m = _kernel_meta(notes, sym)
        assert int(m["vgpr_count"]) <= 96, (sym, m)
        assert int(m["group_segment_fixed_size"]) * 5 <= 160 * 1024, (sym, m)
        assert int(m.get("vgpr_spill_count", "0")) <= 2 and int(m["private_segment_fixed_size"]) <= 16, (sym, m)   # (one value lives in scratch)
