"""ISA lint of the kernels whose MFMAs are inline asm (csrc/gemm_w4m.hip, gemm_w4h.hip, gemm_w4s.hip, attn_bwd_res.hip).  hipcc does not know that those asm statements are
MFMAs, so its hazard recognizer does not separate a VALU write of a register from an MFMA that reads it (found on hardware in round 3:
of_platform.h, of_mfma_acc_guard).  This test cross-compiles the file (no GPU needed) and fails if any v_mfma reads a VGPR / AGPR that a
VALU instruction wrote within the four instructions in front of it without an s_nop in between."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "open_flamingo_amd", "csrc")


def _regs(tok):
    """'v[50:53]' -> ('v', {50..53}); 'a7' -> ('a', {7}); anything else -> None"""
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return m.group(1), {int(m.group(2))}
    return None


ASM_MFMA_SOURCES = [("gemm_w4m.hip", 10000), ("gemm_w4h.hip", 3000), ("gemm_w4s.hip", 500), ("attn_bwd_res.hip", 400)]      # (file, at least this many v_mfma in its ISA)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc to cross-compile")
@pytest.mark.parametrize("src,min_mfma", ASM_MFMA_SOURCES)
def test_no_valu_write_right_in_front_of_an_asm_mfma(tmp_path, src, min_mfma):
    out = tmp_path / (src + ".s")
    _compile_to_asm(src, out)
    window = []          # the last four instructions: (mnemonic, written register set or None)
    n_mfma, bad = 0, []
    for ln, raw in enumerate(open(out), 1):
        line = raw.split(";")[0].strip()
        if not line or line.startswith(".") or line.endswith(":"):
            continue          # labels do not reset the window: the copies that bit sat right in front of a loop label (fall-through)
        parts = line.replace(",", " ").split()
        op, args = parts[0], parts[1:]
        if op.startswith("v_mfma"):
            n_mfma += 1
            reads = [r for r in (_regs(a) for a in args[1:]) if r]
            for wop, wr in window:
                if wr is None:
                    continue
                for kind, regs in reads:
                    if wr[0] == kind and wr[1] & regs:
                        bad.append((ln, wop, line))
            window.append((op, None))           # an MFMA's own result is ordered by the MFMA pipeline (different accumulators back to back)
        elif op.startswith("s_nop"):
            window = []
        elif op.startswith("v_") and args:
            window.append((op, _regs(args[0])))
        elif op.startswith(("ds_read", "global_load", "scratch_load")) and args and _regs(args[0]):
            # a load result replaces what a VALU wrote there (e.g. the load's own address register reused as its destination): the MFMA
            # then reads the LOAD's data, ordered by s_waitcnt -- the earlier VALU write is dead, not a hazard
            kind, regs = _regs(args[0])
            window = [(wop, (wr[0], wr[1] - regs) if wr is not None and wr[0] == kind else wr) for wop, wr in window]
            window.append((op, None))
        else:
            window.append((op, None))
        window = window[-4:]
    assert n_mfma > min_mfma, n_mfma          # gemm_w4m.hip: twenty instantiations (classic + stream-K) x several unrolled stage bodies
    assert not bad, bad[:5]


_ASM_CACHE = {}       # source file -> path of its cross-compiled ISA (gemm_w4m.hip takes ~20 s: three tests read it)


def _compile_to_asm(src, out):
    import shutil
    import tempfile
    if src not in _ASM_CACHE:
        cached = os.path.join(tempfile.mkdtemp(prefix="of_isa_"), src + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-I", CSRC, "-Wno-unused-function",
                        "-fno-fast-math", "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", cached], check=True, capture_output=True)
        _ASM_CACHE[src] = cached
    shutil.copyfile(_ASM_CACHE[src], str(out))


def _instructions(path):
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].strip()
        if not line or line.startswith(".") or line.endswith(":"):
            continue
        parts = line.replace(",", " ").split()
        yield ln, line, parts[0], parts[1:]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc to cross-compile")
@pytest.mark.parametrize("src", [s for s, _ in ASM_MFMA_SOURCES if s != "attn_bwd_res.hip"])      # that kernel: the bank test at the end of this file
def test_accumulators_of_asm_mfmas_are_read_only_after_the_settle_wait(tmp_path, src):
    """The other direction of the same blind spot: of_mfma_acc_settle() is an asm statement with no operand tie to the
    accumulators, so nothing formally stops hipcc from moving a read of an accumulation register (v_accvgpr_read / an `a` source
    operand of a store) above its s_nops.  An 8-pass MFMA (16x16x32 bf16) needs 11 wait states between its issue and a VALU /
    memory read of its result: count them in the cross-compiled ISA (s_nop n = n + 1 wait states, any other instruction 1)."""
    out = tmp_path / (src + ".s")
    _compile_to_asm(src, out)
    # Per basic block, conservatively: a label that follows an MFMA of the same kernel in the listing counts as "an MFMA was just
    # issued" (the block may be entered from the K loop).  Registers (re)written by v_accvgpr_write / v_accvgpr_mov inside the
    # block are exempt -- the accumulator zero-fill blocks read those right away.
    since, bad, n_reads, seen, clean = None, [], 0, False, set()
    for ln, raw in enumerate(open(out), 1):
        line = raw.split(";")[0].strip()
        if not line or line.startswith("."):
            if line.startswith(".L") and line.endswith(":"):
                since, clean = (0 if seen else None), set()
            continue
        if line.endswith(":"):
            seen, since, clean = False, None, set()          # a new kernel
            continue
        parts = line.replace(",", " ").split()
        op, args = parts[0], parts[1:]
        if op.startswith("v_mfma"):
            seen, since = True, 0
            r = _regs(args[0])
            if r:
                clean -= r[1]
            continue
        srcs = args if op.startswith(("global_store", "buffer_store", "ds_write")) else args[1:]
        acc_reads = set()
        for a in srcs:
            r = _regs(a)
            if r and r[0] == "a":
                acc_reads |= r[1]
        if acc_reads and since is not None:
            n_reads += 1
            # (v_accvgpr_mov is what the zero-fill blocks are made of -- a zero written in an earlier block copied around, placed
            # behind the K loop in the listing; a copy of a live MFMA result does not occur and would surface at its consumer)
            if since < 11 and not acc_reads <= clean and not op.startswith("v_accvgpr_mov"):
                bad.append((ln, since, line))
        if op.startswith(("v_accvgpr_write", "v_accvgpr_mov")):
            r = _regs(args[0])
            if r and (not acc_reads or acc_reads <= clean or since is None or since >= 11):
                clean |= r[1]
        if since is not None:
            since += (int(args[0]) + 1) if op == "s_nop" else 1
            if since >= 11:
                seen = False          # every MFMA in front of this point has retired: later blocks start settled
    assert n_reads > 100, n_reads
    assert not bad, bad[:5]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc to cross-compile")
@pytest.mark.parametrize("src", ["gemm_w4m.hip", "gemm_w4h.hip", "gemm_w4s.hip", "gemm_mid.hip", "gemm_pp.hip", "attention.hip"])
def test_m0_is_only_ever_the_lds_dma_destination(tmp_path, src):
    """The inline-asm LDS-DMA (of_platform.h) writes M0 without declaring it: hipcc reserves M0 and refuses it in a clobber list
    ("inline asm clobber list contains reserved registers").  That is sound as long as the compiler itself never keeps a value in M0
    across those statements -- on gfx950 its only M0 uses are the LDS-DMA builtins (which re-write M0 right in front of every use),
    s_movrel / GWS / sendmsg (none in these kernels).  Pinned here: in the cross-compiled ISA every mention of m0 is a scalar write
    of it (`s_mov_b32 m0, ...`, the builtins also `s_add_i32 m0, ...`) and every one of them is followed -- before the next M0 write -- by an `... lds` DMA load."""
    out = tmp_path / (src + ".s")
    _compile_to_asm(src, out)
    pending, n = None, 0
    for ln, line, op, args in _instructions(out):
        if "m0" in args:
            assert op.startswith("s_") and args[0] == "m0" and "m0" not in args[1:], (ln, line)      # written (s_mov / s_add ...), never read
            assert pending is None, ("M0 written twice without a DMA in between", pending, ln)
            pending = ln
            n += 1
        elif op.startswith(("buffer_load", "global_load")) and ("_lds_" in op or (args and args[-1] == "lds")):
            pending = None
        elif op == "s_endpgm":
            assert pending is None, (pending, "M0 written, never used")
    assert n >= 4, n


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc to cross-compile")
def test_nothing_of_the_compilers_lives_in_the_fixed_accumulator_bank(tmp_path):
    """csrc/attn_bwd_res.hip keeps 64 accumulator tiles in FIXED accumulation registers (of_accbank64.h) that hipcc does not know to be
    live.  Found on hardware (round 6): under register pressure hipcc parked VGPR values in a0..a7 between two statements of the bank
    (v_accvgpr_write_b32 aN, vM ... v_accvgpr_read) and chose accumulation registers as destinations of BUILTIN MFMAs -- dK / dV wrong,
    the emulator green.  The kernel therefore issues every MFMA by inline asm and fences the bank (of_accbank64_fence); this test holds
    the cross-compiled ISA to it: accumulation registers are written only by `v_accvgpr_write aN, 0` and by MFMAs that accumulate in
    place, never copied, and MFMA results in VGPRs are not read before 11 wait states have passed."""
    out = tmp_path / "attn_bwd_res.s"
    _compile_to_asm("attn_bwd_res.hip", out)
    bad, n_bank, n_acc_reads, pending, since_bank = [], 0, 0, {}, 99          # pending: VGPR -> wait states since the MFMA that writes it
    for ln, line, op, args in _instructions(out):
        if op.startswith("v_accvgpr_read"):
            n_acc_reads += 1
            if since_bank < 11:
                bad.append((ln, "accumulator read %d wait states behind a bank MFMA" % since_bank, line))
        if op.startswith("v_accvgpr_mov") or (op.startswith("v_accvgpr_write") and args[1] != "0"):
            bad.append((ln, line))
        reads = set()
        if op.startswith("v_mfma"):
            d, c = _regs(args[0]), _regs(args[3])
            if d and d[0] == "a":
                n_bank += 1
                if not (c and c == d):
                    bad.append((ln, line))          # an accumulation-register destination that is not an in-place accumulate of the bank
            for a in args[1:3]:
                r = _regs(a)
                if r and r[0] == "v":
                    reads |= r[1]
        elif op.startswith(("v_", "ds_write", "global_store", "scratch_store", "buffer_store")):
            for a in (args if not op.startswith("v_") else args[1:]):
                r = _regs(a)
                if r and r[0] == "v":
                    reads |= r[1]
        early = [r for r in reads if pending.get(r, 99) < 11]
        if early:
            bad.append((ln, "read of an MFMA result after %d wait states" % min(pending[r] for r in early), line))
        step = (int(args[0]) + 1) if op == "s_nop" else 1
        pending = {r: w + step for r, w in pending.items() if w + step < 11}
        since_bank = min(99, since_bank + step)
        if op.startswith("v_mfma"):
            d = _regs(args[0])
            if d and d[0] == "v":
                for r in d[1]:
                    pending[r] = 0
            elif d:
                since_bank = 0
    assert n_bank >= 240 and n_acc_reads >= 384, (n_bank, n_acc_reads)          # head 128: 16 (1 + 2 + 3 + 4) MFMAs, 256 reads; head 64: half
    assert not bad, bad[:5]
