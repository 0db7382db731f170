"""ISA lint of the one kernel whose MFMAs are inline asm (csrc/gemm_w4m.hip).  hipcc does not know that those asm statements are
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


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc to cross-compile")
def test_no_valu_write_right_in_front_of_an_asm_mfma(tmp_path):
    out = tmp_path / "gemm_w4m.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-I", CSRC, "-Wno-unused-function",
                    "-fno-fast-math", "-S", "--cuda-device-only", os.path.join(CSRC, "gemm_w4m.hip"), "-o", str(out)],
                   check=True, capture_output=True)
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
        else:
            window.append((op, None))
        window = window[-4:]
    assert n_mfma > 5000, n_mfma              # ten instantiations x several unrolled stage bodies
    assert not bad, bad[:5]
