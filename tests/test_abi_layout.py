"""The three statements of the C ABI's argument structs must agree: include/of_hip.h (compiled here with the host C compiler),
open_flamingo_amd/hip/abi.py (the ctypes mirror the package binds with) and the snippet a maintainer would copy from INTEGRATION.md.
Round 4's review found the snippet two fields short of the header (it read cu_limit / sk_grid from whatever followed the struct)."""
import ctypes
import os
import re
import shutil
import subprocess

import pytest

from open_flamingo_amd.hip import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "of_hip.h")


def _c_layout(tmp_path, struct, fields):
    """sizeof and every offsetof of `struct`, printed by a C program compiled against the header"""
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no host C compiler")
    src = tmp_path / "layout.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void) {',
             f'  printf("sizeof %zu\\n", sizeof({struct}));']
    lines += [f'  printf("{f} %zu\\n", offsetof({struct}, {f}));' for f in fields]
    lines += ['  return 0;', '}']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([cc, "-std=c11", "-o", str(exe), str(src)], check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    return dict(zip(out[0::2], (int(v) for v in out[1::2])))


@pytest.mark.parametrize("struct,mirror", [("OfGemmArgs", abi.OfGemmArgs), ("OfAttnArgs", abi.OfAttnArgs),
                                           ("OfXattnFusedArgs", abi.OfXattnFusedArgs), ("OfPackDesc", abi.OfPackDesc)])
def test_ctypes_mirror_has_the_layout_of_the_header(tmp_path, struct, mirror):
    names = [f[0] for f in mirror._fields_]
    c = _c_layout(tmp_path, struct, names)          # a field the header does not have fails to compile
    assert c["sizeof"] == ctypes.sizeof(mirror), (c["sizeof"], ctypes.sizeof(mirror))
    for n in names:
        assert c[n] == getattr(mirror, n).offset, (n, c[n], getattr(mirror, n).offset)
    # ... and the header has no field the mirror lacks: the last mirrored field ends where the struct ends (up to tail padding)
    last = names[-1]
    assert c["sizeof"] - (c[last] + getattr(mirror, last).size) < 8


def test_integration_md_snippet_is_the_mirror():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"class OfGemmArgs\(ctypes\.Structure\):.*?_fields_ = \[(.*?)\]\s*(?:#.*)?\nlib\.of_gemm", text, re.S)
    assert m, "INTEGRATION.md no longer shows the OfGemmArgs binding"
    doc = re.findall(r'\("(\w+)",\s*ctypes\.(\w+)\)', m.group(1))
    want = [(n, t.__name__) for n, t in abi.OfGemmArgs._fields_]
    assert [n for n, _ in doc] == [n for n, _ in want]
    for (n, dt), (_, wt) in zip(doc, want):
        assert ctypes.sizeof(getattr(ctypes, dt)) == ctypes.sizeof(getattr(ctypes, wt)), (n, dt, wt)
    ver = re.search(r"of_abi_version\(\) == (\d+)", text)
    hdr = re.search(r"#define\s+OF_ABI_VERSION\s+(\d+)", open(HEADER).read())
    assert ver and hdr and ver.group(1) == hdr.group(1)
