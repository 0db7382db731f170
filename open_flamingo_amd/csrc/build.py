"""Build libofhip.so (gfx950) in-tree with hipcc.  Also used by __graft_entry__.build().

    python -m open_flamingo_amd.csrc.build            # device library  -> open_flamingo_amd/csrc/libofhip.so
    python -m open_flamingo_amd.csrc.build --emu      # host emulator   -> tests/emu/libofhip_emu.so (tests only)
    python -m open_flamingo_amd.csrc.build --tools    # -DOF_TOOLS_BUILD -> tools/libofhip_tools.so (profiling tools only:
                                                      #    timing ablations / A-B variants of the GEMM kernels, some of
                                                      #    them wrong by design; never loaded by the package)

hipcc cross-compiles gfx950 without a GPU.  One translation unit per .hip file, compiled in parallel,
objects cached by source mtime.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["gemm.hip", "gemm_pp.hip", "gemm_w4.hip", "gemm_w4m.hip", "gemm_w4h.hip", "gemm_w4s.hip", "gemm_mid.hip", "gemm_skinny.hip", "layernorm.hip", "attention.hip", "attn_bwd_res.hip", "xattn_fused.hip", "elementwise.hip", "optim.hip", "loss.hip", "api.hip"]
HEADERS = ["of_platform.h", "gemm_common.h", "gemm_tile256.h", "gemm_w4_epi.h", "attn_core.h", os.path.join(ROOT, "include", "of_hip.h")]
# kernels of_gemm never selects (measured, kept with their tests and records): tools / emulator builds only, not in the product library
TOOLS_ONLY_SOURCES = ["gemm_w4.hip", "gemm_w4s.hip"]
LIB = os.path.join(HERE, "libofhip.so")
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libofhip_emu.so")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return r.stdout + r.stderr


TOOLS_LIB = os.path.join(ROOT, "tools", "libofhip_tools.so")


def build(emu=False, verbose=False, force=False, tools=False):
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s)) and (emu or tools or s not in TOOLS_ONLY_SOURCES)]
    hdrs = [h if os.path.isabs(h) else os.path.join(HERE, h) for h in HEADERS]
    if emu:
        hdrs.append(os.path.join(EMU_DIR, "of_emu.h"))
        objdir = os.path.join(EMU_DIR, "build")
        cc = ["/opt/rocm/lib/llvm/bin/clang++", "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-DOF_HOST_EMU",
              "-I", EMU_DIR, "-I", HERE, "-Wno-unused-function", "-Wno-unknown-attributes", "-pthread"]
        lib = EMU_LIB
        link = ["/opt/rocm/lib/llvm/bin/clang++", "-shared", "-pthread"]
    else:
        objdir = os.path.join(HERE, "build_tools" if tools else "build")
        cc = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-I", HERE,
              "-Wno-unused-function", "-fno-fast-math"]   # erf / tanh / division semantics of the epilogues stay IEEE
        if tools:
            cc.append("-DOF_TOOLS_BUILD")
        lib = TOOLS_LIB if tools else LIB
        link = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"]
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append(cc + ["-c", s, "-o", o])
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        for out in ex.map(_run, jobs):
            if verbose and out.strip():
                print(out)
    if jobs or not os.path.exists(lib):
        _run(link + objs + ["-o", lib])
    return lib


if __name__ == "__main__":
    print(build(emu="--emu" in sys.argv, verbose=True, force="--force" in sys.argv, tools="--tools" in sys.argv))
