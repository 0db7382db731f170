#!/bin/bash
# Build a same-source A/B variant of libofhip into tools/ab/<name>.so with extra -D flags (profiling tools only).
#   tools/build_ab_variant.sh libofhip_builtin_dma.so -DOF_DMA_VIA_BUILTIN
name=$1; shift
cd /root/repo/open_flamingo_amd/csrc; mkdir -p /tmp/ab_$name /root/repo/tools/ab
# the product library's translation units (open_flamingo_amd/csrc/build.py: SOURCES minus the tools-only kernels)
files=$(PYTHONPATH=/root/repo python -c "from open_flamingo_amd.csrc import build as b; print(' '.join(s for s in b.SOURCES if s not in b.TOOLS_ONLY_SOURCES))")
for f in $files; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -I . -Wno-unused-function -fno-fast-math "$@" -c $f -o /tmp/ab_$name/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/ab_$name/*.o -o /root/repo/tools/ab/$name && ls -la /root/repo/tools/ab/$name
