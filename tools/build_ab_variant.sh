#!/bin/bash
# Build a same-source A/B variant of libofhip into tools/ab/<name>.so with extra -D flags (profiling tools only).
#   tools/build_ab_variant.sh libofhip_builtin_dma.so -DOF_DMA_VIA_BUILTIN
name=$1; shift
cd /root/repo/open_flamingo_amd/csrc; mkdir -p /tmp/ab_$name /root/repo/tools/ab
for f in gemm.hip gemm_pp.hip gemm_w4.hip gemm_w4m.hip gemm_w4h.hip gemm_w4s.hip gemm_mid.hip gemm_skinny.hip layernorm.hip attention.hip elementwise.hip optim.hip loss.hip api.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -I . -Wno-unused-function -fno-fast-math "$@" -c $f -o /tmp/ab_$name/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/ab_$name/*.o -o /root/repo/tools/ab/$name && ls -la /root/repo/tools/ab/$name
