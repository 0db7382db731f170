#!/bin/bash
# Cross-compile one csrc/*.hip file to gfx950 assembly (no GPU needed) and print resource usage; the .s lands in /tmp.
f=${1:-gemm_w4.hip}
cd /root/repo/open_flamingo_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -I . -Wno-unused-function -fno-fast-math -S --cuda-device-only $f -o /tmp/${f%.hip}.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs:|AGPRs:|Scratch|error|Occupancy" | sort | uniq -c
