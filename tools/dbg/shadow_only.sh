#!/bin/bash
# usage: tools/dbg/shadow_only.sh [extra -D flags]   -> resource usage of the shadow kernels, ISA in /tmp/shadow_only.s
cd "$(dirname "$0")/../../ray-tracing_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function -fno-slp-vectorize \
  -S --cuda-device-only -Rpass-analysis=kernel-resource-usage "$@" -o /tmp/shadow_only.s ../../tools/dbg/shadow_only.hip 2>&1 \
  | grep -E "error|Function Name|VGPRs:|Occupancy|LDS Size|ScratchSize|SGPRs:" | sed "s/.*remark: //; s/ \[-Rpass.*//" | grep -A5 "k_shadow_\(pass\|affine\)"
