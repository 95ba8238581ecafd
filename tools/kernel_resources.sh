#!/bin/bash
# usage: tools/kernel_resources.sh [-DFLAG ...]   compile-only check of the decoder kernels' registers / scratch
# (-Rpass-analysis=kernel-resource-usage; no GPU needed).  ScratchSize of DecodeKernel<1,0> is what to watch.
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -D__HIP_PLATFORM_AMD__ \
  -mllvm -amdgpu-inline-max-bb=100000 -mllvm -disable-machine-licm -Rpass-analysis=kernel-resource-usage "$@" -c old-kaldi-git_amd/csrc/kh_decoder.hip -o /tmp/kh_dec_chk.o 2>&1 |
  awk '/Function Name/ {n=$0; sub(/.*Name: /,"",n); sub(/ \[.*/,"",n)} /VGPRs:/ {v=$0; sub(/.*VGPRs: /,"",v); sub(/ \[.*/,"",v)} /ScratchSize/ {s=$0; sub(/.*: /,"",s); sub(/ \[.*/,"",s); } /Occupancy/ {print substr(n,1,60), "VGPRs", v, "Scratch", s}'
