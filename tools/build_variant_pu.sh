#!/bin/bash
# usage: tools/build_variant_pu.sh NAME PU [-DFLAG ...]
set -e
cd "$(dirname "$0")/.."
name=$1; pu=$2; shift; shift
P=old-kaldi-git_amd
sed "s/^constexpr int PU = [0-9]*;/constexpr int PU = $pu;/" $P/csrc/kh_decoder.hip > /tmp/kh_decoder_$name.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -D__HIP_PLATFORM_AMD__ -mllvm -amdgpu-inline-max-bb=100000 -I$P/csrc -Iinclude "$@" -c -x hip /tmp/kh_decoder_$name.hip -o /tmp/kh_decoder_$name.o
objs=$(ls $P/build/*.o | grep -v kh_decoder.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libkh_exp_$name.so $objs /tmp/kh_decoder_$name.o
echo tools/libkh_exp_$name.so
