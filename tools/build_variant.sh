#!/bin/bash
# usage: tools/build_variant.sh NAME [-DFLAG ...]  -> tools/libkh_exp_NAME.so (decoder rebuilt with flags)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
P=old-kaldi-git_amd
python -c "import importlib; importlib.import_module('old-kaldi-git_amd.build').build()" >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -D__HIP_PLATFORM_AMD__ -mllvm -amdgpu-inline-max-bb=100000 -mllvm -disable-machine-licm "$@" -c $P/csrc/kh_decoder.hip -o /tmp/kh_decoder_$name.o
objs=$(ls $P/build/*.o | grep -v kh_decoder.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libkh_exp_$name.so $objs /tmp/kh_decoder_$name.o
echo tools/libkh_exp_$name.so
