#!/bin/bash
# build_variant.sh <name> [-DFLAG=..]...   -> build_ab/lib_<name>.so (A/B measurements: MCCONV_LIB selects it)
set -e
cd "$(dirname "$0")/.."
mkdir -p build_ab
name=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wno-unused-function "$@" -o build_ab/lib_$name.so cuda_audio_amd/csrc/mcconv.hip
echo build_ab/lib_$name.so
