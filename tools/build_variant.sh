#!/bin/bash
# tools/build_variant.sh <name> <file.hip> <-Dflags...>: ab/lib_<name>.so = the library with ONE source recompiled with extra flags
# (A/B runs: HOISDF_LIB=ab/lib_<name>.so python tools/mb_....py).  ab/ is git-ignored and travels to the GPU box.
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p ab/obj
obj=ab/obj/${name}_$(basename $src .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wall -Wno-unused-function "$@" -c hoisdf_amd/csrc/$src -o $obj
others=$(ls hoisdf_amd/csrc/build/*.o | grep -v "/$(basename $src .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib_${name}.so $obj $others
echo built ab/lib_${name}.so
