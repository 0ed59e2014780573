#!/bin/bash
# dev: A/B builds that differ in ONE source only (variants/lib_<name>.so; the other objects come from the main build)
# usage: [SRC=chain_bwd] build_chain_variants.sh name1:"-DFLAG.." name2:"..."      (SRC defaults to chain)
set -e
SRC=${SRC:-chain}
cd "$(dirname "$0")/../multitask-end-to-end-video-captioning_amd/csrc"
mkdir -p ../../variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  mkdir -p /tmp/cvar_$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 $flags -c $SRC.hip -o /tmp/cvar_$name/$SRC.o &
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib_$name.so $(ls *.o | grep -v "^$SRC.o\$" | grep -v "^api_exp.o\$\|^decode4.o\$\|^decode_loop.o\$") /tmp/cvar_$name/$SRC.o
done
ls -la ../../variants
