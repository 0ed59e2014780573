#!/bin/bash
# dev: A/B builds that differ in chain.hip only (variants/lib_<name>.so; the other objects come from the main build)
# usage: build_chain_variants.sh name1:"-DFLAG.." name2:"..."
set -e
cd "$(dirname "$0")/../multitask-end-to-end-video-captioning_amd/csrc"
mkdir -p ../../variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  mkdir -p /tmp/cvar_$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 $flags -c chain.hip -o /tmp/cvar_$name/chain.o &
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib_$name.so $(ls *.o | grep -v '^chain.o$') /tmp/cvar_$name/chain.o
done
ls -la ../../variants
