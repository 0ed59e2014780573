#!/bin/bash
# dev: build A/B variants of the library side by side (variants/*.so, shipped to the GPU box by gpurun)
# usage: build_variants.sh name1:"-DFLAG=.. -DFLAG2=.." name2:"..." ...
set -e
cd "$(dirname "$0")/../multitask-end-to-end-video-captioning_amd/csrc"
mkdir -p ../../variants
rm -f ../../variants/*.so
build() { # name flags...
  local name=$1; shift
  local d=/tmp/var_$name; mkdir -p $d
  for f in fwd aux api train attn attn_model attn_chain attn_chain_bwd session chain chain_bwd; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 "$@" -c $f.hip -o $d/$f.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib_$name.so $d/*.o
}
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  build $name $flags
done
ls -la ../../variants
