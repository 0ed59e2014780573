#!/bin/bash
# dev: A/B builds that differ in attn_chain.hip / attn_chain_bwd.hip only (variants/lib_<name>.so; the other objects come from the main build)
# usage: build_achain_variants.sh name1:"-DFLAG.." name2:"..."
set -e
cd "$(dirname "$0")/../multitask-end-to-end-video-captioning_amd/csrc"
mkdir -p ../../variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  mkdir -p /tmp/acvar_$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 $flags -c attn_chain.hip -o /tmp/acvar_$name/attn_chain.o &
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 $flags -c attn_chain_bwd.hip -o /tmp/acvar_$name/attn_chain_bwd.o &
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib_$name.so fwd.o aux.o api.o train.o attn.o attn_model.o session.o chain.o chain_bwd.o \
     /tmp/acvar_$name/attn_chain.o /tmp/acvar_$name/attn_chain_bwd.o
done
ls -la ../../variants
