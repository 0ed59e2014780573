#!/bin/bash
# dev: A/B builds that differ in decode_loop.hip only (variants/lib_<name>.so):  bash tools/build_decloop_variants.sh stamp:"-DS2VT_DL_STAMP"
# (experimental objects: run `make -C multitask-end-to-end-video-captioning_amd/csrc exp` first -- api.o is api.hip built with -DS2VT_EXPERIMENTAL)
set -e
cd "$(dirname "$0")/../multitask-end-to-end-video-captioning_amd/csrc"
mkdir -p ../../variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  mkdir -p /tmp/lvar_$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 $flags -c decode_loop.hip -o /tmp/lvar_$name/decode_loop.o &
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib_$name.so fwd.o aux.o api.o train.o attn.o attn_model.o attn_chain.o attn_chain_bwd.o session.o chain.o chain_bwd.o decode4.o /tmp/lvar_$name/decode_loop.o
done
ls ../../variants
