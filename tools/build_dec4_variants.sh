#!/bin/bash
# dev: A/B builds that differ in decode4.hip only (variants/lib_<name>.so)
# (experimental objects: run `make -C multitask-end-to-end-video-captioning_amd/csrc exp` first -- api.o is api.hip built with -DS2VT_EXPERIMENTAL)
set -e
cd "$(dirname "$0")/../multitask-end-to-end-video-captioning_amd/csrc"
mkdir -p ../../variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  mkdir -p /tmp/dvar_$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 $flags -c decode4.hip -o /tmp/dvar_$name/decode4.o &
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib_$name.so fwd.o aux.o api.o train.o attn.o session.o chain.o chain_bwd.o /tmp/dvar_$name/decode4.o
done
ls ../../variants
