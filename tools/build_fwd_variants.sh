#!/bin/bash
# dev: A/B variants that differ in fwd.hip only (gemm_mfma.h instantiations): compile fwd.hip with the flags, link with the product objects
# usage: build_fwd_variants.sh name1:"-DFLAG=.." name2:"..." ...   (run `make` in csrc first)
set -e
cd "$(dirname "$0")/../multitask-end-to-end-video-captioning_amd/csrc"
mkdir -p ../../variants
rm -f ../../variants/*.so
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( d=/tmp/varf_$name; mkdir -p $d
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 $flags -c fwd.hip -o $d/fwd.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib_$name.so $d/fwd.o $(ls *.o | grep -v '^fwd.o$') ) &
done
wait
ls -la ../../variants
