#!/bin/bash
# dev: build A/B variants of the library side by side (variants/*.so, shipped to the GPU box by gpurun)
set -e
cd "$(dirname "$0")/../multitask-end-to-end-video-captioning_amd/csrc"
mkdir -p ../../variants
build() { # name flags...
  local name=$1; shift
  local d=/tmp/var_$name; mkdir -p $d
  for f in fwd aux api train attn; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 "$@" -c $f.hip -o $d/$f.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib_$name.so $d/*.o
}
build pf24_c0 -DS2VT_PF_BUDGET=32 -DS2VT_ALLREADS=0 &
build pf48_c0 -DS2VT_PF_BUDGET=48 -DS2VT_ALLREADS=0 &
wait
build pf64_c0 -DS2VT_PF_BUDGET=64 -DS2VT_ALLREADS=0 &
build pf48_c1 -DS2VT_PF_BUDGET=48 -DS2VT_ALLREADS=1 &
wait
ls -la ../../variants
