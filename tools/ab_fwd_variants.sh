#!/bin/bash
# dev: the sampler's pick / cell launches and the sampler call with every variants/lib_*.so (S2VT_LIB override), two rounds
for r in 1 2; do
for v in variants/lib_*.so; do
  echo "== $v (round $r)"
  S2VT_LIB=$PWD/$v python tools/tune_dma.py 2>&1 | grep -E "pick M=384 cfg 4:|pick M=256 cfg 6|cell M=384 cfg 18"
  S2VT_LIB=$PWD/$v python tools/quick_step.py 10 2>&1 | tail -2
done; done
