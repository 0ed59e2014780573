#!/bin/bash
# dev: time the bench shapes + one REINFORCE step with every variants/lib_*.so (S2VT_LIB override)
for r in 1 2; do
for v in variants/lib_*.so; do
  echo "== $v (round $r)"; S2VT_LIB=$PWD/$v python tools/tune_tiles.py 2>&1 | grep -E "M=64 LSTM|M=384 LSTM2|M=320 LSTM|PICK M|STORE logits|STORE dO2|dh slab" | cut -c1-150
  S2VT_LIB=$PWD/$v python tools/quick_step.py 10 2>&1 | tail -1
done; done
