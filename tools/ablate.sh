#!/bin/bash
# timing-only ablations of the contraction kernel (outputs are wrong with S2VT_DBG != 0); needs variants/lib_ablate.so
for d in 0 1 2 4 8 16 17 6 7 23 31; do
  echo "== S2VT_DBG=$d"; S2VT_LIB=$PWD/variants/lib_ablate.so S2VT_DBG=$d python tools/tune_tiles.py 2>&1 | grep -E "M=64 LSTM1|M=384 LSTM2|PICK M" | cut -c1-210
done
