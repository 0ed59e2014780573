#!/bin/bash
# timing-only ablations of the contraction kernel (outputs are wrong with S2VT_DBG != 0)
for d in 0 1 2 4 8 3 5 6 7 15; do
  echo "== S2VT_DBG=$d"; S2VT_DBG=$d python tools/tune_tiles.py 2>&1 | grep -E "M=64 LSTM2|M=384 LSTM2|STORE logits" | cut -c1-60
done
