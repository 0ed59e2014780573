#!/bin/bash
# Gated overlap of the persistent backward recurrences with independent weight-gradient contractions (csrc/train.hip, S2VT_OVERLAP=0|1|2):
# step time per workload and setting, two interleaved repetitions.   bash tools/ab_overlap.sh gpurun_out/overlap "xe multitask rl attention"
out=${1:-gpurun_out/overlap}
wls=${2:-"xe multitask rl"}
mkdir -p $out
for rep in 1 2; do
  for w in $wls; do
    for m in 0 2 1; do
      S2VT_OVERLAP=$m python3 bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'workload': '$w', 'overlap': $m, 'rep': $rep, 'ms_per_step': d['ms_per_step'], 'median': d['step_ms']['median'], 'timeouts': d['config']['persistent_recurrence_timeouts'], 'loss': d['config']['loss']}))" >> $out/ab.jsonl
    done
  done
done
cat $out/ab.jsonl
