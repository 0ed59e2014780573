#!/bin/bash
# second pass of tools/ab_sample_groups.sh: the two-group form with the full-size tiles forced onto the half-size launches
out=${1:-gpurun_out/groups}
mkdir -p $out
run() {
  python3 bench.py --workload rl --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'groups': os.environ.get('S2VT_SAMPLE_GROUPS','1'), 'lstm_cfg': os.environ.get('S2VT_GROUP_LSTM_CFG'), 'pick_cfg': os.environ.get('S2VT_GROUP_PICK_CFG'), 'ms_per_step': d['ms_per_step'], 'kernels': [(k['tile'], k['launches'], k['ms']) for k in d['roofline']['all_kernels_warmup'] if k['class'] in (1, 2)]}))" >> $out/ab2.jsonl
}
run
for combo in "11 4" "11 -1" "-1 4" "11 0" "8 4" "6 4"; do
  set -- $combo
  S2VT_SAMPLE_GROUPS=2 S2VT_GROUP_LSTM_CFG=$1 S2VT_GROUP_PICK_CFG=$2 run
done
run
cat $out/ab2.jsonl
