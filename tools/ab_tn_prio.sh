#!/bin/bash
# dev: rotating issue priority in the weight-gradient kernel (S2VT_TN_PRIO = chunks per priority phase, 0 = off) -- step time
# and fetch traffic per launch
export TMPDIR=/tmp
for p in ${@:-0 1}; do
  for i in 1 2; do
    S2VT_TN_PRIO=$p python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('prio $p', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
  done
  export S2VT_TN_PRIO=$p
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/tnprio$p/FETCH_SIZE -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/tnprio$p.log 2>&1
done
