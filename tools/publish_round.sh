#!/bin/bash
# Turn one tools/collect_round.sh directory (+ the three bench lines next to it) into the tracked files under profiles/:
#   bash tools/publish_round.sh gpurun_out/r03final4 r03
# Run it on the SAME sources the collection ran on: the traffic summary is stamped with bench.kernel_signature().
set -e
src=$1; tag=$2
for w in rl xe multitask; do
  f=$(find $src/stats_$w -name "*kernel_stats.csv" | head -1)
  cp $f profiles/${tag}_${w}_kernel_stats.csv
  python3 tools/kernel_stats_md.py $f 12 "rocprofv3 --kernel-trace --stats: bench.py --workload $w --steps 10 --warmup 2 (round ${tag#r}, final build)" > profiles/${tag}_${w}_kernel_stats.md
done
python3 tools/pmc_to_json.py $src/pmc profiles/${tag}_pmc_traffic.json
python3 tools/sq_summary.py $src/sq > profiles/${tag}_sq_counters.txt
[ -s ${src}_bench_rl.json ] && tail -1 ${src}_bench_rl.json > profiles/${tag}_bench_rl.json
[ -s ${src}_bench_xe.json ] && tail -1 ${src}_bench_xe.json > profiles/${tag}_bench_xe.json
[ -s ${src}_bench_mt.json ] && tail -1 ${src}_bench_mt.json > profiles/${tag}_bench_multitask.json
python3 - <<PY
import json
for w in ("rl", "xe", "multitask"):
    d = json.load(open("profiles/${tag}_bench_%s.json" % w))
    print(w, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("traffic"))
PY
