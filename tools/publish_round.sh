#!/bin/bash
# Turn one tools/collect_round.sh directory (+ the bench lines next to it: <dir>_bench_<workload>.json) into the tracked files
# under profiles/:   bash tools/publish_round.sh gpurun_out/r05prof r05
# Run it on the SAME sources the collection ran on: the traffic / SQ summaries are stamped with bench.kernel_signature().
set -e
src=$1; tag=$2
WL="rl xe multitask attention attention32 rl_msvd rl_msvd_eos rl_ref"
for w in $WL; do
  f=$(find $src/stats_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] || continue
  n=$([ $w = rl_ref ] && echo 4 || echo 12)
  cp $f profiles/${tag}_${w}_kernel_stats.csv
  python3 tools/kernel_stats_md.py $f $n "rocprofv3 --kernel-trace --stats: bench.py --workload $w (round ${tag#r}, final build)" > profiles/${tag}_${w}_kernel_stats.md
done
python3 tools/pmc_to_json.py $src/pmc profiles/${tag}_pmc_traffic.json
python3 tools/sq_summary.py $src/sq > profiles/${tag}_sq_counters.txt
python3 tools/sq_to_json.py $src/sq profiles/${tag}_sq_counters.json
for w in xe multitask attention attention32 rl_msvd rl_msvd_eos rl_ref; do
  [ -d $src/pmc_$w ] && python3 tools/pmc_to_json.py $src/pmc_$w profiles/${tag}_pmc_traffic_$w.json
  if [ -d $src/sq_$w ]; then
    python3 tools/sq_summary.py $src/sq_$w > profiles/${tag}_sq_counters_$w.txt
    python3 tools/sq_to_json.py $src/sq_$w profiles/${tag}_sq_counters_$w.json
  fi
done
for w in $WL; do
  [ -s ${src}_bench_$w.json ] && tail -1 ${src}_bench_$w.json > profiles/${tag}_bench_$w.json
done
python3 - <<PY
import json, os
for w in "$WL".split():
    f = "profiles/${tag}_bench_%s.json" % w
    if os.path.exists(f):
        d = json.load(open(f))
        print(w, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("traffic"))
PY
