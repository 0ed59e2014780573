#!/bin/bash
# Turn one tools/collect_round.sh directory (+ the bench lines next to it: <dir>_bench_<workload>.json) into the tracked files
# under profiles/:   bash tools/publish_round.sh gpurun_out/r04final r04
# Run it on the SAME sources the collection ran on: the traffic summaries are stamped with bench.kernel_signature().
set -e
src=$1; tag=$2
for w in rl xe multitask attention attention32 rl_msvd rl_msvd_eos; do
  f=$(find $src/stats_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] || continue
  cp $f profiles/${tag}_${w}_kernel_stats.csv
  python3 tools/kernel_stats_md.py $f 12 "rocprofv3 --kernel-trace --stats: bench.py --workload $w --steps 10 --warmup 2 (round ${tag#r}, final build)" > profiles/${tag}_${w}_kernel_stats.md
done
python3 tools/pmc_to_json.py $src/pmc profiles/${tag}_pmc_traffic.json
python3 tools/sq_summary.py $src/sq > profiles/${tag}_sq_counters.txt
for w in xe multitask attention attention32 rl_msvd rl_msvd_eos; do
  [ -d $src/pmc_$w ] && python3 tools/pmc_to_json.py $src/pmc_$w profiles/${tag}_pmc_traffic_$w.json
  [ -d $src/sq_$w ] && python3 tools/sq_summary.py $src/sq_$w > profiles/${tag}_sq_counters_$w.txt
done
for w in rl xe multitask attention attention32 rl_msvd rl_msvd_eos; do
  [ -s ${src}_bench_$w.json ] && tail -1 ${src}_bench_$w.json > profiles/${tag}_bench_$w.json
done
python3 - <<PY
import json, os
for w in ("rl", "xe", "multitask", "attention", "attention32", "rl_msvd", "rl_msvd_eos"):
    f = "profiles/${tag}_bench_%s.json" % w
    if os.path.exists(f):
        d = json.load(open(f))
        print(w, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("traffic"))
PY
