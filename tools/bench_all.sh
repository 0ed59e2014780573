#!/bin/bash
# The bench line of every workload (after tools/publish_round.sh, so that roofline.traffic / mfma_busy_pct find this build's stamped summaries):
#   bash tools/bench_all.sh gpurun_out/r06lines   ->  <dir>/bench_<workload>.json
out=${1:-gpurun_out/lines}; mkdir -p $out
for w in rl xe multitask attention attention32 rl_msvd rl_msvd_eos rl_ref e2e e2e_xe; do
  extra=$([ $w = rl ] && echo "" || echo "--no-cpu-baseline")
  st=$(case $w in rl_ref) echo "--steps 20 --warmup 3";; e2e|e2e_xe) echo "--steps 10 --warmup 3";; *) echo "";; esac)
  python3 bench.py --workload $w $st $extra 2>/dev/null | tail -1 > $out/bench_$w.json
done
python3 - <<PY
import json
for w in "rl xe multitask attention attention32 rl_msvd rl_msvd_eos rl_ref e2e e2e_xe".split():
    d = json.load(open("$out/bench_%s.json" % w)); r = d["roofline"]
    print(w, d["ms_per_step"], d["value"], r["frac"], r.get("executed_flops_frac"), r.get("traffic"), r.get("mfma_busy_pct"), r["kernel"][:40])
PY
