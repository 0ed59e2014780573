#!/bin/bash
# dev: kernels of a bench workload's step that are not this library's (tensor-library fills / reductions / copies), per step.
# usage (GPU box): bash tools/framework_kernels.sh multitask
R=${GRAFT_REPO_ROOT:-/root/repo}
w=${1:-multitask}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fk -o run -- python3 $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > /tmp/fk.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/fk/**/*kernel_stats.csv", recursive=True)[0]
tot = 0
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "s2vt" in n or "anonymous namespace)::" in n and "at::native" not in n: continue
    per = int(r["Calls"]) / 12; us = float(r["TotalDurationNs"]) / 12 / 1e3
    tot += us
    if per >= 0.5: print(f"{per:5.1f}/step {us:7.1f} us/step  {n[:120]}")
print("total us/step", round(tot, 1))
PY
