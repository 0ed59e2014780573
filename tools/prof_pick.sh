#!/bin/bash
# dev: true kernel durations (rocprofv3 --kernel-trace --stats) of the pick tiles at small M.  usage: prof_pick.sh "16,64" [ncfg]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
MS=${1:-64} NCFG=${2:-8} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o pk -- python3 $R/tools/tune_pick_m.py > /tmp/pk.log 2>&1
tail -3 /tmp/pk.log
python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/pk/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(r["Name"][:100], r["Calls"], r["AverageNs"])
PY
