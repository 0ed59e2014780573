#!/bin/bash
# fetch traffic (FETCH_SIZE x 2) per launch of the many-tile products under tile-order variants: bash tools/ab_supertile_traffic.sh out "0 8,8 ..."
out=${1:-gpurun_out/supt}
export TMPDIR=/tmp
mkdir -p $out
for v in ${2:-0 8,8 6,11 4,8 8,4 4,16 16,4 2,32 12,11}; do
  export S2VT_SUP=$v
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/$v -o run -- python3 tools/sup_traffic.py > $out/$v.log 2>&1
  python3 - $out/$v $v <<'PY'
import csv, glob, sys, collections, re, json
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "gemm_kernel" in r["Kernel_Name"]:
            m = re.search(r"gemm_kernel<([^>]*)>", r["Kernel_Name"])
            agg[m.group(1)].append(float(r["Counter_Value"]))
print(json.dumps({"sup": sys.argv[2], "fetch_MB_per_launch": {k: round(2 * 1024 * sum(v) / len(v) / 1e6, 1) for k, v in agg.items()}}))
PY
done
