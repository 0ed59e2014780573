#!/usr/bin/env python3
"""Aggregate the FETCH_SIZE / WRITE_SIZE passes of tools/collect_pmc.sh into profiles/<tag>_pmc_traffic.json:
per (kernel class : tile name) the average HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (both counters are
in KB; the factor 2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md section HBM)."""
import collections, csv, glob, json, os, re, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench import kernel_signature          # hash of csrc/*.h, *.hip: bench.py only trusts a summary made from THIS build

src, dst = sys.argv[1], sys.argv[2]


from prof_keys import prof_key


agg = {c: collections.defaultdict(list) for c in ("FETCH_SIZE", "WRITE_SIZE")}
for c in agg:
    for f in glob.glob(f"{src}/{c}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                agg[c][prof_key(r["Kernel_Name"])].append(float(r["Counter_Value"]))
out = {}
for k in sorted(set(agg["FETCH_SIZE"]) | set(agg["WRITE_SIZE"])):
    f, w = agg["FETCH_SIZE"].get(k, []), agg["WRITE_SIZE"].get(k, [])
    fetch = 2.0 * 1024.0 * (sum(f) / len(f)) if f else 0.0
    write = 1024.0 * (sum(w) / len(w)) if w else 0.0
    out[k] = {"bytes_per_launch": round(fetch + write), "fetch_bytes": round(fetch), "write_bytes": round(write), "launches_sampled": max(len(f), len(w))}
stamped = dict(out)
stamped["_stamp"] = {"kernel_signature": kernel_signature(), "command": "tools/collect_pmc.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"}
json.dump(stamped, open(dst, "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["bytes_per_launch"])[:14]:
    print(f"{k:<34} {v['bytes_per_launch'] / 1e6:10.1f} MB/launch  (fetch {v['fetch_bytes'] / 1e6:.1f}, write {v['write_bytes'] / 1e6:.1f}; n={v['launches_sampled']})")
