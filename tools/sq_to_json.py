#!/usr/bin/env python3
"""Aggregate the SQ-counter passes of tools/collect_round.sh (p1 / p2 / p3, separate rocprofv3 --pmc runs) into
profiles/<tag>_sq_counters[_<workload>].json, keyed like the launch profiler's rows (kernel class : tile name) and stamped with the hash
of the kernel sources, so bench.py can print `roofline.mfma_busy_pct` for the dominant kernel of THIS build (null when stale).

  mfma_busy_pct          SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x dispatch cycles), dispatch cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8
                         XCDs): the share of the chip's matrix-pipe cycles that issued MFMAs = MFMA utilisation against the peak at the clock
                         the launch ran at
  mfma_busy_pct_per_wave the same cycles / the waves' own lifetime (SQ_WAVE_CYCLES x 4): what tools/sq_summary.py prints as mfma%
  wait_pct, issue_stall_pct, lds_bank_conflict_ratio: as tools/sq_summary.py"""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench import kernel_signature
from prof_keys import prof_key

src, dst = sys.argv[1], sys.argv[2]
N_SIMD = 256 * 4
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{src}/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[prof_key(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for key, c in agg.items():
    a = {k: sum(v) / len(v) for k, v in c.items()}
    wc = a.get("SQ_WAVE_CYCLES", 0) * 4
    gpu = a.get("GRBM_GUI_ACTIVE", 0) / 8
    if wc <= 0:
        continue
    ent = {"launches_sampled": len(c.get("SQ_WAVE_CYCLES", [])), "waves": round(a.get("SQ_WAVES", 0)),
           "mfma_busy_pct_per_wave": round(100 * a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / wc, 1),
           "wait_pct": round(100 * a.get("SQ_WAIT_ANY", 0) * 4 / wc, 1), "issue_stall_pct": round(100 * a.get("SQ_WAIT_INST_ANY", 0) * 4 / wc, 1),
           "lds_bank_conflict_ratio": round(a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_ACTIVE_INST_LDS", 1), 1), 3),
           "dispatch_cycles": round(gpu)}
    ent["mfma_busy_pct"] = round(100 * a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (N_SIMD * gpu), 1) if gpu > 0 else None
    out[key] = ent
stamped = dict(out)
stamped["_stamp"] = {"kernel_signature": kernel_signature(), "command": "tools/collect_round.sh (rocprofv3 --pmc SQ_* / GRBM_GUI_ACTIVE, separate passes)"}
json.dump(stamped, open(dst, "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -(kv[1]["dispatch_cycles"] * kv[1]["launches_sampled"]))[:14]:
    print(f"{k:<34} mfma busy {v['mfma_busy_pct']} % of the chip's pipe cycles ({v['mfma_busy_pct_per_wave']} % per wave), wait {v['wait_pct']} %, bank conflicts {v['lds_bank_conflict_ratio']}")
