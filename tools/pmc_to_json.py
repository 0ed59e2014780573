#!/usr/bin/env python3
"""Aggregate the FETCH_SIZE / WRITE_SIZE passes of tools/collect_pmc.sh into profiles/<tag>_pmc_traffic.json:
per (kernel class : tile name) the average HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (both counters are
in KB; the factor 2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md section HBM)."""
import collections, csv, glob, json, os, re, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import kernel_signature          # hash of csrc/*.h, *.hip: bench.py only trusts a summary made from THIS build

src, dst = sys.argv[1], sys.argv[2]


def prof_key(kname):
    m = re.search(r"gemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false), (\d+), (\d+), (true|false)(?:, (true|false))?>", kname)
    if m:
        WM, WN, TM, TN, NG, EPI, _, BKT, PW = [int(x) for x in m.groups()[:6]] + [0] + [int(x) for x in m.groups()[7:9]]
        bt = m.groups()[9] == "true"
        sfx = (f"k{BKT}" if BKT != 32 else "") + (f"+{PW}" if PW else "") + ("[live]" if m.groups()[10] == "true" else "")
        if EPI == 3:
            return f"1:gw{WM * TM * 16}x{TN * 16}u({WM}x{WN}){sfx}"
        if EPI == 1:
            return f"1:{WM * TM * 16}x{WN * (TN // 4) * 16}u({WM}x{WN}){sfx}"
        if bt:
            return f"4:nt{WM * TM * 16}x{WN * TN * 16}({WM}x{WN}){sfx}"
        return f"{EPI}:{WM * TM * 16}x{WN * TN * 16}({WM}x{WN}){sfx}"
    m = re.search(r"gemm_tn_kernel<(\d+), (\d+), (\d+), (\d+), (true|false)>", kname)
    if m:
        WM, WN, TM, TN = [int(x) for x in m.groups()[:4]]
        return f"3:tn{WM * TM * 16}x{WN * TN * 16}({WM}x{WN})"
    if "gemm_tn_dma_kernel" in kname:
        return "3:tn128x128(dma)"
    m = re.search(r"lstm_chain_kernel<(\d+), (\d+), (\d+)>", kname)
    if m:
        ng, tmw, nc = [int(x) for x in m.groups()]
        return f"5:chain{2 if nc == 2 else ''}(ng{ng},m{tmw * 64 * nc})"
    m = re.search(r"lstm_chain4_live_kernel<(\d+), (\d+)>", kname)
    if m:
        ng, tpp = [int(x) for x in m.groups()]
        return f"5:chain4(ng{ng},m{tpp * 64})[live]"
    m = re.search(r"lstm_bwd_chain4_live_kernel<(\d+), (\d+)>", kname)
    if m:
        ng, tpp = [int(x) for x in m.groups()]
        return f"6:bchain4(ng{ng},m{tpp * 64})[live]"
    m = re.search(r"lstm_chain4_kernel<(\d+), (\d+)>", kname)
    if m:
        ng, tpp = [int(x) for x in m.groups()]
        return f"5:chain4(ng{ng},m{tpp * 64})"
    m = re.search(r"lstm_bwd_chain4_kernel<(\d+), (\d+)>", kname)
    if m:
        ng, tpp = [int(x) for x in m.groups()]
        return f"6:bchain4(ng{ng},m{tpp * 64})"
    m = re.search(r"lstm_bwd_chain_kernel<(\d+), (\d+), (\d+)>", kname)
    if m:
        ng, tmw, nc = [int(x) for x in m.groups()]
        return f"6:bchain{2 if nc == 2 else ''}(ng{ng},m{384 if nc == 2 else tmw * 64})"
    m = re.search(r"attn_chain_kernel<(\d+)>", kname)
    if m:
        return f"9:attn_chain(ng{m.group(1)})"
    m = re.search(r"attn_bwd_chain_kernel<(\d+)>", kname)
    if m:
        return f"10:attn_bchain(ng{m.group(1)})"
    if "attn_fwd_kernel" in kname:
        return "7:attn_fwd(score+softmax+ctx)"
    if "attn_bwd_kernel" in kname:
        return "8:attn_bwd"
    return "x:" + re.sub(r"^void ", "", kname).split("(")[0][-48:]


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
