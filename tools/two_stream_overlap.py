#!/usr/bin/env python3
"""Read a rocprofv3 --kernel-trace directory of a run with S2VT_SAMPLE_GROUPS=2 and report, for the sampler's decode loop (the LSTM_GW cell step and the
PICK kernel), how much of their execution overlapped a kernel of the OTHER stream, and the wall time of one 20-step loop."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
def kind(n):
    if "gemm_kernel<" in n:
        a = n.split("gemm_kernel<")[1].split(">")[0].replace(" ", "").split(",")
        return {"2": "pick", "3": "cell"}.get(a[5])
    return None
dec = [(s, e, kind(n), q) for s, e, n, q in rows if kind(n)]
print(f"{len(rows)} kernels traced, {len(dec)} decode-loop launches, queues: {sorted(set(q for *_, q in dec))}")
tot = {"pick": [0, 0, 0], "cell": [0, 0, 0]}
for i, (s, e, k, q) in enumerate(dec):
    ov = 0
    for s2, e2, k2, q2 in dec[max(0, i - 6):i + 7]:
        if q2 != q:
            ov += max(0, min(e, e2) - max(s, s2))
    tot[k][0] += 1; tot[k][1] += e - s; tot[k][2] += min(ov, e - s)
for k, (n, dur, ov) in tot.items():
    if n:
        print(f"{k}: {n} launches, avg {dur / n / 1e3:.1f} us, {100.0 * ov / dur:.0f} % of their time beside a decode kernel of the other stream")
# wall time of the decode loops: gaps > 200 us between consecutive decode launches separate the sampler calls
loops, start, last = [], dec[0][0], dec[0][1]
for s, e, k, q in dec[1:]:
    if s - last > 200000:
        loops.append(last - start); start = s
    last = max(last, e)
loops.append(last - start)
print("decode-loop wall times (us):", [round(x / 1e3) for x in loops])
