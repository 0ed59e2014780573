#!/usr/bin/env python3
"""dev: from a rocprofv3 --kernel-trace directory, the last step's kernels with their queue and the share of each one's run time during which a
kernel of ANOTHER queue was running (the gated overlap of train.hip: a weight-gradient contraction beside a persistent backward recurrence)."""
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
def short(n):
    n = re.sub(r"^void ", "", n).replace("s2vt::(anonymous namespace)::", "").replace("s2vt::", "")
    return n.split("(")[0][:60]
idx = [i for i, r in enumerate(rows) if "adam_tf_kernel" in r[2]]
lo, hi = (idx[-2] + 1, idx[-1] + 1) if len(idx) >= 2 else (0, len(rows))
step = rows[lo:hi]
t0 = step[0][0]
for s, e, n, q in step:
    ov = sum(max(0, min(e, e2) - max(s, s2)) for s2, e2, n2, q2 in step if q2 != q)
    if e - s > 20000 or ov:
        print(f"q{q} {(s - t0) / 1e3:8.1f} us {(e - s) / 1e3:7.1f} us  {100.0 * min(ov, e - s) / (e - s):3.0f} % beside another queue  {short(n)}")
print(f"step wall {(step[-1][1] - t0) / 1e3:.1f} us, kernel time sum {sum(e - s for s, e, _, _ in step) / 1e3:.1f} us")
