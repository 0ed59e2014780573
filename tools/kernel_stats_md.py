#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats csv -> the markdown table kept under profiles/ (per-step figures).
usage: kernel_stats_md.py run_kernel_stats.csv STEPS_TRACED 'title' > profiles/<tag>_kernel_stats.md"""
import csv, sys

src, steps, title = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows = list(csv.DictReader(open(src)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {title}\n\n{steps} steps traced; kernel time {tot / steps / 1e6:.2f} ms per step.\n")
print("| kernel | calls | ms/step | avg us | % |\n|---|---|---|---|---|")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:26]:
    t = float(r["TotalDurationNs"])
    print(f"| `{r['Name'][:90]}` | {r['Calls']} | {t / steps / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {100 * t / tot:.1f} |")
