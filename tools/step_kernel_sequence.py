#!/usr/bin/env python3
"""dev: the kernel sequence of ONE step from a rocprofv3 --kernel-trace directory (the last full step of the run), names shortened --
to find the tensor-library / runtime kernels (at::native::*, __amd_rocclr_*) that still sit between the library's launches.
  rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python3 bench.py --workload xe --steps 3 --warmup 1 --no-cpu-baseline
  python3 tools/step_kernel_sequence.py DIR"""
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("s2vt::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("s2vt::", "")
    return n.split("(")[0][:70]
# a step ends with adam_tf_kernel: take the launches between the last two
idx = [i for i, r in enumerate(rows) if "adam_tf_kernel" in r[2]]
lo, hi = (idx[-2] + 1, idx[-1] + 1) if len(idx) >= 2 else (0, len(rows))
foreign = 0
for s, e, n in rows[lo:hi]:
    tag = "" if ("s2vt::" in n or "(anonymous namespace)::" in n and "at::" not in n) else "   <-- not a library kernel"
    foreign += bool(tag)
    print(f"{(s - rows[lo][0]) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  {short(n)}{tag}")
print(f"{hi - lo} launches in the step, {foreign} of them not library kernels")
