for w in rl multitask; do for d in 0 1; do
S2VT_DMA=$d python bench.py --workload $w --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/ab_${w}_dma$d.json
python3 - <<PY
import json
d=json.loads(open("gpurun_out/ab_${w}_dma$d.json").read().strip().splitlines()[-1])
print("$w dma=$d ms/step", d["ms_per_step"])
for r in d["roofline"]["all_kernels_warmup"]:
    if r["class"] in (1,2): print("   ", r)
PY
done; done
