#!/bin/bash
# A/B of the L2-sized supertile order (fwd.hip, gemm_mfma.h xcd_map == 2): step time + per-kernel table, then fetch traffic.
#   bash tools/ab_supertile.sh gpurun_out/sup
out=${1:-gpurun_out/sup}
export TMPDIR=/tmp
mkdir -p $out
for v in 0 default 8,8 6,11 4,16 16,4; do
  if [ "$v" = default ]; then unset S2VT_SUP; else export S2VT_SUP=$v; fi
  python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
rows = {r['tile']: (r['ms'] / r['launches'] * 1e3, r['tflops']) for r in d['roofline']['all_kernels_warmup'] if r['class'] in (0, 4)}
print(json.dumps({'sup': '$v', 'ms_per_step': d['ms_per_step'], 'us_per_launch_tflops': rows}))" | tee -a $out/ab.jsonl
done
for v in 0 default; do
  if [ "$v" = default ]; then unset S2VT_SUP; else export S2VT_SUP=$v; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_$v/FETCH_SIZE -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_$v.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_$v/WRITE_SIZE -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline >> $out/pmc_$v.log 2>&1
  python3 tools/pmc_to_json.py $out/pmc_$v $out/traffic_$v.json | head -8
done
