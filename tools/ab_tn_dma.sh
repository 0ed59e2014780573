# A/B of the LDS-DMA forms of the weight-gradient tile (S2VT_TN_DMA=0: register staging, n>=1: kTnDma[n-1]) and of the
# slab-count rule (S2VT_TN_OVH = fixed cost per workgroup in chunks; S2VT_TN_WGS = the old rule): parity tests, bench lines
mkdir -p gpurun_out/dma; export TMPDIR=/tmp
python -m pytest tests/test_gpu_timed_tiles.py tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -1
run() {  # tag, env...
  tag=$1; shift
  for w in rl xe multitask; do
    env "$@" python bench.py --workload $w --steps 60 --no-cpu-baseline > gpurun_out/dma/$w.$tag.json 2>/dev/null
  done
}
run reg S2VT_TN_DMA=0
run dma_wgs1024 S2VT_TN_WGS=1024
run dma_ovh4 S2VT_TN_OVH=4
run dma_ovh12 S2VT_TN_OVH=12
run dma_ovh30 S2VT_TN_OVH=30
run dma32_ovh6 S2VT_TN_DMA=2 S2VT_TN_OVH=6
run dma32_wgs1024 S2VT_TN_DMA=2 S2VT_TN_WGS=1024
python - <<'PY'
import json,glob
for tag in ('reg','dma_wgs1024','dma_ovh4','dma_ovh12','dma_ovh30','dma32_ovh6','dma32_wgs1024'):
    for w in ('rl','xe','multitask'):
        d=json.loads(open(f'gpurun_out/dma/{w}.{tag}.json').read().strip().splitlines()[-1])
        print(tag,w,d['ms_per_step'],[ (k['tile'],k['tflops']) for k in d['roofline']['all_kernels_warmup'] if k['class']==3][:1])
PY
