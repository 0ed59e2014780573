# A/B of the LDS-DMA weight-gradient tile (S2VT_TN_DMA=0/1): parity tests, then bench lines of the three workloads
mkdir -p gpurun_out/dma; export TMPDIR=/tmp
python -m pytest tests/test_gpu_timed_tiles.py tests/test_gpu_train.py tests/test_gpu_fwd.py -m gpu -x -q 2>&1 | tail -3
for x in 0 1; do
  for w in rl xe multitask; do
    S2VT_TN_DMA=$x python bench.py --workload $w --steps 80 --no-cpu-baseline > gpurun_out/dma/$w$x.json 2>/dev/null
  done
done
python - <<'PY'
import json
for x in (0,1):
    for w in ('rl','xe','multitask'):
        d=json.loads(open(f'gpurun_out/dma/{w}{x}.json').read().strip().splitlines()[-1])
        print(x,w,d['ms_per_step'],[ (k['tile'],k['tflops']) for k in d['roofline']['all_kernels_warmup'] if k['class']==3])
PY
