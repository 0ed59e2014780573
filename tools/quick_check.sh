mkdir -p gpurun_out/dma2; export TMPDIR=/tmp
python -m pytest tests/test_gpu_fwd.py tests/test_gpu_fullsize.py tests/test_gpu_replay_train.py tests/test_gpu_beam.py -m gpu -x -q 2>&1 | tail -2
for w in rl xe multitask; do python bench.py --workload $w --steps 80 --no-cpu-baseline > gpurun_out/dma2/$w.json 2>/dev/null; done
python - <<'PY'
import json
for w in ('rl','xe','multitask'):
    d=json.loads(open(f'gpurun_out/dma2/{w}.json').read().strip().splitlines()[-1])
    print(w,d['ms_per_step'],d['roofline']['kernel'],d['roofline']['achieved'],d['roofline']['frac'],[ (k['tile'],k['launches'],k['ms'],k['tflops']) for k in d['roofline']['all_kernels_warmup'] if k['class']==3])
PY
