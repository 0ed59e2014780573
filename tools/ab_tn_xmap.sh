mkdir -p gpurun_out/xmap; export TMPDIR=/tmp
python -m pytest tests/test_gpu_timed_tiles.py tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -2
for x in 0 1; do
  S2VT_TN_XMAP=$x python bench.py --steps 80 --no-cpu-baseline > gpurun_out/xmap/rl$x.json 2>/dev/null
  S2VT_TN_XMAP=$x python bench.py --workload xe --steps 80 --no-cpu-baseline > gpurun_out/xmap/xe$x.json 2>/dev/null
  S2VT_TN_XMAP=$x rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/xmap/fetch$x -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/xmap/pmc$x.log 2>&1
done
python - <<'PY'
import json,glob,csv,collections
for x in (0,1):
    for w in ('rl','xe'):
        d=json.loads(open(f'gpurun_out/xmap/{w}{x}.json').read().strip().splitlines()[-1])
        print(x,w,d['ms_per_step'],[ (k['tile'],k['tflops']) for k in d['roofline']['all_kernels_warmup'] if k['class']==3])
    agg=collections.defaultdict(list)
    for f in glob.glob(f'gpurun_out/xmap/fetch{x}/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            if 'gemm_tn_kernel' in r['Kernel_Name']: agg[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(x,k,len(v),sum(v)/len(v)*2*1024/1e6 if max(v)<1e7 else sum(v)/len(v))
PY
