#!/bin/bash
# Row-group pipelining of the sampler's decode loop (S2VT_SAMPLE_GROUPS=1|2|3, csrc/api.hip::sample_decode): step time of the rl workload per
# setting (two runs each, interleaved), the ids' bit-exactness (tests), and a rocprofv3 kernel trace of the two-group form.
#   bash tools/ab_sample_groups.sh gpurun_out/groups
out=${1:-gpurun_out/groups}
mkdir -p $out
export TMPDIR=/tmp
for rep in 1 2; do
  for g in 1 2 3; do
    S2VT_SAMPLE_GROUPS=$g python3 bench.py --workload rl --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'groups': $g, 'rep': $rep, 'ms_per_step': d['ms_per_step'], 'median': d['step_ms']['median'], 'kernels': [(k['tile'], k['launches'], k['ms']) for k in d['roofline']['all_kernels_warmup'] if k['class'] in (1, 2)]}))" >> $out/ab.jsonl
  done
done
S2VT_SAMPLE_GROUPS=2 python3 -m pytest tests/test_gpu_fullsize.py::test_sampler_determinism_and_shard_independence tests/test_gpu_fwd.py::test_sampler_token_ids_bit_exact -x -q 2>&1 | tail -3 > $out/ids_groups2.log
S2VT_SAMPLE_GROUPS=3 python3 -m pytest tests/test_gpu_fullsize.py::test_sampler_determinism_and_shard_independence -x -q 2>&1 | tail -3 > $out/ids_groups3.log
export S2VT_SAMPLE_GROUPS=2
rocprofv3 --kernel-trace --output-format csv -d $out/trace2 -o run -- python3 bench.py --workload rl --steps 3 --warmup 2 --no-cpu-baseline > $out/trace2.log 2>&1
unset S2VT_SAMPLE_GROUPS
python3 tools/two_stream_overlap.py $out/trace2 > $out/overlap.txt 2>&1
cat $out/ab.jsonl $out/ids_groups2.log $out/ids_groups3.log $out/overlap.txt
