#!/bin/bash
# One round's profile set, on the GPU box from the repo root:  bash tools/collect_round.sh gpurun_out/r04prof
#   kernel stats (rocprofv3 --kernel-trace --stats) of the bench workloads, HBM traffic (separate --pmc passes,
#   MI355X_MICROARCH.md section HBM) of every workload, SQ counters of the default, the XE and the attention workload.
set -x
out=${1:-gpurun_out/prof}
export TMPDIR=/tmp
mkdir -p $out
for w in rl xe multitask attention attention32 rl_msvd rl_msvd_eos; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$w -o run -- python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $out/stats_$w.log 2>&1
done
for w in rl xe multitask attention attention32 rl_msvd rl_msvd_eos; do
  sfx=$([ $w = rl ] && echo "" || echo "_$w")
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc$sfx/$c -o run -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc$sfx.$c.log 2>&1
  done
done
for w in rl xe attention; do
  sfx=$([ $w = rl ] && echo "" || echo "_$w")
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out/sq$sfx/p1 -o run -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $out/sq$sfx.p1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA --kernel-trace --output-format csv -d $out/sq$sfx/p2 -o run -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $out/sq$sfx.p2.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/sq$sfx/p3 -o run -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $out/sq$sfx.p3.log 2>&1
done
find $out -name "*.csv" | head -40
