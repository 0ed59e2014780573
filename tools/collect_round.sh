#!/bin/bash
# One round's profile set, on the GPU box from the repo root:  bash tools/collect_round.sh gpurun_out/r05prof
#   bench lines (the JSON the driver records) of every workload, kernel stats (rocprofv3 --kernel-trace --stats), HBM traffic (separate
#   --pmc passes, MI355X_MICROARCH.md section HBM) and SQ counters (three separate --pmc passes) of every workload.
#   rl_ref (the reference's own default configuration, 0.1 s per step) runs fewer steps.
set -x
out=${1:-gpurun_out/prof}
export TMPDIR=/tmp
mkdir -p $out
WL="rl xe multitask attention attention32 rl_msvd rl_msvd_eos rl_ref"
steps() { [ $1 = rl_ref ] && echo "--steps 3 --warmup 1" || echo "--steps 10 --warmup 2"; }
psteps() { [ $1 = rl_ref ] && echo "--steps 1 --warmup 1" || echo "--steps 2 --warmup 1"; }
for w in $WL; do
  extra=$([ $w = rl ] && echo "" || echo "--no-cpu-baseline")
  bs=$([ $w = rl_ref ] && echo "--steps 20 --warmup 3" || echo "")
  python3 bench.py --workload $w $bs $extra 2> $out/bench_$w.err | tail -1 > ${out}_bench_$w.json
done
for w in $WL; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$w -o run -- python3 bench.py --workload $w $(steps $w) --no-cpu-baseline > $out/stats_$w.log 2>&1
done
for w in $WL; do
  sfx=$([ $w = rl ] && echo "" || echo "_$w")
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc$sfx/$c -o run -- python3 bench.py --workload $w $(psteps $w) --no-cpu-baseline > $out/pmc$sfx.$c.log 2>&1
  done
done
for w in rl xe multitask attention attention32 rl_ref; do
  sfx=$([ $w = rl ] && echo "" || echo "_$w")
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out/sq$sfx/p1 -o run -- python3 bench.py --workload $w $(psteps $w) --no-cpu-baseline > $out/sq$sfx.p1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA --kernel-trace --output-format csv -d $out/sq$sfx/p2 -o run -- python3 bench.py --workload $w $(psteps $w) --no-cpu-baseline > $out/sq$sfx.p2.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/sq$sfx/p3 -o run -- python3 bench.py --workload $w $(psteps $w) --no-cpu-baseline > $out/sq$sfx.p3.log 2>&1
done
# keep what gpurun merges back small: the traces themselves stay on the box, the per-kernel summaries travel
for d in $(find $out -name "*kernel_trace.csv"); do rm -f $d; done
find $out -name "*agent_info.csv" -delete
du -sh $out
