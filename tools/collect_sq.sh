#!/bin/bash
# SQ counters per kernel of the bench step (separate passes, --pmc with the kernel trace only).
set -e
out=${1:-gpurun_out/sq}
export TMPDIR=/tmp
mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out/p1 -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out.p1.log 2>&1 || true
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA --kernel-trace --output-format csv -d $out/p2 -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out.p2.log 2>&1 || true
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p3 -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out.p3.log 2>&1 || true
ls $out/*
