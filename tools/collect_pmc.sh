#!/bin/bash
# HBM traffic of every kernel of the bench step, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
# SEPARATE rocprofv3 --pmc passes (TCC slots), no trace domains besides the kernel trace; FETCH_SIZE is doubled
# afterwards (gfx950 counts 128-B requests as 64 B).  Run on the GPU box from the repo root:
#   bash tools/collect_pmc.sh gpurun_out/pmc_r01
set -e
out=${1:-gpurun_out/pmc}
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out.$c.log 2>&1 || true
done
ls $out/*
