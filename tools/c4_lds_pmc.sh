# dev: LDS bank-conflict counters of lstm_chain4_kernel, with and without its staged epilogue (variants/lib_base.so, lib_noepi.so)
export TMPDIR=/tmp
for v in base noepi; do
  S2VT_LIB=$PWD/variants/lib_$v.so rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/c4pmc_$v -o run -- python3 tools/c4_run.py > gpurun_out/c4pmc_$v.log 2>&1
  python3 - gpurun_out/c4pmc_$v $v <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "chain4" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v)/len(v)) for k, v in agg.items()})
PY
done
