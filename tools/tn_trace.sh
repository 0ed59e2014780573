# per-launch durations of the weight-gradient kernels under a few settings (kernel trace of 5 rl steps each)
export TMPDIR=/tmp; mkdir -p gpurun_out/tntrace
i=0
for cfg in "S2VT_TN_DMA=1 S2VT_TN_OVH=12" "S2VT_TN_DMA=2 S2VT_TN_OVH=6" "S2VT_TN_DMA=1 S2VT_TN_WGS=1024" "S2VT_TN_DMA=3 S2VT_TN_OVH=12" "S2VT_TN_DMA=1 S2VT_TN_WGS=1536"; do
  export $cfg
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tntrace/c$i -o run -- python3 bench.py --workload ${W:-rl} --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/tntrace/c$i.log 2>&1
  unset S2VT_TN_DMA S2VT_TN_OVH S2VT_TN_WGS
  echo "c$i: $cfg"
  python3 - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/tntrace/c$i/*kernel_trace.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'gemm_tn' in r['Kernel_Name']:
        agg[(r['Kernel_Name'].split('(')[0][-42:], r['Grid_Size_X'], r['Grid_Size_Y'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in agg.items(): print('  ',k,len(v),round(sum(v)/len(v),1))
PY
  i=$((i+1))
done
