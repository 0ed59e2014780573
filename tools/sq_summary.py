#!/usr/bin/env python3
"""Per-kernel SQ counter summary (averages per launch) from tools/collect_sq.sh output."""
import collections, csv, glob, re, sys
src = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{src}/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"(gemm_kernel|gemm_tn_kernel|gemm_tn_dma_kernel|lstm_chain_kernel|lstm_chain4_kernel|lstm_chain4_live_kernel|lstm_bwd_chain_kernel|lstm_bwd_chain4_kernel|lstm_bwd_chain4_live_kernel|decode_lstm4_kernel|attn_chain_kernel|attn_bwd_chain_kernel)<([^>]*)>", k)
        name = (m.group(1).replace("gemm_", "") + "<" + m.group(2).replace(" ", "") + ">") if m else k.split("(")[0][-40:]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for name, c in agg.items():
    a = {k: sum(v) / len(v) for k, v in c.items()}
    n = len(c.get("SQ_WAVE_CYCLES", [])) or 1
    wc = a.get("SQ_WAVE_CYCLES", 0) * 4          # quad-cycles -> cycles, summed over waves
    if wc <= 0: continue
    waves = a.get("SQ_WAVES", 1)
    rows.append((wc * n, name, n, waves, wc / waves, a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / wc if wc else 0,
                 a.get("SQ_WAIT_ANY", 0) * 4 / wc, a.get("SQ_WAIT_INST_ANY", 0) * 4 / wc, a.get("SQ_ACTIVE_INST_ANY", 0) * 4 / wc,
                 a.get("SQ_INSTS_VALU", 0) / waves, a.get("SQ_INSTS_MFMA", 0) / waves, a.get("SQ_INSTS_LDS", 0) / waves,
                 a.get("SQ_INSTS_SALU", 0) / waves, a.get("SQ_INSTS_VMEM_RD", 0) / waves, a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_ACTIVE_INST_LDS", 1), 1),
                 a.get("GRBM_GUI_ACTIVE", 0) / 8))
print(f"{'kernel':<46}{'n':>5}{'waves':>7}{'cyc/wave':>10}{'mfma%':>7}{'wait%':>7}{'istall%':>8}{'active%':>8}{'valu/w':>8}{'mfma/w':>8}{'lds/w':>7}{'salu/w':>8}{'vmem/w':>7}{'bankcf':>7}{'gpu_cyc':>9}")
for r in sorted(rows, reverse=True)[:20]:
    print(f"{r[1]:<46}{r[2]:>5}{r[3]:>7.0f}{r[4]:>10.0f}{100*r[5]:>7.1f}{100*r[6]:>7.1f}{100*r[7]:>8.1f}{100*r[8]:>8.1f}{r[9]:>8.0f}{r[10]:>8.0f}{r[11]:>7.0f}{r[12]:>8.0f}{r[13]:>7.0f}{r[14]:>7.2f}{r[15]:>9.0f}")
