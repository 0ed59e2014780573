"""rocprofv3 kernel name -> the launch profiler's (kernel class : tile name) key of bench.py's tables (shared by pmc_to_json.py and
sq_to_json.py)."""
import re


def prof_key(kname):
    m = re.search(r"gemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false), (\d+), (\d+), (true|false)(?:, (true|false))?(?:, (\d+))?>", kname)
    if m:
        WM, WN, TM, TN, NG, EPI, _, BKT, PW = [int(x) for x in m.groups()[:6]] + [0] + [int(x) for x in m.groups()[7:9]]
        bt = m.groups()[9] == "true"
        dm = int(m.groups()[11] or 0)                      # round 6: LDS-DMA ring stages (fwd.hip names "<tile>[k64]+<PW>dma<DM>")
        sfx = (f"k{BKT}" if BKT != 32 else "") + (f"+{PW}" if PW else "") + (f"dma{dm}" if dm else "") + ("[live]" if m.groups()[10] == "true" else "")
        if EPI == 3:
            return f"1:gw{WM * TM * 16}x{TN * 16}u({WM}x{WN}){sfx}"
        if EPI == 1:
            return f"1:{WM * TM * 16}x{WN * (TN // 4) * 16}u({WM}x{WN}){sfx}"
        if bt:
            return f"4:nt{WM * TM * 16}x{WN * TN * 16}({WM}x{WN}){sfx}"
        return f"{EPI}:{WM * TM * 16}x{WN * TN * 16}({WM}x{WN}){sfx}"
    m = re.search(r"gemm_tn_kernel<(\d+), (\d+), (\d+), (\d+), (true|false)>", kname)
    if m:
        WM, WN, TM, TN = [int(x) for x in m.groups()[:4]]
        return f"3:tn{WM * TM * 16}x{WN * TN * 16}({WM}x{WN})"
    if "gemm_tn_dma_kernel" in kname:
        return "3:tn128x128(dma)"
    m = re.search(r"lstm_chain_kernel<(\d+), (\d+), (\d+)>", kname)
    if m:
        ng, tmw, nc = [int(x) for x in m.groups()]
        return f"5:chain{2 if nc == 2 else ''}(ng{ng},m{tmw * 64 * nc})"
    m = re.search(r"lstm_chain4_live_kernel<(\d+), (\d+)>", kname)
    if m:
        ng, tpp = [int(x) for x in m.groups()]
        return f"5:chain4(ng{ng},m{tpp * 64})[live]"
    m = re.search(r"lstm_bwd_chain4_live_kernel<(\d+), (\d+)>", kname)
    if m:
        ng, tpp = [int(x) for x in m.groups()]
        return f"6:bchain4(ng{ng},m{tpp * 64})[live]"
    m = re.search(r"lstm_chain4_kernel<(\d+), (\d+)>", kname)
    if m:
        ng, tpp = [int(x) for x in m.groups()]
        return f"5:chain4(ng{ng},m{tpp * 64})"
    m = re.search(r"lstm_bwd_chain4_kernel<(\d+), (\d+)>", kname)
    if m:
        ng, tpp = [int(x) for x in m.groups()]
        return f"6:bchain4(ng{ng},m{tpp * 64})"
    m = re.search(r"lstm_bwd_chain_kernel<(\d+), (\d+), (\d+)>", kname)
    if m:
        ng, tmw, nc = [int(x) for x in m.groups()]
        return f"6:bchain{2 if nc == 2 else ''}(ng{ng},m{384 if nc == 2 else tmw * 64})"
    m = re.search(r"decode_loop_kernel<(\d+)>", kname)
    if m:                                                   # the persistent decode loop: the launch profiler files it under class 2 (vocabulary pick)
        tpp = int(m.group(1))
        return f"2:decloop(m{64 if tpp == 1 else tpp * 64})"
    m = re.search(r"attn_chain_kernel<(\d+)>", kname)
    if m:
        return f"9:attn_chain(ng{m.group(1)})"
    m = re.search(r"attn_bwd_chain_kernel<(\d+)>", kname)
    if m:
        return f"10:attn_bchain(ng{m.group(1)})"
    if "attn_fwd_kernel" in kname:
        return "7:attn_fwd(score+softmax+ctx)"
    if "attn_bwd_kernel" in kname:
        return "8:attn_bwd"
    return "x:" + re.sub(r"^void ", "", kname).split("(")[0][-48:]
