#!/usr/bin/env python3
"""Build-time audit of the hand-pipelined loads (cdna_hip_programming.md §5.7 item 1): between an
inline-asm global_load_dwordx4 and the next hand-written `s_waitcnt vmcnt`, NO instruction may read
the load's destination registers (a compiler copy of a ring slot before its data has landed is
silent garbage).  Also requires zero scratch / spills in the pipelined kernels.
usage: check_asm_ring.py file.s"""
import re
import sys


def regs(tok):
    m = re.match(r'[va]\[(\d+):(\d+)\]', tok)
    if m:
        return {f"{tok[0]}{i}" for i in range(int(m.group(1)), int(m.group(2)) + 1)}
    m = re.match(r'([va])(\d+)$', tok)
    return {tok} if m else set()


def main(path):
    s = open(path).read()
    bad = 0
    for m in re.finditer(r'^(_ZN4s2vt\w+):\n(.*?)\n\.Lfunc_end', s, re.S | re.M):
        name, body = m.group(1), m.group(2).split('\n')
        pending = {}          # reg -> line of the asm load
        in_asm = False
        for n, line in enumerate(body):
            t = line.strip()
            if t.startswith(';;#ASMSTART'):
                in_asm = True; continue
            if t.startswith(';;#ASMEND'):
                in_asm = False; continue
            if not t or t.startswith(';') or t.startswith('.'):
                continue
            ops = re.split(r'[,\s]+', t)
            if in_asm and ops[0] == 'global_load_dwordx4':
                for r in regs(ops[1]):
                    pending[r] = n
                continue
            if in_asm and ops[0] == 's_waitcnt':
                # hand-counted in-order wait: all but the N youngest asm loads have landed
                mm = re.search(r'vmcnt\((\d+)\)', t)
                keep = int(mm.group(1)) if mm else 0
                lines = sorted(set(pending.values()))
                young = set(lines[len(lines) - keep:]) if keep else set()
                pending = {r: ln for r, ln in pending.items() if ln in young}
                continue
            if pending and not in_asm:
                srcs = set()
                for o in ops[2:] if len(ops) > 2 else []:
                    srcs |= regs(o)
                if ops[0].startswith(('ds_write', 'global_store', 'buffer_store')):
                    srcs |= regs(ops[1]) | (regs(ops[2]) if len(ops) > 2 else set())
                hit = srcs & set(pending)
                # registers still in flight because their wait only covers older chunks are legitimately
                # pending; a READ of them is the bug
                if hit:
                    print(f"{name}: line {n}: '{t}' reads in-flight ring register(s) {sorted(hit)}")
                    bad += 1
                    for r in hit:
                        pending.pop(r, None)
    for n, p in re.findall(r'\.name:\s+(_ZN4s2vt11gemm_kernel\S+)\n(?:.*\n){0,40}?\s+\.private_segment_fixed_size:\s+(\d+)', s):
        if int(p) and 'Lb1E' in n:
            print(f"{n}: scratch {p} bytes in a pipelined kernel"); bad += 1
    print("asm ring audit:", "FAILED" if bad else "ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
