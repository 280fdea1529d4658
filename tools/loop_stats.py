#!/usr/bin/env python3
"""Static shape of a decode kernel's byte loop, from the built library (no GPU): the innermost loop around the nine decoder
steps of a byte, its instruction mix, and the distances between consecutive steps.  python tools/loop_stats.py [kernel ...]"""
import os, re, sys
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import count_instr as ci

def stats(ins, want=9):
    addrs = [a for a, m, o in ins]
    idx = {a: i for i, a in enumerate(addrs)}
    muls = [i for i, (a, m, o) in enumerate(ins) if m == 's_mul_hi_u32']
    loops = []
    for i, (a, m, o) in enumerate(ins):
        if m.startswith('s_cbranch') or m == 's_branch':
            mm = re.match(r'(-?\d+)', o)
            if mm:
                off = int(mm.group(1))
                if off > 32767: off -= 65536
                tgt = a + 4 + off * 4
                if tgt <= a and tgt in idx: loops.append((idx[tgt], i))
    best = None
    for lo, hi in loops:
        n = sum(1 for k in muls if lo <= k <= hi)
        if n >= want and (best is None or hi - lo < best[1] - best[0]): best = (lo, hi, n)
    if not best: return None
    lo, hi, n = best
    c = Counter()
    for a, m, o in ins[lo:hi + 1]:
        k = 'salu' if m.startswith('s_') else 'valu' if m.startswith('v_') else 'lds' if m.startswith('ds_') else 'vmem'
        if m.startswith('s_cbranch') or m == 's_branch': k = 'branch'
        if m == 's_waitcnt': k = 'wait'
        if m == 's_nop': k = 'nop'
        if m in ('v_readlane_b32', 'v_writelane_b32', 'v_readfirstlane_b32'): k = m[2:-4]
        c[k] += 1
    ml = [k for k in muls if lo <= k <= hi]
    return dict(instr=hi - lo + 1, steps=n, bytes=addrs[hi] - addrs[lo], mix=dict(c), between=[ml[i + 1] - ml[i] for i in range(len(ml) - 1)], head=ml[0] - lo, tail=hi - ml[-1])

if __name__ == '__main__':
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'zpaqsharp_amd', 'libzpaqhip.so')
    funcs = ci.disassemble(lib)
    for name in sys.argv[1:] or ['zh_decode_nb_min', 'zh_decode_nb_mid', 'zh_decode_c2_min', 'zh_decode_c2_mid']:
        print(name, stats(funcs[name]))
