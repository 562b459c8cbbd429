#!/usr/bin/env python3
"""Instruction-class sequence of the loops of a kernel in a gfx950 .s file: for every backward branch, the run-length-compressed
op classes between target and branch.  usage: tools/isa_seq.py file.s kernel-substring [must-contain-op] [must-not-contain-op]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if re.match(r'^[A-Za-z_].*:', l) and sys.argv[2] in l and not l.startswith('.')][0]
labels = {}
end = len(lines)
for i in range(start, len(lines)):
    m = re.match(r'^(\.LBB\d+_\d+):', lines[i])
    if m: labels[m.group(1)] = i
    if lines[i].strip().startswith('s_endpgm'): end = i; break
need = sys.argv[3] if len(sys.argv) > 3 else None
forbid = sys.argv[4] if len(sys.argv) > 4 else None
def cls(op):
    for pre, k in (('v_mfma', 'MFMA'), ('v_pk', 'pk'), ('v_readlane', 'LANE'), ('v_writelane', 'LANE'), ('v_', 'VALU'), ('ds_read', 'dsr'), ('ds_write', 'dsw'),
                   ('scratch', 'SCRATCH'), ('buffer_load', 'vld'), ('buffer_store', 'vst'), ('global_', 'glb'), ('s_waitcnt', 'wait'), ('s_barrier', 'BAR'),
                   ('s_cbranch', 'br'), ('s_branch', 'br'), ('s_load', 'sld'), ('s_', 's')):
        if op.startswith(pre): return k
    return op
seen = set()
for i in range(start, end):
    m = re.match(r'\s+s_c?branch\w*\s+(\.LBB\d+_\d+)', lines[i])
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        if (a, i) in seen: continue
        seen.add((a, i))
        ops = [l.strip().split()[0] for l in lines[a:i + 1] if l.strip() and not l.strip().startswith(';') and not re.match(r'^\.', l.strip()) and not l.strip().endswith(':')]
        if need and not any(o.startswith(need) for o in ops): continue
        if forbid and any(o.startswith(forbid) for o in ops): continue
        out, prev, c = [], None, 0
        for o in ops:
            k = cls(o)
            if k == prev: c += 1
            else:
                if prev: out.append(prev + (f"x{c}" if c > 1 else ""))
                prev, c = k, 1
        out.append(prev + (f"x{c}" if c > 1 else ""))
        print(f"[{a}-{i}] {len(ops)} instrs: " + ' '.join(out))
