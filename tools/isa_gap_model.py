#!/usr/bin/env python3
"""Gap model of an fp32-MFMA loop on gfx950 (tools/mfma_shadow_probe.hip, profiles/r04_mfma_shadow_probe.txt):
in ONE wave a v_mfma_f32_32x32x2_f32 occupies the pipe for 64 cycles and NO vector-ALU instruction of the same wave overlaps with it:
a group of k VALU instructions (v_*, v_pk_*, v_accvgpr_*) between two MFMAs costs ~17 + 4.5 k cycles; LDS reads, global loads, SALU
and waits for data that already arrived are free.  Prints, for the hottest loops of a kernel, the VALU groups between MFMAs and the
modelled matrix-pipe efficiency of the loop for a wave alone on its SIMD.
usage: tools/isa_gap_model.py file.s kernel-substring [min_mfma]"""
import re, sys

def main():
    path, want = sys.argv[1], sys.argv[2]
    min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    lines = open(path).read().splitlines()
    kernels, cur = {}, None
    for l in lines:
        m = re.match(r"^(\S+):\s*(;.*)?$", l)
        if m and not m.group(1).startswith(".L") and not l.startswith("\t"):
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is not None:
            cur.append(l)
    for name, body in kernels.items():
        if want not in name or not any("v_mfma" in b for b in body):
            continue
        labels = {}
        for i, l in enumerate(body):
            m = re.match(r"^(\.LBB\S+):", l)
            if m:
                labels[m.group(1)] = i
        loops = []
        for i, l in enumerate(body):
            m = re.match(r"^\s+s_cbranch_\S+\s+(\.LBB\S+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        print(name)
        for a, b in loops:
            ops = []
            for l in body[a:b + 1]:
                t = l.strip()
                if not t or t.startswith(";") or t.startswith(".") or re.match(r"^\.LBB", t):
                    continue
                ops.append(t.split()[0])
            nm = sum(o.startswith("v_mfma") for o in ops)
            if nm < min_mfma:
                continue
            groups, k, other = [], 0, {}
            for o in ops:
                if o.startswith("v_mfma"):
                    if k:
                        groups.append(k)
                    k = 0
                elif o.startswith("v_"):
                    k += 1
                else:
                    key = o.split("_")[0] + "_" + o.split("_")[1] if "_" in o else o
                    other[key] = other.get(key, 0) + 1
            if k:
                groups.append(k)
            nv = sum(groups)
            cost = sum(17 + 4.5 * g for g in groups)
            eff = nm * 64 / (nm * 64 + cost)
            print(f"  loop @{a}-{b}: {nm} MFMA, {nv} VALU in {len(groups)} groups {groups}")
            print(f"     modelled VALU cost {cost:.0f} cycles vs {nm*64} MFMA cycles => pipe efficiency {eff:.3f} (one wave per SIMD); other: {other}")

main()
