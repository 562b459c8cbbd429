#!/usr/bin/env python3
"""Instruction mix of the hottest loop of each kernel in a gfx950 .s file (hipcc -save-temps): for every backward branch the body
between its target label and the branch is a loop; the one holding the most v_mfma instructions is reported, by instruction class.
usage: tools/isa_loop_stats.py file.s [kernel-name-substring]"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "accvgpr_mov"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return "lds_write"
    if op.startswith("buffer_load") or op.startswith("global_load"):
        return "vmem_load"
    if op.startswith("buffer_store") or op.startswith("global_store"):
        return "vmem_store"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    lines = open(path).read().splitlines()
    kernels, cur, name = {}, None, None
    for l in lines:
        m = re.match(r"^(\S+):\s*(;.*)?$", l)
        if m and not m.group(1).startswith(".L") and not l.startswith("\t"):
            name = m.group(1)
            cur = kernels.setdefault(name, [])
            continue
        if cur is not None:
            cur.append(l)
        if l.strip().startswith(".end_amdhsa_kernel") or l.strip() == "s_endpgm":
            pass
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
            m = re.match(r"^\s+s_cbranch_\S+\s+(\.LBB\S+)", l) or re.match(r"^\s+s_branch\s+(\.LBB\S+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        # innermost loops that hold MFMAs: no other loop strictly inside
        inner = [(a, b) for a, b in loops if not any((c > a or d < b) and c >= a and d <= b for c, d in loops if (c, d) != (a, b))]
        print(name[:110])
        for a, b in inner:
            seg = body[a:b + 1]
            if not any("v_mfma" in x for x in seg):
                continue
            cnt = collections.Counter()
            for x in seg:
                x = x.strip()
                if not x or x.startswith(";") or x.startswith(".") or x.endswith(":"):
                    continue
                cnt[classify(x.split()[0])] += 1
            tot = sum(cnt.values())
            print(f"  loop @{a}-{b}: {tot} instrs", dict(cnt.most_common()),
                  "| non-MFMA per MFMA: %.2f" % ((tot - cnt["mfma"]) / max(1, cnt["mfma"])))
        regs = [l.strip() for l in body if re.search(r"\.(vgpr_count|sgpr_count|agpr_count)|NumVgprs|NumAgprs|ScratchSize|Occupancy", l)]
        for r in regs[:8]:
            print("  ", r)


if __name__ == "__main__":
    main()
