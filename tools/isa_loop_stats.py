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
        best = None
        for i, l in enumerate(body):
            m = re.match(r"^\s+s_cbranch_\S+\s+(\.LBB\S+)", l) or re.match(r"^\s+s_branch\s+(\.LBB\S+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                seg = body[labels[m.group(1)]:i + 1]
                n = sum("v_mfma" in s for s in seg)
                if best is None or n > best[0]:
                    best = (n, seg)
        if best is None:
            continue
        cnt = collections.Counter()
        for s in best[1]:
            s = s.strip()
            if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
                continue
            cnt[classify(s.split()[0])] += 1
        regs = [l.strip() for l in body if re.search(r"\.(vgpr_count|sgpr_count|agpr_count)|NumVgprs|NumAgprs|ScratchSize|Occupancy", l)]
        print(name[:110])
        tot = sum(cnt.values())
        print("  loop instrs:", tot, dict(cnt.most_common()))
        if cnt["mfma"]:
            print("  non-MFMA per MFMA: %.2f" % ((tot - cnt["mfma"]) / cnt["mfma"]))
        for r in regs[:8]:
            print("  ", r)


if __name__ == "__main__":
    main()
