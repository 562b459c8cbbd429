#!/usr/bin/env python3
"""ISA lint: wide VMEM stores whose data registers are overwritten by the very next vector-ALU instructions.

gfx950 reads the data of a buffer_store_dwordx3 / x4 over several cycles; a VALU instruction that overwrites one of those registers
needs wait states behind the store.  hipcc 7.2 inserts them when the store's scalar offset is an immediate / `off`, but NOT when it is
an SGPR (seen twice in this library: seam_pwpc.hip round 5 -- wrong fourth channels -- and seam_pwh.hip round 6 -- NaNs).  This
script compiles every csrc/*.hip to assembly and reports each `buffer_store_dwordx{3,4} v[a:b], ..., sN offen` that is followed
within WINDOW instructions (s_nop k counts k + 1) by a VALU write to v[a:b].  Exit code 1 if any is found.
usage: isa_store_hazard.py [file.hip ...]"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "seam-match-rcnn_amd", "csrc")
WINDOW = 2
files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
store = re.compile(r"^\s*buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(s\d+|m0)\b")
dst = re.compile(r"^\s*(v_\w+)\s+(v\[(\d+):(\d+)\]|v(\d+))")
bad = 0


def compile_asm(f):
    return subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", f"-I{ROOT}/include", f"-I{CSRC}", "-S", "--cuda-device-only",
                           f, "-o", "-"], capture_output=True, text=True)


from concurrent.futures import ThreadPoolExecutor      # noqa: E402
with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as pool:
    asms = list(pool.map(compile_asm, files))
for f, asm in zip(files, asms):
    if asm.returncode:
        print(f"{f}: compile failed\n{asm.stderr[-2000:]}")
        bad += 1
        continue
    lines = [l for l in asm.stdout.splitlines() if l.strip() and not l.strip().startswith((";", "."))]
    func = "?"
    n_sgpr = 0
    for i, l in enumerate(lines):
        if re.match(r"^[_A-Za-z][\w$]*:", l):
            func = l.split(":")[0]
        m = store.match(l)
        if not m:
            continue
        n_sgpr += 1
        lo, hi = int(m.group(1)), int(m.group(2))
        slots = 0
        for nxt in lines[i + 1:i + 1 + 8]:
            t = nxt.strip()
            if t.startswith("s_nop"):
                slots += int(t.split()[1]) + 1
                continue
            if slots >= WINDOW:
                break
            d = dst.match(nxt)
            if d and not t.startswith(("v_cmp", "v_cmpx")):
                a, b = (int(d.group(3)), int(d.group(4))) if d.group(3) else (int(d.group(5)), int(d.group(5)))
                if a <= hi and b >= lo:
                    print(f"{os.path.basename(f)}: {func[:60]}: `{l.strip()}` followed by `{t}`")
                    bad += 1
                    break
            slots += 1
    print(f"{os.path.basename(f)}: {n_sgpr} wide stores with an SGPR offset checked")
sys.exit(1 if bad else 0)
