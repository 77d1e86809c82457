#!/usr/bin/env python
"""The inline-assembly DPP FMACs of prep.hip (fmac_bcast) are invisible to the compiler's hazard recogniser: on gfx9 a DPP read of a
VGPR needs 2 wait states behind the VALU instruction that wrote it.  This compiles prep.hip to assembly and checks every
v_fmac_*_dpp: none of the two instruction slots before it may write its DPP source register (s_nop N counts as N + 1 slots).
usage: python tools/check_dpp_hazard.py   (exit code 1 and a listing when a hazard is found)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "approximategps.jl_amd", "csrc", "prep.hip")


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check(asm_text):
    bad, n = [], 0
    lines = [l.split(";")[0].strip() for l in asm_text.split("\n")]
    lines = [l for l in lines if l and not l.startswith(".") and not l.endswith(":")]
    for i, l in enumerate(lines):
        m = re.match(r"(v_fmac_f(?:32|64)_dpp|v_mov_b(?:32|64)_dpp)\s+(\S+),\s*(\S+?)(?:,|\s)", l)
        if not m:
            continue
        n += 1
        src = regs(m.group(3))
        slots, j = 0, i - 1
        while slots < 2 and j >= 0:
            p = lines[j]
            if p.startswith("s_nop"):
                slots += int(p.split()[1]) + 1
            else:
                slots += 1
                if p.startswith("v_") and not p.startswith("v_cmp"):
                    dst = regs(p.split(None, 1)[1].split(",")[0]) if len(p.split(None, 1)) > 1 else set()
                    if dst & src and slots <= 2:
                        bad.append((i, p, l))
            j -= 1
    return n, bad


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "prep.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", SRC, "-o", out],
                       check=True, stderr=subprocess.DEVNULL)
        n, bad = check(open(out).read())
    print(f"{n} DPP instructions checked, {len(bad)} with their source written inside the 2 wait states")
    for i, p, l in bad[:20]:
        print("  ", p, "->", l)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
