#!/usr/bin/env python3
"""Basic-block census of one kernel in hipcc's -S output: instructions inside inline-asm
regions (the hand-written decision bodies) versus compiler-generated glue around them.

usage: asm_blocks.py k.s decode_fast_kernel [-v]
"""
import re, sys

def main():
    path, key = sys.argv[1], sys.argv[2]
    verbose = "-v" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % key, l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks = []  # [label, asm_instrs, glue_instrs, text]
    cur = ["entry", 0, 0, []]
    in_asm = False
    for l in lines[start + 1:end]:
        s = l.strip()
        if not s:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            blocks.append(cur)
            cur = [m.group(1), 0, 0, []]
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if s.startswith(";") or s.startswith("."):
            continue
        if in_asm:
            cur[1] += 1
        else:
            cur[2] += 1
        cur[3].append(("A " if in_asm else "  ") + s.split(";")[0].rstrip())
    blocks.append(cur)
    ta = sum(b[1] for b in blocks)
    tg = sum(b[2] for b in blocks)
    print(f"{len(blocks)} blocks, {ta} asm instrs, {tg} glue instrs")
    for b in blocks:
        print(f"{b[0]:12s} asm {b[1]:4d} glue {b[2]:4d}")
        if verbose:
            for t in b[3]:
                print("      " + t)

main()
