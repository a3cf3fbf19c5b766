#!/usr/bin/env python3
"""Exact executed-instruction profile of the generated asm loop on one bench stream (CPU emulation).

    python3 tools/emu/profile.py [--size N] [--kind text] [--dict D] [--regions K]

Prints instructions per output byte by class (scalar ALU, vector ALU, branch, ...), taken branches per
byte, and the K heaviest code regions (a region = the instructions after one label)."""
import argparse
import os
import struct
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import asmprog  # noqa: E402
from lzma_rs_amd import workloads as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1 << 20)
    ap.add_argument("--kind", default="text")
    ap.add_argument("--dict", type=int, default=1 << 16)
    ap.add_argument("--regions", type=int, default=0)
    ap.add_argument("--index", type=int, default=0)
    ap.add_argument("--streams", type=int, default=1, help="average over this many streams (indices index .. index + streams - 1)")
    ap.add_argument("--json", default="", help="also write the mix to this file (bench.py reads profiles/r03_instruction_mix_<config>.json)")
    a = ap.parse_args()
    emu = None
    tot = {}
    executed = comp_bytes = 0
    t = time.time()
    for index in range(a.index, a.index + a.streams):
        plain = W.make_plain(a.kind, a.size, seed=W.SEED0 ^ index)
        comp = W.compress_alone(plain, dict_size=a.dict, known_size=True)
        props = comp[0]
        lc, lp, pb = props % 9, (props // 9) % 5, props // 45
        ds = struct.unpack("<I", comp[1:5])[0]
        us = struct.unpack("<Q", comp[5:13])[0]
        if emu is None:
            emu = asmprog.AsmLoop(lp0=(lp == 0))
        r = emu.decode_raw(comp[13:], lc, lp, pb, ds, us, out_cap=a.size)
        assert r["status"] == "OK" and r["out"] == plain, r["status"]
        executed += r["executed"]
        comp_bytes += len(comp) - 13
    mix = emu.mix()                      # (the counters accumulate over the streams)
    n = float(a.size) * a.streams
    print("%s %d x %d B, dict %d: compressed %d B, %d instructions executed (%.1f s), bit-exact"
          % (a.kind, a.streams, a.size, a.dict, comp_bytes, executed, time.time() - t))
    print("per output byte: " + "  ".join("%s %.2f" % (k, mix[k] / n) for k in ("salu", "valu", "branch", "misc", "lds", "vmem", "total", "taken") if k in mix))
    if a.json:
        import json
        import bench
        with open(a.json, "w") as f:
            json.dump({"kernel_source_sha256": bench.kernel_source_hash(),
                       "workload": "%s, %d streams of %d B (indices %d..%d of bench.py's recipe), dict %d"
                                   % (a.kind, a.streams, a.size, a.index, a.index + a.streams - 1, a.dict),
                       "how": "tools/emu/profile.py: the generated symbol loop executed instruction by instruction on the CPU "
                              "(bit-exact output), every executed instruction counted by class, averaged over the streams",
                       "per_output_byte": {k: round(mix.get(k, 0) / n, 4) for k in ("salu", "valu", "branch", "misc", "lds", "vmem", "total", "taken")},
                       "executed_instructions": executed, "compressed_bytes": comp_bytes}, f, indent=1)
            f.write("\n")
    if a.regions:
        c, tk = emu.counts()
        reg = {}
        for i, text in enumerate(emu.prog.text):
            k = emu.prog.region[i]
            d = reg.setdefault(k, dict(salu=0, valu=0, branch=0, misc=0, lds=0, vmem=0, other=0, total=0, taken=0))
            d[asmprog.classify(text)] += int(c[i])
            d["total"] += int(c[i])
            d["taken"] += int(tk[i])
        rows = sorted(reg.items(), key=lambda kv: -kv[1]["total"])[:a.regions]
        print("%-28s %9s %8s %8s %8s %8s" % ("region", "instr/B", "salu/B", "valu/B", "branch/B", "taken/B"))
        for k, d in rows:
            print("%-28s %9.3f %8.3f %8.3f %8.3f %8.3f" % (k.replace("%=", ""), d["total"] / n, d["salu"] / n, d["valu"] / n, d["branch"] / n, d["taken"] / n))


if __name__ == "__main__":
    main()
