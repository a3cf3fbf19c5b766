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
sys.path.insert(0, os.path.join(ROOT, "tools"))
import asmprog  # noqa: E402
from lzma_rs_amd import workloads as W  # noqa: E402


ROLES = ("core", "update", "normtest", "normstub", "book")
CLASSES = ("salu", "valu", "branch", "misc", "lds", "vmem")


def sections_table(emu, n, a, executed):
    """Where the executed instructions sit: loop section (rows) x role in a decision (columns) and x instruction class.
    core = the instructions of a decision itself (bound, compare, select, symbol), update = probability updates, normtest = the
    `range < 2^24` compare + branch after every decision, normstub = the normalisation proper (one per input byte), book = the rest
    (state, indices, guards, row swaps, stores, copies, jumps)."""
    import json
    c, tk = emu.counts()
    tab = {}
    for i, text in enumerate(emu.prog.text):
        sec, role = emu.prog.tag[i]
        d = tab.setdefault(sec or "?", {})
        k = asmprog.classify(text)
        d[role] = d.get(role, 0) + int(c[i])
        d["_" + k] = d.get("_" + k, 0) + int(c[i])
        d["_taken"] = d.get("_taken", 0) + int(tk[i])
    rows = sorted(tab.items(), key=lambda kv: -sum(v for k, v in kv[1].items() if not k.startswith("_")))
    hdr = "%-18s %8s | " % ("section", "instr/B") + " ".join("%8s" % r for r in ROLES) + " | " + " ".join("%7s" % k for k in CLASSES) + " %7s" % "taken"
    print(hdr)
    out_rows = {}
    tot = {}
    for sec, d in rows:
        total = sum(d.get(r, 0) for r in ROLES)
        if not total:
            continue
        print("%-18s %8.3f | " % (sec, total / n) + " ".join("%8.3f" % (d.get(r, 0) / n) for r in ROLES) + " | " +
              " ".join("%7.3f" % (d.get("_" + k, 0) / n) for k in CLASSES) + " %7.3f" % (d.get("_taken", 0) / n))
        out_rows[sec] = {"total": round(total / n, 4), "by_role": {r: round(d.get(r, 0) / n, 4) for r in ROLES},
                         "by_class": {k: round(d.get("_" + k, 0) / n, 4) for k in CLASSES}, "taken_branches": round(d.get("_taken", 0) / n, 4)}
        for k, v in d.items():
            tot[k] = tot.get(k, 0) + v
    total = sum(tot.get(r, 0) for r in ROLES)
    print("%-18s %8.3f | " % ("all", total / n) + " ".join("%8.3f" % (tot.get(r, 0) / n) for r in ROLES) + " | " +
          " ".join("%7.3f" % (tot.get("_" + k, 0) / n) for k in CLASSES) + " %7.3f" % (tot.get("_taken", 0) / n))
    import bench
    with open(a.sections, "w") as f:
        json.dump({"kernel_source_sha256": bench.kernel_source_hash(),
                   "workload": "%s, %d streams of %d B (indices %d..%d of bench.py's recipe), dict %d"
                               % (a.kind, a.streams, a.size, a.index, a.index + a.streams - 1, a.dict),
                   "how": "tools/emu/profile.py --sections: the generated symbol loop executed instruction by instruction on the CPU (bit-exact "
                          "output); every executed instruction attributed to the loop section and the role the generator emitted it for "
                          "(tools/gen_fast_loop.py: Gen.sec / @role); instructions per output byte",
                   "roles": {"core": "the decision itself (bound, compare, select, symbol bit)", "update": "probability updates",
                             "normtest": "range < 2^24 compare + branch after every decision", "normstub": "normalisation proper (one per input byte)",
                             "book": "state, indices, guards, row swaps, stores, copies, jumps"},
                   "sections": out_rows,
                   "all": {"total": round(total / n, 4), "by_role": {r: round(tot.get(r, 0) / n, 4) for r in ROLES},
                           "by_class": {k: round(tot.get("_" + k, 0) / n, 4) for k in CLASSES}}}, f, indent=1)
        f.write("\n")


# What an instruction costs a wave at four waves per SIMD (cycles), measured: experiments/microbench/order_slots.hip and
# shadow_slots.hip (profiles/r04_kernel_ab.txt section 3, profiles/r03_shadow_slots_and_spec_walk.txt).  A price list, not a simulator: it
# ranks the loop's sections by what they cost rather than by how many instructions they issue.
PRICE = {"v_readlane": 17.3, "valu": 7.0, "salu": 5.0, "cmp_branch": 13.7, "branch": 4.0, "taken": 6.0, "lds": 8.0, "vmem": 8.0, "misc": 2.0}


def cost_table(emu, n):
    """estimated cycles per output byte by section: every executed instruction at the measured price of its kind (a scalar compare whose
    next instruction is a conditional branch: the pair's price)"""
    c, tk = emu.counts()
    text = emu.prog.text
    tab = {}
    for i, t in enumerate(text):
        if not c[i]:
            continue
        sec = emu.prog.tag[i][0] or "?"
        op = t.split()[0]
        k = asmprog.classify(t)
        if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
            w = PRICE["v_readlane"]
        elif op.startswith("s_cmp") and i + 1 < len(text) and text[i + 1].split()[0].startswith("s_cbranch"):
            w = PRICE["cmp_branch"] - PRICE["branch"]
        elif k == "branch":
            w = PRICE["branch"]
        else:
            w = PRICE.get(k, 5.0)
        d = tab.setdefault(sec, [0.0, 0.0])
        d[0] += w * int(c[i]) + PRICE["taken"] * int(tk[i])
        d[1] += int(c[i])
    total = sum(v[0] for v in tab.values())
    print("%-18s %10s %8s %8s" % ("section", "cycles/B", "share", "instr/B"))
    for sec, v in sorted(tab.items(), key=lambda kv: -kv[1][0]):
        print("%-18s %10.1f %7.1f%% %8.2f" % (sec, v[0] / n, 100.0 * v[0] / total, v[1] / n))
    print("%-18s %10.1f" % ("all", total / n))


def form_of(text):
    """opcode + the operand kinds that decide which issue class a vector instruction falls in: s = a scalar register among its sources,
    l = a 32-bit literal (an integer outside the inline range -16 .. 64)"""
    import re
    t = text.replace(",", " ").split()
    op, ops = t[0], t[1:]
    if not op.startswith("v_"):
        return op
    flags = ""
    srcs = ops[1:]
    import gen_fast_loop as G
    scalar_names = set(G.OPS_INOUT_S) | set(G.OPS_IN_S)
    srcs = ["s0" if (o.startswith("%[") and o[2:-1] in scalar_names) else o for o in srcs]   # (asm operands the compiler places: scalar ones count as SGPRs)
    if op.startswith("v_readlane") and srcs[-1] == "m0":
        flags += "m0"        # (the lane select in m0 does not go through the SGPR file: priced like a constant lane)
    elif any(re.match(r"^(s\d+|s\[\d+:\d+\]|vcc|m0|exec)$", o) for o in srcs):
        flags += "s"
    for o in srcs:
        try:
            v = int(o, 0)
        except ValueError:
            continue
        if v < -16 or v > 64:
            flags += "l"
            break
    return op + ("/" + flags if flags else "")


def pipes_table(emu, n, prices_path, cycles_per_byte, quiet=False):
    """The loop's executed instructions by opcode form (per output byte), and -- with the measured issue cost of each form (pipe cycles per
    wave64 instruction at the pipe's peak: experiments/microbench/pipe_peaks.hip, profiles/r05_pipe_peaks.txt) -- the cycles per output byte
    each issue pipe of a SIMD is busy for ONE wave; four waves share a SIMD, so utilisation = 4 x that / the cycles a byte takes a wave."""
    import json
    c, tk = emu.counts()
    hist = {}
    for i, text in enumerate(emu.prog.text):
        if c[i]:
            f = form_of(text)
            hist[f] = hist.get(f, 0) + int(c[i])
    prices = json.load(open(prices_path)) if prices_path and os.path.exists(prices_path) else None
    pipe = {"valu": 0.0, "salu": 0.0, "branch": 0.0}
    unknown = {}
    if quiet:
        import io
        import contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            return pipes_table(emu, n, prices_path, cycles_per_byte)
    print("%-34s %9s %9s" % ("form", "instr/B", "cycles/B"))
    for f, k in sorted(hist.items(), key=lambda kv: -kv[1]):
        cyc = None
        if prices:
            cls = "valu" if f.startswith("v_") else "branch" if f.startswith(("s_cbranch", "s_branch", "s_setpc", "s_call")) else "salu" if f.startswith("s_") else None
            pr = prices["forms"].get(f, prices["forms"].get(f.split("/")[0]))
            if cls and pr is None:
                pr = prices["default"][cls]
                if cls == "valu":
                    unknown[f] = k / n
            if cls:
                cyc = pr * k / n
                pipe[cls] += cyc
        if k / n >= 0.02:
            print("%-34s %9.3f %9s" % (f, k / n, "%.2f" % cyc if cyc is not None else ""))
    if prices:
        taken = sum(int(x) for x in tk) / n
        pipe["branch"] += taken * prices.get("taken_extra", 0.0)
        print("pipe cycles per output byte for one wave: " + "  ".join("%s %.1f" % kv for kv in pipe.items()))
        if cycles_per_byte:
            print("utilisation at four waves per SIMD and %.0f cycles per byte per wave: " % cycles_per_byte +
                  "  ".join("%s %.0f %%" % (k, 400.0 * v / cycles_per_byte) for k, v in pipe.items()))
        if unknown:
            print("(priced at the class default: " + ", ".join("%s %.3f" % kv for kv in sorted(unknown.items(), key=lambda kv: -kv[1])) + ")")
    return hist, pipe


def section_pipes_table(emu, n, prices_path, out_path, cycles_per_byte, a):
    """per loop section and role, the cycles per output byte each ISSUE PIPE is busy for one wave -- every executed
    instruction at the measured issue cost of its form (profiles/r05_pipe_prices.json), scalar ALU, branch unit (a taken branch costs
    `taken_extra` more) and vector ALU apart.  The scalar column (salu + branch) is the work list while the scalar pipe binds."""
    import json
    prices = json.load(open(prices_path))
    c, tk = emu.counts()
    tab, instr = {}, {}
    for i, text in enumerate(emu.prog.text):
        if not c[i]:
            continue
        f = form_of(text)
        cls = "valu" if f.startswith("v_") else "branch" if f.startswith(("s_cbranch", "s_branch", "s_setpc", "s_call")) else "salu" if f.startswith("s_") else None
        if cls is None:
            continue
        pr = prices["forms"].get(f, prices["forms"].get(f.split("/")[0]))
        if pr is None:
            pr = prices["default"][cls]
        sec, role = emu.prog.tag[i]
        key = (sec or "?", role or "book")
        d = tab.setdefault(key, {"salu": 0.0, "branch": 0.0, "valu": 0.0})
        d[cls] += pr * int(c[i]) + (prices.get("taken_extra", 0.0) * int(tk[i]) if cls == "branch" else 0.0)
        e = instr.setdefault(key, {"salu": 0, "branch": 0, "valu": 0, "taken": 0})
        e[cls] += int(c[i])
        e["taken"] += int(tk[i])
    secs = {}
    for (sec, role), d in tab.items():
        s_ = secs.setdefault(sec, {"salu": 0.0, "branch": 0.0, "valu": 0.0})
        for k in d:
            s_[k] += d[k]
    print("%-18s %-9s | %9s %9s %9s | %8s %8s %8s %8s" % ("section", "role", "salu c/B", "branch c/B", "valu c/B", "salu /B", "branch/B", "taken/B", "valu /B"))
    rows = {}
    for sec, s_ in sorted(secs.items(), key=lambda kv: -(kv[1]["salu"] + kv[1]["branch"])):
        print("%-18s %-9s | %9.2f %9.2f %9.2f |" % (sec, "all", s_["salu"] / n, s_["branch"] / n, s_["valu"] / n))
        rows[sec] = {"pipe_cycles_per_byte": {k: round(v / n, 3) for k, v in s_.items()}, "roles": {}}
        for role in ROLES:
            d = tab.get((sec, role))
            if not d:
                continue
            e = instr[(sec, role)]
            print("%-18s %-9s | %9.2f %9.2f %9.2f | %8.3f %8.3f %8.3f %8.3f" % ("", role, d["salu"] / n, d["branch"] / n, d["valu"] / n, e["salu"] / n,
                                                                            e["branch"] / n, e["taken"] / n, e["valu"] / n))
            rows[sec]["roles"][role] = {"pipe_cycles_per_byte": {k: round(v / n, 3) for k, v in d.items()},
                                        "instructions_per_byte": {k: round(v / n, 4) for k, v in e.items()}}
    tot = {k: sum(s_[k] for s_ in secs.values()) / n for k in ("salu", "branch", "valu")}
    print("%-18s %-9s | %9.2f %9.2f %9.2f |" % ("all", "", tot["salu"], tot["branch"], tot["valu"]))
    if cycles_per_byte:
        print("at four waves per SIMD and %.0f cycles per byte per wave: scalar pipe (salu + branch) %.1f %% busy, vector pipe %.1f %%"
              % (cycles_per_byte, 400.0 * (tot["salu"] + tot["branch"]) / cycles_per_byte, 400.0 * tot["valu"] / cycles_per_byte))
    if out_path:
        import bench
        with open(out_path, "w") as f:
            json.dump({"kernel_source_sha256": bench.kernel_source_hash(),
                       "workload": "%s, %d streams of %d B (indices %d..%d of bench.py's recipe), dict %d"
                                   % (a.kind, a.streams, a.size, a.index, a.index + a.streams - 1, a.dict),
                       "how": "tools/emu/profile.py --section-pipes: the generated loop executed on the CPU emulator (bit-exact output); every "
                              "executed instruction attributed to its loop section and role and priced at the measured issue cost of its form "
                              "(profiles/r05_pipe_prices.json): cycles per output byte each issue pipe is busy for ONE wave",
                       "cycles_per_byte_per_wave": cycles_per_byte or None,
                       "all": {k: round(v, 3) for k, v in tot.items()}, "sections": rows}, f, indent=1)
            f.write("\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1 << 20)
    ap.add_argument("--kind", default="text")
    ap.add_argument("--dict", type=int, default=1 << 16)
    ap.add_argument("--regions", type=int, default=0)
    ap.add_argument("--index", type=int, default=0)
    ap.add_argument("--streams", type=int, default=1, help="average over this many streams (indices index .. index + streams - 1)")
    ap.add_argument("--sections", default="", help="write the per-section / per-role table (instructions per output byte) to this JSON file and print it")
    ap.add_argument("--section-pipes", default=None, nargs="?", const="", help="per section and role: cycles per byte each issue pipe is busy (optionally written to this JSON file)")
    ap.add_argument("--cost", action="store_true", help="estimated cycles per output byte by section, from the measured price of each kind of instruction")
    ap.add_argument("--pipes", action="store_true", help="executed instructions by opcode form; with --prices: pipe cycles per byte")
    ap.add_argument("--prices", default=os.path.join(ROOT, "profiles", "r05_pipe_prices.json"))
    ap.add_argument("--cycles-per-byte", type=float, default=0.0, help="measured cycles one output byte takes a wave (kernel ms x clock / stream size)")
    ap.add_argument("--config", default="", help="bench.py's recipe of that name (lzma64k, dict8m, xz: the .xz files' blocks -- LZMA2 units -- through "
                                                 "the front end's packet walk) instead of --kind / --size / --dict")
    ap.add_argument("--json", default="", help="also write the mix to this file (bench.py reads profiles/r03_instruction_mix_<config>.json)")
    a = ap.parse_args()
    emu = None
    tot = {}
    executed = comp_bytes = 0
    t = time.time()
    xz = a.config == "xz"
    if a.config:
        import bench
        cfg = bench.CONFIGS[a.config]
        a.kind, a.size, a.dict = "text", cfg["size"], cfg["dict"]
    for index in range(a.index, a.index + a.streams):
        if xz:   # one .xz file of the recipe: every block's LZMA2 payload is one decode unit (milzma_xz_plan, as bench.py plans them)
            import bench
            import lzma_rs_amd as M
            plain = bench.xz_plain(index, a.size)
            comp = W.compress_xz_blocks(plain, block_size=1 << 20, dict_size=a.dict, check="crc64")
            units, _ = M.xz_plan(comp)
            if emu is None:
                emu = asmprog.AsmLoop(lp0=True)
            got = b""
            for u in units:
                r = emu.decode_lzma2(comp[u.in_off:u.in_off + u.in_len], int(u.out_cap) + 300)
                assert r["status"] == "OK", r["status"]
                got += r["out"]
                executed += r["executed"]
                comp_bytes += int(u.in_len)
            assert got == plain
            continue
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
    pipes = None
    if a.pipes or a.json:
        _, pipes = pipes_table(emu, n, a.prices, a.cycles_per_byte, quiet=not a.pipes)
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
                       "pipe_cycles_per_output_byte": {k: round(v, 2) for k, v in (pipes or {}).items()},
                       "pipe_cycles_how": "every executed instruction at the measured issue cost of its form (cycles of its pipe one wave64 instruction "
                                          "occupies at the pipe's peak: profiles/r05_pipe_prices.json, experiments/microbench/pipe_peaks.hip); one wave's "
                                          "share -- four waves share a SIMD's vector pipe and its turn on the CU's scalar pipe",
                       "executed_instructions": executed, "compressed_bytes": comp_bytes}, f, indent=1)
            f.write("\n")
    if a.sections:
        sections_table(emu, n, a, executed)
    if a.cost:
        cost_table(emu, n)
    if a.section_pipes is not None:
        section_pipes_table(emu, n, a.prices, a.section_pipes, a.cycles_per_byte, a)

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
