#!/usr/bin/env python3
"""Assembles profiles/r06_bench_matrix.md from what experiments/gpu_calls/r6_final.sh left under gpurun_out/r6_final (one MI355X)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
O = os.path.join(ROOT, "gpurun_out", "r6_final")
O2 = os.path.join(ROOT, "gpurun_out", "r6_final2")   # the last host-side changes (same kernel sources): the default line, push mode, whole-file calls


def path_of(name):
    p2 = os.path.join(O2, name)
    return p2 if os.path.exists(p2) else os.path.join(O, name)


def last_json(name):
    p = path_of(name)
    if not os.path.exists(p):
        return None
    lines = [l for l in open(p) if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def main():
    out = []
    d = last_json("bench_default.json")
    out.append("# Round 6 bench matrix (one MI355X; lc3/lp0/pb2, known-size headers, 4096 x 1 MiB per launch unless noted)\n")
    out.append("Every line: all units OK and the CRC-32 of every unit's output (computed on the GPU) equal to `zlib.crc32` of the regenerated plaintext.")
    out.append("GB/s = decompressed output.  Source: `experiments/gpu_calls/r6_final.sh` and `r6_final2.sh` (`gpurun_out/r6_final*/*`; the JSON lines are kept as `profiles/r06_bench_*.json`).")
    out.append("Kernel source hash of every line: `%s`.\n" % (d["roofline"]["kernel_source_sha256"] if d else "?"))
    out.append("| workload | command | GB/s | kernel ms | note |")
    out.append("|---|---|---|---|---|")
    if d:
        ri = d.get("roofline_issue") or {}
        pmc = ri.get("pmc") or {}
        out.append("| configs[1]: 4096 x 1 MiB `.lzma`, dict 64 KiB, text | `python bench.py` | **%.2f** | %.1f | headline; HBM roofline frac %.5f, PMC traffic %s GB per launch; issue pipes priced: vector %.2f, scalar + branch %.2f; from the counters: vector %s, scalar %s, %s branches per cycle per CU |"
                   % (d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"],
                      ("%.1f" % (d["roofline"]["traffic"] / 1e9)) if d["roofline"].get("traffic") else "null",
                      ri.get("valu_busy", 0), ri.get("salu_plus_branch_busy", 0), pmc.get("valu_busy"), pmc.get("salu_busy"), pmc.get("branch_per_cycle_per_cu")))
        oc = d.get("other_configs") or {}
        for k, label in (("dict8m", "configs[2]: same, dict 8 MiB"), ("xz", "configs[3]: 1024 x 4 MiB `.xz` (4 blocks each, CRC64)"), ("unknown_size", "configs[1] with liblzma's own headers (no size, end marker)")):
            c = oc.get(k)
            if c:
                out.append("| %s | the same run -> `other_configs.%s` | **%.2f** | %.1f | %s |"
                           % (label, k, c["value"], c["kernel_ms"],
                              ("every guess wrong: %.2f GB/s; arriving in four views: %.2f" % (c["every_guess_wrong"]["value"], c["fed_in_four_views"]["value"]))
                              if k == "unknown_size" and "every_guess_wrong" in c else ""))
        cb = d.get("cpu_baseline") or {}
        if cb:
            out.append("| CPU: the C restatement of the reference, %d threads / 1 thread; liblzma on the same sample | `cpu_baseline` | %.3f / %.3f; %.3f | | %s |"
                       % (cb["cores"], cb["value"], cb["one_thread"]["value"], cb["liblzma"]["value"], cb["liblzma"]["note"].split(";")[-1].strip()))
    for kind, what in (("random", "SURVEY 8d worst case: incompressible data (every symbol a literal: 8+ decisions per byte)"),
                       ("repeat", "SURVEY 8d best case: one long repetition (273-byte matches: the copy path, no decisions to speak of)"),
                       ("zeros", "all zeros (rep0 matches of length 273)")):
        k = last_json("bench_%s.json" % kind)
        if k:
            out.append("| class `%s`, dict 64 KiB | `python bench.py --kind %s` | %.2f | %.1f | %s; compressed %.1f MB per launch |"
                       % (kind, kind, k["value"], k["roofline"]["kernel_ms"], what, k["config"]["compressed_bytes_per_gpu"] / 1e6))
    k = last_json("bench_distinct0.json")
    if k:
        out.append("| configs[1], 4096 DISTINCT streams (no tiling) | `python bench.py --distinct 0` | %.2f | %.1f | the headline does not depend on the 512-distinct tiling |"
                   % (k["value"], k["roofline"]["kernel_ms"]))
    p = os.path.join(O, "lclp_classes.txt")
    if os.path.exists(p):
        rows = [json.loads(l) for l in open(p) if l.startswith("{")]
        base = next((r for r in rows if r["props"] == "lc3/lp0/pb2"), None)
        for r in rows:
            variant = "LP0" if r["props"] in ("lc3/lp0/pb2", "lc0/lp0/pb0") else "PB4" if r["props"] == "lc3/lp0/pb4" else \
                "HB0" if re.match(r"lc[4-8]/lp0/pb[0-2]", r["props"]) else "HBM"
            out.append("| property class %s (greedy-parse streams of `tests/lzma_enc.py`, ratio %.2f) | `experiments/lclp_bench.py` | %.2f | %.1f | loop variant %s%s |"
                       % (r["props"], r["compressed_ratio"], r["GBps"], r["kernel_ms"], variant,
                          (": %+.1f %% of lc3/lp0/pb2 on the same recipe" % (100.0 * (r["kernel_ms"] / base["kernel_ms"] - 1))) if base and r is not base else ""))
    s = last_json("streams_bench.json")
    if s:
        out.append("| push mode: 4096 `.lzma` streams arriving in four pieces each from host memory, result buffers handed over (PCIe both ways) | `experiments/streams_bench.py` | %.2f | | wall %.3f s = writes %.3f + finish %.3f (round 5: 8.7 GB/s, finish 0.125 s; 12.2 .. 12.7 by box); `profiles/r06_streams.txt` |"
                   % (s["GBps"], s["seconds"], s["writes_s"], s["finish_s"]))
    for name, label in (("batch_api_lzma.txt", "one whole-file call, 4096 x 1 MiB `.lzma`, host buffers in, per-file result buffers out"),
                        ("batch_api_xz.txt", "one whole-file call, 1024 x 4 MiB `.xz`")):
        p = path_of(name)
        if os.path.exists(p):
            best = 0.0
            kms = ""
            for l in open(p):
                m = re.search(r"one call, groups over lanes.*= ([0-9.]+) GB/s .*kernel ([0-9.]+) ms", l)
                if m and float(m.group(1)) > best:
                    best, kms = float(m.group(1)), m.group(2)
            if best:
                out.append("| %s | `experiments/batch_api_bench.py` | %.2f | %s | streamed launch (the waves deliver while they decode; input in two parts); PCIe-inclusive, never `value`; round 5: 16.4 / 15.7 (`profiles/r06_batch_api.txt`) |" % (label, best, kms))
    text = "\n".join(out) + "\n"
    with open(os.path.join(ROOT, "profiles", "r06_bench_matrix.md"), "w") as f:
        f.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
