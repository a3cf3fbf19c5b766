#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of one bench workload into profiles/<round>_pmc_<config>.json (what bench.py's
roofline.traffic reads, keyed by the kernel source hash).

    python tools/make_pmc_profile.py <config> <dir with pass_*/**/*counter_collection.csv> <out.json>

FETCH_SIZE is doubled (128-byte requests tallied at 64: MI355X_MICROARCH.md, re-calibrated on this kernel's byte loads with
experiments/pmc_calib.hip in round 2: profiles/r02_pmc_lzma64k.json `calibration`); WRITE_SIZE is exact for byte stores."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    config, src, out = sys.argv[1:4]
    # unknown_size (bench.py --unknown-size): the growable-output launch is the time-sliced kernel; the FIRST dispatch of a pass is the
    # timed step (what follows -- the verification step, the every-guess-wrong sequence with its RESUME launches -- is not the figure)
    first_only = config == "unknown_size"
    key = "decode_fast_asm_sliced_kernel" if first_only else "decode_fast_asm_kernel"
    counters, dur, passes = {}, [], {}
    newest = {}          # one CSV per pass directory: the most recent (a repeated pass leaves the older attempt's file behind)
    for path in glob.glob(os.path.join(src, "pass_*", "**", "*counter_collection.csv"), recursive=True):
        name = path[len(src):].strip("/").split("/")[0]
        if name not in newest or os.path.getmtime(path) > os.path.getmtime(newest[name]):
            newest[name] = path
    for path in sorted(newest.values()):
        per = defaultdict(lambda: defaultdict(float))
        ms_of = {}
        for r in csv.DictReader(open(path)):
            if key not in r["Kernel_Name"]:
                continue
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            ms_of[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if not per:
            continue
        if first_only:
            first = min(per, key=lambda d: int(d))
            per = {first: per[first]}
        dur.extend(ms_of[d] for d in per)
        name = path[len(src):].strip("/").split("/")[0]
        passes[name] = len(per)
        for c in sorted({c for d in per.values() for c in d}):
            counters[c] = sum(d.get(c, 0.0) for d in per.values()) / len(per)
    cfg = bench.CONFIGS["lzma64k" if config == "unknown_size" else config]
    out_bytes = cfg["streams"] * cfg["size"]
    derived = {}
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        fetch = counters["FETCH_SIZE"] * 1024 * 2
        write = counters["WRITE_SIZE"] * 1024
        derived = {"fetch_bytes_per_launch_corrected": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
                   "output_bytes_per_launch": out_bytes,
                   "traffic_note": "rocprofv3 FETCH_SIZE x 2 (128-B requests tallied at 64 B; calibrated on byte loads, "
                                   "profiles/r02_pmc_lzma64k.json) + WRITE_SIZE (exact for byte stores), separate --pmc passes of this "
                                   "kernel source on the bench's own batch (%d distinct streams tiled over %d).  Writes = the output, "
                                   "once; reads = one 128-B line of the stream's own LZ77 window per match from beyond the XCD's "
                                   "4 MiB L2 plus the compressed input: byte-granular dictionary reads, ~3 %% of the HBM roofline"
                                   % (cfg["distinct"], cfg["streams"])}
    for k in ("SQ_INSTS_SALU", "SQ_INSTS_VALU", "SQ_INSTS_BRANCH"):
        if k in counters:
            derived[k.lower() + "_per_output_byte"] = counters[k] / out_bytes
    # where a wave's cycles go (MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, all in quad-cycles):
    # WAIT_ANY = parked in s_waitcnt, WAIT_INST_ANY = an instruction is ready but cannot issue (dependency / pipe / arbitration),
    # ACTIVE_INST_* = issuing an instruction of that kind
    if "SQ_WAVE_CYCLES" in counters:
        wc = counters["SQ_WAVE_CYCLES"]
        derived["wave_cycle_shares"] = {k[3:].lower(): round(counters[k] / wc, 4) for k in
                                        ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_SCA",
                                         "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MISC")
                                        if k in counters}
    # the issue pipes as the counters see them (the priced model of bench.py's roofline_issue beside it):
    #   cycles of the launch = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs), 1024 SIMDs, 256 CUs
    #   vector pipe: SQ_ACTIVE_INST_VALU counts issue quads (4 cycles) summed over all SIMDs
    #   scalar pipe: SQ_INST_CYCLES_SALU = cycles the CU's scalar unit spent on scalar ALU instructions, summed over all CUs
    if all(k in counters for k in ("GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_INST_CYCLES_SALU", "SQ_INSTS_BRANCH")):
        cyc = counters["GRBM_GUI_ACTIVE"] / 8.0
        derived["pipe_busy_from_counters"] = {
            "launch_cycles": cyc,
            "valu_busy": round(counters["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc), 4),
            "salu_busy": round(counters["SQ_INST_CYCLES_SALU"] / (256.0 * cyc), 4),
            "branch_per_cycle_per_cu": round(counters["SQ_INSTS_BRANCH"] / (256.0 * cyc), 4),
            "how": "valu_busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x cycles); salu_busy = SQ_INST_CYCLES_SALU / (256 CUs x cycles); "
                   "branch_per_cycle_per_cu = SQ_INSTS_BRANCH / (256 x cycles); cycles = GRBM_GUI_ACTIVE / 8 XCDs"}
    with open(out, "w") as f:
        json.dump({"kernel_source_sha256": bench.kernel_source_hash(), "config": config,
                   "command": "rocprofv3 --pmc <one counter set per pass> --kernel-trace --output-format csv -- python bench.py %s "
                              "--steps 1 --warmup 0 --no-cpu-baseline --no-verify --other-configs none "
                              "(experiments/gpu_calls/r6_evidence.sh)" % ("--unknown-size" if first_only else "--config " + config),
                   "dispatches_per_pass": passes, "kernel_ms_under_pmc": round(sum(dur) / max(1, len(dur)), 2),
                   "counters_per_launch": counters, "derived": derived}, f, indent=1)
        f.write("\n")
    print(open(out).read()[:1500])


if __name__ == "__main__":
    main()
