#!/usr/bin/env python3
"""Tuning aid: times the decode kernel of several library builds on the same batch.

    python experiments/ab_bench.py [--steps K] [--streams N] [--distinct D] [--dict B] [--kind text] lib.so [lib.so ...]

The compressed batch is generated once and cached under /tmp; every library runs in its own
process (MILZMA_LIB), all outputs are CRC-checked on the GPU against the plaintext's CRC-32.
Prints one line per library: median kernel ms, GB/s decompressed.  --wavetime: the library was
built with -DMILZMA_WAVETIME (per-wave start / end clock and hardware ids in the results)."""
import argparse
import json
import os
import pickle
import subprocess
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def batch_path(a):
    return "/tmp/milzma_ab_%s_%d_%d_%d_%s.pkl" % (a.kind, a.distinct, a.size, a.dict, a.props.replace(",", ""))


def worker(a):
    import ctypes
    import torch
    import lzma_rs_amd as M
    from lzma_rs_amd import workloads as W
    import bench
    path = batch_path(a)
    bench.PROPS = tuple(int(x) for x in a.props.split(","))
    if os.path.exists(path):
        with open(path, "rb") as f:
            blob, unit_bytes, crcs = pickle.load(f)
        units_d = (M.Unit * a.distinct).from_buffer_copy(unit_bytes)
    else:
        units_d, blob, _, _ = bench.build_batch(a.distinct, a.size, a.kind, a.dict, 0, bench.effective_cores())
        crcs = [zlib.crc32(W.make_plain(a.kind, a.size, W.SEED0 ^ k)) for k in range(a.distinct)]
        with open(path, "wb") as f:
            pickle.dump((blob, bytes(units_d), crcs), f)
    n, distinct = a.streams, a.distinct
    dev = torch.device("cuda", 0)
    ctx = M.Context(0)
    units = (M.Unit * n)()
    reps = (n + distinct - 1) // distinct
    d_in = torch.frombuffer(bytearray(blob), dtype=torch.uint8).repeat(reps).to(dev)
    for k in range(n):
        src = units_d[k % distinct]
        u = M.Unit()
        ctypes.memmove(ctypes.byref(u), ctypes.byref(src), ctypes.sizeof(M.Unit))
        u.in_off = src.in_off + (k // distinct) * len(blob)
        u.out_off = k * a.size
        units[k] = u
    d_out = torch.zeros(n * a.size, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ms = []
    for i in range(a.steps + 1):
        res, t, _ = ctx.decode_units(units, d_in.data_ptr(), d_out.data_ptr(), stream)
        if i:
            ms.append(t)
    bad = sum(1 for r in res if r.status != 0 or r.out_len != a.size)
    if not a.wavetime:
        c32, _ = ctx.crc_units(units, res, d_out.data_ptr(), stream)
        bad += sum(1 for k in range(n) if c32[k] != crcs[k % distinct])
    ms.sort()
    med = ms[len(ms) // 2]
    out = {"lib": os.path.basename(os.environ.get("MILZMA_LIB", "default")), "ms": round(med, 2), "min_ms": round(ms[0], 2),
           "GBps": round(n * a.size / med / 1e6, 3), "bad": bad}
    if a.wavetime:
        import numpy as np
        st = np.array([r.err_a for r in res], dtype=np.float64)
        en = np.array([r.err_b for r in res], dtype=np.float64)
        ids = np.array([r.chunks for r in res], dtype=np.uint32)
        t0 = st.min()
        dur = (en - st) / 100.0  # us (100 MHz)
        out["wave_us"] = {"min": dur.min(), "mean": dur.mean(), "p50": float(np.median(dur)), "p99": float(np.percentile(dur, 99)),
                          "max": dur.max(), "last_end": (en.max() - t0) / 100.0, "mean_end": (en.mean() - t0) / 100.0,
                          "last_start": (st.max() - t0) / 100.0}
        cu = (ids >> 8) & 0xF
        sh = (ids >> 12) & 1
        se = (ids >> 13) & 7
        xcc = ids >> 24
        key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        cnt = np.bincount(key.astype(np.int64))
        cnt = cnt[cnt > 0]
        out["waves_per_cu"] = {"cus": int(len(cnt)), "min": int(cnt.min()), "max": int(cnt.max())}
        by = {}
        for k_, d_ in zip(key, dur):  # do slow waves cluster on CUs?
            by.setdefault(int(k_), []).append(d_)
        m = np.array([np.mean(v) for v in by.values()])
        out["cu_mean_us"] = {"min": m.min(), "max": m.max(), "std": m.std()}
        slot = ids & 0xF
        out["mean_us_by_wave_slot"] = {int(w): round(float(dur[slot == w].mean()), 0) for w in np.unique(slot)}
        simd = (ids >> 4) & 3
        out["mean_us_by_simd"] = {int(w): round(float(dur[simd == w].mean()), 0) for w in np.unique(simd)}
        comp = np.array([units[k].in_len for k in range(n)], dtype=np.float64)
        out["corr_dur_vs_compressed_len"] = float(np.corrcoef(dur, comp)[0, 1])
        out["compressed_len"] = {"min": comp.min(), "mean": comp.mean(), "max": comp.max()}
    print(json.dumps(out))
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--distinct", type=int, default=512)
    ap.add_argument("--size", type=int, default=1 << 20)
    ap.add_argument("--dict", type=int, default=1 << 16)
    ap.add_argument("--kind", default="text")
    ap.add_argument("--props", default="3,0,2")
    ap.add_argument("--wavetime", action="store_true")
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("libs", nargs="*")
    a = ap.parse_args()
    if a.worker:
        return worker(a)
    for lib in a.libs or [""]:
        env = dict(os.environ)
        if lib:
            env["MILZMA_LIB"] = os.path.abspath(lib)
        args = [sys.executable, os.path.abspath(__file__), "--worker"] + [x for x in sys.argv[1:] if not x.endswith(".so")]
        r = subprocess.run(args, env=env, capture_output=True, text=True)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED rc=%d %s" % (r.returncode, r.stderr[-400:])
        print(line, flush=True)


if __name__ == "__main__":
    main()
