#!/usr/bin/env python3
"""Cycle attribution of the fast kernel (tuning aid): runs a batch through libmilzma_prof.so
(`make -C lzma_rs_amd/csrc prof`), whose successful units report s_memtime deltas in err_a/err_b:
  err_a = literal decode cycles << 32 | match-copy issue cycles
  err_b = cycles stalled finishing a pending copy before a literal << 32 | whole-unit cycles
Usage: MILZMA_LIB=lzma_rs_amd/libmilzma_prof.so python experiments/prof_attrib.py [streams] [kind]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MILZMA_KERNEL", "fast")  # the attribution macros live in the C++ fast kernel
os.environ.setdefault("MILZMA_LIB", os.path.join(ROOT, "lzma_rs_amd", "libmilzma_prof.so"))
import torch  # noqa: E402
import lzma_rs_amd as M  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = sys.argv[2] if len(sys.argv) > 2 else "text"
distinct = 64
units_d, blob, comp_total, _ = bench.build_batch(distinct, 1 << 20, kind, 1 << 16, 0, bench.effective_cores())
ctx = M.Context(0)
import ctypes
units = (M.Unit * n)()
reps = (n + distinct - 1) // distinct
d_in = torch.frombuffer(bytearray(blob), dtype=torch.uint8).repeat(reps).cuda()
for k in range(n):
    src = units_d[k % distinct]
    u = M.Unit()
    ctypes.memmove(ctypes.byref(u), ctypes.byref(src), ctypes.sizeof(M.Unit))
    u.in_off = src.in_off + (k // distinct) * len(blob)
    u.out_off = k << 20
    units[k] = u
d_out = torch.empty(n << 20, dtype=torch.uint8, device="cuda")
for _ in range(2):
    res, ms, _ = ctx.decode_units(units, d_in.data_ptr(), d_out.data_ptr(), 0)
lit = sum(r.err_a >> 32 for r in res) / n
copy = sum(r.err_a & 0xFFFFFFFF for r in res) / n
dist = sum(r.err_b >> 32 for r in res) / n
total = 16 * sum(r.err_b & 0xFFFFFFFF for r in res) / n
lend = sum(r.out_flushed >> 32 for r in res) / n
print("kernel %.1f ms; per unit s_memtime ticks: total %.0f | literal decode %.0f (%.1f%%) | new-match len decode %.0f (%.1f%%) | "
      "distance decode %.0f (%.1f%%) | copy issue %.0f (%.1f%%)" % (ms, total, lit, 100 * lit / total, lend, 100 * lend / total,
                                                                    dist, 100 * dist / total, copy, 100 * copy / total))
