#!/usr/bin/env python3
"""Tuning aid: average s_waitcnt vmcnt(0) time at the two sites that complete a pending match (before a
matched literal / before the next match's copy).  Needs a library built from a MILZMA_GEN_WAITPROF=1
generator run with -DMILZMA_WAITPROF (see the Makefile comment); successful units then report
err_a = wait cycles << 32 | events (matched literal), err_b = the same for the copy site.
Usage: MILZMA_LIB=lzma_rs_amd/libmilzma_wp.so python experiments/wait_prof.py [streams]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import lzma_rs_amd as M  # noqa: E402
import bench  # noqa: E402
import ctypes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dict_size = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
distinct = 64
units_d, blob_d, comp_d, gen_s = bench.build_batch(distinct, 1 << 20, "text", dict_size, first_index=0, processes=bench.effective_cores())
ctx = M.Context(0)
dev = torch.device("cuda", 0)
units = (M.Unit * n)()
reps = (n + distinct - 1) // distinct
d_in = torch.frombuffer(bytearray(blob_d), dtype=torch.uint8).repeat(reps).to(dev)
for k in range(n):
    u = M.Unit()
    ctypes.memmove(ctypes.byref(u), ctypes.byref(units_d[k % distinct]), ctypes.sizeof(M.Unit))
    u.in_off = units_d[k % distinct].in_off + (k // distinct) * len(blob_d)
    u.out_off = k << 20
    units[k] = u
d_out = torch.empty(n << 20, dtype=torch.uint8, device=dev)
for _ in range(2):
    res, ms, _ = ctx.decode_units(units, d_in.data_ptr(), d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
wm = sum(r.err_a >> 32 for r in res) / n
nm = sum(r.err_a & 0xFFFFFFFF for r in res) / n
wc = sum(r.err_b >> 32 for r in res) / n
nc = sum(r.err_b & 0xFFFFFFFF for r in res) / n
print("kernel %.1f ms, %d streams, dict %d" % (ms, n, dict_size))
# s_memtime counts SHADER cycles on gfx950 (~2.4 GHz under this load), not 100 MHz ticks: round 1 read it as 10 ns ticks and
# overstated these waits 24-fold; profiles/r02_wait_attribution.txt carries that wrong unit in its labels.
CLOCK_HZ = 2.4e9
print("matched-literal site: %.0f events/stream, %.0f cycles each (s_memtime: shader cycles), %.2f ms per stream" % (nm, wm / max(nm, 1), wm / CLOCK_HZ * 1e3))
print("copy site:            %.0f events/stream, %.0f cycles each, %.2f ms per stream" % (nc, wc / max(nc, 1), wc / CLOCK_HZ * 1e3))
