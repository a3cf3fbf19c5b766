#!/usr/bin/env python3
"""Tuning probe (needs a library built with -DMILZMA_SLICE_DEBUG): where the waves of a time-sliced launch sit and when their units end.
    MILZMA_LIB=lzma_rs_amd/variants/libmilzma_slicedbg.so MILZMA_SLICE=1 python experiments/slice_probe.py [streams=4096]"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import lzma_rs_amd as M  # noqa: E402
from lzma_rs_amd import workloads as W  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
plains = [W.make_plain("text", 1 << 20, seed=900 + i) for i in range(16)]
comps = [W.compress_alone(p, dict_size=65536, known_size=True) for p in plains]
units, blob, out_off = [], bytearray(), 0
for k in range(n):
    c = comps[k % 16]
    u, hl = M.lzma_read_header(c)
    u.in_off, u.in_len, u.out_off, u.out_cap = len(blob), len(c) - hl, out_off, 1 << 20
    blob += c[hl:] + b"\0" * (-(len(c) - hl) % 64)
    out_off += 1 << 20
    units.append(u)
arr = (M.Unit * n)(*units)
ctx = M.Context(0)
d_in = torch.frombuffer(bytearray(blob) + bytearray(512), dtype=torch.uint8).cuda()
d_out = torch.zeros(out_off + 512, dtype=torch.uint8, device="cuda")
for rep in range(2):
    res, kms, launches = ctx.decode_units(arr, d_in.data_ptr(), d_out.data_ptr())
torch.cuda.synchronize()
print("kernel ms", kms, "launches", launches)
per_simd = collections.defaultdict(list)
ends = []
for r in res:
    assert r.status == 0
    hw = r.chunks
    slot, simd, cu, sh, se, xcc = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, hw >> 24
    per_simd[(xcc, se, sh, cu, simd)].append(slot)
    ends.append(r.err_b)
print("SIMDs seen:", len(per_simd), " units per SIMD:", collections.Counter(len(v) for v in per_simd.values()))
print("slot sets (first 12):", [sorted(v) for v in list(per_simd.values())[:12]])
print("distinct (slot & 3) per SIMD:", collections.Counter(len({s & 3 for s in v}) for v in per_simd.values()))
e0 = min(ends)
ms = sorted((e - e0) / 1e5 for e in ends)   # s_memrealtime: 100 MHz
print("unit end times relative to the first (ms): p0 %.1f p10 %.1f p50 %.1f p90 %.1f p100 %.1f" % (ms[0], ms[len(ms) // 10], ms[len(ms) // 2], ms[len(ms) * 9 // 10], ms[-1]))
