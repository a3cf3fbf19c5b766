#!/usr/bin/env python3
"""Throughput of the property classes liblzma cannot write (lc + lp > 4: the generic kernel's HBM-spill class) next to
lc + lp = 4 and the headline class, on streams with real match structure.

    python experiments/lclp_bench.py [--streams 4096] [--size 262144] [--distinct 32] [--steps 3] lc,lp,pb [lc,lp,pb ...]

Streams are built with the tests' greedy LZ parser + symbol encoder (tests/lzma_enc.py) from bench.py's "text"
plaintext, `distinct` of them on the host cores, tiled over `streams` slots (every slot reads its own copy and
writes its own slice); every output is CRC-checked on the GPU.  One JSON line per property set."""
import argparse
import ctypes
import json
import multiprocessing
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _one(job):
    import lzma_enc as E
    from lzma_rs_amd import workloads as W
    lc, lp, pb, size, index = job
    plain = W.make_plain("text", size, W.SEED0 ^ index)
    enc = E.LzmaSymbolEncoder(lc, lp, pb)
    enc.encode(E.lz_parse(plain, dict_size=1 << 16))
    return E.lzma_header(lc, lp, pb, 1 << 16, len(plain)) + enc.finish(), zlib.crc32(plain)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--size", type=int, default=1 << 18)
    ap.add_argument("--distinct", type=int, default=32)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("props", nargs="+")
    a = ap.parse_args()
    import torch
    import lzma_rs_amd as M
    import bench
    for spec in a.props:
        lc, lp, pb = (int(x) for x in spec.split(","))
        with multiprocessing.get_context("fork").Pool(min(bench.effective_cores(), a.distinct)) as pool:
            made = pool.map(_one, [(lc, lp, pb, a.size, i) for i in range(a.distinct)], chunksize=1)
        blobs, units_d, off = [], [], 0
        for comp, _ in made:
            u, hl = M.lzma_read_header(comp)
            payload = comp[hl:]
            u.in_off, u.in_len, u.out_cap = off, len(payload), a.size
            units_d.append(u)
            pad = (-len(payload)) % 256
            blobs.append(payload + bytes(pad))
            off += len(payload) + pad
        blob = b"".join(blobs)
        n, d = a.streams, a.distinct
        units = (M.Unit * n)()
        comp_total = 0
        for k in range(n):
            u = M.Unit()
            ctypes.memmove(ctypes.byref(u), ctypes.byref(units_d[k % d]), ctypes.sizeof(M.Unit))
            u.in_off = units_d[k % d].in_off + (k // d) * len(blob)
            u.out_off = k * a.size
            units[k] = u
            comp_total += u.in_len
        dev = torch.device("cuda", 0)
        ctx = M.Context(0)
        d_in = torch.frombuffer(bytearray(blob), dtype=torch.uint8).repeat((n + d - 1) // d).to(dev)
        d_out = torch.zeros(n * a.size + 512, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        ms = []
        for i in range(a.steps + 1):
            res, t, launches = ctx.decode_units(units, d_in.data_ptr(), d_out.data_ptr(), stream)
            if i:
                ms.append(t)
        c32, _ = ctx.crc_units(units, res, d_out.data_ptr(), stream)
        bad = sum(1 for k in range(n) if res[k].status != 0 or res[k].out_len != a.size or c32[k] != made[k % d][1])
        ms.sort()
        med = ms[len(ms) // 2]
        print(json.dumps({"props": "lc%d/lp%d/pb%d" % (lc, lp, pb), "streams": n, "stream_bytes": a.size, "distinct": d,
                          "compressed_ratio": round(comp_total / (n * a.size), 3), "kernel_ms": round(med, 2), "launches": launches,
                          "GBps": round(n * a.size / med / 1e6, 3), "bad": bad}), flush=True)
        ctx.close()
        del d_in, d_out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
