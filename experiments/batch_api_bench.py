#!/usr/bin/env python3
"""The whole-file batch entry points as timing probes (not bench.py lines), host buffers in and out:
  xz   (default): configs[4] of BASELINE.json, N .xz files of 4 MiB (text | 200 KB random | text; 1 MiB blocks;
                  CRC64) through milzma_xz_decompress_batch
  lzma:           configs[1] through milzma_lzma_decompress_batch, N .lzma files of 1 MiB
After the two single-call runs: `calls` calls of the same batch with TWO in flight (two contexts, milzma_*_batch_async /
milzma_batch_wait): the copies and the hand-over of one call overlap the decode kernel of the other.
Usage: python experiments/batch_api_bench.py [files=1024] [distinct=32] [xz|lzma] [calls=2] [lzma file size=1048576]"""
import ctypes
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import lzma_rs_amd as M  # noqa: E402
from lzma_rs_amd import workloads as W  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
distinct = int(sys.argv[2]) if len(sys.argv) > 2 else 32
mode = sys.argv[3] if len(sys.argv) > 3 else "xz"
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 2
fsize = int(sys.argv[5]) if len(sys.argv) > 5 else 1 << 20


def one(i):
    if mode == "lzma":
        plain = W.make_plain("text", fsize, seed=100 + i)
        return W.compress_alone(plain, dict_size=65536, known_size=True), plain
    half = (4 * 1048576 - 200_000) // 2
    plain = W.make_plain("text", half, seed=100 + i) + W.make_plain("random", 200_000, seed=200 + i) + \
        W.make_plain("text", 4 * 1048576 - 200_000 - half, seed=300 + i)
    return W.compress_xz_blocks(plain, block_size=1 << 20, check="crc64"), plain


t0 = time.time()
with ThreadPoolExecutor(16) as ex:
    made = list(ex.map(one, range(distinct)))
print("generated %d distinct files in %.1f s" % (distinct, time.time() - t0))
files = [made[i % distinct][0] for i in range(n)]
ctx = M.Context(0)
lib = M.lib()
bufs = [M._as_buffer(d) for d in files]
ptrs = (ctypes.c_void_p * n)(*[b[0] for b in bufs])
lens = (ctypes.c_size_t * n)(*[b[1] for b in bufs])
def one_call(label):
    for rep in range(3):
        outs = (M._COutput * n)()
        t0 = time.time()
        if mode == "lzma":
            lib.milzma_lzma_decompress_batch(ctx._h, n, ptrs, lens, None, outs)
        else:
            lib.milzma_xz_decompress_batch(ctx._h, n, ptrs, lens, outs)
        dt = time.time() - t0
        total = sum(outs[i].len for i in range(n))
        ok = sum(1 for i in range(n) if outs[i].kind == 0)
        launches = ctypes.c_uint32(0)
        kms = lib.milzma_last_kernel_ms(ctx._h, ctypes.byref(launches))
        print("%s, run %d: %d files, %d ok, %.2f GiB out in %.3f s = %.2f GB/s (the context's last decode call: kernel %.1f ms, %d launches)" %
              (label, rep, n, ok, total / 2**30, dt, total / dt / 1e9, kms, launches.value))
        if os.environ.get("BATCH_VERIFY_ALL"):   # every file of every run against its plaintext's CRC-32 (16 host threads)
            import zlib
            want = [zlib.crc32(made[k][1]) for k in range(distinct)] if rep == 0 else want_crc
            globals()["want_crc"] = want

            def crc_of(k):
                return zlib.crc32((ctypes.c_char * outs[k].len).from_address(ctypes.cast(outs[k].data, ctypes.c_void_p).value)) if outs[k].len else 0
            with ThreadPoolExecutor(16) as ex:
                got = list(ex.map(crc_of, range(n)))
            bad = [k for k in range(n) if outs[k].kind != 0 or outs[k].len != len(made[k % distinct][1]) or got[k] != want[k % distinct]]
            print("   verified all %d files: %d bad%s" % (n, len(bad), (" (first: %d)" % bad[0]) if bad else ""))
            assert not bad
        elif rep == 0:
            for k in (0, n // 2, n - 1):
                assert ctypes.string_at(outs[k].data, outs[k].len) == made[k % distinct][1], "file %d differs from its plaintext" % k
        for i in range(n):
            if outs[i].data:
                lib.milzma_free(ctypes.cast(outs[i].data, ctypes.c_void_p))


os.environ["MILZMA_NO_GROUPS"] = "1"
one_call("one call, one launch (MILZMA_NO_GROUPS=1)")
os.environ.pop("MILZMA_NO_GROUPS")
one_call("one call, groups over lanes (default)")


# ---- `calls` calls, two in flight -------------------------------------------------------------------------------
ctxs = [ctx, M.Context(0)]
warm = (M._COutput * n)()          # the second context's first call allocates its pinned / device staging (~0.5 s): not timed
if mode == "lzma":
    lib.milzma_lzma_decompress_batch(ctxs[1]._h, n, ptrs, lens, None, warm)
else:
    lib.milzma_xz_decompress_batch(ctxs[1]._h, n, ptrs, lens, warm)
for i in range(n):
    if warm[i].data:
        lib.milzma_free(ctypes.cast(warm[i].data, ctypes.c_void_p))
for ncalls in [2] + sorted({2, calls}):   # (the first pair grows the output-buffer pool to two calls' worth: shown, not the figure)
    slots = [None, None]
    t0 = time.time()
    total = ok = 0

    def finish(k):
        global total, ok
        rc = lib.milzma_batch_wait(ctxs[k]._h)
        o = slots[k]
        assert rc == 0
        total += sum(o[i].len for i in range(n))
        ok += sum(1 for i in range(n) if o[i].kind == 0)
        for i in range(n):
            if o[i].data:
                lib.milzma_free(ctypes.cast(o[i].data, ctypes.c_void_p))
        slots[k] = None

    for c in range(ncalls):
        k = c % 2
        if slots[k] is not None:
            finish(k)
        slots[k] = (M._COutput * n)()
        if mode == "lzma":
            r = lib.milzma_lzma_decompress_batch_async(ctxs[k]._h, n, ptrs, lens, None, slots[k])
        else:
            r = lib.milzma_xz_decompress_batch_async(ctxs[k]._h, n, ptrs, lens, slots[k])
        assert r == 0
    for k in ((ncalls % 2), 1 - (ncalls % 2)):
        if slots[k] is not None:
            finish(k)
    dt = time.time() - t0
    print("%d calls x %d files, two in flight: %d ok, %.2f GiB out in %.3f s = %.2f GB/s" % (ncalls, n, ok, total / 2**30, dt, total / dt / 1e9))


# ---- ONE call with two chip-fulls of files: the library groups it over the context and its peer ----------------------
n2 = 2 * n
ptrs2 = (ctypes.c_void_p * n2)(*([b[0] for b in bufs] * 2))
lens2 = (ctypes.c_size_t * n2)(*([b[1] for b in bufs] * 2))
for label, env in (("grouped (default)", None), ("one launch (MILZMA_NO_GROUPS=1)", "1")):
    if env:
        os.environ["MILZMA_NO_GROUPS"] = env
    for rep in range(2):
        outs2 = (M._COutput * n2)()
        t0 = time.time()
        if mode == "lzma":
            lib.milzma_lzma_decompress_batch(ctx._h, n2, ptrs2, lens2, None, outs2)
        else:
            lib.milzma_xz_decompress_batch(ctx._h, n2, ptrs2, lens2, outs2)
        dt = time.time() - t0
        total = sum(outs2[i].len for i in range(n2))
        ok = sum(1 for i in range(n2) if outs2[i].kind == 0)
        for i in range(n2):
            if outs2[i].data:
                lib.milzma_free(ctypes.cast(outs2[i].data, ctypes.c_void_p))
    print("one call x %d files, %s: %d ok, %.2f GiB out in %.3f s = %.2f GB/s" % (n2, label, ok, total / 2**30, dt, total / dt / 1e9))
os.environ.pop("MILZMA_NO_GROUPS", None)
