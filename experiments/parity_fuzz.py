#!/usr/bin/env python3
"""Differential fuzz: the GPU library against the CPU oracle on thousands of damaged / odd inputs.

    python experiments/parity_fuzz.py [--seed S] [--rounds R] [--kernel asm|generic]

Each round builds a few hundred cases per entry point from a pool of valid streams (all data classes, many
property sets, dictionary sizes, known / unknown sizes) by: header edits (props, dictionary size, declared size:
smaller, larger, huge, marker), byte flips anywhere, truncation, trailing garbage, concatenation; LZMA2 streams and
XZ containers get the same treatment plus chunk-header / block-header / index / footer edits.  Every case is decoded
through the batch entry points and compared with the oracle: error kind, full message, bytes delivered to the
writer, reader position.  A mismatch prints the case (hex, up to 200 bytes) and exits non-zero.  Options
(UnpackedSize modes, memlimit) are fuzzed through the single-file entry point on a subset.
This is a search tool (minutes of GPU time), not part of the test suite; what it finds becomes a test."""
import argparse
import lzma
import os
import random
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402,F401
import lzma_rs_amd as M  # noqa: E402
import oracle_py as orc  # noqa: E402
from lzma_rs_amd import workloads as W  # noqa: E402


def pool_lzma(rng):
    out = []
    for kind in ("text", "random", "repeat", "zeros"):
        for size in (0, 1, 50, 3000, 40000):
            plain = W.make_plain(kind, size, seed=rng.randrange(1 << 30))
            for lc, lp, pb in ((3, 0, 2), (0, 0, 0), (4, 0, 4), (1, 2, 3), (0, 4, 1), (2, 1, 4), (3, 0, 4)):
                if rng.random() < 0.35:
                    ds = rng.choice((1 << 12, 1 << 16, 1 << 20))
                    out.append(W.compress_alone(plain, dict_size=ds, lc=lc, lp=lp, pb=pb, known_size=rng.random() < 0.5))
    return out


def pool_lzma2_xz(rng):
    l2, xz = [], []
    for kind in ("text", "random", "repeat"):
        for size in (0, 10, 5000, 70000, 300000):
            plain = W.make_plain(kind, size, seed=rng.randrange(1 << 30))
            if rng.random() < 0.5:
                plain = plain + W.make_plain("random", 70000, seed=3) + plain[:1000]
            lc, lp, pb = rng.choice(((3, 0, 2), (0, 0, 0), (4, 0, 4), (1, 2, 3), (2, 2, 0)))
            f = [{"id": lzma.FILTER_LZMA2, "dict_size": rng.choice((1 << 12, 1 << 16)), "lc": lc, "lp": lp, "pb": pb}]
            l2.append(lzma.compress(plain, format=lzma.FORMAT_RAW, filters=f))
            chk = rng.choice((lzma.CHECK_NONE, lzma.CHECK_CRC32, lzma.CHECK_CRC64, lzma.CHECK_SHA256))
            xz.append(lzma.compress(plain, format=lzma.FORMAT_XZ, check=chk, filters=f))
            if size >= 5000:
                xz.append(W.compress_xz_blocks(plain, block_size=1 << 16, dict_size=1 << 12, check=rng.choice(("crc32", "crc64", "none"))))
    return l2, xz


def damage(rng, b, header_len):
    b = bytearray(b)
    r = rng.random()
    if r < 0.25 and len(b) > header_len:      # byte flips in the body
        for _ in range(rng.randint(1, 4)):
            b[rng.randrange(header_len, len(b))] = rng.randrange(256)
    elif r < 0.4 and len(b):                  # flips anywhere (headers included)
        for _ in range(rng.randint(1, 3)):
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
    elif r < 0.55:                            # truncation
        b = b[:rng.randrange(len(b) + 1)]
    elif r < 0.65:                            # trailing bytes
        b += bytes(rng.randrange(256) for _ in range(rng.randint(1, 40)))
    elif r < 0.72:                            # bit flip near the end (range coder tail / marker / footer)
        if len(b):
            b[max(0, len(b) - 1 - rng.randrange(min(len(b), 24)))] ^= 1 << rng.randrange(8)
    return bytes(b)


def edit_lzma_header(rng, b):
    if len(b) < 13:
        return b
    b = bytearray(b)
    r = rng.random()
    if r < 0.15:
        b[0] = rng.randrange(256)                                      # props
    elif r < 0.3:
        b[1:5] = struct.pack("<I", rng.choice((0, 1, 4095, 4096, 5000, 1 << 16, 1 << 30, 0xFFFFFFFF)))
    elif r < 0.6:
        cur = struct.unpack("<Q", b[5:13])[0]
        choices = [0, 1, 0xFFFFFFFFFFFFFFFF, 1 << 32, (1 << 32) - 1, 1 << 40]
        if cur != 0xFFFFFFFFFFFFFFFF:
            choices += [max(0, cur - 1), cur + 1, cur // 2, cur + 300]
        b[5:13] = struct.pack("<Q", rng.choice(choices))
    return bytes(b)


def same(tag, comp, dec, ref):
    ok = (dec.kind, dec.msg) == (ref.kind, ref.msg) and dec.data == ref.out and dec.in_consumed == ref.in_consumed
    if not ok:
        print("MISMATCH [%s] %d bytes: %s%s" % (tag, len(comp), comp[:200].hex(), "..." if len(comp) > 200 else ""))
        print("  gpu   :", dec)
        print("  oracle:", ref)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--kernel", default="asm")
    a = ap.parse_args()
    if a.kernel == "generic":
        os.environ["MILZMA_KERNEL"] = "generic"
    rng = random.Random(a.seed)
    ctx = M.Context(0)
    total = bad = 0
    for rnd in range(a.rounds):
        lz = pool_lzma(rng)
        l2, xz = pool_lzma2_xz(rng)
        cases = []
        for _ in range(600):
            c = rng.choice(lz)
            if rng.random() < 0.5:
                c = edit_lzma_header(rng, c)
            if rng.random() < 0.15:
                c = c + rng.choice(lz)
            cases.append(damage(rng, c, 13))
        for comp, d in zip(cases, ctx.lzma_batch(cases)):
            total += 1
            bad += not same("lzma", comp, d, orc.lzma_decompress(comp))
        cases = [damage(rng, rng.choice(l2), 0) for _ in range(300)]
        for comp, d in zip(cases, ctx.lzma2_batch(cases)):
            total += 1
            bad += not same("lzma2", comp, d, orc.lzma2_decompress(comp))
        cases = [damage(rng, rng.choice(xz), 0) for _ in range(300)]
        for comp, d in zip(cases, ctx.xz_batch(cases)):
            total += 1
            bad += not same("xz", comp, d, orc.xz_decompress(comp))
        # options through the single-file entry point
        for _ in range(60):
            c = rng.choice(lz)
            mode = rng.choice((0, 1, 2))
            cur = struct.unpack("<Q", c[5:13])[0] if len(c) >= 13 else 0
            provided = rng.choice((None, 0, 1, 100, cur if cur != 0xFFFFFFFFFFFFFFFF else 77, 1 << 33))
            memlimit = rng.choice((None, None, 0, 1, 100, 4096, 70000))
            comp = c if mode != 2 else c[:5] + c[13:]
            if rng.random() < 0.3:
                comp = damage(rng, comp, 5)
            us = M.UnpackedSize(mode, provided)
            d = ctx.lzma(comp, M.Options(unpacked_size=us, memlimit=memlimit))
            total += 1
            bad += not same("lzma opts mode=%d provided=%r memlimit=%r" % (mode, provided, memlimit), comp, d,
                            orc.lzma_decompress(comp, mode, provided, memlimit))
        print("round %d: %d cases so far, %d mismatches" % (rnd, total, bad), flush=True)
    ctx.close()
    if bad:
        raise SystemExit("parity fuzz: %d of %d cases differ from the oracle" % (bad, total))
    print("parity fuzz: %d cases, all equal to the oracle (kernel %s, seed %d)" % (total, a.kernel, a.seed))


if __name__ == "__main__":
    main()
