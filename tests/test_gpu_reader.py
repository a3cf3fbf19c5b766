"""The crate's one-shot entry points over a reader that shows its input piece by piece ("what's missing" 3): the reference reads
byte by byte (src/decode/rangecoder.rs:59-69; one fill_buf per symbol, src/decode/lzma.rs:497), so after a stream of known size -- or an
LZMA2 stream's end byte -- the reader stands right behind it whatever its buffer size.  Until round 5 the Rust shim decoded what one
fill_buf showed and, if that was not everything, read the rest and decoded again: the reader ended up at ITS end.  With fed input
(MILZMA_DECODE_FEED) and its tail rule the pieces are decoded as they are shown: lzma_rs_amd.decompress_reader is the shim's `run_fed` in
Python; verdict, message, bytes AND reader position against the oracle's one-shot call, for buffers of 1 byte to everything."""
import lzma
import os
import random

import pytest

import lzma_rs_amd as M
import oracle_py as orc
import test_gpu_parity as P
from lzma_rs_amd import workloads as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    if os.environ.get("MILZMA_TEST_KEEP_ENV") != "1":   # (stress runs keep MILZMA_SLICE=2 / MILZMA_QUANTUM: every unit parked at every quantum)
        for k in ("MILZMA_KERNEL", "MILZMA_SPILL", "MILZMA_SLICE"):
            os.environ.pop(k, None)
    c = M.Context(0)
    yield c
    c.close()


BUFS = (1, 7, 19, 20, 21, 64, 4096, 1 << 20)


def check(ctx, kind, data, ref, what):
    for buf in BUFS:
        if len(data) // buf > 2500:
            continue
        d, pos = M.decompress_reader(ctx, kind, data, buf)
        assert (d.kind, d.msg) == (ref.kind, ref.msg), (what, buf, d.msg, ref.msg)
        assert d.data == ref.out, (what, buf, len(d.data), len(ref.out))
        assert pos == ref.in_consumed, (what, buf, pos, ref.in_consumed)


def test_lzma_files_behind_a_buffered_reader(ctx):
    rng = random.Random(11)
    behind = b"-- whatever follows the stream in the reader: another member, a footer, a protocol frame --" * 3
    for i in range(10):
        lc, lp, pb = [(3, 0, 2), (0, 2, 0), (1, 1, 4), (4, 0, 2), (2, 0, 3)][i % 5]
        plain = W.make_plain(["text", "random", "repeat"][i % 3], rng.randint(1, 20000), seed=60 + i)
        comp = W.compress_alone(plain, dict_size=1 << 16, known_size=True, lc=lc, lp=lp, pb=pb)
        comp = comp[:orc.lzma_decompress(comp).in_consumed]              # (known size, no end marker: the stream ends by its size)
        for data, what in ((comp + behind, "known size, bytes behind it"), (comp, "known size"), (comp[:len(comp) * 2 // 3], "truncated")):
            ref = orc.lzma_decompress(data)
            check(ctx, M.KIND_RAW_LZMA, data, ref, (i, what))
        if i < 5:
            marked = W.compress_alone(plain, dict_size=1 << 16, known_size=False, lc=lc, lp=lp, pb=pb)
            k = 13 + rng.randrange(len(marked) - 13)
            damaged = marked[:k] + bytes([marked[k] ^ 0x20]) + marked[k + 1:]
            for data, what in ((marked, "end marker"), (marked + b"x", "a byte behind the marker"), (damaged, "damaged")):
                ref = orc.lzma_decompress(data)
                check(ctx, M.KIND_RAW_LZMA, data, ref, (i, what))
    for data in (b"", b"\x5d", b"\x5d\x00\x00\x01\x00" + bytes(8), b"\x5d\x00\x00\x01\x00" + bytes(8) + bytes(3), bytes([255]) * 30):
        check(ctx, M.KIND_RAW_LZMA, data, orc.lzma_decompress(data), ("short", data[:6]))


def test_lzma2_streams_behind_a_buffered_reader(ctx):
    rng = random.Random(12)
    behind = b"bytes behind the end byte" * 4
    streams = [lzma.compress(W.make_plain("text", 30000, seed=5) + W.make_plain("random", 9000, seed=6), format=lzma.FORMAT_RAW,
                             filters=[{"id": lzma.FILTER_LZMA2, "dict_size": 1 << 16}])]
    streams += [P.random_lzma2_stream(rng, props_pool=[(3, 0, 2), (0, 0, 0), (1, 2, 1), (2, 1, 2)]) for _ in range(8)]
    for i, c in enumerate(streams):
        for data, what in ((c + behind, "bytes behind the end byte"), (c, "whole"), (c[:len(c) - 1], "no end byte"), (c[:len(c) // 2], "half")):
            ref = orc.lzma2_decompress(data)
            check(ctx, M.KIND_LZMA2, data, ref, (i, what))
