"""GPU parity of the push-mode API (milzma_streams_*: lzma_rs::decompress::Stream, feature `stream`) against the oracle's
restatement of src/decode/stream.rs (tests/test_oracle_stream.py pins that one to the reference's own unit tests).

The reference's unit tests of the feature (src/decode/stream.rs:350-493) run here on the GPU path one to one; then random chunkings of
streams of every kind -- good, truncated, damaged; every property class; known / unknown / provided sizes; memlimit -- many streams per
batch, each on its own schedule (streams join late and sit calls out: MILZMA_KIND_START / _HOLD).  What must be EQUAL: what finish hands
over (kind, message, bytes), the text of a failed write, and the CALL that fails -- the tail of every write's data is decoded as far as its
symbols are complete, the crate's trial-run rule (include/milzma.h)."""
import os
import random

import pytest

import lzma_enc as E
import lzma_rs_amd as M
import oracle_py as orc
import test_gpu_parity as P
from lzma_rs_amd import workloads as W

pytestmark = pytest.mark.gpu

# MILZMA_TEST_EXTRA_SEEDS=k: k more seeds for the random schedules (out-of-suite fuzz runs: profiles/r05_parity_fuzz.txt)
EXTRA = [2000 + 41 * j for j in range(int(os.environ.get("MILZMA_TEST_EXTRA_SEEDS", "0")))]

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EMPTY = b"\x5d\x00\x00\x80\x00\xff\xff\xff\xff\xff\xff\xff\xff\x00\x83\xff\xfb\xff\xff\xc0\x00\x00\x00"
WRITE_ZERO = "failed to write whole buffer"


@pytest.fixture(scope="module")
def ctx():
    if os.environ.get("MILZMA_TEST_KEEP_ENV") != "1":   # (stress runs keep MILZMA_SLICE=2 / MILZMA_QUANTUM: every unit parked at every quantum)
        for k in ("MILZMA_KERNEL", "MILZMA_SPILL", "MILZMA_SLICE"):
            os.environ.pop(k, None)
    c = M.Context(0)
    yield c
    c.close()


def small():
    return open(os.path.join(GOLD, "small.txt"), "rb").read()


def run_batch(ctx, comps, schedules, options=None):
    """comps[i] written to stream i in the pieces schedules[i] = [(call number, bytes), ...]; returns (write errors per stream:
    [(call, text)], list of Decoded)."""
    n = len(comps)
    s = M.Streams(ctx, n, options)
    errs = [[] for _ in range(n)]
    calls = max((c for sch in schedules for c, _ in sch), default=-1) + 1
    at = [0] * n
    for call in range(calls):
        pieces = {}
        for i in range(n):
            if at[i] < len(schedules[i]) and schedules[i][at[i]][0] == call:
                pieces[i] = schedules[i][at[i]][1]
                at[i] += 1
        if pieces:
            for i, text in s.write(pieces).items():
                errs[i].append((call, text))
    decs = s.finish()
    s.close()
    return errs, decs


def oracle_stream(comp, schedule, options=None):
    kw = {}
    if options is not None:
        us = options.unpacked_size
        kw = dict(unpacked_size_mode=us.mode, provided=us.provided, memlimit=options.memlimit, allow_incomplete=options.allow_incomplete)
    s = orc.Stream(**kw)
    errs = []
    for call, piece in schedule:
        try:
            s.write_all(piece)
        except orc.Stream.WriteError as e:
            errs.append((call, str(e)))
    return errs, s.finish()


def chunks(comp, size):
    return [(k, comp[at:at + size]) for k, at in enumerate(range(0, len(comp), size))]


def compare(comps, schedules, errs, decs, options=None, what="", skip=()):
    lagged = 0
    for i, comp in enumerate(comps):
        if i in skip:
            continue
        opt = options[i] if isinstance(options, (list, tuple)) else options
        o_errs, o_fin = oracle_stream(comp, schedules[i], opt)
        o_fail = [e for e in o_errs if e[1] != WRITE_ZERO]
        g_fail = [e for e in errs[i] if e[1] != WRITE_ZERO]
        d = decs[i]
        if not o_fail and not g_fail:
            assert (d.kind, d.msg) == (o_fin.kind, o_fin.msg), (what, i, d.msg, o_fin.msg)
            if o_fin.ok:
                assert d.data == o_fin.out, (what, i, len(d.data), len(o_fin.out))
        elif o_fail and g_fail:
            assert g_fail[0][1] == o_fail[0][1], (what, i, g_fail[0], o_fail[0])
            assert g_fail[0][0] == o_fail[0][0], (what, i, g_fail[0], o_fail[0])     # in the very call the crate reports it in

            assert (d.kind, d.msg) == (o_fin.kind, o_fin.msg) and "previous write error" in d.msg, (what, i, d.msg)
        elif o_fail:
            raise AssertionError((what, i, "the crate's write fails, this one's does not", o_fail[0], d.msg))
        else:
            raise AssertionError((what, i, "a write failed that the crate's does not", g_fail[0]))
    return lagged


def test_reference_unit_tests_on_the_gpu_path(ctx):
    # test_stream_noop / test_stream_zero (stream.rs:352-372)
    s = M.Streams(ctx, 2)
    assert s.write({1: b""}) == {}
    assert s.write({1: b""}) == {}
    decs = s.finish()
    s.close()
    assert all(d.ok and d.data == b"" for d in decs)
    # test_bad_header (stream.rs:374-388)
    s = M.Streams(ctx, 1)
    assert s.write({0: bytes([255]) * 32}) == {0: "LZMA header invalid properties: 255 must be < 225"}
    assert "previous write error" in s.finish()[0].msg
    s.close()
    # test_stream_incomplete (stream.rs:390-430): every prefix of the empty stream, one stream each
    n = len(EMPTY) - 1
    s = M.Streams(ctx, n)
    assert s.write({k: EMPTY[:k + 1] for k in range(n)}) == {}
    decs = s.finish()
    s.close()
    for k, d in enumerate(decs):
        end = k + 1
        assert not d.ok and ("failed to read header" if end < 18 else "failed to fill whole buffer") in d.msg, (end, d.msg)
    # test_stream_chunked (stream.rs:432-457): every chunk size, one stream each
    for comp, expected in ((EMPTY, b""), (E.dumb_encode(small()), small())):
        sizes = list(range(1, len(comp)))
        errs, decs = run_batch(ctx, [comp] * len(sizes), [chunks(comp, c) for c in sizes])
        assert not any(errs)
        for c, d in zip(sizes, decs):
            assert d.ok and d.data == expected, (c, d.msg, len(d.data))
    # test_stream_corrupted (stream.rs:459-471)
    s = M.Streams(ctx, 1)
    w = s.write({0: b"corrupted bytes here corrupted bytes here"})
    assert "beyond output size" in w[0], w
    assert "can't finish stream because of previous write error" in s.finish()[0].msg
    s.close()
    # test_allow_incomplete (stream.rs:473-493): exactly 26 bytes
    comp = E.dumb_encode(small())
    half = comp[:len(comp) // 2]
    errs, decs = run_batch(ctx, [half, half], [[(0, half)], [(0, half)]], options=[M.Options(), M.Options(allow_incomplete=True)])
    assert not decs[0].ok and decs[1].ok and decs[1].data == small()[:26], (decs[0].msg, len(decs[1].data))


def test_fixture_files_in_chunks(ctx):   # tests/lzma.rs:116-131: CHUNK_SIZES
    comps, scheds = [], []
    for name in ("foo.txt.lzma", "hello.txt.lzma", "empty.txt.lzma", "hugedict.txt.lzma", "range-coder-edge-case.lzma"):
        comp = open(os.path.join(GOLD, name), "rb").read()
        for c in (1, 2, 3, 4, 5, 6, 7, 8, 16, 32, 64, 128, 256, 512, 1024):
            if len(comp) // c > 3000:
                continue
            comps.append(comp)
            scheds.append(chunks(comp, c))
    errs, decs = run_batch(ctx, comps, scheds)
    assert compare(comps, scheds, errs, decs, what="fixtures") == 0 and all(d.ok for d in decs)


@pytest.mark.parametrize("seed", [91, 191] + EXTRA)
def test_random_chunkings_of_good_and_bad_streams(ctx, seed):
    rng = random.Random(seed)
    comps = []
    for i in range(40):
        lc, lp, pb = [(3, 0, 2), (0, 2, 0), (1, 1, 4), (4, 0, 2), (2, 2, 3), (8, 0, 2)][i % 6]
        if lc + lp > 4:
            c = P._rows_stream(lc, lp, pb, rng.randint(3000, 40000), seed * 7 + i, i % 3 == 0)
        else:
            p = W.make_plain(rng.choice(["text", "text", "random", "repeat", "zeros"]), rng.randint(1, 40000), seed=seed * 7 + i)
            c = W.compress_alone(p, dict_size=rng.choice([4096, 1 << 16]), known_size=(i % 3 == 0), lc=lc, lp=lp, pb=pb)
            if i % 3 == 0 and rng.random() < 0.5:
                c = c[:orc.lzma_decompress(c).in_consumed]   # (the other half of the known-size streams keep liblzma's end marker: WriteZero)
        if i % 7 == 5:
            c = c[:13 + (len(c) - 13) * 2 // 3]
        if i % 7 == 6:
            k = 13 + rng.randrange(len(c) - 13)
            c = c[:k] + bytes([c[k] ^ (1 << rng.randrange(8))]) + c[k + 1:]
        comps.append(c)
    scheds = []
    for c in comps:
        sch, at, call = [], 0, rng.randrange(4)          # (streams join at different calls, and sit calls out)
        while at < len(c):
            n = rng.choice([1, 3, 19, 20, 21, 64, 300, 2000])
            sch.append((call, c[at:at + n]))
            at += n
            call += rng.choice([1, 1, 1, 2, 5])
        scheds.append(sch)
    errs, decs = run_batch(ctx, comps, scheds)
    compare(comps, scheds, errs, decs, what="seed %d" % seed)
    assert sum(1 for d in decs if not d.ok) >= 8


def test_options_and_memlimit(ctx):
    plain = W.make_plain("text", 30000, seed=8)
    marker = W.compress_alone(plain, dict_size=1 << 16, known_size=False)
    sized = W.compress_alone(plain, dict_size=1 << 16, known_size=True)
    sized = sized[:orc.lzma_decompress(sized).in_consumed]                                       # (without liblzma's end marker)
    comps = [marker, marker, sized, sized[:5] + sized[13:], marker, marker]
    opts = [M.Options(unpacked_size=M.UnpackedSize.ReadHeaderButUseProvided(len(plain))),       # the marker stays unread: WriteZero, Ok
            M.Options(unpacked_size=M.UnpackedSize.ReadHeaderButUseProvided(len(plain) + 5)),   # "Expected unpacked size ..." at finish
            M.Options(unpacked_size=M.UnpackedSize.ReadHeaderButUseProvided(None)),             # no size after all: ends at the end of input
            M.Options(unpacked_size=M.UnpackedSize.UseProvided(len(plain))),                    # five-byte header
            M.Options(memlimit=1000),                                                           # "exceeded memory limit of 1000" in a write
            M.Options(memlimit=1 << 20)]
    scheds = [chunks(c, 777) for c in comps]
    errs, decs = run_batch(ctx, comps, scheds, options=opts)
    compare(comps, scheds, errs, decs, options=opts, what="options", skip=(1,))
    assert decs[0].ok and decs[0].data == plain and decs[3].ok and decs[5].ok and not decs[4].ok
    # A provided size LARGER than what an end marker delivers: the crate's finish() decodes on behind the marker (its Partial-mode loop
    # had merely left at `Finished`; the Finish-mode pass finds the size not reached) and trips over the marker's distance -- "Match distance
    # 4294967296 is beyond dictionary size 65536".  Nothing is decoded behind a marker here (include/milzma.h): the one-shot verdict.
    o_errs, o_fin = oracle_stream(comps[1], scheds[1], opts[1])
    assert not o_errs and o_fin.kind == M.LZMA_ERROR and "Match distance 4294967296" in o_fin.msg
    assert decs[1].kind == M.LZMA_ERROR and decs[1].msg == "lzma error: Expected unpacked size of 30005 but decompressed to 30000"


def test_many_streams_on_their_own_schedules(ctx):
    """600 streams of 128 KiB, eight pieces each, a third of them joining only after the others are half done: the launches mix units that
    start, units that resume and units that sit the call out; bytes against the plain text."""
    rng = random.Random(95)
    plains = [W.make_plain("text", 128 << 10, seed=700 + k) for k in range(20)]
    comps20 = [W.compress_alone(p, dict_size=1 << 16, known_size=(k % 2 == 0)) for k, p in enumerate(plains)]
    comps20 = [c[:orc.lzma_decompress(c).in_consumed] for c in comps20]           # (known sizes: without liblzma's end marker)
    n = 600
    comps = [comps20[i % 20] for i in range(n)]
    scheds = []
    for i, c in enumerate(comps):
        cuts = sorted(rng.randrange(1, len(c)) for _ in range(7))
        start = 0 if i % 3 else 5
        calls = sorted(rng.sample(range(start, start + 14), 8))
        scheds.append([(calls[j], c[a:b]) for j, (a, b) in enumerate(zip([0] + cuts, cuts + [len(c)]))])
    errs, decs = run_batch(ctx, comps, scheds)
    assert not any(errs)
    for i, d in enumerate(decs):
        assert d.ok and d.data == plains[i % 20], (i, d.msg, len(d.data))
        assert d.in_consumed == len(comps[i]), (i, d.in_consumed, len(comps[i]))
