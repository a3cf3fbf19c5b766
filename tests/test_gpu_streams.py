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


def compare(comps, schedules, errs, decs, options=None, what=""):
    """Every write error of every stream -- WriteZero included: which calls, which texts --, and what finish hands over, against the oracle's
    Stream on the same schedule."""
    for i, comp in enumerate(comps):
        opt = options[i] if isinstance(options, (list, tuple)) else options
        o_errs, o_fin = oracle_stream(comp, schedules[i], opt)
        d = decs[i]
        assert errs[i] == o_errs, (what, i, errs[i][:3], o_errs[:3])       # the very calls the crate's write_all fails in, with its texts
        assert (d.kind, d.msg) == (o_fin.kind, o_fin.msg), (what, i, d.msg, o_fin.msg)
        if o_fin.ok:
            assert d.data == o_fin.out, (what, i, len(d.data), len(o_fin.out))
        if any(t != WRITE_ZERO for _, t in o_errs):
            assert "previous write error" in d.msg, (what, i, d.msg)


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
    compare(comps, scheds, errs, decs, what="fixtures")
    assert all(d.ok for d in decs)


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
    compare(comps, scheds, errs, decs, options=opts, what="options")
    assert decs[0].ok and decs[0].data == plain and decs[3].ok and decs[5].ok and not decs[4].ok
    # A provided size LARGER than what an end marker delivers: the crate's Partial-mode loop had merely left at `Finished`; finish()'s
    # Finish-mode pass finds the size not reached, decodes on from the marker's state and trips over the marker's distance.
    assert decs[1].kind == M.LZMA_ERROR and decs[1].msg == "lzma error: Match distance 4294967296 is beyond dictionary size 65536"


def test_bytes_written_behind_an_end_marker(ctx):
    """lzma.rs:493-495, :507-509: at an end marker that ends a write's data the crate's Partial-mode loop merely leaves; the stream stays
    in State::Data and whatever is written later is decoded on from the marker's state (rep[0] = 0xFFFF_FFFF, the state after a match).
    Good streams, garbage, another stream's bytes, zeros, a second marker -- in pieces of every size -- against the oracle's Stream."""
    rng = random.Random(613)
    plain = W.make_plain("text", 5000, seed=3)
    marker = W.compress_alone(plain, dict_size=1 << 16, known_size=False)
    other = W.compress_alone(W.make_plain("text", 3000, seed=4), dict_size=1 << 12, known_size=False)
    tails = [b"\x00" * 40, b"\x00" * 7, bytes(rng.randrange(256) for _ in range(64)), other, other[13:], marker[13:], EMPTY[13:], EMPTY,
             b"\xff" * 30, bytes([0, 0, 0, 0, 1]) + bytes(rng.randrange(256) for _ in range(300)), b"\x00" * 19 + b"\x80", b""]
    heads = [marker, EMPTY, W.compress_alone(plain, dict_size=1 << 16, known_size=False, lc=4, lp=0, pb=2),
             W.compress_alone(plain, dict_size=1 << 16, known_size=False, lc=0, lp=2, pb=4)]
    comps, scheds, opts = [], [], []
    for head in heads:
        for tail in tails:
            for how in range(4):
                comp = head + tail
                if how == 0:      # the tail in one write
                    sch = [(0, head), (1, tail)]
                elif how == 1:    # ... byte by byte (the trial runs hide the distance error until 20 bytes are at hand)
                    sch = [(0, head)] + [(1 + k, tail[k:k + 1]) for k in range(len(tail))]
                elif how == 2:    # ... with the marker's own last bytes (no write ends at the marker: "more bytes are available")
                    sch = [(0, head[:-3]), (1, head[-3:] + tail)]
                else:             # ... in random pieces, the head too
                    sch, at, call = [], 0, 0
                    cuts = sorted(set([len(head)] + [rng.randrange(1, len(comp)) for _ in range(6)]))
                    for c in cuts + [len(comp)]:
                        if c > at:
                            sch.append((call, comp[at:c]))
                            call += 1
                            at = c
                sch = [(c, b) for c, b in sch if b]
                comps.append(comp)
                scheds.append(sch)
                opts.append([M.Options(), M.Options(allow_incomplete=True),
                             M.Options(unpacked_size=M.UnpackedSize.ReadHeaderButUseProvided(len(plain) + 100))][len(comps) % 3])
    errs, decs = run_batch(ctx, comps, scheds, options=opts)
    compare(comps, scheds, errs, decs, options=opts, what="behind a marker")
    texts = {t for e in errs for _, t in e} | {d.msg for d in decs}
    assert any("Match distance 4294967296 is beyond dictionary size" in t for t in texts), texts
    assert any("Found end-of-stream marker but more bytes are available" in t for t in texts), texts
    # (behind a marker `code` is 0, so the next is_match decision always says "literal" -- a matched one, 2^32 back: whatever follows a marker
    #  that ended a write fails there as soon as 20 bytes are at hand, or at finish; only allow_incomplete gets away with fewer)
    for comp, sch, d in zip(comps, scheds, decs):
        assert not (d.ok and len(d.data) > len(plain)), (len(comp), len(d.data))


def test_bytes_behind_a_declared_size_and_headers_read_through_stream_tmp(ctx):
    """Which write is refused (ErrorKind::WriteZero) when bytes follow a stream that has reached its declared size depends on the crate's
    partial-input buffer (lzma.rs:420-433, :457-495: once a write has ended inside a symbol, every iteration first fills that buffer to 20
    bytes from the symbol's first byte on -- bytes BEHIND the stream's end too, which then count as taken), and a five-byte header
    (UnpackedSize::UseProvided) that arrives in pieces is read through Stream.tmp, whose leftover is the next write() call's first input
    (stream.rs:233-270, :312-317).  Trailing bytes of 1 .. 60, cuts everywhere around the stream's end and inside the header, the calls
    and the texts against the oracle's Stream."""
    rng = random.Random(2718)
    comps, scheds, opts = [], [], []
    for k in range(120):
        size = rng.choice([1, 7, 300, 5000, 20000])
        plain = W.make_plain(rng.choice(["text", "random", "zeros"]), size, seed=40 + k)
        sized = W.compress_alone(plain, dict_size=1 << 16, known_size=True, lc=(4 if k % 5 == 0 else 3))
        sized = sized[:orc.lzma_decompress(sized).in_consumed]                    # (without liblzma's end marker)
        trail = bytes(rng.randrange(256) for _ in range(rng.choice([1, 2, 5, 19, 20, 21, 40, 60])))
        if k % 2:                                                                 # a five-byte header + a provided size
            comp, opt = sized[:5] + sized[13:] + trail, M.Options(unpacked_size=M.UnpackedSize.UseProvided(size))
        else:
            comp, opt = sized + trail, M.Options()
        end = len(comp) - len(trail)
        cuts = set()
        how = k % 4
        if how == 0:       # one cut somewhere in the last bytes of the stream, then the trail in one piece / in small ones
            cuts = {max(1, end - rng.randrange(1, 30))} | ({end + 3, end + 11} if k % 8 == 0 else set())
        elif how == 1:     # byte by byte through the header, then a few larger pieces
            cuts = set(range(1, min(len(comp), 26))) | {rng.randrange(1, len(comp)) for _ in range(3)}
        elif how == 2:     # small pieces all the way
            at = 0
            while at < len(comp):
                at += rng.choice([1, 2, 3, 5, 8, 19, 20, 21])
                cuts.add(at)
        else:              # header in two pieces (3 + the rest of an 18-byte Stream.tmp ...), everything else at once
            cuts = {3, rng.choice([9, 10, 11, 17, 18, 19])}
        cuts = sorted(c for c in cuts if 0 < c < len(comp))
        sch = [(n, comp[a:b]) for n, (a, b) in enumerate(zip([0] + cuts, cuts + [len(comp)]))]
        comps.append(comp)
        scheds.append(sch)
        opts.append(opt)
    errs, decs = run_batch(ctx, comps, scheds, options=opts)
    compare(comps, scheds, errs, decs, options=opts, what="behind a declared size")
    zero = sum(1 for e in errs if e and e[0][1] == WRITE_ZERO)
    assert 30 <= zero < len(comps) and all(d.ok for d in decs), zero           # some writes are refused, some rests fit the crate's buffer


def test_get_output_between_writes(ctx):
    """Stream::get_output (stream.rs:102-116): the sink holds every completed flush of the ring (lzbuffer.rs:264-267) -- whole multiples of
    the dictionary size -- while the stream runs, and there is none after a failed write."""
    plain = W.make_plain("text", 70000, seed=21)
    comps = [W.compress_alone(plain, dict_size=4096, known_size=False), W.compress_alone(plain, dict_size=1 << 16, known_size=True),
             W.compress_alone(plain, dict_size=1 << 20, known_size=False), b"corrupted bytes here corrupted bytes here" * 3, EMPTY[:9]]
    n = len(comps)
    s = M.Streams(ctx, n)
    os_ = [orc.Stream() for _ in range(n)]
    piece, seen = 1500, 0
    for at in range(0, max(len(c) for c in comps), piece):
        pieces = {i: c[at:at + piece] for i, c in enumerate(comps) if c[at:at + piece]}
        g = s.write(pieces)
        for i, b in pieces.items():
            try:
                os_[i].write_all(b)
                assert i not in g, (i, at, g[i])
            except orc.Stream.WriteError as e:
                assert g.get(i) == str(e), (i, at, g.get(i), str(e))
            if g.get(i, WRITE_ZERO) == WRITE_ZERO:       # how much of the piece Stream::write took: all of it, or what the crate reckons
                assert s.taken(i) == os_[i].last_taken(), (i, at, s.taken(i), os_[i].last_taken())   # to lie in front of the stream's end
        for i in range(n):
            want = os_[i].get_output()
            got = s.output(i)
            assert got == want, (i, at, None if got is None else len(got), None if want is None else len(want))
            seen += 1 if want else 0
    assert seen > 20 and s.output(3) is None
    decs = s.finish()
    s.close()
    for i in range(n):
        ref = os_[i].finish()
        assert (decs[i].kind, decs[i].msg) == (ref.kind, ref.msg), (i, decs[i].msg, ref.msg)
        if ref.ok:
            assert decs[i].data == ref.out == plain


def test_a_late_stream_with_more_literal_rows_than_the_slab_has(ctx):
    """the batch's literal-row slab has one stride, fixed by its first launch of the class; a .lzma stream that joins later with
    lc + lp = 9 .. 12 (one legal header byte) used to fail the whole batch.  It gets a batch of its own behind the same calls."""
    rng = random.Random(77)
    n = 700                                        # (a batch this large lays its slab out for lc + lp <= 8: milzma_streams_open)
    plains = [W.make_plain("text", 8000, seed=900 + k) for k in range(4)]
    base = [W.compress_alone(plains[k % 4], dict_size=1 << 16, known_size=(k % 2 == 0), lc=(4 if k >= 4 else 3)) for k in range(8)]
    comps = [base[i % 8] for i in range(n)]
    rich = {5: (8, 4, 2), 17: (8, 2, 0), 40: (7, 4, 4), 63: (8, 4, 4), 64: (8, 0, 2)}
    for i, (lc, lp, pb) in rich.items():
        comps[i] = P._rows_stream(lc, lp, pb, 9000 + i, 31 * i, i % 2 == 0)
    comps[63] = comps[63][:len(comps[63]) * 2 // 3]                              # (a truncated one: "failed to fill whole buffer" at finish)
    scheds = []
    for i, c in enumerate(comps):
        start = 3 if i in rich else rng.randrange(2)                          # the rich streams join after the slab is laid out
        cuts = sorted(rng.randrange(1, len(c)) for _ in range(4))
        scheds.append([(start + 2 * j, c[a:b]) for j, (a, b) in enumerate(zip([0] + cuts, cuts + [len(c)])) if b > a])
    errs, decs = run_batch(ctx, comps, scheds)
    compare(comps, scheds, errs, decs, what="rich literal rows")
    assert all(decs[i].ok for i in range(n) if i != 63) and not decs[63].ok


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
