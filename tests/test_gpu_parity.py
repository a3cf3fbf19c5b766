"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bit-exact is the bar: error kind, full message, bytes delivered to the writer and the reader's
final position must all equal what the oracle (= the reference's behaviour) produces.  Every test
here needs a real MI355X and calls through libmilzma.so; nothing falls back to the oracle.
"""
import hashlib
import lzma
import os
import random
import struct

import pytest

import lzma_enc as E
import lzma_rs_amd as M
import oracle_py as orc
from lzma_rs_amd import workloads as W

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    with open(os.path.join(GOLD, name), "rb") as f:
        return f.read()


@pytest.fixture(scope="module", params=["asm", "generic"])
def ctx(request):
    """Every test runs twice: with the lane-resident-model kernel whose symbol loop is asm (the default; units
    outside its class still fall through to the generic kernel) and with the generic kernel only."""
    os.environ["MILZMA_KERNEL"] = request.param
    c = M.Context(0)
    yield c
    c.close()
    os.environ.pop("MILZMA_KERNEL", None)


def same(dec, ref, check_consumed=True):
    """dec: lzma_rs_amd.Decoded, ref: oracle_py.OracleResult"""
    assert (dec.kind, dec.msg) == (ref.kind, ref.msg), (dec, ref)
    assert dec.data == ref.out, (len(dec.data), len(ref.out))
    # the reader's position, on success AND on error (round 4: a truncated stream used to leave it one byte beyond the end of the
    # input -- nothing compared it for failed decodes)
    if check_consumed:
        assert dec.in_consumed == ref.in_consumed, (dec, ref)


# ---- the reference's own fixtures (tests/lzma.rs, tests/xz.rs) -------------------------------

@pytest.mark.parametrize("name", ["hello.txt", "empty.txt", "foo.txt"])
def test_lzma_fixtures(ctx, name):
    comp = gold(name + ".lzma")
    d = ctx.lzma(comp)
    assert d.ok and d.data == gold(name)
    same(d, orc.lzma_decompress(comp))


def test_lzma_hugedict_and_edge_case(ctx):
    d = ctx.lzma(gold("hugedict.txt.lzma"))
    assert d.ok and d.data == gold("foo.txt")
    d = ctx.lzma(gold("range-coder-edge-case.lzma"))
    assert d.ok and len(d.data) == 3040092
    assert hashlib.sha256(d.data).hexdigest() == \
        "1bb292093eef1b21af67a24468ab40cfde2a616b2f859f90d3861568efd3b0f6"


@pytest.mark.parametrize("name", ["foo.txt", "good-1-lzma2-1", "good-1-lzma2-2", "good-1-lzma2-3",
                                  "good-1-lzma2-4", "hello.txt", "empty.txt", "block-check-crc32.txt"])
def test_xz_fixtures(ctx, name):
    comp = gold(name + ".xz")
    d = ctx.xz(comp)
    assert d.ok and d.data == gold(name)
    same(d, orc.xz_decompress(comp))


def test_xz_block_check_crc32_invalid(ctx):  # tests/xz.rs:123-146
    buf = bytearray(gold("block-check-crc32.txt.xz"))
    buf[0x54:0x58] = bytes([0x67, 0x45, 0x23, 0x01])
    d = ctx.xz(bytes(buf))
    assert d.msg == "xz error: Invalid footer CRC32: expected 0x01234567 but got 0x8b0d303e"
    same(d, orc.xz_decompress(bytes(buf)))


def test_crate_shaped_api(ctx):
    import io
    out = io.BytesIO()
    M.lzma_decompress(io.BytesIO(gold("foo.txt.lzma")), out, ctx=ctx)
    assert out.getvalue() == gold("foo.txt")
    out = bytearray()
    M.xz_decompress(gold("foo.txt.xz"), out, ctx=ctx)
    assert bytes(out) == gold("foo.txt")
    with pytest.raises(M.HeaderTooShort):
        M.lzma_decompress(b"", bytearray(), ctx=ctx)
    # known-size stream followed by other data: the reader stops where the reference stops
    comp = W.compress_alone(b"hello hello hello hello", known_size=True)
    inp = io.BytesIO(comp + b"TRAILING")
    out = io.BytesIO()
    M.lzma_decompress(inp, out, ctx=ctx)
    ref = orc.lzma_decompress(comp + b"TRAILING")
    assert out.getvalue() == ref.out and inp.tell() == ref.in_consumed


# ---- options (tests/lzma.rs:237-356) ----------------------------------------------------------

def test_option_matrix_and_memlimit(ctx):
    data = b"Some data"
    n = len(data)
    US = M.UnpackedSize
    cases = [
        (E.dumb_encode(data, unpacked_size=n), None, (orc.READ_FROM_HEADER, None)),
        (E.dumb_encode(data, write_size=False), M.Options(US.UseProvided(n)), (orc.USE_PROVIDED, n)),
        (E.dumb_encode(data, unpacked_size=n), M.Options(US.ReadHeaderButUseProvided(n)),
         (orc.READ_HEADER_BUT_USE_PROVIDED, n)),
        (E.dumb_encode(data), M.Options(US.ReadHeaderButUseProvided(n)), (orc.READ_HEADER_BUT_USE_PROVIDED, n)),
        (E.dumb_encode(data), M.Options(US.ReadHeaderButUseProvided(None)),
         (orc.READ_HEADER_BUT_USE_PROVIDED, None)),
    ]
    for comp, opts, (mode, provided) in cases:
        d = ctx.lzma(comp, opts)
        assert d.ok and d.data == data
        same(d, orc.lzma_decompress(comp, mode, provided))
    comp = E.dumb_encode(data)
    for memlimit in (0, 4, 8, 9, 100):
        d = ctx.lzma(comp, M.Options(memlimit=memlimit))
        same(d, orc.lzma_decompress(comp, memlimit=memlimit))
    d = ctx.lzma(comp, M.Options(US.ReadHeaderButUseProvided(None), memlimit=0))
    assert "exceeded memory limit of 0" in d.msg
    # memlimit below the dictionary size, hit in the middle of a long match
    big, _ = E.encode_lzma([("lit", 65)] + [("match", 200, 1)] * 40 + [("marker",)], dict_size=1 << 16)
    for memlimit in (150, 4096, 5000):
        same(ctx.lzma(big, M.Options(memlimit=memlimit)), orc.lzma_decompress(big, memlimit=memlimit))


# ---- generated streams vs the oracle (and liblzma) --------------------------------------------

@pytest.mark.parametrize("kind", ["text", "random", "repeat", "zeros"])
def test_generated_streams_all_props(ctx, kind):
    plain = W.make_plain(kind, 150_000, seed=11)
    comps = []
    for dict_size in (4096, 65536, 1 << 23):
        for lc, lp, pb in [(3, 0, 2), (0, 2, 0), (4, 0, 4), (1, 3, 1), (0, 0, 0)]:
            comp = W.compress_alone(plain, dict_size=dict_size, lc=lc, lp=lp, pb=pb)
            comps.append(comp)
            comps.append(comp[:5] + struct.pack("<Q", len(plain)) + comp[13:])  # known size
    decs = ctx.lzma_batch(comps)
    for comp, d in zip(comps, decs):
        assert d.ok and d.data == plain
        same(d, orc.lzma_decompress(comp))


def test_symbol_streams_every_lclppb(ctx):
    # liblzma refuses lc+lp > 4; the symbol encoder covers the whole 225-value props space
    rng = random.Random(5)
    comps = []
    for trial in range(120):
        lc, lp, pb = rng.choice([0, 3, 4, 8]), rng.choice([0, 1, 2, 4]), rng.choice([0, 1, 2, 4])
        enc = E.LzmaSymbolEncoder(lc, lp, pb)
        n = 0
        for _ in range(rng.randint(0, 300)):
            r = rng.random()
            if n == 0 or r < 0.4:
                s = ("lit", rng.randrange(256))
            elif r < 0.7:
                s = ("match", rng.randint(2, rng.choice([5, 20, 273])), rng.randint(1, n))
            elif r < 0.85:
                idx = rng.randint(0, 3)
                s = ("rep", idx, rng.randint(2, 40)) if enc.rep[idx] + 1 <= n else ("lit", 7)
            else:
                s = ("shortrep",) if enc.rep[0] + 1 <= n else ("lit", 9)
            enc.encode([s])
            n = len(enc.out)
        marker = rng.random() < 0.5
        if marker:
            enc.encode([("marker",)])
        comps.append(E.lzma_header(lc, lp, pb, 1 << 20, None if marker else n) + enc.finish())
    decs = ctx.lzma_batch(comps)
    for comp, d in zip(comps, decs):
        same(d, orc.lzma_decompress(comp))
        assert d.ok


def test_lclp_above_four_at_size(ctx):
    """lc + lp > 4 (legal in a .lzma header, lzma.rs:96-161; liblzma cannot write it) and the lc + lp = 4 class at sizes where the
    literal table is really exercised: 256 KiB..1 MiB of text + binary with real match structure (greedy LZ parse, symbol
    encoder), every literal row of the 2^(lc+lp) x 0x300 table reachable; the spill class keeps its table in HBM scratch."""
    rnd = random.Random(77)
    comps, plains = [], []
    for (lc, lp, pb), size in [((8, 0, 2), 1 << 20), ((4, 4, 0), 1 << 18), ((5, 2, 4), 1 << 18), ((8, 4, 4), 1 << 17),
                               ((4, 0, 2), 1 << 18), ((0, 4, 0), 1 << 18), ((2, 2, 4), 1 << 18)]:
        plain = W.make_plain("text", size - 40000, seed=lc * 100 + lp * 10 + pb) + rnd.randbytes(20000) + bytes(range(256)) * 78 + b"x" * 32
        plain = plain[:size]
        syms = E.lz_parse(plain, dict_size=1 << 16)
        known = (lc + lp) % 2 == 0
        enc = E.LzmaSymbolEncoder(lc, lp, pb)
        enc.encode(syms)
        if not known:
            enc.encode([("marker",)])
        comps.append(E.lzma_header(lc, lp, pb, 1 << 16, len(plain) if known else None) + enc.finish())
        plains.append(plain)
    decs = ctx.lzma_batch(comps)
    for comp, plain, d in zip(comps, plains, decs):
        assert d.ok and d.data == plain, (comp[0], d)
        same(d, orc.lzma_decompress(comp))
    # and cut short / with a byte damaged in the middle: same verdict, same bytes delivered, as the oracle
    bad = [c[:len(c) * 2 // 3] for c in comps] + [c[:len(c) // 2] + bytes([c[len(c) // 2] ^ 0x55]) + c[len(c) // 2 + 1:] for c in comps]
    for comp, d in zip(bad, ctx.lzma_batch(bad)):
        same(d, orc.lzma_decompress(comp))


def test_symbol_streams_stress(ctx):
    # many longer random symbol sequences inside the fast kernels' property class (pb <= 2, lc + lp <= 3):
    # every symbol kind after every other, short / 64+ byte / self-overlapping matches, all rep indices,
    # known-size and end-marker termination; a third of them with a few corrupted payload bytes
    rng = random.Random(2026)
    comps = []
    for trial in range(400):
        lc = rng.choice([0, 1, 2, 3])
        lp = rng.randint(0, 3 - lc)
        pb = rng.randint(0, 2)
        enc = E.LzmaSymbolEncoder(lc, lp, pb)
        n = 0
        for _ in range(rng.randint(1, 1500)):
            r = rng.random()
            if n == 0 or r < 0.5:
                s = ("lit", rng.choice([rng.randrange(256), 0x20, 0x65]))
            elif r < 0.75:
                dist = rng.randint(1, n) if rng.random() < 0.5 else rng.randint(1, min(n, 40))
                s = ("match", rng.randint(2, rng.choice([3, 8, 70, 273])), dist)
            elif r < 0.9:
                idx = rng.randint(0, 3)
                s = ("rep", idx, rng.randint(2, rng.choice([6, 90]))) if enc.rep[idx] + 1 <= n else ("lit", 7)
            else:
                s = ("shortrep",) if enc.rep[0] + 1 <= n else ("lit", 9)
            enc.encode([s])
            n = len(enc.out)
        marker = rng.random() < 0.4
        if marker:
            enc.encode([("marker",)])
        comp = bytearray(E.lzma_header(lc, lp, pb, rng.choice([1 << 12, 1 << 16, 1 << 20]), None if marker else n) +
                         enc.finish())
        if trial % 3 == 2 and len(comp) > 20:
            for _ in range(rng.randint(1, 3)):
                comp[rng.randrange(13, len(comp))] ^= 1 << rng.randrange(8)
        comps.append(bytes(comp))
    decs = ctx.lzma_batch(comps)
    for comp, d in zip(comps, decs):
        same(d, orc.lzma_decompress(comp))


def test_error_sites_match_oracle(ctx):
    lits = [("lit", c) for c in b"abcdefgh"]
    many = [("lit", i & 0xFF) for i in range(6000)]
    cases = [
        E.encode_lzma(lits + [("match", 4, 9), ("marker",)])[0],
        E.encode_lzma(many + [("match", 4, 5000), ("marker",)], dict_size=4096)[0],
        E.encode_lzma(lits + [("marker",)])[0] + b"\x00",
        E.encode_lzma([("lit", 1), ("lit", 2), ("lit", 3), ("match", 7, 3)], unpacked_size=5)[0],
        E.encode_lzma([("lit", 1)] * 5000 + [("match", 273, 3)], unpacked_size=5100, dict_size=4096)[0],
        gold("hello.txt.lzma")[:17],
        gold("foo.txt.lzma")[:30000],
        gold("foo.txt.lzma")[:13],
        b"\xff" + gold("hello.txt.lzma")[1:],
        E.encode_lzma(lits, unpacked_size=None)[0],
    ]
    for n in range(14, 35):
        cases.append(gold("hello.txt.lzma")[:n])
    decs = ctx.lzma_batch(cases)
    for comp, d in zip(cases, decs):
        same(d, orc.lzma_decompress(comp))
    for comp in cases[:4]:
        same(ctx.lzma(comp), orc.lzma_decompress(comp))


def test_corrupted_streams_same_verdict(ctx):  # fuzz/fuzz_targets/decompress_lzma.rs, compare_xz.rs
    rng = random.Random(99)
    base = gold("foo.txt.lzma")
    cases = []
    for _ in range(64):
        b = bytearray(base)
        for _ in range(rng.randint(1, 3)):
            b[rng.randrange(13, len(b))] = rng.randrange(256)
        cases.append(bytes(b))
    for comp, d in zip(cases, ctx.lzma_batch(cases)):
        same(d, orc.lzma_decompress(comp))
    base = gold("foo.txt.xz")
    cases = []
    for _ in range(48):
        b = bytearray(base)
        for _ in range(rng.randint(1, 3)):
            b[rng.randrange(len(b))] = rng.randrange(256)
        cases.append(bytes(b))
    for comp, d in zip(cases, ctx.xz_batch(cases)):
        same(d, orc.xz_decompress(comp))


# ---- LZMA2 / XZ ---------------------------------------------------------------------------------

def test_lzma2_streams(ctx):
    plain = W.make_plain("text", 300_000, seed=3) + W.make_plain("random", 150_000, seed=4) + \
        W.make_plain("text", 300_000, seed=5)
    cases = []
    for lc, lp, pb in [(3, 0, 2), (4, 0, 0), (0, 4, 4), (2, 2, 1)]:
        filt = [{"id": lzma.FILTER_LZMA2, "dict_size": 65536, "lc": lc, "lp": lp, "pb": pb}]
        cases.append(lzma.compress(plain, format=lzma.FORMAT_RAW, filters=filt))
    stored = b"".join(E.lzma2_stored_chunk(plain[i:i + 0x10000], True)
                      for i in range(0, len(plain), 0x10000)) + b"\x00"
    cases.append(stored)
    cases.append(b"\x00")
    for comp, d in zip(cases, ctx.lzma2_batch(cases)):
        ref = orc.lzma2_decompress(comp)
        same(d, ref)
        assert d.ok
    same(ctx.lzma2(cases[0]), orc.lzma2_decompress(cases[0]))


def test_lzma2_error_sites(ctx):
    enc = E.LzmaSymbolEncoder(3, 0, 2).encode([("lit", 65), ("match", 3, 5)])
    dist_err = E.lzma2_stored_chunk(b"0123456789", True) + \
        E.lzma2_lzma_chunk(enc.finish(), 4, 0xE0, props=0x5D) + b"\x00"
    enc = E.LzmaSymbolEncoder(3, 0, 2).encode([("lit", c) for c in b"abcdef"] + [("match", 2, 5)])
    c1 = enc.take_chunk()
    enc.stored(b"ZZ", True)
    enc.encode([("lit", 0x41)])
    c2 = enc.take_chunk()
    match_err = E.lzma2_lzma_chunk(c1, 8, 0xE0, props=0x5D) + E.lzma2_stored_chunk(b"ZZ", True) + \
        E.lzma2_lzma_chunk(c2, 1, 0x80) + b"\x00"
    enc = E.LzmaSymbolEncoder(0, 0, 0).encode([("lit", c) for c in b"xyz"] + [("match", 5, 2)])
    no_reset = E.lzma2_lzma_chunk(enc.finish(), 8, 0x80) + b"\x00"
    enc = E.LzmaSymbolEncoder(3, 0, 2).encode([("lit", c) for c in b"hello"])
    payload = enc.finish()
    probe = orc.lzma2_decompress(E.lzma2_lzma_chunk(payload, 5, 0xE0, props=0x5D) + b"\x00")
    used = probe.in_consumed - 1 - 6
    leftover = bytes([0xE0]) + struct.pack(">H", 4) + struct.pack(">H", used + 3 - 1) + b"\x5d" + \
        payload[:used] + E.lzma2_stored_chunk(b"!", False) + b"\x00"
    size_mismatch = E.lzma2_lzma_chunk(payload, 4, 0xE0, props=0x5D) + b"\x00"
    cases = [b"", b"\x03\x00\x00", b"\x01\x00", b"\x01\x00\x04abc", b"\xe0\x00\x04\x00",
             b"\xe0\x00\x04\x00\x09", b"\xe0\x00\x04\x00\x09\xe1",
             b"\xe0\x00\x04\x00\x09" + bytes([E.props_byte(4, 1, 0)]), b"\xe0\x00\x04\x00\x09\x5d\x00\x00",
             dist_err, match_err, no_reset, leftover, size_mismatch]
    for comp, d in zip(cases, ctx.lzma2_batch(cases)):
        same(d, orc.lzma2_decompress(comp))


def test_xz_multiblock_and_checks(ctx):
    plain = W.make_plain("text", 500_000, seed=21) + W.make_plain("random", 100_000, seed=22) + \
        W.make_plain("repeat", 300_000, seed=23)
    filt = [{"id": lzma.FILTER_LZMA2, "dict_size": 65536, "lc": 3, "lp": 0, "pb": 2}]
    cases = [W.compress_xz_blocks(plain, block_size=1 << 18, check=c) for c in ("crc64", "crc32", "none")]
    cases.append(lzma.compress(plain, format=lzma.FORMAT_XZ, check=lzma.CHECK_SHA256, filters=filt))
    cases.append(lzma.compress(plain, format=lzma.FORMAT_XZ, check=lzma.CHECK_CRC64, filters=filt))
    # damage in the middle of a block, a truncated file, trailing bytes: the planned units cannot all
    # be trusted, the exact walk must still produce the reference's verdict
    bad = bytearray(cases[0])
    bad[len(bad) // 2] ^= 0xFF
    cases.append(bytes(bad))
    cases.append(cases[0][:-3])
    cases.append(cases[0] + b"\x00\x00\x00\x00")
    for comp, d in zip(cases, ctx.xz_batch(cases)):
        same(d, orc.xz_decompress(comp))
    assert ctx.xz(cases[0]).data == plain


def test_batch_grows_output_for_unknown_sizes(ctx):
    # end-marker streams give no size up front: the batch guesses an output slice, and the streams that
    # overflow it (here: 300 KB of zeros in ~100 bytes) are decoded again, together, with more room
    plains = [W.make_plain("zeros", 300_000 + 1000 * i, seed=i) for i in range(12)] + \
             [W.make_plain("repeat", 200_000, seed=50), W.make_plain("text", 100_000, seed=51)]
    comps = [W.compress_alone(p, dict_size=65536, known_size=False) for p in plains]
    for comp, plain, d in zip(comps, plains, ctx.lzma_batch(comps)):
        assert d.ok and d.data == plain
        same(d, orc.lzma_decompress(comp))
    raw = [lzma.compress(p, format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "dict_size": 65536}])
           for p in plains[:6]]
    for comp, plain, d in zip(raw, plains, ctx.lzma2_batch(raw)):
        assert d.ok and d.data == plain
        same(d, orc.lzma2_decompress(comp))


def test_truncated_at_every_length(ctx):
    # the reader hits EOF at every byte position (inside every kind of symbol, across the 64-byte input
    # window refills): same error, same bytes delivered, same reader position as the reference
    plain = W.make_plain("text", 5000, seed=31)
    comp = W.compress_alone(plain, dict_size=4096)
    known = comp[:5] + struct.pack("<Q", len(plain)) + comp[13:]
    cases = [known[:n] for n in range(0, len(known))] + [comp[:n] for n in range(len(comp) - 40, len(comp))]
    for c, d in zip(cases, ctx.lzma_batch(cases)):
        same(d, orc.lzma_decompress(c))


LZMA2_PROPS_POOL = [(3, 0, 2), (3, 0, 2), (0, 0, 0), (1, 2, 1), (2, 1, 2), (0, 3, 0), (4, 0, 2), (0, 4, 1)]


def random_lzma2_stream(rng, max_chunks=7, max_syms=120, props_pool=LZMA2_PROPS_POOL):
    """One LZMA2 stream of random packets: LZMA chunks that end at arbitrary symbols (every control byte the format has, properties from
    `props_pool`), stored chunks and dictionary resets in between; ends with the end byte."""
    lc, lp, pb = rng.choice(props_pool)
    enc = E.LzmaSymbolEncoder(lc, lp, pb)
    stream = b""
    first, props_sent = True, False
    for chunk in range(rng.randint(1, max_chunks)):
        r = rng.random()
        if r < 0.25:
            data = bytes(rng.randrange(256) for _ in range(rng.randint(1, 300)))
            reset = first or rng.random() < 0.2
            enc.stored(data, reset)
            stream += E.lzma2_stored_chunk(data, reset)
            first = False
            continue
        if first:
            control = 0xE0
        elif not props_sent:
            control = rng.choice([0xC0, 0xE0])  # the first LZMA chunk has to carry the properties
        else:
            control = rng.choice([0x80, 0x80, 0xA0, 0xC0, 0xE0])
        props_sent = True
        props = None
        if control >= 0xC0:
            lc, lp, pb = rng.choice(props_pool)
            props = E.props_byte(lc, lp, pb)
        if control >= 0xA0:
            enc.reset_state(lc, lp, pb)
        if control == 0xE0:
            enc.stored(b"", True)
        before = enc.total_len
        n = len(enc.out)
        for _ in range(rng.randint(1, max_syms)):
            r = rng.random()
            if n == 0 or r < 0.45:
                sym = ("lit", rng.randrange(256))
            elif r < 0.75:
                sym = ("match", rng.randint(2, rng.choice([4, 30, 273])), rng.randint(1, n))
            elif r < 0.9:
                idx = rng.randint(0, 3)
                sym = ("rep", idx, rng.randint(2, 90)) if enc.rep[idx] + 1 <= n else ("lit", 1)
            else:
                sym = ("shortrep",) if enc.rep[0] + 1 <= n else ("lit", 2)
            enc.encode([sym])
            n = len(enc.out)
        payload = enc.take_chunk()
        stream += E.lzma2_lzma_chunk(payload, enc.total_len - before, control, props=props)
        first = False
    return stream + b"\x00"


def test_lzma2_random_chunk_sequences(ctx):
    # chunk boundaries at arbitrary symbols: state, reps and the literal context cross from one range
    # coder to the next; stored chunks and dictionary resets in between (lzma2.rs:84-229)
    rng = random.Random(77)
    cases = [random_lzma2_stream(rng) for trial in range(40)]
    for comp, d in zip(cases, ctx.lzma2_batch(cases)):
        ref = orc.lzma2_decompress(comp)
        same(d, ref)
        assert d.ok, ref.msg


# ---- the unit-level ABI with device-resident buffers (what bench.py times) ---------------------

def test_decode_units_device_resident(ctx):
    import torch
    comps, plains = W.make_lzma_batch(24, size=1 << 18, kind="text", dict_size=65536, known_size=True,
                                      keep_plain=True)
    n = len(comps)
    units = (M.Unit * n)()
    in_off, blobs = 0, []
    for i, c in enumerate(comps):
        u, hl = M.lzma_read_header(c)
        payload = c[hl:]
        u.in_off, u.in_len = in_off, len(payload)
        u.out_off, u.out_cap = i * (1 << 18), 1 << 18
        units[i] = u
        pad = (-len(payload)) % 256
        blobs.append(payload + bytes(pad))
        in_off += len(payload) + pad
    d_in = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_out = torch.zeros(n << 18, dtype=torch.uint8, device="cuda")
    res, ms, launches = ctx.decode_units(units, d_in.data_ptr(), d_out.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream)
    assert launches == 1 and ms > 0
    host = d_out.cpu().numpy().tobytes()
    for i in range(n):
        assert res[i].status == M.ST_OK and res[i].out_len == 1 << 18
        assert host[i << 18:(i + 1) << 18] == plains[i]
        ref = orc.lzma_decompress(comps[i])
        assert res[i].in_consumed + 13 == ref.in_consumed
    # a slice that is too small is reported, not overrun
    units[0].out_cap = 1000
    guard = d_out.clone()
    res, _, _ = ctx.decode_units(units, d_in.data_ptr(), d_out.data_ptr(), 0)
    assert res[0].status == M.ST_OUT_FULL and res[1].status == M.ST_OK
    assert torch.equal(d_out[1000:1 << 18], guard[1000:1 << 18])


def test_crc_units_on_device(ctx):
    # milzma_crc_units: CRC-32 / CRC-64(XZ) of device-resident decoded output (validate_block_check,
    # src/decode/xz.rs:292-333) against zlib and the oracle, for lengths around the 64-chunk split and
    # for unaligned output slices
    import torch
    import zlib
    rng = random.Random(8)
    lens = [0, 1, 2, 15, 16, 17, 63, 64, 65, 1023, 1024, 1025, 4097, 65536, 70001, (1 << 20) + 3]
    plains = [bytes(rng.getrandbits(8) for _ in range(min(n, 5000))) * (n // 5000 + 1) for n in lens]
    plains = [p[:n] for p, n in zip(plains, lens)]
    comps = [E.dumb_encode(p, unpacked_size=len(p)) if len(p) < 3000 else W.compress_alone(p, dict_size=65536)
             for p in plains]
    comps = [c[:5] + struct.pack("<Q", len(p)) + c[13:] for c, p in zip(comps, plains)]
    n = len(comps)
    units = (M.Unit * n)()
    in_off, out_off, blobs = 0, 0, []
    for i, c in enumerate(comps):
        u, hl = M.lzma_read_header(c)
        payload = c[hl:]
        skew = (1, 3, 5, 63)[i % 4] if i % 2 else 0  # odd units: unaligned input and output slices
        u.in_off, u.in_len = in_off + skew, len(payload)
        out_off += 7 if i % 2 else 0                 # (the CRC kernel's byte-wise path)
        u.out_off, u.out_cap = out_off, len(plains[i]) + 32
        units[i] = u
        blobs.append(bytes(skew) + payload + bytes((-(len(payload) + skew)) % 256))
        in_off += len(blobs[-1])
        out_off = (out_off + len(plains[i]) + 32 + 255) & ~255
    d_in = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_out = torch.zeros(out_off + 256, dtype=torch.uint8, device="cuda")
    res, _, _ = ctx.decode_units(units, d_in.data_ptr(), d_out.data_ptr(), 0)
    c32, c64 = ctx.crc_units(units, res, d_out.data_ptr(), 0)
    for i in range(n):
        assert res[i].status == M.ST_OK and res[i].out_len == len(plains[i]), (i, res[i].status)
        assert c32[i] == zlib.crc32(plains[i]), (i, lens[i])
        assert c64[i] == orc.crc64(plains[i]), (i, lens[i])
    # a failed unit reports 0
    units[3].out_cap = 4
    res, _, _ = ctx.decode_units(units, d_in.data_ptr(), d_out.data_ptr(), 0)
    c32, c64 = ctx.crc_units(units, res, d_out.data_ptr(), 0)
    assert res[3].status == M.ST_OUT_FULL and c32[3] == 0 and c64[3] == 0
    assert c32[4] == zlib.crc32(plains[4])


# ---- the benchmark's own workloads at full size (BASELINE.json configs[1] / [2] / [3]) -------------------------

@pytest.mark.parametrize("dict_size", [1 << 16, 1 << 23])
def test_full_size_bench_streams_vs_oracle(ctx, dict_size):
    """32 complete 1 MiB text streams of the bench recipe, byte for byte and reader position against the oracle:
    16 ring wraps at 64 KiB (lzbuffer.rs:257-297), matches further back than 64 KiB at 8 MiB (lzma.rs:513-521)."""
    import torch
    n, size = 32, 1 << 20
    comps, plains = W.make_lzma_batch(n, size=size, kind="text", dict_size=dict_size, known_size=True, keep_plain=True)
    units = (M.Unit * n)()
    in_off, blobs = 0, []
    for i, c in enumerate(comps):
        u, hl = M.lzma_read_header(c)
        payload = c[hl:]
        u.in_off, u.in_len = in_off, len(payload)
        u.out_off, u.out_cap = i * size, size
        units[i] = u
        blobs.append(payload + bytes((-len(payload)) % 256))
        in_off += len(blobs[-1])
    d_in = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_out = torch.zeros(n * size, dtype=torch.uint8, device="cuda")
    res, _, _ = ctx.decode_units(units, d_in.data_ptr(), d_out.data_ptr(), 0)
    host = d_out.cpu().numpy().tobytes()
    far = 0
    for i in range(n):
        ref = orc.lzma_decompress(comps[i])
        assert ref.ok and ref.out == plains[i]
        assert res[i].status == M.ST_OK and res[i].out_len == size
        assert host[i * size:(i + 1) * size] == ref.out
        assert res[i].in_consumed + 13 == ref.in_consumed
        far += len(comps[i])
    if dict_size > 1 << 16:  # the bigger window must have been used (better ratio than the 64 KiB recipe's 0.367)
        assert far / (n * size) < 0.36


def test_xz_bench_recipe_device_resident_and_whole_file(ctx):
    """configs[3]: a 4 MiB .xz of four 1 MiB blocks (text | 200 KB random | text: LZMA2 with stored chunks, CRC64).
    Whole-file entry point against the oracle; and the blocks planned with milzma_xz_plan, decoded device-resident,
    their CRC-64 computed on the GPU against the check fields stored in the file."""
    import torch
    import bench
    size = 4 << 20
    plain = bench.xz_plain(5, size)
    comp = W.compress_xz_blocks(plain, block_size=1 << 20, dict_size=1 << 16, check="crc64")
    ref = orc.xz_decompress(comp)
    assert ref.ok and ref.out == plain
    d = ctx.xz(comp)
    same(d, ref)
    units_l, check = M.xz_plan(comp)
    assert check == 4 and len(units_l) == 4
    units = (M.Unit * 4)(*units_l)
    d_in = torch.frombuffer(bytearray(comp + bytes(512)), dtype=torch.uint8).cuda()
    d_out = torch.zeros(size + 512, dtype=torch.uint8, device="cuda")
    res, _, _ = ctx.decode_units(units, d_in.data_ptr(), d_out.data_ptr(), 0)
    _, c64 = ctx.crc_units(units, res, d_out.data_ptr(), 0)
    assert d_out[:size].cpu().numpy().tobytes() == plain
    for k, u in enumerate(units_l):
        assert res[k].status == M.ST_OK and res[k].out_len == 1 << 20 and res[k].in_consumed == u.in_len
        end = u.in_off + u.in_len
        end = (end + 3) & ~3  # block padding
        assert struct.unpack("<Q", comp[end:end + 8])[0] == c64[k] == orc.crc64(plain[k << 20:(k + 1) << 20])


def test_hostile_declared_sizes(ctx):
    """A header may declare any size (lzma.rs:126-161 reads it unchecked): the reference streams and fails when the
    input ends; so must the library, without sizing anything by the declared number."""
    plain = W.make_plain("text", 5000, seed=3)
    comp = W.compress_alone(plain, dict_size=65536, known_size=False)
    for declared in (1 << 32, (1 << 32) - 1000, 1 << 40):
        lie = comp[:5] + struct.pack("<Q", declared) + comp[13:]
        same(ctx.lzma(lie), orc.lzma_decompress(lie))
    good = [W.compress_alone(W.make_plain("text", 3000 + i, seed=i), dict_size=65536, known_size=True) for i in range(6)]
    lie = comp[:5] + struct.pack("<Q", 1 << 36) + comp[13:]
    outs = ctx.lzma_batch(good[:3] + [lie] + good[3:])
    for d, c in zip(outs, good[:3] + [lie] + good[3:]):   # the liar does not take its neighbours down
        same(d, orc.lzma_decompress(c))


@pytest.mark.parametrize("lc,lp,pb", [(3, 0, 4), (3, 0, 3), (4, 0, 4), (2, 2, 3), (0, 4, 0), (1, 2, 4)])
def test_property_classes_at_size(ctx, lc, lp, pb):
    """text streams of 256 KiB in the property classes served by the PB4 variant (16 position states) and the LC4
    instantiation (lc + lp = 4) of the asm kernel, device-resident, against the oracle (rangecoder.rs:153-270 LenDecoder
    per pos_state; lzma.rs:526-561 literal rows)."""
    import torch
    n, size = 6, 1 << 18
    plains = [W.make_plain("text", size, seed=1000 + i + lc * 7 + lp * 5 + pb) for i in range(n)]
    comps = [W.compress_alone(p, dict_size=1 << 16, lc=lc, lp=lp, pb=pb, known_size=bool(i & 1)) for i, p in enumerate(plains)]
    units = (M.Unit * n)()
    in_off, blobs = 0, []
    for i, c in enumerate(comps):
        u, hl = M.lzma_read_header(c)
        payload = c[hl:]
        u.in_off, u.in_len = in_off, len(payload)
        u.out_off, u.out_cap = i * size, size
        units[i] = u
        blobs.append(payload + bytes((-len(payload)) % 256))
        in_off += len(blobs[-1])
    d_in = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_out = torch.zeros(n * size, dtype=torch.uint8, device="cuda")
    res, _, launches = ctx.decode_units(units, d_in.data_ptr(), d_out.data_ptr(), 0)
    assert launches == 1
    host = d_out.cpu().numpy().tobytes()
    for i in range(n):
        ref = orc.lzma_decompress(comps[i])
        assert ref.ok and ref.out == plains[i]
        assert res[i].status == M.ST_OK and res[i].out_len == size and host[i * size:(i + 1) * size] == ref.out
        assert res[i].in_consumed + 13 == ref.in_consumed


def test_decode_units_async_overlaps_and_matches_sync(ctx):
    """milzma_decode_units_async / _wait: returns before the GPU is done, the caller's unit array may go away, one batch
    per context, same results as the synchronous call"""
    import torch
    comps, plains = W.make_lzma_batch(8, size=1 << 18, kind="text", dict_size=65536, known_size=True, keep_plain=True)
    n = len(comps)
    units = (M.Unit * n)()
    in_off, blobs = 0, []
    for i, c in enumerate(comps):
        u, hl = M.lzma_read_header(c)
        payload = c[hl:]
        u.in_off, u.in_len, u.out_off, u.out_cap = in_off, len(payload), i << 18, 1 << 18
        units[i] = u
        blobs.append(payload + bytes((-len(payload)) % 256))
        in_off += len(blobs[-1])
    d_in = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_out = torch.zeros(n << 18, dtype=torch.uint8, device="cuda")
    st = torch.cuda.Stream()
    ctx.decode_units_async(units, d_in.data_ptr(), d_out.data_ptr(), st.cuda_stream)
    with pytest.raises(M.InfraError):
        ctx.decode_units_async(units, d_in.data_ptr(), d_out.data_ptr(), st.cuda_stream)  # one batch per context
    del units  # (the library keeps its own copy of the descriptors)
    res, ms, launches = ctx.decode_units_wait(n)
    assert launches == 1 and ms > 0
    host = d_out.cpu().numpy().tobytes()
    for i in range(n):
        assert res[i].status == M.ST_OK and host[i << 18:(i + 1) << 18] == plains[i]
    with pytest.raises(M.InfraError):
        ctx.decode_units_wait(n)  # nothing in flight


def test_differential_fuzz_round(ctx):
    """one round of experiments/parity_fuzz.py (1260 damaged / odd .lzma, LZMA2 and .xz inputs and option combinations
    through the batch and single-file entry points) against the oracle; profiles/r02_parity_fuzz.txt records the
    75 000-case search it comes from (fuzz/fuzz_targets/decompress_*.rs, compare_*.rs of the reference)"""
    import importlib.util
    import random as _random
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("parity_fuzz", os.path.join(root, "experiments", "parity_fuzz.py"))
    pf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pf)
    rng = _random.Random(20260927)
    lz = pf.pool_lzma(rng)
    l2, xz = pf.pool_lzma2_xz(rng)
    bad = 0
    cases = [pf.damage(rng, pf.edit_lzma_header(rng, rng.choice(lz)) if rng.random() < 0.5 else rng.choice(lz), 13) for _ in range(600)]
    for comp, d in zip(cases, ctx.lzma_batch(cases)):
        bad += not pf.same("lzma", comp, d, orc.lzma_decompress(comp))
    cases = [pf.damage(rng, rng.choice(l2), 0) for _ in range(300)]
    for comp, d in zip(cases, ctx.lzma2_batch(cases)):
        bad += not pf.same("lzma2", comp, d, orc.lzma2_decompress(comp))
    cases = [pf.damage(rng, rng.choice(xz), 0) for _ in range(300)]
    for comp, d in zip(cases, ctx.xz_batch(cases)):
        bad += not pf.same("xz", comp, d, orc.xz_decompress(comp))
    assert bad == 0


# ---- several GPUs behind one handle (milzma_multi_*): with the devices present (one here) the results must be those of the
#      single-device entry points, whatever the partition ----------------------------------------------------------------
@pytest.mark.parametrize("shares", [1, 3])
def test_multi_device_entry_points_match_single_device(ctx, monkeypatch, shares):
    """shares = 3: MILZMA_MULTI_REPLICAS makes the handle hold three contexts on device 0, each treated as a device of its own, so
    that the partition, the per-device workers and the merge of their results run with several shares on a one-GPU box."""
    import ctypes
    if shares > 1:
        monkeypatch.setenv("MILZMA_MULTI_REPLICAS", str(shares))
    m = M.MultiContext(1)          # mask 1 = device 0
    try:
        assert m.devices == [0] * shares
        rng = random.Random(12)
        plains = [W.make_plain(rng.choice(["text", "random", "repeat"]), rng.randint(1, 90000), seed=i) for i in range(24)]
        comps = [W.compress_alone(p, dict_size=1 << 16, known_size=(i % 2 == 0)) for i, p in enumerate(plains)]
        comps[5] = comps[5][:len(comps[5]) // 2]                     # a truncated one and a damaged one ride along
        comps[9] = comps[9][:40] + b"\xff" + comps[9][41:]
        # whole-file batches
        for a, b in zip(m.lzma_batch(comps), ctx.lzma_batch(comps)):
            assert (a.kind, a.msg, a.data, a.in_consumed) == (b.kind, b.msg, b.data, b.in_consumed)
        xzs = [lzma.compress(p, format=lzma.FORMAT_XZ, check=lzma.CHECK_CRC64) for p in plains[:8]] + [gold("good-1-lzma2-4.xz")]
        xzs[3] = xzs[3][:-1] + b"\x00"
        for a, b in zip(m.xz_batch(xzs), ctx.xz_batch(xzs)):
            assert (a.kind, a.msg, a.data) == (b.kind, b.msg, b.data)
        raws = [lzma.compress(p, format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "dict_size": 1 << 16}]) for p in plains[:6]]
        for a, b in zip(m.lzma2_batch(raws), ctx.lzma2_batch(raws)):
            assert (a.kind, a.msg, a.data, a.in_consumed) == (b.kind, b.msg, b.data, b.in_consumed)
        # unit level, host-resident: the multi call packs each device's share itself
        units, blob, out_off = [], bytearray(), 0
        for comp, p in zip(comps, plains):
            u, hl = M.lzma_read_header(comp)
            u.in_off, u.in_len, u.out_off, u.out_cap = len(blob) + 3, len(comp) - hl, out_off, len(p) + 300
            blob += b"\x00\x00\x00" + comp[hl:]
            out_off += len(p) + 300 + 17
            units.append(u)
        arr = (M.Unit * len(units))(*units)
        r1, o1 = m.decode_units_host(arr, bytes(blob), out_off)
        r2, o2 = ctx.decode_units_host(arr, bytes(blob), out_off)
        for u, a, b in zip(units, r1, r2):
            assert (a.status, a.out_len, a.out_flushed, a.in_consumed, a.err_a, a.err_b) == (b.status, b.out_len, b.out_flushed, b.in_consumed, b.err_a, b.err_b)
            n = min(a.out_len, u.out_cap)
            assert o1[u.out_off:u.out_off + n] == o2[u.out_off:u.out_off + n]
        # unit level, device-resident
        import torch
        d_in = torch.frombuffer(bytearray(blob) + bytearray(512), dtype=torch.uint8).cuda()
        d_out = torch.zeros(out_off + 512, dtype=torch.uint8, device="cuda")
        r3 = m.decode_units(arr, [i % shares for i in range(len(units))], [d_in.data_ptr()] * shares, [d_out.data_ptr()] * shares)
        got = d_out.cpu().numpy().tobytes()
        for u, a, b in zip(units, r3, r2):
            assert (a.status, a.out_len, a.in_consumed) == (b.status, b.out_len, b.in_consumed)
            n = min(a.out_len, u.out_cap)
            assert got[u.out_off:u.out_off + n] == bytes(o2[u.out_off:u.out_off + n])
        assert all(ms > 0 for ms in m.kernel_ms())
        # one ingest point: everything resident on device index `root`; the other devices' shares are packed, cross device to device
        # (hipMemcpyPeer), are decoded there and come back into the slices the descriptors name
        # ... by every way back the entry has: the decoding waves' own stores into the root's slices (default), one copy behind the decode
        # (MILZMA_ROOTED_STREAM=0), and the same copy because peer access to the root is denied (MILZMA_ROOTED_PEER=0: what a device that
        # cannot reach the root's memory does) -- the switches are read per call
        for way in ({}, {"MILZMA_ROOTED_STREAM": "0"}, {"MILZMA_ROOTED_PEER": "0"}):
            for k, v in way.items():
                monkeypatch.setenv(k, v)
            for root in range(shares):
                d_out2 = torch.zeros(out_off + 512, dtype=torch.uint8, device="cuda")
                torch.cuda.synchronize()     # (the library's own streams do not order themselves behind torch's)
                r4, (t_in, t_dec, t_out) = m.decode_units_rooted(root, arr, d_in.data_ptr(), d_out2.data_ptr())
                got = d_out2.cpu().numpy().tobytes()
                for u, a, b in zip(units, r4, r2):
                    assert (a.status, a.out_len, a.out_flushed, a.in_consumed, a.err_a, a.err_b) == (b.status, b.out_len, b.out_flushed, b.in_consumed, b.err_a, b.err_b)
                    n = min(a.out_len, u.out_cap)
                    assert got[u.out_off:u.out_off + n] == bytes(o2[u.out_off:u.out_off + n]), (way, root)
                assert t_dec > 0 and (shares == 1 or (t_in > 0 and t_out > 0))
            for k in way:
                monkeypatch.delenv(k)
        with pytest.raises(M.InfraError):
            m.decode_units_rooted(shares, arr, d_in.data_ptr(), d_out.data_ptr())     # no such root
        # descriptor checks as in the single-device host call
        bad = (M.Unit * 1)(units[0])
        bad[0].in_len = len(blob) + 1
        with pytest.raises(M.InfraError):
            m.decode_units_host(bad, bytes(blob), out_off)
        with pytest.raises(M.InfraError):
            m.decode_units(arr, [shares] * len(units), [d_in.data_ptr()] * shares, [d_out.data_ptr()] * shares)   # no such device index
    finally:
        m.close()
    with pytest.raises(M.InfraError):
        M.MultiContext(1 << 40)        # a device that is not there: no partial sets


def test_xz_batch_mixes_planned_and_on_demand_blocks_with_a_large_output(ctx):
    """files whose Index can be planned ahead next to files that must be decoded on demand while the planned output
    (tens of MiB, several 64 MiB-chunks of D2H) is still coming back: the on-demand decodes reuse the context's output
    buffer and must not disturb bytes that have not been copied yet (advisor finding, round 2)."""
    big = [W.make_plain("text", 3 << 20, seed=100 + i) for i in range(24)]
    files = [W.compress_xz_blocks(p, block_size=1 << 20, dict_size=1 << 16, check="crc64") for p in big]
    odd_plain = W.make_plain("text", 200000, seed=7)
    odd = lzma.compress(odd_plain, format=lzma.FORMAT_XZ, check=lzma.CHECK_CRC32)
    unplannable = odd[:-2] + b"YY"                       # footer magic broken: no plan, walked (and failed) on demand
    mixed = [odd] + files[:12] + [unplannable] + files[12:] + [odd]
    decs = ctx.xz_batch(mixed)
    want = [odd_plain] + big[:12] + [None] + big[12:] + [odd_plain]
    for d, w in zip(decs, want):
        if w is None:
            same(d, orc.xz_decompress(unplannable))
        else:
            assert d.ok and d.data == w


def test_whole_file_batches_async_two_in_flight(ctx):
    """milzma_*_decompress_batch_async / milzma_batch_wait: two contexts, two calls in flight, the results of the synchronous calls"""
    other = M.Context(0)
    try:
        plains = [W.make_plain("text", 30000 + 977 * i, seed=50 + i) for i in range(40)]
        a = [W.compress_alone(p, dict_size=1 << 16, known_size=True) for p in plains[:20]]
        b = [lzma.compress(p, format=lzma.FORMAT_XZ, check=lzma.CHECK_CRC32) for p in plains[20:]]
        a[3] = a[3][:100]
        ctx.batch_async("lzma", a)
        other.batch_async("xz", b)
        with pytest.raises(M.InfraError):
            ctx.batch_async("lzma", a)                 # one batch in flight per context
        ra, rb = ctx.batch_wait(), other.batch_wait()
        for x, y in zip(ra, ctx.lzma_batch(a)):
            assert (x.kind, x.msg, x.data, x.in_consumed) == (y.kind, y.msg, y.data, y.in_consumed)
        for x, p in zip(rb, plains[20:]):
            assert x.ok and x.data == p
    finally:
        other.close()


def test_large_batch_calls_are_grouped_over_lanes(ctx, monkeypatch):
    """>= 8192 decode units in one whole-file call: the library cuts it into chip-sized groups that alternate between the context and
    a second context of the same device (uploads in group order; the download of one group under the other's kernel).  With
    MILZMA_LANES=4: groups of 512..2048 units on four contexts at once (8300 files = 5 groups, 1100 files = 2).  Same results as
    small calls, every file in place."""
    plains = [W.make_plain("text" if i % 3 else "random", 200 + (i * 37) % 900, seed=i) for i in range(64)]
    comps = [W.compress_alone(p, dict_size=1 << 12, known_size=(i % 2 == 0)) for i, p in enumerate(plains)]
    comps[7] = comps[7][:30]                                     # a truncated one in every 64
    files = [comps[i % 64] for i in range(8300)]
    ref = [orc.lzma_decompress(c) for c in comps]

    def check(decs):
        for i, d in enumerate(decs):
            r = ref[i % 64]
            assert (d.kind, d.msg, d.data) == (r.kind, r.msg, r.out), i

    check(ctx.lzma_batch(files))
    xz = [lzma.compress(p * 3, format=lzma.FORMAT_XZ, check=lzma.CHECK_CRC32) for p in plains[:32]]
    many = [xz[i % 32] for i in range(8200)]
    for i, d in enumerate(ctx.xz_batch(many)):
        assert d.ok and d.data == plains[i % 32] * 3, i
    monkeypatch.setenv("MILZMA_LANES", "4")
    check(ctx.lzma_batch(files))
    check(ctx.lzma_batch(files[:1100]))
    for i, d in enumerate(ctx.xz_batch(many[:1500])):
        assert d.ok and d.data == plains[i % 32] * 3, i


def test_time_sliced_launches_match_ordinary_ones(monkeypatch):
    """The persistent, time-sliced form of the fast kernel (launches that are not a whole number of chip-fulls): every unit parked at
    every quantum and taken up again by whichever wave is free (MILZMA_SLICE=2, a 2 KiB quantum) -- and the automatic form, 4100
    units on a chip that holds 4096 waves -- give the oracle's results for good, truncated and damaged .lzma streams of all
    property classes, LZMA2 streams with every chunk kind, and .xz files."""
    monkeypatch.delenv("MILZMA_KERNEL", raising=False)   # (its own contexts; the module's `ctx` fixture may be alive with MILZMA_KERNEL=generic)
    rng = random.Random(77)
    plains, comps = [], []
    for i in range(40):
        kind = rng.choice(["text", "random", "repeat", "zeros"])
        lc, lp, pb = rng.choice([(3, 0, 2), (0, 2, 0), (1, 1, 4), (4, 0, 2), (2, 2, 3), (0, 0, 0)])
        p = W.make_plain(kind, rng.randint(1, 60000), seed=1000 + i)
        c = W.compress_alone(p, dict_size=rng.choice([4096, 1 << 16, 1 << 20]), known_size=(i % 2 == 0), lc=lc, lp=lp, pb=pb)
        if i % 7 == 3:
            c = c[:len(c) * 3 // 5]                                  # truncated
        if i % 7 == 5:
            k = 13 + len(c) // 2
            c = c[:k] + bytes([c[k] ^ 0x5A]) + c[k + 1:]            # damaged in the middle
        plains.append(p)
        comps.append(c)
    ref = [orc.lzma_decompress(c) for c in comps]
    raws = [lzma.compress(W.make_plain("text", 3 << 20, seed=5), format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "dict_size": 1 << 16}]),
            lzma.compress(os.urandom(300000) + b"abc" * 50000, format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "dict_size": 1 << 20}])]
    raws.append(raws[0][:len(raws[0]) // 2])
    ref2 = [orc.lzma2_decompress(r) for r in raws]
    xzs = [lzma.compress(p * 3, format=lzma.FORMAT_XZ, check=lzma.CHECK_CRC64) for p in plains[:6]]

    def check(c):
        for d, r in zip(c.lzma_batch(comps), ref):
            assert (d.kind, d.msg, d.data, d.in_consumed) == (r.kind, r.msg, r.out, r.in_consumed)
        for d, r in zip(c.lzma2_batch(raws), ref2):
            assert (d.kind, d.msg, d.data, d.in_consumed) == (r.kind, r.msg, r.out, r.in_consumed)
        for d, p in zip(c.xz_batch(xzs), plains):
            assert d.ok and d.data == p * 3

    monkeypatch.setenv("MILZMA_SLICE", "2")
    monkeypatch.setenv("MILZMA_QUANTUM", "2048")
    c = M.Context(0)
    try:
        check(c)
    finally:
        c.close()
    monkeypatch.delenv("MILZMA_SLICE")
    monkeypatch.delenv("MILZMA_QUANTUM")
    c = M.Context(0)       # automatic: 4100 streams do not fill a whole number of rounds -> sliced, 128 KiB quanta
    try:
        many = [comps[i % 40] for i in range(4100)]
        for i, d in enumerate(c.lzma_batch(many)):
            r = ref[i % 40]
            assert (d.kind, d.msg, d.data, d.in_consumed) == (r.kind, r.msg, r.out, r.in_consumed), i
    finally:
        c.close()


# ---- growable output: units parked at the end of their slice and resumed in a larger one (milzma_decode_units_ex) ----------------

def _resume_until_done(ctx, units, d_in, make_out, first_out, max_rounds=40):
    """GROW, then as long as units are parked: a fresh, larger output buffer, the parked units' bytes moved into their new slices on
    the device, RESUME.  Returns (results, final output tensor, rounds, parked-ever count)."""
    import torch
    n = len(units)
    d_out = first_out
    res, _, _ = ctx.decode_units_ex(units, d_in.data_ptr(), d_out.data_ptr(), M.DECODE_GROW)
    rounds, ever = 0, 0
    while True:
        parked = [i for i in range(n) if res[i].status == M.ST_OUT_FULL and res[i].err_a == M.PARKED]
        if not parked:
            return res, d_out, rounds, ever
        ever += len(parked)
        rounds += 1
        assert rounds < max_rounds
        # everything gets a new place in a new buffer (finished units keep their bytes too: the test compares at the end)
        old = [(units[i].out_off, units[i].out_cap) for i in range(n)]
        total = 0
        for i in range(n):
            cap = units[i].out_cap
            if i in set(parked):
                assert res[i].out_len <= cap, (res[i].out_len, cap)
                cap = cap * 3 + 512
            units[i].out_off, units[i].out_cap = total, cap
            total += (cap + 255) & ~255
        new_out = make_out(total)
        lens = [min(res[i].out_len, old[i][1]) for i in range(n)]
        ctx.move_units(d_out.data_ptr(), [o[0] for o in old], new_out.data_ptr(), [units[i].out_off for i in range(n)], lens)
        d_out = new_out
        res, _, _ = ctx.decode_units_ex(units, d_in.data_ptr(), d_out.data_ptr(), M.DECODE_RESUME, results=res)


def test_growable_units_park_and_resume(ctx):
    """Unit level, device resident: RAW streams of every class (unknown and known sizes, all property classes of the fast kernels) and
    LZMA2 units (compressed, stored and dictionary-reset chunks) start in slices of a few hundred bytes to a few KiB and are parked /
    moved / resumed until they end: bytes, status, reader position equal the oracle's; nothing is decoded twice (the sum of the
    kernels' work is checked through in_consumed never going backwards)."""
    import torch
    rng = random.Random(4)
    comps, refs, kinds = [], [], []
    for i in range(36):
        kind = ["text", "zeros", "repeat", "random"][i % 4]
        lc, lp, pb = [(3, 0, 2), (0, 2, 0), (1, 1, 4), (4, 0, 2), (2, 2, 3), (3, 0, 2)][i % 6]
        p = W.make_plain(kind, rng.randint(1, 90000), seed=300 + i)
        c = W.compress_alone(p, dict_size=rng.choice([4096, 1 << 16]), known_size=(i % 3 == 0), lc=lc, lp=lp, pb=pb)
        if i % 9 == 7:
            c = c[:len(c) * 2 // 3]                      # truncated: the error must be the oracle's, wherever the unit was parked before
        comps.append(c)
        refs.append(orc.lzma_decompress(c))
        kinds.append(M.KIND_RAW_LZMA)
    raws = [lzma.compress(W.make_plain("text", 200000, seed=9) + os.urandom(70000) + b"xyz" * 30000, format=lzma.FORMAT_RAW,
                          filters=[{"id": lzma.FILTER_LZMA2, "dict_size": 1 << 16}]),
            lzma.compress(os.urandom(150000), format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "dict_size": 1 << 12}])]
    for r in raws:
        comps.append(r)
        refs.append(orc.lzma2_decompress(r))
        kinds.append(M.KIND_LZMA2)
    n = len(comps)
    units = (M.Unit * n)()
    in_off, blobs, total = 0, [], 0
    for i, c in enumerate(comps):
        if kinds[i] == M.KIND_RAW_LZMA:
            u, hl = M.lzma_read_header(c)
        else:
            u, hl = M.Unit(), 0
            u.kind = M.KIND_LZMA2
        payload = c[hl:]
        u.in_off, u.in_len = in_off, len(payload)
        u.out_cap = [300, 1000, 4096, 70000][i % 4]
        u.out_off = total
        total += (u.out_cap + 255) & ~255
        units[i] = u
        pad = (-len(payload)) % 256
        blobs.append(payload + bytes(pad))
        in_off += len(payload) + pad
    d_in = torch.frombuffer(bytearray(b"".join(blobs) + bytes(512)), dtype=torch.uint8).cuda()
    make = lambda nbytes: torch.zeros(nbytes + 512, dtype=torch.uint8, device="cuda")
    res, d_out, rounds, ever = _resume_until_done(ctx, units, d_in, make, make(total))
    host = d_out.cpu().numpy().tobytes()
    fast = os.environ.get("MILZMA_KERNEL") != "generic"
    for i in range(n):
        ref, r = refs[i], res[i]
        kind, msg = M.result_message(r, kinds[i])
        if r.status == M.ST_OUT_FULL:       # only the generic kernel's units may end like this (they cannot be parked)
            assert not fast and r.err_a == 0, (i, r.status, r.err_a)
            continue
        assert (kind, msg) == (ref.kind, ref.msg), (i, msg, ref.msg)
        got = host[units[i].out_off:units[i].out_off + min(r.out_flushed, units[i].out_cap)]
        assert got == ref.out, (i, len(got), len(ref.out))
        if ref.ok:
            assert r.in_consumed + (13 if kinds[i] == M.KIND_RAW_LZMA else 0) == ref.in_consumed, i
    if fast:
        assert ever >= n // 2 and rounds >= 3, (ever, rounds)


def _rows_stream(lc, lp, pb, size, seed, known):
    """a .lzma stream of a property set liblzma cannot write (greedy LZ parse + the tests' symbol encoder)"""
    plain = W.make_plain("text", size - 2048, seed=seed) + bytes(range(256)) * 8
    enc = E.LzmaSymbolEncoder(lc, lp, pb)
    enc.encode(E.lz_parse(plain, dict_size=1 << 16))
    if not known:
        enc.encode([("marker",)])
    return E.lzma_header(lc, lp, pb, 1 << 16, len(plain) if known else None) + enc.finish()


def test_grow_batch_of_mixed_literal_row_classes(ctx):
    """ADVICE r4 (two high findings): the literal rows of lc + lp >= 4 units live in ONE slab per batch, indexed by unit with one stride.
    A GROW batch in which the units with the MOST rows (lc 8, known size, roomy slices) finish at once while units with fewer rows (lc + lp
    4 .. 6, unknown size, tiny slices) park: the RESUME launches see only the parked subset -- the stride must stay the one the first
    launch chose (it used to be re-derived: rows read out of another unit's) --, and an LZMA2 unit whose chunk asks for lc + lp = 4 is
    promoted into the slab's class by a launch of its own in the middle (it used to wipe the whole slab: parked units resumed on fresh
    probabilities).  Bytes, verdicts and reader positions are the oracle's; then the same files through the whole-file batch call."""
    import torch
    specs = [((8, 0, 2), True), ((4, 0, 2), False), ((8, 0, 0), True), ((1, 3, 0), False), ((6, 0, 2), False), ((5, 2, 4), False),
             ((3, 0, 2), False), ((8, 4, 4), True), ((4, 0, 4), False), ((0, 4, 1), False)]
    comps, refs, kinds = [], [], []
    for k, ((lc, lp, pb), known) in enumerate(specs):
        c = _rows_stream(lc, lp, pb, 60000 + 3000 * k, 1200 + k, known)
        if k == 5:
            c = c[:len(c) * 3 // 4]                      # a truncated one: the oracle's error, wherever it was parked before
        comps.append(c)
        refs.append(orc.lzma_decompress(c))
        kinds.append(M.KIND_RAW_LZMA)
    flt = [{"id": lzma.FILTER_LZMA2, "dict_size": 1 << 16, "lc": 2, "lp": 2, "pb": 2}]
    l2 = lzma.compress(W.make_plain("text", 90000, seed=77), format=lzma.FORMAT_RAW, filters=flt)
    comps.append(l2)
    refs.append(orc.lzma2_decompress(l2))
    kinds.append(M.KIND_LZMA2)
    n = len(comps)
    units = (M.Unit * n)()
    in_off, blobs, total = 0, [], 0
    for i, c in enumerate(comps):
        if kinds[i] == M.KIND_RAW_LZMA:
            u, hl = M.lzma_read_header(c)
        else:
            u, hl = M.Unit(), 0
            u.kind = M.KIND_LZMA2
        payload = c[hl:]
        u.in_off, u.in_len = in_off, len(payload)
        known = kinds[i] == M.KIND_RAW_LZMA and specs[i][1]
        u.out_cap = len(refs[i].out) + 64 if known else [400, 1500, 5000][i % 3]
        u.out_off = total
        total += (u.out_cap + 255) & ~255
        units[i] = u
        pad = (-len(payload)) % 256
        blobs.append(payload + bytes(pad))
        in_off += len(payload) + pad
    d_in = torch.frombuffer(bytearray(b"".join(blobs) + bytes(512)), dtype=torch.uint8).cuda()
    make = lambda nbytes: torch.zeros(nbytes + 512, dtype=torch.uint8, device="cuda")
    res, d_out, rounds, ever = _resume_until_done(ctx, units, d_in, make, make(total))
    host = d_out.cpu().numpy().tobytes()
    fast = os.environ.get("MILZMA_KERNEL") != "generic"
    for i in range(n):
        ref, r = refs[i], res[i]
        kind, msg = M.result_message(r, kinds[i])
        if r.status == M.ST_OUT_FULL:       # only the generic kernel's units may end like this (they cannot be parked)
            assert not fast and r.err_a == 0, (i, r.status, r.err_a)
            continue
        assert (kind, msg) == (ref.kind, ref.msg), (i, msg, ref.msg)
        got = host[units[i].out_off:units[i].out_off + min(r.out_flushed, units[i].out_cap)]
        assert got == ref.out, (i, len(got), len(ref.out))
        if ref.ok:
            assert r.in_consumed + (13 if kinds[i] == M.KIND_RAW_LZMA else 0) == ref.in_consumed, i
    if fast:
        assert ever >= 6 and rounds >= 3, (ever, rounds)
        # a RESUME whose descriptor does not fit the parked state is refused (nothing launched): park once more, then lie about a slice
        for i in range(n - 1):
            if not specs[i][1]:
                units[i].out_cap = 300
        res2, _, _ = ctx.decode_units_ex(units, d_in.data_ptr(), d_out.data_ptr(), M.DECODE_GROW)
        parked = [i for i in range(n) if res2[i].status == M.ST_OUT_FULL and res2[i].err_a == M.PARKED]
        assert parked
        units[parked[0]].out_cap = max(1, res2[parked[0]].out_len - 8)
        with pytest.raises(M.InfraError):
            ctx.decode_units_ex(units, d_in.data_ptr(), d_out.data_ptr(), M.DECODE_RESUME, results=res2)
    # the whole-file call: unknown-size members get a guessed slice and are grown (park / move / resume inside the library)
    many = [comps[i % (n - 1)] for i in range(3 * (n - 1))]
    for i, d in enumerate(ctx.lzma_batch(many)):
        same(d, refs[i % (n - 1)])


def test_thousands_of_wrong_guesses_are_resumed_not_redecoded(ctx):
    """4200 marker-terminated streams whose output is a thousand times their input (zeros / repeats: the whole-file path's first
    slice, 6 x the payload or 64 KiB, is wrong for every one of them) next to text streams whose guess holds: the batch call parks
    and resumes them -- bytes and reader position are the oracle's for every file."""
    plains = [W.make_plain("zeros", 150_000 + 4096 * i, seed=i) for i in range(6)] + \
             [W.make_plain("repeat", 120_000 + 1000 * i, seed=40 + i) for i in range(6)] + \
             [W.make_plain("text", 30_000, seed=90), W.make_plain("text", 90_000, seed=91)]
    comps = [W.compress_alone(p, dict_size=65536, known_size=False) for p in plains]
    refs = [orc.lzma_decompress(c) for c in comps]
    many = [comps[i % len(comps)] for i in range(4200)]
    for i, d in enumerate(ctx.lzma_batch(many)):
        r = refs[i % len(comps)]
        assert (d.kind, d.msg, d.in_consumed) == (r.kind, r.msg, r.in_consumed), i
        assert d.data == r.out, i
    # a declared size that lies (too small a guess is impossible then: the header is believed up to what the payload can plausibly
    # expand to) and single-file calls take the same path
    d = ctx.lzma(comps[0])
    assert d.ok and d.data == plains[0]
    raw = lzma.compress(plains[7] * 4, format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "dict_size": 1 << 16}])
    d = ctx.lzma2(raw)
    assert d.ok and d.data == plains[7] * 4


# ---- streamed launches: the whole-file batch calls whose waves write their output to the caller's buffers themselves -------------

@pytest.mark.parametrize("pinned", ["1", "0"])
def test_streamed_whole_file_batches_match_the_oracle(monkeypatch, pinned):
    """The streamed form of the batch calls (one time-sliced launch, output spans written to the host by the waves --
    into page-locked result buffers, or with MILZMA_PINNED_OUT=0 into a staging buffer a host thread copies from --, the .lzma input
    in two parts) is the default only for large uniform batches; MILZMA_STREAM_MIN sends small ones down the same path: equal-sized
    streams of every class, unknown sizes (parked and resumed afterwards), truncated and damaged ones, LZMA2 streams and .xz files
    (good, with a lying Index, the 34 malformed ones) must come out exactly as the oracle says."""
    import test_xz_literals as X
    monkeypatch.delenv("MILZMA_KERNEL", raising=False)   # (the module's `ctx` fixture may be alive with MILZMA_KERNEL=generic: this test brings its own context)
    monkeypatch.setenv("MILZMA_STREAM_MIN", "2,1,1")   # (third field: ragged batches too -- the unknown-size members' guessed slices differ by kind)
    monkeypatch.setenv("MILZMA_SPAN", "65536")
    monkeypatch.setenv("MILZMA_PINNED_OUT", pinned)
    rng = random.Random(8)
    c = M.Context(0)
    try:
        for known in (True, False):
            plains = [W.make_plain(["text", "random", "repeat", "text"][i % 4], 300_000, seed=600 + i) for i in range(40)]
            comps = [W.compress_alone(p, dict_size=1 << 16, known_size=known) for p in plains]
            comps[7] = comps[7][:len(comps[7]) // 2]
            comps[11] = comps[11][:2000] + bytes([comps[11][2000] ^ 0x40]) + comps[11][2001:]
            decs = c.lzma_batch(comps)
            assert c.last_call_paths() & M.PATH_STREAMED, c.last_call_paths()   # (the switch is read per call since round 5: it used to be cached)
            for comp, d in zip(comps, decs):
                r = orc.lzma_decompress(comp)
                assert (d.kind, d.msg, d.data, d.in_consumed) == (r.kind, r.msg, r.out, r.in_consumed)
        raws = [lzma.compress(W.make_plain("text", 250_000, seed=700 + i) + os.urandom(70_000), format=lzma.FORMAT_RAW,
                              filters=[{"id": lzma.FILTER_LZMA2, "dict_size": 1 << 16}]) for i in range(12)]
        decs = c.lzma2_batch(raws)
        assert c.last_call_paths() & M.PATH_STREAMED, c.last_call_paths()
        for comp, d in zip(raws, decs):
            r = orc.lzma2_decompress(comp)
            assert (d.kind, d.msg, d.data, d.in_consumed) == (r.kind, r.msg, r.out, r.in_consumed)
        # .xz: multi-block files of one block size; a file whose Index understates a block; the malformed ones
        blocks = [W.make_plain("text", 200_000, seed=800 + i) for i in range(24)]
        xzs = [W.compress_xz_blocks(b"".join(blocks[3 * i:3 * i + 3]), block_size=200_000, check="crc64") for i in range(8)]
        # (the Index understates block 1 by 8 bytes: the unit still fits its slice and ends OK, but longer than its place in the file's
        #  buffer -- the streamed path must deliver the block's TRUE bytes before the walk reports the Index; and by 4000: OUT_FULL)
        for lie in (8, 4000):
            bl = [X.block(blocks[j], check=4) for j in range(3)]
            idx = X.index([(bl[0][1], bl[0][2]), (bl[1][1], bl[1][2] - lie), (bl[2][1], bl[2][2])])
            xzs.append(X.xz_file(check=4, blocks=bl, idx=idx))
        xzs += [data for _name, (data, _k, _m) in sorted(X.CASES.items())]
        decs = c.xz_batch(xzs)
        assert c.last_call_paths() & M.PATH_STREAMED, c.last_call_paths()
        for comp, d in zip(xzs, decs):
            r = orc.xz_decompress(comp)
            assert (d.kind, d.msg, d.data) == (r.kind, r.msg, r.out)
    finally:
        c.close()
