"""The generated gfx950 symbol loop (tools/gen_fast_loop.py -> lzma_rs_amd/csrc/fast_loop_asm.inc), executed
instruction by instruction on the CPU by tools/emu (a functional emulator of the instruction subset the loop
uses) and compared with the oracle: bytes, final status and reader position.  This is how a generator change
is checked without a GPU; the `-m gpu` parity tests remain the proof on hardware.  Reference behaviour:
src/decode/lzma.rs:255-593, src/decode/rangecoder.rs, src/decode/lzbuffer.rs:167-321."""
import lzma
import os
import random
import struct
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "emu"))

import asmprog  # noqa: E402
import oracle_py as orc  # noqa: E402
from lzma_rs_amd import workloads as W  # noqa: E402


def _hdr(comp):
    props = comp[0]
    ds = struct.unpack("<I", comp[1:5])[0]
    us = struct.unpack("<Q", comp[5:13])[0]
    return props % 9, (props // 9) % 5, props // 45, max(ds, 4096), (None if us == 0xFFFFFFFFFFFFFFFF else us)


@pytest.fixture(scope="module")
def loops():
    m = {True: asmprog.AsmLoop(lp0=True), False: asmprog.AsmLoop(lp0=False), "lc4": asmprog.AsmLoop(lp0=False, pb4=True, hbm=True),
         "hb0": asmprog.AsmLoop(lp0=True, pb4=False, hbm=True),   # (round 6: lc >= 4 with lp == 0 and pb <= 2)
         "pb4": asmprog.AsmLoop(lp0=False, pb4=True)}     # ("lc4": lc + lp = 4 runs in the HBM variant since round 4)
    yield m
    for a in set(m.values()):
        a.close()


def _pick(loops, lc, lp, pb):
    """the loop variant AsmDecoder::process runs for this property set"""
    if lc + lp >= 4:
        return loops["hb0" if lp == 0 and pb <= 2 else "lc4"]
    return loops["pb4" if pb > 2 else lp == 0]


def _check(loops, comp, plain):
    lc, lp, pb, ds, us = _hdr(comp)
    r = _pick(loops, lc, lp, pb).decode_raw(comp[13:], lc, lp, pb, ds, us, out_cap=max(len(plain), 1))
    ref = orc.lzma_decompress(comp)
    assert ref.out == plain
    assert r["status"] == "OK" and r["out"] == plain
    assert r["in_consumed"] + 13 == ref.in_consumed


@pytest.mark.parametrize("kind", ["text", "random", "repeat", "zeros"])
@pytest.mark.parametrize("known", [True, False])
def test_emulated_loop_bench_classes(loops, kind, known):
    for size in (1, 100, 5000, 70000):
        plain = W.make_plain(kind, size, seed=W.SEED0 ^ size)
        _check(loops, W.compress_alone(plain, dict_size=65536, known_size=known), plain)


@pytest.mark.parametrize("lc,lp,pb", [(3, 0, 2), (0, 0, 0), (1, 2, 1), (2, 1, 2), (0, 3, 0), (3, 0, 0), (3, 0, 4), (3, 0, 3),
                                      (0, 0, 4), (1, 2, 3), (2, 1, 4), (0, 3, 4), (4, 0, 4), (4, 0, 0), (0, 4, 2), (2, 2, 3),
                                      (1, 3, 4), (3, 1, 2)])
def test_emulated_loop_props_and_near_distances(loops, lc, lp, pb):
    """small dictionary: exercises the reverse-tree distance slots (4..13) and rep matches"""
    rnd = random.Random(lc * 100 + lp * 10 + pb)
    plain = (W.make_plain("text", 30000, seed=lc * 100 + lp * 10 + pb) + bytes(rnd.randrange(256) for _ in range(3000)) + b"abc" * 5000 +
             bytes(4000) + b"abcdefg" * 2000)
    filt = [{"id": lzma.FILTER_LZMA1, "lc": lc, "lp": lp, "pb": pb, "dict_size": 1 << 12}]
    _check(loops, lzma.compress(plain, format=lzma.FORMAT_ALONE, filters=filt), plain)


def test_emulated_loop_truncated_and_oversized_declared(loops):
    plain = W.make_plain("text", 20000, seed=7)
    comp = W.compress_alone(plain, dict_size=65536, known_size=True)
    lc, lp, pb, ds, us = _hdr(comp)
    r = loops[True].decode_raw(comp[13:len(comp) // 2], lc, lp, pb, ds, us, out_cap=len(plain))
    ref = orc.lzma_decompress(comp[:len(comp) // 2])
    assert r["status"] == "INPUT_EOF" and ref.kind != 0
    assert r["out"][:len(ref.out)] == ref.out          # what the reference flushed is a prefix of what was decoded
    r = loops[True].decode_raw(comp[13:], lc, lp, pb, ds, us - 5, out_cap=len(plain))
    assert r["status"] in ("OK", "SIZE_MISMATCH") and r["out"][:us - 5] == plain[:us - 5]


@pytest.mark.parametrize("kind", ["text", "random"])
def test_emulated_loop_output_limit(loops, kind):
    """the output slice ends before the stream does: the loop stops exactly at the limit, whether a literal
    (append_literal, lzbuffer.rs:206-217) or a match (append_lz) hits it, and everything before it is right"""
    plain = W.make_plain(kind, 6000, seed=11)
    comp = W.compress_alone(plain, dict_size=65536, known_size=True)
    lc, lp, pb, ds, us = _hdr(comp)
    for cap in (1, 2, 100, 273, 274, 1000, 5726, 5727, 5999):
        r = loops[True].decode_raw(comp[13:], lc, lp, pb, ds, us, out_cap=cap)
        assert r["status"] == "OUT_FULL", (cap, r["status"])
        assert r["len"] == cap and r["out"] == plain[:cap]


def test_marker_before_a_declared_size(loops):
    """a stream that ends with the marker before the size its header declares: Finished by the marker
    (lzma.rs:372-377), then the known-size check after the loop fails it (lzma.rs:513-521)"""
    plain = W.make_plain("text", 5000, seed=3)
    comp = W.compress_alone(plain, dict_size=65536, known_size=False)
    for declared in (5001, 6000, 1 << 32, 1 << 40):
        lie = comp[:5] + struct.pack("<Q", declared) + comp[13:]
        ref = orc.lzma_decompress(lie)
        assert ref.msg == "lzma error: Expected unpacked size of %d but decompressed to 5000" % declared
        lc, lp, pb, ds, us = _hdr(lie)
        r = loops[True].decode_raw(lie[13:], lc, lp, pb, ds, us, out_cap=8192)
        assert r["status"] == "SIZE_MISMATCH" and r["len"] == 5000 and r["in_consumed"] + 13 == ref.in_consumed


@pytest.mark.parametrize("lc,lp,pb", [(4, 0, 2), (0, 4, 0), (2, 2, 4)])
def test_emulated_lc4_through_the_row_caches(loops, lc, lp, pb):
    """lc + lp = 4 (16 literal rows) runs in the HBM variant since round 4: eight register rows and eight LDS rows cache the 16 rows
    of the slab.  Binary data with many short matches makes plain and matched literals land in every row: both caches must miss,
    evict (write back) and hit, and the output is the oracle's."""
    rnd = random.Random(lc * 10 + lp)
    blk = rnd.randbytes(4000)
    plain = b"".join(blk[i:i + rnd.randint(3, 40)] + rnd.randbytes(rnd.randint(1, 6)) for i in range(0, 3900, 17)) + \
        W.make_plain("text", 20000, seed=1)
    filt = [{"id": lzma.FILTER_LZMA1, "lc": lc, "lp": lp, "pb": pb, "dict_size": 1 << 16}]
    comp = lzma.compress(plain, format=lzma.FORMAT_ALONE, filters=filt)
    _check(loops, comp, plain)
    emu = _pick(loops, lc, lp, pb)
    emu.reset_counts()
    emu.decode_raw(comp[13:], lc, lp, pb, 1 << 16, None, out_cap=len(plain) + 8)
    c, _ = emu.counts()
    by_op = {}
    for i, text in enumerate(emu.prog.text):
        by_op[text.split()[0]] = by_op.get(text.split()[0], 0) + int(c[i])
    # plain rows: loads (misses) and stores (evictions) of 8 bytes per lane; matched rows: 16 bytes per lane
    assert by_op["buffer_load_dwordx2"] > 50 and by_op["buffer_store_dwordx2"] > 30
    assert by_op["buffer_load_dwordx4"] > 20 and by_op["buffer_store_dwordx4"] > 10


@pytest.mark.parametrize("lc,lp,pb", [(3, 0, 2), (1, 2, 1), (3, 0, 4), (4, 0, 2)])   # the LP0, GEN, PB4 and HBM variants of the loop
def test_emulated_reader_at_every_cut(loops, lc, lp, pb):
    """the loop's reader (end-aligned last window, "reader at EOF" state) against the oracle for EVERY prefix of short streams: known
    size, unknown size with marker, unknown size without marker (finished only if the reader is at EOF with code == 0), lengths
    around the 64-byte window: status, bytes decoded and reader position"""
    emu = _pick(loops, lc, lp, pb)
    plain = W.make_plain("text", 700, seed=5) + bytes(range(200))
    for known, marker in ((True, False), (False, True)):
        comp = W.compress_alone(plain, dict_size=4096, known_size=known, lc=lc, lp=lp, pb=pb)
        assert _hdr(comp)[:3] == (lc, lp, pb)
        _, _, _, ds, us = _hdr(comp)
        for cut in list(range(14, 150)) + list(range(len(comp) - 80, len(comp) + 1)):
            part = comp[:cut]
            ref = orc.lzma_decompress(part)
            r = emu.decode_raw(part[13:], lc, lp, pb, ds, us, out_cap=len(plain) + 8)
            if ref.ok:
                assert r["status"] == "OK" and r["out"] == ref.out and r["in_consumed"] + 13 == ref.in_consumed, (known, cut, r["status"])
            else:
                assert r["status"] != "OK", (known, cut)
                assert r["out"][:len(ref.out)] == ref.out
                if "failed to fill" in ref.msg:
                    assert r["status"] in ("INPUT_EOF", "RC_INIT") and r["in_consumed"] == len(part) - 13, (known, cut, r)
    # unknown size, no marker: a stream cut exactly where the encoder's flush ends decodes "successfully" iff code == 0 there
    raw = W.compress_alone(plain, dict_size=4096, known_size=False, lc=lc, lp=lp, pb=pb)
    body = raw[:5] + b"\xff" * 8 + raw[13:]
    for cut in range(len(body) - 12, len(body) + 1):
        part = body[:cut]
        ref = orc.lzma_decompress(part)
        _, _, _, ds, us = _hdr(part)
        r = emu.decode_raw(part[13:], lc, lp, pb, ds, us, out_cap=len(plain) + 300)
        assert (r["status"] == "OK") == ref.ok, (cut, r["status"], ref)
        if ref.ok:
            assert r["out"] == ref.out and r["in_consumed"] + 13 == ref.in_consumed


@pytest.mark.parametrize("lc,lp,pb", [(3, 0, 2), (1, 2, 1), (3, 0, 4), (4, 0, 2)])   # LP0, GEN, PB4, LC4
def test_emulated_loop_yields_at_quanta(loops, lc, lp, pb):
    """time-sliced launches: the loop leaves at the first symbol top at or beyond qtop (exit QUANTUM) and is re-entered after the
    kernel's resume (reader re-seeked, tables rebuilt).  Whatever the quantum -- every symbol, odd sizes, larger than the stream --
    status, bytes and reader position are those of the uninterrupted run, for good, truncated and size-mismatched streams and at
    the output limit."""
    emu = _pick(loops, lc, lp, pb)
    rng = random.Random(lc * 100 + lp * 10 + pb)
    cases = []
    for kind, n in (("text", 6000), ("random", 1500), ("repeat", 9000), ("zeros", 3000)):
        plain = W.make_plain(kind, n, seed=rng.randrange(1 << 20))
        for known in (True, False):
            comp = W.compress_alone(plain, dict_size=4096, known_size=known, lc=lc, lp=lp, pb=pb)
            cases.append((comp, len(plain) + 8))
            cases.append((comp[:len(comp) * 2 // 3], len(plain) + 8))                    # truncated
        comp = W.compress_alone(plain, dict_size=4096, known_size=True, lc=lc, lp=lp, pb=pb)
        cases.append((comp[:5] + struct.pack("<Q", n - 7) + comp[13:], len(plain) + 8))   # declared size too small
        cases.append((comp, n - 100))                                                     # output limit inside the stream
    most = 0
    for comp, cap in cases:
        _, _, _, ds, us = _hdr(comp)
        base = emu.decode_raw(comp[13:], lc, lp, pb, ds, us, out_cap=cap)
        for q in (1, 97, 2500, 1 << 20):
            r = emu.decode_raw(comp[13:], lc, lp, pb, ds, us, out_cap=cap, quantum=q)
            assert (r["status"], r["out"], r["len"], r["in_consumed"]) == (base["status"], base["out"], base["len"], base["in_consumed"]), q
            if q == 1 and base["len"] > 50:
                assert r["yields"] >= 1          # (a stream of a few long matches has few symbol tops)
                most = max(most, r["yields"])
            if q == 1 << 20:
                assert r["yields"] == 0
    assert most > 1000                           # quantum 1 on random data: a yield at (nearly) every symbol


@pytest.mark.parametrize("lc,lp,pb", [(3, 0, 2), (1, 2, 1), (3, 0, 4), (4, 0, 2)])   # the LP0, GEN, PB4 and HBM variants of the loop
def test_emulated_loop_fed_in_views(loops, lc, lp, pb):
    """MILZMA_DECODE_FEED at the level of the loop (round 5): the input arrives in views of arbitrary lengths (down to a byte more than the
    one before).  With the FEED bit the loop leaves at the first symbol top with fewer than FEED_MARGIN bytes of its view left -- never in
    the middle of a symbol --, is re-entered on the longer view, and runs the whole payload last without the bit: bytes, verdict and reader
    position are the oracle's, for good streams of known and unknown size and for a truncated one."""
    import gen_fast_loop as G
    loop = _pick(loops, lc, lp, pb)
    rng = random.Random(lc * 100 + lp * 10 + pb)
    for kind, known, size in (("text", True, 90000), ("text", False, 40000), ("random", True, 9000), ("repeat", False, 50000)):
        plain = W.make_plain(kind, size, seed=77 + size)
        comp = W.compress_alone(plain, dict_size=1 << 16, lc=lc, lp=lp, pb=pb, known_size=known)
        pay = comp[13:]
        # (many views: the dangerous stop is the one whose view ends 33 .. 95 bytes into its last-but-one window -- the symbol that starts
        #  near that window's end must not be begun; plus 32 views one byte apart)
        cuts = sorted(set(rng.randrange(16, len(pay)) for _ in range(260)) | set(range(200, min(len(pay), 232))))
        r = loop.decode_raw(pay, lc, lp, pb, 1 << 16, len(plain) if known else None, out_cap=len(plain) + 300, feed_views=cuts)
        ref = orc.lzma_decompress(comp)
        assert r["status"] == "OK" and r["out"] == plain and r["in_consumed"] + 13 == ref.in_consumed, (kind, known)
        assert len(r["feeds"]) >= len(cuts) - 33     # (a view a byte longer than a stop inside the margin stops again at once: fine, and exercised)
        for view, pos in r["feeds"]:
            assert 0 <= view - pos < G.FEED_MARGIN
    plain = W.make_plain("text", 60000, seed=9)
    comp = W.compress_alone(plain, dict_size=1 << 16, lc=lc, lp=lp, pb=pb, known_size=True)
    cut = comp[:len(comp) // 2]
    r = loop.decode_raw(cut[13:], lc, lp, pb, 1 << 16, len(plain), out_cap=len(plain) + 300, feed_views=[50, 4000, len(cut) - 13 - 7])
    ref = orc.lzma_decompress(cut)
    assert r["status"] == "INPUT_EOF" and r["in_consumed"] + 13 == ref.in_consumed and r["out"][:len(ref.out)] == ref.out


@pytest.mark.parametrize("lc,lp,pb", [(3, 0, 2), (1, 2, 1), (3, 0, 4), (4, 0, 2)])   # the LP0, GEN, PB4 and HBM variants of the loop
def test_emulated_loop_goes_on_behind_an_end_marker(loops, lc, lp, pb):
    """The crate's Partial mode at an end marker (lzma.rs:493-495, :507-509; MILZMA_KIND_PARTIAL): the loop of a write merely leaves at
    `Finished`, and bytes written later are decoded on from the marker's state -- rep[0] = 0xFFFF_FFFF, the state after a match.  The
    loop, re-entered from that state on a longer view, against the oracle's Stream fed the same two pieces."""
    loop = _pick(loops, lc, lp, pb)
    rng = random.Random(4000 + lc * 100 + lp * 10 + pb)
    plain = W.make_plain("text", 6000, seed=5)
    head = W.compress_alone(plain, dict_size=1 << 16, lc=lc, lp=lp, pb=pb, known_size=False)
    other = W.compress_alone(W.make_plain("text", 2000, seed=6), dict_size=1 << 16, lc=lc, lp=lp, pb=pb, known_size=False)
    tails = [b"\x00" * 40, b"\xff" * 40, other[13:], head[13:], other, bytes([0, 0, 0, 0, 1]) + bytes(rng.randrange(256) for _ in range(200))]
    tails += [bytes(rng.randrange(256) for _ in range(80)) for _ in range(24)]
    seen = set()
    for tail in tails:
        o = orc.Stream()
        o.write_all(head)
        text = None
        try:
            o.write_all(tail)
        except orc.Stream.WriteError as e:
            text = str(e)
        fin = o.finish()
        if text is None:
            text = fin.msg if not fin.ok else None
        r = loop.decode_raw(head[13:] + tail, lc, lp, pb, 1 << 16, None, out_cap=1 << 20, feed_views=[len(head) - 13], partial=True)
        assert len(r["feeds"]) >= 1 and r["feeds"][-1] == (len(head) - 13, len(head) - 13)     # parked right behind the marker
        st = r["status"]
        seen.add(st)
        if st == "OK":
            assert text is None and r["out"] == fin.out, (tail[:8], len(r["out"]), len(fin.out))
            continue
        want = {"MATCH_DIST_DICT": "Match distance %d is beyond dictionary size 65536" % (r["rep0"] + 1),
                "LZ_DIST_DICT": "LZ distance %d is beyond dictionary size 65536" % (r["rep0"] + 1),
                "MATCH_DIST_OUT": "Match distance %d is beyond output size %d" % (r["rep0"] + 1, r["len"]),
                "LZ_DIST_OUT": "LZ distance %d is beyond output size %d" % (r["rep0"] + 1, r["len"]),
                "MARKER_TRAILING": "Found end-of-stream marker but more bytes are available",
                "INPUT_EOF": "failed to fill whole buffer"}[st]
        assert text is not None and want in text, (tail[:8], st, text)
    # (behind a marker `code` is 0 -- is_finished_ok demanded it --, so the next decision is_match always says "literal": a matched one, whose
    #  match byte lies rep[0] + 1 = 2^32 back.  Whatever is written behind a marker ends there.)
    assert seen == {"MATCH_DIST_DICT"}, seen


def test_emulated_hbm_variant_lclp_above_four():
    """The HBM variant of the loop (lc + lp > 4: the 2^(lc+lp) literal rows in a slab in memory, eight register rows and eight LDS
    rows as direct-mapped caches over it, tags in two VGPRs' lanes): streams with real match structure for every kind of property set
    it may meet (lc up to 8, lp up to 4, pb up to 4; also sets the other variants own, where it must agree with them), known and
    marker-terminated, truncated, with a quantum that parks the walk every 2500 bytes -- against the oracle."""
    import random
    import lzma_enc as E
    emu_hbm = asmprog.AsmLoop(lp0=False, pb4=True, hbm=True)
    emu_hb0 = asmprog.AsmLoop(lp0=True, pb4=False, hbm=True)   # round 6: the same slab behind the LP0 variant's bookkeeping (lp == 0, pb <= 2)
    try:
        rnd = random.Random(3)
        for (lc, lp, pb, emu) in [(8, 0, 2, emu_hbm), (4, 4, 0, emu_hbm), (5, 2, 4, emu_hbm), (8, 4, 4, emu_hbm), (3, 0, 2, emu_hbm), (4, 0, 2, emu_hbm),
                                  (0, 0, 0, emu_hbm), (8, 0, 2, emu_hb0), (4, 0, 2, emu_hb0), (6, 0, 0, emu_hb0), (5, 0, 1, emu_hb0), (3, 0, 2, emu_hb0)]:
            size = 90_000
            plain = (W.make_plain("text", size - 30000, seed=lc * 100 + lp * 10 + pb) + rnd.randbytes(15000) + bytes(range(256)) * 58 + b"x" * 152)[:size]
            for known in (True, False):
                enc = E.LzmaSymbolEncoder(lc, lp, pb)
                enc.encode(E.lz_parse(plain, dict_size=1 << 16))
                if not known:
                    enc.encode([("marker",)])
                comp = E.lzma_header(lc, lp, pb, 1 << 16, len(plain) if known else None) + enc.finish()
                ref = orc.lzma_decompress(comp)
                assert ref.ok and ref.out == plain
                for quantum in (None, 2500):
                    r = emu.decode_raw(comp[13:], lc, lp, pb, 1 << 16, len(plain) if known else None, out_cap=len(plain) + 300, quantum=quantum)
                    assert r["status"] == "OK" and r["out"] == plain and r["in_consumed"] + 13 == ref.in_consumed, (lc, lp, pb, known, quantum)
                cut = comp[:len(comp) * 2 // 3]
                ref = orc.lzma_decompress(cut)
                r = emu.decode_raw(cut[13:], lc, lp, pb, 1 << 16, len(plain) if known else None, out_cap=len(plain) + 300)
                assert r["status"] == "INPUT_EOF" and r["in_consumed"] + 13 == ref.in_consumed
                assert r["out"][:len(ref.out)] == ref.out
    finally:
        emu_hbm.close()
        emu_hb0.close()


_KNOB_SCRIPT = r"""
import os, struct, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tools", "emu")); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import asmprog, oracle_py as orc
from lzma_rs_amd import workloads as W
loops = {True: asmprog.AsmLoop(lp0=True), False: asmprog.AsmLoop(lp0=False)}
for kind, lp, known in (("text", 0, True), ("text", 1, False), ("random", 0, True), ("repeat", 0, False)):
    plain = W.make_plain(kind, 150000, seed=11)
    lc = 3 - lp                                                  # (lc + lp <= 3: the LP0 / GEN variants' classes)
    comp = W.compress_alone(plain, dict_size=1 << 16, lc=lc, lp=lp, pb=2, known_size=known)
    ref = orc.lzma_decompress(comp)
    for quantum in (None, 3000):
        r = loops[lp == 0].decode_raw(comp[13:], lc, lp, 2, 1 << 16, len(plain) if known else None, out_cap=len(plain) + 300, quantum=quantum)
        assert r["status"] == "OK" and r["out"] == plain and r["in_consumed"] + 13 == ref.in_consumed, (kind, lp, known, quantum)
    cut = comp[:len(comp) // 2]
    ref = orc.lzma_decompress(cut)
    r = loops[lp == 0].decode_raw(cut[13:], lc, lp, 2, 1 << 16, len(plain) if known else None, out_cap=len(plain) + 300)
    assert r["status"] == "INPUT_EOF" and r["in_consumed"] + 13 == ref.in_consumed and r["out"][:len(ref.out)] == ref.out
print("ok")
"""


KNOB_SETS = [
    {"MILZMA_GEN_SSHADOW": "state,rep,wb"}, {"MILZMA_GEN_LENDEFER": "1"}, {"MILZMA_GEN_LENDEFER": "1", "MILZMA_GEN_ALIGNLAZY": "1"},
    {"MILZMA_GEN_R11S": "tree"}, {"MILZMA_GEN_SWAP2": "0", "MILZMA_GEN_DISPMAD": "0", "MILZMA_GEN_MLGUARD": "0", "MILZMA_GEN_EARLYLDS": "0",
                                  "MILZMA_GEN_NBPRE": "0"},
    # round 5 (profiles/r05_kernel_ab.txt): the loop as round 4 shipped it (the symbol in an SGPR, four instructions per shadow), and the
    # forms that were measured and rejected: single decisions all scalar, tree walks in form A with shadows, 2^24 in an SGPR
    {"MILZMA_GEN_SYM_M0": "0", "MILZMA_GEN_SHADOW": "4"}, {"MILZMA_GEN_S1": "single"}, {"MILZMA_GEN_FORMB": "none", "MILZMA_GEN_FORMA2": "1"},
    {"MILZMA_GEN_K24S": "1"},
    # round 6: the direct-bit chains on the vector ALU, all bits / all but the last four of a chain
    {"MILZMA_GEN_VDIRECT": "1"}, {"MILZMA_GEN_VDIRECT": "1", "MILZMA_GEN_VDIRECT_S": "4"},
    # ... and the loop without the quotient blocks (round 6's kernel up to hash 5f7ed0723658b699), and with them from five bits on only
    {"MILZMA_GEN_QDIRECT": "0"}, {"MILZMA_GEN_QDIRECT": "5"}]


def _knob_id(k):
    return "+".join(sorted(x[11:] for x in k))


@pytest.fixture(scope="module")
def knob_runs():
    """one process per knob set (the generator reads its switches at import), four at a time; each test waits for its own"""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=4)
    jobs = {}
    for knobs in KNOB_SETS:
        env = {k: v for k, v in os.environ.items() if not k.startswith("MILZMA_GEN_")}
        env.update(knobs)
        jobs[_knob_id(knobs)] = pool.submit(subprocess.run, [sys.executable, "-c", _KNOB_SCRIPT % {"root": ROOT}], env=env, capture_output=True,
                                            text=True, timeout=900)
    yield jobs
    pool.shutdown(wait=True, cancel_futures=True)


@pytest.mark.parametrize("knobs", KNOB_SETS, ids=_knob_id)
def test_emulated_loop_under_measured_and_rejected_knobs(knob_runs, knobs):
    """The generator switches of round 4's A/Bs (profiles/r04_kernel_ab.txt sections 3-4: scalar bookkeeping in shadows, the two deferred
    tree updates, range >> 11 on the scalar ALU; and the loop WITHOUT the batch that shipped) still generate loops that decode bit-exactly:
    what the profile says was measured can be rebuilt and measured again.  (The generator reads its switches at import: a process each.)"""
    r = knob_runs[_knob_id(knobs)].result()
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-4000:]


def test_quotient_block_arithmetic_against_get_bit():
    """The quotient blocks of the direct-bit chains (tools/gen_fast_loop.py: QDIRECT, Gen.direct_quotient) in numpy, against RangeDecoder::get_bit
    (rangecoder.rs:71-82) bit by bit: behind a normalisation the range is r << 8; for every j = 2 .. 6, ranges of that form and codes below, at and above
    the range (a damaged stream can get there: a direct bit on an odd range leaves code == range), and at every multiple of range >> j and its
    neighbours: a lane answers exactly when code < range, then code, range and the inverted bits are what j serial bits leave; no lane answers otherwise
    (the block then runs the serial bits)."""
    import numpy as np
    rng = np.random.default_rng(11)
    for j in range(2, 7):
        m = (1 << j) - 1
        n = 200000
        R = (rng.integers(1 << 16, 1 << 24, n, dtype=np.uint64) << np.uint64(8)).astype(np.uint64)      # normalised: [2^24, 2^32), low byte zero
        code = rng.integers(0, 1 << 32, n, dtype=np.uint64)
        r = R >> np.uint64(j)
        k = rng.integers(0, 1 << j, n, dtype=np.uint64)
        edge = (k * r + rng.integers(-2, 3, n).astype(np.int64).astype(np.uint64)) & np.uint64(0xFFFFFFFF)
        pick = rng.integers(0, 4, n)
        code = np.where(pick == 0, edge, np.where(pick == 1, code % R, np.where(pick == 2, R + (code & np.uint64(3)), code)))
        # serial: j times get_bit; acc = 2 acc + (code < range) as the loop accumulates it (the inverted bit)
        sr, sc, acc = R.copy(), code.copy(), np.zeros(n, dtype=np.uint64)
        for _ in range(j):
            sr = sr >> np.uint64(1)
            lt = sc < sr
            sc = np.where(lt, sc, sc - sr)
            acc = (acc << np.uint64(1)) | lt.astype(np.uint64)
        # the block: 64 lanes, candidate c = (m - L) & m, 32-bit wrapping arithmetic
        lanes = np.arange(64, dtype=np.uint64)
        c = (np.uint64(m) ^ lanes) & np.uint64(m)
        prod = (c[None, :] * r[:, None]) & np.uint64(0xFFFFFFFF)
        diff = (code[:, None] - prod) & np.uint64(0xFFFFFFFF)
        ans = diff < r[:, None]
        any_lane = ans.any(axis=1)
        first = ans.argmax(axis=1).astype(np.uint64)
        assert (any_lane == (code < R)).all(), j
        q = first ^ np.uint64(m)
        qc = (code - q * r) & np.uint64(0xFFFFFFFF)
        ok = any_lane
        assert (qc[ok] == sc[ok]).all() and (r[ok] == sr[ok]).all() and (first[ok] == acc[ok]).all(), j
