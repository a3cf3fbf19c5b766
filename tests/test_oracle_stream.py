"""The oracle's Stream restatement (oracle/lzma_oracle.c: src/decode/stream.rs + the Partial mode of DecoderState::process_mode) pinned to
the reference's own unit tests of the feature (src/decode/stream.rs:350-493: their inline vectors, tests/files/small.txt through the crate's
literal-only encoder, restated in tests/lzma_enc.py and itself pinned here to the 23-byte vector the tests carry) and to tests/lzma.rs's
round trips in chunks (:116-131)."""
import os
import random

import pytest

import lzma_enc as E
import oracle_py as orc
from lzma_rs_amd import workloads as W

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# lzma_compress of nothing (stream.rs:395-396, :436): header lc3/lp0/pb2, dict 0x800000, no size; five bytes of range coder; the end marker
EMPTY = b"\x5d\x00\x00\x80\x00\xff\xff\xff\xff\xff\xff\xff\xff\x00\x83\xff\xfb\xff\xff\xc0\x00\x00\x00"
MAX_HEADER_LEN, START_BYTES = 13, 5


def small():
    return open(os.path.join(GOLD, "small.txt"), "rb").read()


def test_the_encoder_restatement_writes_the_reference_vector():
    assert E.dumb_encode(b"") == EMPTY


def test_stream_noop_and_zero():                      # stream.rs:352-372
    s = orc.Stream()
    assert s.get_output() == b""
    r = s.finish()
    assert r.ok and r.out == b""
    s = orc.Stream()
    s.write_all(b"")
    s.write_all(b"")
    r = s.finish()
    assert r.ok and r.out == b""


def test_bad_header():                                # stream.rs:374-388: write_all(...).unwrap() panics with this text
    s = orc.Stream()
    with pytest.raises(orc.Stream.WriteError, match="LZMA header invalid properties: 255 must be < 225"):
        s.write_all(bytes([255]) * 32)


def test_stream_incomplete():                         # stream.rs:390-430
    for end in range(1, MAX_HEADER_LEN + START_BYTES):
        s = orc.Stream()
        s.write_all(EMPTY[:end])
        r = s.finish()
        assert not r.ok and "failed to read header" in r.msg, (end, r.msg)
    for end in range(MAX_HEADER_LEN + START_BYTES, len(EMPTY)):
        s = orc.Stream()
        s.write_all(EMPTY[:end])
        r = s.finish()
        assert not r.ok and "failed to fill whole buffer" in r.msg, (end, r.msg)


def test_stream_chunked():                            # stream.rs:432-457: every chunk size
    for comp, expected in ((EMPTY, b""), (E.dumb_encode(small()), small())):
        for chunk in range(1, len(comp)):
            s = orc.Stream()
            for at in range(0, len(comp), chunk):
                s.write_all(comp[at:at + chunk])
            r = s.finish()
            assert r.ok and r.out == expected, (chunk, r.msg)


def test_stream_corrupted():                          # stream.rs:459-471
    s = orc.Stream()
    with pytest.raises(orc.Stream.WriteError, match="beyond output size"):
        s.write_all(b"corrupted bytes here corrupted bytes here")
    r = s.finish()
    assert not r.ok and "can't finish stream because of previous write error" in r.msg


def test_allow_incomplete():                          # stream.rs:473-493: half the stream decodes to exactly 26 bytes
    plain = small()
    comp = E.dumb_encode(plain)
    half = comp[:len(comp) // 2]
    s = orc.Stream()
    s.write_all(half)
    assert not s.finish().ok
    s = orc.Stream(allow_incomplete=True)
    s.write_all(half)
    r = s.finish()
    assert r.ok and r.out == plain[:26], (len(r.out), r.msg)


def test_round_trips_in_chunks_equal_the_one_shot_decode():   # tests/lzma.rs:116-131 (chunks of the compressed file), :66-90 (options)
    rng = random.Random(5)
    for name in ("foo.txt.lzma", "hello.txt.lzma", "empty.txt.lzma", "hugedict.txt.lzma", "range-coder-edge-case.lzma"):
        comp = open(os.path.join(GOLD, name), "rb").read()
        ref = orc.lzma_decompress(comp)
        assert ref.ok
        for chunk in (1, 2, 3, 7, 19, 20, 21, 100, 4096, len(comp)):
            if name.startswith("range") and chunk < 19:
                continue                                            # (600 KB a byte at a time: minutes)
            s = orc.Stream()
            for at in range(0, len(comp), chunk):
                s.write_all(comp[at:at + chunk])
            r = s.finish()
            assert r.ok and r.out == ref.out, (name, chunk, r.msg)
    # liblzma's streams with real matches, every property class, known and unknown sizes, random chunkings
    for i in range(12):
        lc, lp, pb = [(3, 0, 2), (0, 2, 0), (1, 1, 4), (4, 0, 2)][i % 4]
        plain = W.make_plain(["text", "random", "repeat"][i % 3], rng.randint(1, 30000), seed=40 + i)
        comp = W.compress_alone(plain, dict_size=4096 if i % 2 else 1 << 16, known_size=(i % 3 == 0), lc=lc, lp=lp, pb=pb)
        ref = orc.lzma_decompress(comp)
        s = orc.Stream()
        at = 0
        while at < len(comp):
            n = rng.choice([1, 5, 19, 20, 64, 1000])
            try:
                s.write_all(comp[at:at + n])
            except orc.Stream.WriteError as err:
                # tests/lzma.rs:71-87: WriteZero once the declared size is reached and bytes remain (liblzma's end marker behind a
                # size that was patched into the header); the stream itself is intact
                assert i % 3 == 0 and "failed to write whole buffer" in str(err), (i, str(err))
                break
            at += n
            assert ref.out.startswith(s.get_output())               # (the sink only ever holds a prefix: whole rings)
        r = s.finish()
        assert (r.kind, r.msg, r.out) == (ref.kind, ref.msg, ref.out), (i, r.msg, ref.msg)


def test_a_truncated_stream_fails_at_finish_like_the_one_shot_decode_and_not_before():
    plain = W.make_plain("text", 20000, seed=3)
    comp = W.compress_alone(plain, dict_size=1 << 16, known_size=True)
    cut = comp[:len(comp) * 2 // 3]
    s = orc.Stream()
    for at in range(0, len(cut), 37):
        s.write_all(cut[at:at + 37])
    r = s.finish()
    ref = orc.lzma_decompress(cut)
    assert (r.kind, r.msg) == (ref.kind, ref.msg) and "failed to fill whole buffer" in r.msg


def test_bytes_behind_a_known_size_stream_are_a_write_zero_error():
    """process_mode leaves its loop at once when the declared size is reached (lzma.rs:441-445): write() then takes nothing and write_all
    reports ErrorKind::WriteZero."""
    plain = W.make_plain("text", 5000, seed=4)
    comp = W.compress_alone(plain, dict_size=1 << 16, known_size=True)
    s = orc.Stream()
    s.write_all(comp[:len(comp) - 6])     # (the six bytes of liblzma's end marker stay behind: the declared size is reached before them)
    with pytest.raises(orc.Stream.WriteError, match="failed to write whole buffer"):
        s.write_all(b"more bytes that nobody reads" * 2)


def test_behind_an_end_marker_by_hand():
    """What the crate does with bytes written BEHIND an end marker that ended a write, derived by hand from the source (no test of the
    reference covers it): process_next returns Finished only after `rep[0] = 0xFFFF_FFFF` and the state after a match are in place
    (lzma.rs:365-377), the Partial-mode loop merely `break`s (lzma.rs:493-495, :507-509) and Stream stays in State::Data.  is_finished_ok
    demanded code == 0 (rangecoder.rs:48-50), so the next is_match decision finds code < bound: a literal (lzma.rs:287-292) -- in a state
    >= 7, a matched one: `output.last_n(rep[0] + 1)` = last_n(2^32) (lzma.rs:540-541), beyond ANY dictionary (lzbuffer.rs:241-245).
    With fewer than 20 bytes at hand the trial run fails on exactly that and the bytes wait in the partial-input buffer (lzma.rs:498-505);
    the write that brings the 20th byte, or finish, reports it.  A dead stream then refuses every byte (stream.rs:230, :324: Ok(0))."""
    text = 'LzmaError("Match distance 4294967296 is beyond dictionary size 8388608")'
    s = orc.Stream()
    s.write_all(EMPTY)
    s.write_all(b"\x00" * 19)                                   # 19 bytes: the trial run fails, nothing is reported
    assert s.last_taken() == 19 and s.get_output() == b""
    with pytest.raises(orc.Stream.WriteError) as e:
        s.write_all(b"\x00")                                    # the 20th
    assert str(e.value) == text
    assert s.get_output() is None                               # the state is gone, and the sink with it
    with pytest.raises(orc.Stream.WriteError, match="failed to write whole buffer"):
        s.write_all(b"x")
    assert s.last_taken() == 0
    r = s.finish()
    assert not r.ok and r.msg == "lzma error: can't finish stream because of previous write error"
    # ... reported by finish when fewer than 20 bytes followed (Finish mode takes no trial run, lzma.rs:462-481)
    s = orc.Stream()
    s.write_all(EMPTY)
    s.write_all(b"\xaa" * 5)
    r = s.finish()
    assert not r.ok and r.msg == "lzma error: Match distance 4294967296 is beyond dictionary size 8388608"
    # ... unless the caller allows an incomplete stream: no last pass at all (stream.rs:130-140)
    s = orc.Stream(allow_incomplete=True)
    s.write_all(EMPTY)
    s.write_all(b"\xaa" * 5)
    r = s.finish()
    assert r.ok and r.out == b""
    # bytes behind the marker in the SAME write: the range decoder's reader is not at its end (lzma.rs:375-381)
    s = orc.Stream()
    with pytest.raises(orc.Stream.WriteError, match="Found end-of-stream marker but more bytes are available"):
        s.write_all(EMPTY + b"\x00")
    # a provided size that the marker does not deliver: finish decodes on from the marker's state
    s = orc.Stream(unpacked_size_mode=orc.READ_HEADER_BUT_USE_PROVIDED, provided=7)
    s.write_all(EMPTY)
    r = s.finish()
    assert not r.ok and r.msg == "lzma error: Match distance 4294967296 is beyond dictionary size 8388608"
