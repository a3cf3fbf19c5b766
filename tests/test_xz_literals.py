"""Malformed .xz files against EXPECTED (kind, message) pairs written out as literals.

The library's XZ container walk (lzma_rs_amd/csrc/host.cpp) and the oracle's (oracle/lzma_oracle.c) were written by the
same hand from the same reading of the reference, so agreement between them on an error path proves little.  Every
expectation below is therefore spelled out here, from the reference's own format strings:
  src/xz/header.rs (StreamHeader::parse), src/xz/mod.rs (StreamFlags::parse, CheckMethod::try_from),
  src/decode/xz.rs:18-94 (decode_stream: footer), :96-171 (check_index), :196-290 (read_block),
  :292-333 (validate_block_check), :335-357 (decode_filter), :359-449 (read_block_header), :451-466 (get_multibyte),
and BOTH decoders are held to it independently: the oracle in the CPU test, the GPU library in the `-m gpu` test.
Numbers inside a message (sizes, CRCs) are computed here from the crafted bytes with zlib / plain arithmetic, never
taken from either implementation.
"""
import lzma
import struct
import zlib

import pytest

import oracle_py as orc

XZ, IO = 4, 1   # error::Error::XzError / IoError (the crate's Display prefixes "xz error: " / "io error: ")
MAGIC = b"\xfd7zXZ\x00"
EOF_MSG = "io error: failed to fill whole buffer"
PLAIN = b"The quick brown fox jumps over the lazy dog. " * 40


def crc32(b):
    return struct.pack("<I", zlib.crc32(b))


def multibyte(v):
    out = bytearray()
    while True:
        if v < 0x80:
            out.append(v)
            return bytes(out)
        out.append(0x80 | (v & 0x7F))
        v >>= 7


def lzma2_payload(plain, dict_size=1 << 16):
    return lzma.compress(plain, format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "dict_size": dict_size}])


def stream_header(flags=b"\x00\x01", crc=None):
    return MAGIC + flags + (crc32(flags) if crc is None else crc)


def block_header(flags=0x00, packed=None, unpacked=None, filter_id=0x21, props=b"\x10", props_size=None, pad=None, crc=None,
                 raw_body=None):
    """size byte | flags | [packed] | [unpacked] | filter id | size of properties | properties | zero padding | CRC32"""
    body = bytes([flags])
    if packed is not None:
        body += multibyte(packed)
    if unpacked is not None:
        body += multibyte(unpacked)
    body += multibyte(filter_id) + multibyte(len(props) if props_size is None else props_size) + props
    if raw_body is not None:
        body = raw_body
    total = (1 + len(body) + 4 + 3) & ~3            # with the size byte and the CRC, padded to a multiple of four
    body += (b"\x00" * (total - 5 - len(body))) if pad is None else pad
    hdr = bytes([total // 4 - 1]) + body
    return hdr + (crc32(hdr) if crc is None else crc)


def block(plain=PLAIN, check=1, header=None, payload=None, padding=None, check_bytes=None):
    header = block_header() if header is None else header
    payload = lzma2_payload(plain) if payload is None else payload
    pad = b"\x00" * ((-(len(header) + len(payload))) % 4) if padding is None else padding
    if check_bytes is None:
        check_bytes = {0: b"", 1: crc32(plain), 4: struct.pack("<Q", orc.crc64(plain)), 10: b"\x00" * 32}[check]
    unpadded = len(header) + len(payload) + len(check_bytes)
    return header + payload + pad + check_bytes, unpadded, len(plain)


def index(records, n=None, pad=None, crc=None):
    body = b"\x00" + multibyte(len(records) if n is None else n)
    for unpadded, unpacked in records:
        body += multibyte(unpadded) + multibyte(unpacked)
    body += (b"\x00" * ((-len(body)) % 4)) if pad is None else pad
    return body + (crc32(body) if crc is None else crc)


def footer(index_len, flags=b"\x00\x01", backward=None, crc=None, magic=b"YZ"):
    tail = struct.pack("<I", index_len // 4 - 1 if backward is None else backward) + flags
    return (crc32(tail) if crc is None else crc) + tail + magic


def xz_file(check=1, blocks=None, idx=None, **foot):
    flags = bytes([0, check])
    blocks = [block(check=check)] if blocks is None else blocks
    idx = index([(u, p) for _, u, p in blocks]) if idx is None else idx
    foot.setdefault("flags", flags)
    return stream_header(flags) + b"".join(b for b, _, _ in blocks) + idx + footer(len(idx), **foot)


def cases():
    """name -> (file bytes, kind, message)"""
    c = {}
    good = xz_file()
    c["good"] = (good, 0, "")
    # ---- stream header (src/xz/header.rs, src/xz/mod.rs)
    c["bad_magic"] = (b"\xfd7zXY\x00" + good[6:], XZ, "xz error: Invalid XZ magic, expected [253, 55, 122, 88, 90, 0]")
    c["header_crc"] = (stream_header(b"\x00\x01", crc=b"\x01\x02\x03\x04") + good[12:], XZ,
                       "xz error: Invalid header CRC32: expected 0x04030201 but got 0x%08x" % zlib.crc32(b"\x00\x01"))
    c["stream_flags_null_byte"] = (stream_header(b"\x01\x01") + good[12:], XZ, "xz error: Invalid null byte in Stream Flags: 1")
    c["check_method_unknown"] = (stream_header(b"\x00\x02") + good[12:], XZ,
                                 "xz error: Invalid check method 2, expected one of [0x00, 0x01, 0x04, 0x0A]")
    c["truncated_in_header"] = (good[:9], IO, EOF_MSG)
    # ---- block header (src/decode/xz.rs:359-449, :196-222)
    c["block_flags_reserved"] = (xz_file(blocks=[block(header=block_header(flags=0x04))]), XZ,
                                 "xz error: Invalid block flags 4, reserved bits (mask 0x3C) must be zero")
    c["unknown_filter"] = (xz_file(blocks=[block(header=block_header(filter_id=0x03))]), XZ, "xz error: Unknown filter id 3")
    c["filter_props_exceed_header"] = (xz_file(blocks=[block(header=block_header(props_size=200, props=b"\x10"))]), XZ,
                                       "xz error: Size of filter properties exceeds block header size (200 > 7)")   # ((2 << 2) - 1: xz.rs:210)
    # (6 properties bytes declared, 4 left inside the 7-byte take() window of the header: read_exact fails there)
    c["filter_props_short"] = (xz_file(blocks=[block(header=block_header(props_size=6, props=b"\x10"))]), XZ,
                               "xz error: Could not read filter properties of size 6: failed to fill whole buffer")
    c["filter_props_two_bytes"] = (xz_file(blocks=[block(header=block_header(props=b"\x10\x00"))]), XZ,
                                   "xz error: Invalid properties for filter Lzma2")
    c["block_header_padding"] = (xz_file(blocks=[block(header=block_header(pad=b"\x00\x01\x00"))]), XZ,
                                 "xz error: Invalid block header padding, must be null bytes")
    good_hdr = block_header()
    c["block_header_crc"] = (xz_file(blocks=[block(header=block_header(crc=b"\xaa\xbb\xcc\xdd"))]), XZ,
                             "xz error: Invalid header CRC32: expected 0xddccbbaa but got 0x%08x" % zlib.crc32(good_hdr[:-4]))
    c["multibyte_too_long"] = (xz_file(blocks=[block(header=block_header(flags=0x40, raw_body=b"\x40" + b"\xff" * 9 + b"\x21\x01\x10"))]),
                               XZ, "xz error: Invalid multi-byte encoding")
    # ---- block body (src/decode/xz.rs:224-290)
    pay = lzma2_payload(PLAIN)
    c["compressed_size_field"] = (xz_file(blocks=[block(header=block_header(flags=0x40, packed=len(pay) + 1))]), XZ,
                                  "xz error: Invalid compressed size: expected %d but got %d" % (len(pay) + 1, len(pay)))
    c["unpacked_size_field"] = (xz_file(blocks=[block(header=block_header(flags=0x80, unpacked=len(PLAIN) + 7))]), XZ,
                                "xz error: Invalid decompressed size: expected %d but got %d" % (len(PLAIN) + 7, len(PLAIN)))
    npad = (-(len(good_hdr) + len(pay))) % 4
    assert npad > 0
    c["block_padding"] = (xz_file(blocks=[block(padding=b"\x01" + b"\x00" * (npad - 1))]), XZ,
                          "xz error: Invalid block padding, must be null bytes")
    c["block_crc32"] = (xz_file(blocks=[block(check_bytes=b"\x78\x56\x34\x12")]), XZ,
                        "xz error: Invalid block CRC32, expected 0x12345678 but got 0x%08x" % zlib.crc32(PLAIN))
    c["block_crc64"] = (xz_file(check=4, blocks=[block(check=4, check_bytes=bytes(range(1, 9)))]), XZ,
                        "xz error: Invalid block CRC64, expected 0x0807060504030201 but got 0x%016x" % orc.crc64(PLAIN))
    c["sha256"] = (xz_file(check=10, blocks=[block(check=10)]), XZ, "xz error: Unsupported SHA-256 checksum (not yet implemented)")
    c["check_none_ok"] = (xz_file(check=0, blocks=[block(check=0)]), 0, "")
    # ---- index (src/decode/xz.rs:96-171)
    b0 = block()
    c["index_record_count"] = (xz_file(blocks=[b0], idx=index([(b0[1], b0[2])], n=2)), XZ, "xz error: Expected 2 records but got 1 records")
    c["index_unpadded"] = (xz_file(blocks=[b0], idx=index([(b0[1] + 4, b0[2])])), XZ,
                           "xz error: Invalid index for record 0: unpadded size (%d) does not match index (%d)" % (b0[1], b0[1] + 4))
    c["index_unpacked"] = (xz_file(blocks=[b0], idx=index([(b0[1], b0[2] - 1)])), XZ,
                           "xz error: Invalid index for record 0: unpacked size (%d) does not match index (%d)" % (b0[2], b0[2] - 1))
    body = b"\x00" + multibyte(1) + multibyte(b0[1]) + multibyte(b0[2])
    ipad = (-len(body)) % 4
    assert ipad > 0
    c["index_padding"] = (xz_file(blocks=[b0], idx=index([(b0[1], b0[2])], pad=b"\x00" * (ipad - 1) + b"\x05")), XZ,
                          "xz error: Invalid index padding, must be null bytes")
    c["index_crc"] = (xz_file(blocks=[b0], idx=index([(b0[1], b0[2])], crc=b"\x11\x22\x33\x44")), XZ,
                      "xz error: Invalid index CRC32: expected 0x44332211 but got 0x%08x" % zlib.crc32(body + b"\x00" * ipad))
    # ---- footer (src/decode/xz.rs:47-94)
    ilen = len(index([(b0[1], b0[2])]))
    c["backward_size"] = (xz_file(blocks=[b0], backward=ilen // 4 + 2), XZ,
                          "xz error: Invalid index size: expected %d but got %d" % ((ilen // 4 + 3) * 4, ilen))
    c["footer_flags_differ"] = (xz_file(blocks=[b0], flags=b"\x00\x04"), XZ,
                                "xz error: Flags in header (StreamFlags { check_method: Crc32 }) does not match footer "
                                "(StreamFlags { check_method: Crc64 })")
    tail = struct.pack("<I", ilen // 4 - 1) + b"\x00\x01"
    c["footer_crc"] = (xz_file(blocks=[b0], crc=b"\xef\xbe\xad\xde"), XZ,
                       "xz error: Invalid footer CRC32: expected 0xdeadbeef but got 0x%08x" % zlib.crc32(tail))
    c["footer_magic"] = (xz_file(blocks=[b0], magic=b"YY"), XZ, "xz error: Invalid footer magic, expected [89, 90]")
    c["trailing_data"] = (good + b"\x00", XZ, "xz error: Unexpected data after last XZ block")
    c["truncated_before_footer_magic"] = (good[:-2], IO, EOF_MSG)
    c["truncated_in_index"] = (good[:len(good) - 12 - 3], IO, EOF_MSG)
    # ---- two blocks: the first is delivered, the second's check fails (output before the error stays written)
    b1 = block(plain=PLAIN[::-1], check_bytes=b"\x00\x00\x00\x00")
    c["second_block_crc32"] = (xz_file(blocks=[b0, b1]), XZ,
                               "xz error: Invalid block CRC32, expected 0x00000000 but got 0x%08x" % zlib.crc32(PLAIN[::-1]))
    return c


CASES = cases()
WRITTEN = {"good": PLAIN, "check_none_ok": PLAIN, "second_block_crc32": PLAIN}   # bytes the writer has received at return
for _n in ("index_record_count", "index_unpadded", "index_unpacked", "index_padding", "index_crc", "backward_size", "footer_flags_differ",
           "footer_crc", "footer_magic", "trailing_data", "truncated_before_footer_magic", "truncated_in_index"):
    WRITTEN[_n] = PLAIN                                                            # (the block itself was fine)


def test_the_crafted_good_files_are_valid_xz():
    for name in ("good", "check_none_ok"):
        assert lzma.decompress(CASES[name][0], format=lzma.FORMAT_XZ) == PLAIN
    assert len(CASES) >= 30


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_against_literal_expectations(name):
    data, kind, msg = CASES[name]
    r = orc.xz_decompress(data)
    assert (r.kind, r.msg) == (kind, msg), (name, r)
    assert r.out == WRITTEN.get(name, b""), (name, len(r.out))


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["asm", "generic"])
def test_gpu_library_against_literal_expectations(kernel):
    import os
    import lzma_rs_amd as M
    os.environ["MILZMA_KERNEL"] = kernel
    ctx = M.Context(0)
    try:
        names = sorted(CASES)
        single = {n: ctx.xz(CASES[n][0]) for n in names}
        batch = dict(zip(names, ctx.xz_batch([CASES[n][0] for n in names])))   # planned-ahead blocks + on-demand fallbacks, mixed
        for n in names:
            for d in (single[n], batch[n]):
                assert (d.kind, d.msg) == CASES[n][1:], (n, d)
                assert d.data == WRITTEN.get(n, b""), (n, len(d.data))
    finally:
        ctx.close()
        os.environ.pop("MILZMA_KERNEL", None)
