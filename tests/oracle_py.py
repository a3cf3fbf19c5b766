"""ctypes binding of oracle/liblzma_oracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference crate's decode path
(oracle/lzma_oracle.c).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module; the product package never does.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(_ORACLE_DIR, "liblzma_oracle.so")

KIND_NAMES = {0: "Ok", 1: "IoError", 2: "HeaderTooShort", 3: "LzmaError", 4: "XzError"}

READ_FROM_HEADER = 0
READ_HEADER_BUT_USE_PROVIDED = 1
USE_PROVIDED = 2


class _Options(ctypes.Structure):
    _fields_ = [
        ("unpacked_size_mode", ctypes.c_int),
        ("provided_is_some", ctypes.c_int),
        ("provided", ctypes.c_uint64),
        ("memlimit_is_some", ctypes.c_int),
        ("memlimit", ctypes.c_uint64),
    ]


class _Result(ctypes.Structure):
    _fields_ = [
        ("kind", ctypes.c_int),
        ("msg", ctypes.c_char * 384),
        ("out", ctypes.POINTER(ctypes.c_uint8)),
        ("out_len", ctypes.c_size_t),
        ("in_consumed", ctypes.c_size_t),
    ]


def build():
    """(Re)build the oracle .so if it is missing or older than its sources."""
    srcs = [os.path.join(_ORACLE_DIR, f) for f in ("lzma_oracle.c", "lzma_oracle.h", "Makefile")]
    if os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs):
        return _SO
    subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.orc_free.argtypes = [ctypes.c_void_p]
        _lib.orc_crc32.restype = ctypes.c_uint32
        _lib.orc_crc32.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        _lib.orc_crc64.restype = ctypes.c_uint64
        _lib.orc_crc64.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        _lib.orc_bench_lzma_batch.restype = ctypes.c_int64
        _lib.orc_bench_lzma_batch.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    return _lib


class OracleResult:
    """kind (0 = Ok), kind_name, msg (full Display string), out (bytes the sink
    received, also on error), in_consumed (reader position at return)."""

    def __init__(self, kind, msg, out, in_consumed):
        self.kind = kind
        self.kind_name = KIND_NAMES[kind]
        self.msg = msg
        self.out = out
        self.in_consumed = in_consumed

    @property
    def ok(self):
        return self.kind == 0

    def __repr__(self):
        return "OracleResult(kind=%s, msg=%r, out_len=%d, in_consumed=%d)" % (
            self.kind_name, self.msg, len(self.out), self.in_consumed)


def _take(res):
    out = ctypes.string_at(res.out, res.out_len) if res.out_len else b""
    if res.out:
        lib().orc_free(ctypes.cast(res.out, ctypes.c_void_p))
    return OracleResult(res.kind, res.msg.decode("utf-8", "replace"), out, res.in_consumed)


def _opts(unpacked_size_mode=READ_FROM_HEADER, provided=None, memlimit=None):
    o = _Options()
    o.unpacked_size_mode = unpacked_size_mode
    o.provided_is_some = 0 if provided is None else 1
    o.provided = 0 if provided is None else provided
    o.memlimit_is_some = 0 if memlimit is None else 1
    o.memlimit = 0 if memlimit is None else memlimit
    return o


def lzma_decompress(data, unpacked_size_mode=READ_FROM_HEADER, provided=None, memlimit=None):
    res = _Result()
    o = _opts(unpacked_size_mode, provided, memlimit)
    lib().orc_lzma_decompress(data, ctypes.c_size_t(len(data)), ctypes.byref(o), ctypes.byref(res))
    return _take(res)


def lzma_raw_decompress(data, lc, lp, pb, dict_size, unpacked_size=None, memlimit=None):
    res = _Result()
    lib().orc_lzma_raw_decompress(
        data, ctypes.c_size_t(len(data)), ctypes.c_uint32(lc), ctypes.c_uint32(lp),
        ctypes.c_uint32(pb), ctypes.c_uint32(dict_size),
        ctypes.c_int(0 if unpacked_size is None else 1),
        ctypes.c_uint64(0 if unpacked_size is None else unpacked_size),
        ctypes.c_int(0 if memlimit is None else 1),
        ctypes.c_uint64(0 if memlimit is None else memlimit), ctypes.byref(res))
    return _take(res)


def lzma2_decompress(data):
    res = _Result()
    lib().orc_lzma2_decompress(data, ctypes.c_size_t(len(data)), ctypes.byref(res))
    return _take(res)


def xz_decompress(data):
    res = _Result()
    lib().orc_xz_decompress(data, ctypes.c_size_t(len(data)), ctypes.byref(res))
    return _take(res)


def crc32(data):
    return lib().orc_crc32(data, len(data))


def crc64(data):
    return lib().orc_crc64(data, len(data))


class Stream:
    """The oracle's restatement of lzma_rs::decompress::Stream over a Vec<u8> sink (src/decode/stream.rs): write_all() raises
    StreamWriteError(text of the io::Error) like Write::write_all returns Err; finish() -> OracleResult."""

    class WriteError(Exception):
        pass

    def __init__(self, unpacked_size_mode=READ_FROM_HEADER, provided=None, memlimit=None, allow_incomplete=False):
        L = lib()
        L.orc_stream_new.restype = ctypes.c_void_p
        L.orc_stream_new.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_stream_write_all.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        L.orc_stream_output.restype = ctypes.c_size_t
        L.orc_stream_output.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8))]
        L.orc_stream_finish.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        o = _opts(unpacked_size_mode, provided, memlimit)
        self._h = L.orc_stream_new(ctypes.byref(o), 1 if allow_incomplete else 0)

    def write_all(self, data):
        msg = ctypes.create_string_buffer(384)
        if lib().orc_stream_write_all(self._h, bytes(data), len(data), msg) != 0:
            raise Stream.WriteError(msg.value.decode("utf-8", "replace"))

    def last_taken(self):
        """bytes the last write_all got rid of (the Ok(n) of Stream::write summed)"""
        lib().orc_stream_last_taken.restype = ctypes.c_size_t
        lib().orc_stream_last_taken.argtypes = [ctypes.c_void_p]
        return lib().orc_stream_last_taken(self._h)

    def get_output(self):
        """Stream::get_output: the sink's bytes, or None after a failed write"""
        lib().orc_stream_has_output.argtypes = [ctypes.c_void_p]
        if not lib().orc_stream_has_output(self._h):
            return None
        p = ctypes.POINTER(ctypes.c_uint8)()
        n = lib().orc_stream_output(self._h, ctypes.byref(p))
        return ctypes.string_at(p, n) if n else b""

    def finish(self):
        res = _Result()
        lib().orc_stream_finish(self._h, ctypes.byref(res))
        self._h = None
        return _take(res)
