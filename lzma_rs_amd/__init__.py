"""lzma_rs_amd -- MI355X-native batched LZMA / LZMA2 / XZ decoding.

Python face of the C ABI in include/milzma.h (ctypes, no torch types).  It mirrors the
decode surface of the reference crate gendx/lzma-rs (src/lib.rs:44-105):

    lzma_decompress(input, output)                      src/lib.rs:44-49
    lzma_decompress_with_options(input, output, opts)   src/lib.rs:52-60
    lzma2_decompress(input, output)                     src/lib.rs:83-88
    xz_decompress(input, output)                        src/lib.rs:100-105
    decompress.Options / decompress.UnpackedSize        src/decode/options.rs
    error.Error {IoError, HeaderTooShort, LzmaError, XzError}   src/error.rs

`input` is bytes-like or a binary file object (the reference's `R: io::BufRead`); `output` is a
binary file object or bytearray (its `W: io::Write`).  On error the bytes the reference would
already have pushed to `W` are still written before the exception is raised, and a seekable input
is left where the reference would have left its reader.

All decoding runs in HIP kernels on the GPU; importing this package never touches the GPU, and
there is no CPU fallback: creating a Context without an MI355X raises InfraError.
"""
import ctypes
import os
import sys

from . import workloads  # noqa: F401  (synthetic stream generator, host-only)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("MILZMA_LIB") or os.path.join(_HERE, "libmilzma.so")

# ---- error kinds / statuses (include/milzma.h) ---------------------------------------------
OK, IO_ERROR, HEADER_TOO_SHORT, LZMA_ERROR, XZ_ERROR, INFRA_ERROR = range(6)
KIND_RAW_LZMA, KIND_LZMA2 = 0, 1
SIZE_UNKNOWN = 0xFFFFFFFFFFFFFFFF
NO_LIMIT = 0xFFFFFFFFFFFFFFFF
ST_OK = 0
ST_OUT_FULL = 32
PARKED = 1                      # Result.err_a of a unit that stopped for room and can be resumed (MILZMA_PARKED)
DECODE_GROW, DECODE_RESUME, DECODE_FEED = 1, 2, 4   # flags of decode_units_ex (MILZMA_DECODE_*)
ST_NEED_INPUT = 37              # ... DECODE_FEED: parked within 20 bytes of the end of its input view (err_a == PARKED)
STREAMS_AS_READER = 0x100       # or-ed into Streams' kind: finish() = the verdict of the one-shot call over a reader (MILZMA_STREAMS_AS_READER)
KIND_LAST_VIEW = 0x80           # or-ed into Unit.kind in a DECODE_FEED call: the view ends where the stream ends
KIND_PARTIAL = 0x10             # ... an end marker that ends a view which is not the last does not end the unit (the crate's Partial mode)


class Error(Exception):
    """error::Error (src/error.rs:8-17); str() is the reference's Display string."""
    kind = None

    def __init__(self, msg, written=0, in_consumed=0):
        super().__init__(msg)
        self.written = written
        self.in_consumed = in_consumed


class IoError(Error):
    kind = IO_ERROR


class HeaderTooShort(Error):
    kind = HEADER_TOO_SHORT


class LzmaError(Error):
    kind = LZMA_ERROR


class XzError(Error):
    kind = XZ_ERROR


PATH_STREAMED, PATH_TWO_PART_INPUT, PATH_CLASSIC, PATH_GROUPED = 1, 2, 4, 8   # milzma_last_call_paths


class InfraError(Error):
    """Not a reference error: no GPU, HIP failure, bad argument."""
    kind = INFRA_ERROR


_ERRORS = {IO_ERROR: IoError, HEADER_TOO_SHORT: HeaderTooShort, LZMA_ERROR: LzmaError,
           XZ_ERROR: XzError, INFRA_ERROR: InfraError}


class UnpackedSize:
    """decompress::UnpackedSize (src/decode/options.rs:22-43)."""
    READ_FROM_HEADER = 0
    READ_HEADER_BUT_USE_PROVIDED = 1
    USE_PROVIDED = 2

    def __init__(self, mode=0, provided=None):
        self.mode = mode
        self.provided = provided

    @classmethod
    def ReadFromHeader(cls):
        return cls(cls.READ_FROM_HEADER)

    @classmethod
    def ReadHeaderButUseProvided(cls, x):
        return cls(cls.READ_HEADER_BUT_USE_PROVIDED, x)

    @classmethod
    def UseProvided(cls, x):
        return cls(cls.USE_PROVIDED, x)


class Options:
    """decompress::Options (src/decode/options.rs:3-20)."""

    def __init__(self, unpacked_size=None, memlimit=None, allow_incomplete=False):
        self.unpacked_size = unpacked_size or UnpackedSize.ReadFromHeader()
        self.memlimit = memlimit
        self.allow_incomplete = allow_incomplete  # stream API only (Streams); no effect on the one-shot calls (as in the crate)


# ---- ctypes mirror of the ABI structs --------------------------------------------------------
class Unit(ctypes.Structure):
    _fields_ = [("in_off", ctypes.c_uint64), ("in_len", ctypes.c_uint64),
                ("out_off", ctypes.c_uint64), ("out_cap", ctypes.c_uint64),
                ("unpacked_size", ctypes.c_uint64), ("memlimit", ctypes.c_uint64),
                ("dict_size", ctypes.c_uint32), ("lc", ctypes.c_uint8), ("lp", ctypes.c_uint8),
                ("pb", ctypes.c_uint8), ("kind", ctypes.c_uint8)]


class Result(ctypes.Structure):
    _fields_ = [("status", ctypes.c_uint32), ("chunks", ctypes.c_uint32),
                ("out_len", ctypes.c_uint64), ("out_flushed", ctypes.c_uint64),
                ("in_consumed", ctypes.c_uint64), ("err_a", ctypes.c_uint64),
                ("err_b", ctypes.c_uint64)]


class _COptions(ctypes.Structure):
    _fields_ = [("unpacked_size_mode", ctypes.c_int32), ("provided_is_some", ctypes.c_int32),
                ("provided", ctypes.c_uint64), ("memlimit_is_some", ctypes.c_int32),
                ("allow_incomplete", ctypes.c_int32), ("memlimit", ctypes.c_uint64)]


class _COutput(ctypes.Structure):
    _fields_ = [("data", ctypes.POINTER(ctypes.c_uint8)), ("len", ctypes.c_size_t),
                ("in_consumed", ctypes.c_size_t), ("kind", ctypes.c_int32),
                ("msg", ctypes.c_char * 388)]


EXPORTS = [
    "milzma_abi_version", "milzma_create", "milzma_destroy", "milzma_last_error",
    "milzma_decode_units", "milzma_decode_units_async", "milzma_decode_units_wait", "milzma_decode_units_host", "milzma_last_kernel_ms", "milzma_last_call_paths", "milzma_crc_units",
    "milzma_result_message", "milzma_default_options", "milzma_free",
    "milzma_lzma_decompress", "milzma_lzma2_decompress", "milzma_xz_decompress",
    "milzma_lzma_decompress_batch", "milzma_lzma2_decompress_batch", "milzma_xz_decompress_batch",
    "milzma_lzma_read_header", "milzma_crc32", "milzma_crc64", "milzma_xz_plan",
    "milzma_multi_create", "milzma_multi_destroy", "milzma_multi_devices", "milzma_multi_last_error",
    "milzma_multi_last_kernel_ms", "milzma_partition", "milzma_multi_decode_units_host", "milzma_multi_decode_units",
    "milzma_multi_lzma_decompress_batch", "milzma_multi_lzma2_decompress_batch", "milzma_multi_xz_decompress_batch",
    "milzma_lzma_decompress_batch_async", "milzma_lzma2_decompress_batch_async", "milzma_xz_decompress_batch_async",
    "milzma_batch_wait",
    "milzma_decode_units_ex", "milzma_move_units", "milzma_pool_trim",
    "milzma_multi_decode_units_rooted", "milzma_multi_last_transfer_ms",
    "milzma_streams_open", "milzma_streams_write", "milzma_streams_write_error", "milzma_streams_finish", "milzma_streams_close",
    "milzma_streams_last_error", "milzma_streams_output", "milzma_streams_write_taken",
]

_lib = None


def lib():
    """Loads libmilzma.so (built in-tree by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7; whichever HIP runtime is loaded first serves the
    # whole process (same SONAME), and torch cannot initialise on top of the system one.  Load
    # torch first so that device tensors and this library share one runtime.
    if "torch" not in sys.modules and not os.environ.get("MILZMA_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(_LIB_PATH):
        raise InfraError("libmilzma.so is not built: run `make -C lzma_rs_amd/csrc` "
                         "(or __graft_entry__.build()); there is no fallback path")
    L = ctypes.CDLL(_LIB_PATH)
    vp, u32, u64, sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_size_t
    L.milzma_abi_version.restype = u32
    L.milzma_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.milzma_destroy.argtypes = [vp]
    L.milzma_last_error.restype = ctypes.c_char_p
    L.milzma_last_error.argtypes = [vp]
    L.milzma_decode_units.argtypes = [vp, ctypes.POINTER(Unit), u32, vp, vp, ctypes.POINTER(Result), vp]
    L.milzma_decode_units_ex.argtypes = [vp, ctypes.POINTER(Unit), u32, vp, vp, ctypes.POINTER(Result), vp, u32]
    L.milzma_move_units.argtypes = [vp, u32, vp, ctypes.POINTER(u64), vp, ctypes.POINTER(u64), ctypes.POINTER(u64), vp]
    L.milzma_pool_trim.restype = sz
    L.milzma_pool_trim.argtypes = [sz]
    L.milzma_decode_units_async.argtypes = [vp, ctypes.POINTER(Unit), u32, vp, vp, vp]
    L.milzma_decode_units_wait.argtypes = [vp, ctypes.POINTER(Result)]
    L.milzma_decode_units_host.argtypes = [vp, ctypes.POINTER(Unit), u32, vp, sz, vp, sz,
                                           ctypes.POINTER(Result)]
    L.milzma_crc_units.argtypes = [vp, ctypes.POINTER(Unit), u32, vp, ctypes.POINTER(Result),
                                   ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64), vp]
    L.milzma_last_kernel_ms.restype = ctypes.c_float
    L.milzma_last_kernel_ms.argtypes = [vp, ctypes.POINTER(u32)]
    L.milzma_last_call_paths.restype = u32
    L.milzma_last_call_paths.argtypes = [vp]
    L.milzma_result_message.argtypes = [ctypes.POINTER(Result), u32, ctypes.c_char_p, sz]
    L.milzma_free.argtypes = [vp]
    L.milzma_lzma_decompress.argtypes = [vp, vp, sz, ctypes.POINTER(_COptions), ctypes.POINTER(_COutput)]
    L.milzma_lzma2_decompress.argtypes = [vp, vp, sz, ctypes.POINTER(_COutput)]
    L.milzma_xz_decompress.argtypes = [vp, vp, sz, ctypes.POINTER(_COutput)]
    L.milzma_lzma_decompress_batch.argtypes = [vp, u32, ctypes.POINTER(vp), ctypes.POINTER(sz),
                                               ctypes.POINTER(_COptions), ctypes.POINTER(_COutput)]
    L.milzma_lzma2_decompress_batch.argtypes = [vp, u32, ctypes.POINTER(vp), ctypes.POINTER(sz),
                                                ctypes.POINTER(_COutput)]
    L.milzma_xz_decompress_batch.argtypes = [vp, u32, ctypes.POINTER(vp), ctypes.POINTER(sz),
                                             ctypes.POINTER(_COutput)]
    L.milzma_lzma_decompress_batch_async.argtypes = L.milzma_lzma_decompress_batch.argtypes
    L.milzma_lzma2_decompress_batch_async.argtypes = L.milzma_lzma2_decompress_batch.argtypes
    L.milzma_xz_decompress_batch_async.argtypes = L.milzma_xz_decompress_batch.argtypes
    L.milzma_batch_wait.argtypes = [vp]
    L.milzma_lzma_read_header.argtypes = [vp, sz, ctypes.POINTER(_COptions), ctypes.POINTER(Unit),
                                          ctypes.POINTER(sz), ctypes.POINTER(_COutput)]
    L.milzma_xz_plan.argtypes = [vp, sz, ctypes.POINTER(Unit), u32, ctypes.POINTER(u32), ctypes.POINTER(u32)]
    L.milzma_multi_create.argtypes = [u64, ctypes.POINTER(vp)]
    L.milzma_multi_destroy.argtypes = [vp]
    L.milzma_multi_devices.restype = u32
    L.milzma_multi_devices.argtypes = [vp, ctypes.POINTER(ctypes.c_int), u32]
    L.milzma_multi_last_error.restype = ctypes.c_char_p
    L.milzma_multi_last_error.argtypes = [vp]
    L.milzma_multi_last_kernel_ms.restype = ctypes.c_float
    L.milzma_multi_last_kernel_ms.argtypes = [vp, u32, ctypes.POINTER(u32)]
    L.milzma_partition.argtypes = [ctypes.POINTER(u64), ctypes.POINTER(u32), u32, u32, ctypes.POINTER(u32)]
    L.milzma_multi_decode_units_host.argtypes = [vp, ctypes.POINTER(Unit), u32, vp, sz, vp, sz, ctypes.POINTER(Result)]
    L.milzma_multi_decode_units.argtypes = [vp, ctypes.POINTER(Unit), u32, ctypes.POINTER(u32), ctypes.POINTER(vp),
                                            ctypes.POINTER(vp), ctypes.POINTER(Result)]
    L.milzma_multi_decode_units_rooted.argtypes = [vp, u32, ctypes.POINTER(Unit), u32, vp, vp, ctypes.POINTER(Result)]
    L.milzma_multi_last_transfer_ms.restype = None
    L.milzma_multi_last_transfer_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    L.milzma_multi_lzma_decompress_batch.argtypes = [vp, u32, ctypes.POINTER(vp), ctypes.POINTER(sz),
                                                     ctypes.POINTER(_COptions), ctypes.POINTER(_COutput)]
    L.milzma_multi_lzma2_decompress_batch.argtypes = [vp, u32, ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.POINTER(_COutput)]
    L.milzma_multi_xz_decompress_batch.argtypes = [vp, u32, ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.POINTER(_COutput)]
    L.milzma_crc32.restype = u32
    L.milzma_crc32.argtypes = [vp, sz]
    L.milzma_crc64.restype = u64
    L.milzma_crc64.argtypes = [vp, sz]
    L.milzma_streams_open.argtypes = [vp, u32, u32, ctypes.POINTER(_COptions), ctypes.POINTER(vp)]
    L.milzma_streams_write.argtypes = [vp, u32, ctypes.POINTER(u32), ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.POINTER(ctypes.c_int32)]
    L.milzma_streams_write_error.restype = ctypes.c_char_p
    L.milzma_streams_write_error.argtypes = [vp, u32]
    L.milzma_streams_finish.argtypes = [vp, ctypes.POINTER(_COutput)]
    L.milzma_streams_close.argtypes = [vp]
    L.milzma_streams_last_error.restype = ctypes.c_char_p
    L.milzma_streams_last_error.argtypes = [vp]
    L.milzma_streams_write_taken.restype = ctypes.c_uint64
    L.milzma_streams_write_taken.argtypes = [vp, ctypes.c_uint32]
    L.milzma_streams_output.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint64, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64),
                                        ctypes.POINTER(ctypes.c_int32)]
    _lib = L
    return L


def _c_options(options):
    o = _COptions()
    if options is None:
        return o
    us = options.unpacked_size
    o.unpacked_size_mode = us.mode
    o.provided_is_some = 0 if us.provided is None else 1
    o.provided = 0 if us.provided is None else us.provided
    o.memlimit_is_some = 0 if options.memlimit is None else 1
    o.memlimit = 0 if options.memlimit is None else options.memlimit
    o.allow_incomplete = 1 if options.allow_incomplete else 0
    return o


class Decoded:
    """Outcome of one whole-file call: bytes for the writer, reader advance, error (or None)."""

    def __init__(self, cout):
        self.data = ctypes.string_at(cout.data, cout.len) if cout.len else b""
        self.in_consumed = cout.in_consumed
        self.kind = cout.kind
        self.msg = cout.msg.decode("utf-8", "replace")
        if cout.data:
            lib().milzma_free(ctypes.cast(cout.data, ctypes.c_void_p))

    @property
    def ok(self):
        return self.kind == OK

    def error(self):
        if self.kind == OK:
            return None
        return _ERRORS[self.kind](self.msg, written=len(self.data), in_consumed=self.in_consumed)

    def __repr__(self):
        return "Decoded(kind=%d, msg=%r, len=%d, in_consumed=%d)" % (
            self.kind, self.msg, len(self.data), self.in_consumed)


def _as_buffer(data):
    """bytes-like -> (ctypes pointer, length, keepalive)."""
    if isinstance(data, (bytes, bytearray)):
        buf = (ctypes.c_char * len(data)).from_buffer_copy(data) if len(data) else ctypes.c_char_p(b"")
        return ctypes.cast(buf, ctypes.c_void_p), len(data), buf
    mv = memoryview(data).cast("B")
    buf = (ctypes.c_char * len(mv)).from_buffer_copy(mv) if len(mv) else ctypes.c_char_p(b"")
    return ctypes.cast(buf, ctypes.c_void_p), len(mv), buf


class Context:
    """milzma_ctx bound to one GPU.  Raises InfraError when no MI355X / HIP runtime is usable."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        L = lib()
        if L.milzma_create(device, ctypes.byref(self._h)) != OK:
            raise InfraError("milzma_create failed: " + (L.milzma_last_error(None) or b"").decode())
        self.device = device

    def close(self):
        if self._h:
            lib().milzma_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        return (lib().milzma_last_error(self._h) or b"").decode()

    # ---- unit-level batch API (device pointers as ints, e.g. torch_tensor.data_ptr()) --------
    def decode_units(self, units, d_in, d_out, stream=0):
        """units: ctypes array of Unit.  Returns (ctypes array of Result, kernel_ms, launches)."""
        n = len(units)
        results = (Result * n)()
        r = lib().milzma_decode_units(self._h, units, n, ctypes.c_void_p(d_in), ctypes.c_void_p(d_out),
                                      results, ctypes.c_void_p(stream))
        if r != OK:
            raise InfraError("milzma_decode_units: " + self.last_error())
        launches = ctypes.c_uint32()
        ms = lib().milzma_last_kernel_ms(self._h, ctypes.byref(launches))
        return results, ms, launches.value

    def last_call_paths(self):
        """milzma_last_call_paths: PATH_* bits of the way the most recent whole-file batch call took (streamed launch / classic rounds ...)"""
        return int(lib().milzma_last_call_paths(self._h))

    def decode_units_ex(self, units, d_in, d_out, flags, results=None, stream=0):
        """milzma_decode_units_ex: DECODE_GROW parks units that run out of room (Result.status == ST_OUT_FULL, err_a == PARKED);
        DECODE_RESUME (with the previous call's `results`) continues them in their new, larger slices.
        DECODE_FEED: every unit's (in_off, in_len) is a view of a stream that goes on; units that come within 20 bytes of its end park
        (ST_NEED_INPUT, err_a == PARKED) and are resumed with a view that starts at their first unused byte (include/milzma.h).
        Returns (results, kernel_ms, launches); `results` is updated in place when given."""
        n = len(units)
        if results is None:
            results = (Result * n)()
        r = lib().milzma_decode_units_ex(self._h, units, n, ctypes.c_void_p(d_in), ctypes.c_void_p(d_out), results,
                                         ctypes.c_void_p(stream), flags)
        if r != OK:
            raise InfraError("milzma_decode_units_ex: " + self.last_error())
        launches = ctypes.c_uint32()
        ms = lib().milzma_last_kernel_ms(self._h, ctypes.byref(launches))
        return results, ms, launches.value

    def move_units(self, d_src, src_off, d_dst, dst_off, lens, stream=0):
        """milzma_move_units: d_dst[dst_off[i], +lens[i]) = d_src[src_off[i], +lens[i]) on the device, one launch."""
        n = len(lens)
        A = ctypes.c_uint64 * n
        r = lib().milzma_move_units(self._h, n, ctypes.c_void_p(d_src), A(*src_off), ctypes.c_void_p(d_dst), A(*dst_off), A(*lens),
                                    ctypes.c_void_p(stream))
        if r != OK:
            raise InfraError("milzma_move_units: " + self.last_error())

    def decode_units_async(self, units, d_in, d_out, stream=0):
        """Enqueue only (milzma_decode_units_async); finish with decode_units_wait(len(units))."""
        r = lib().milzma_decode_units_async(self._h, units, len(units), ctypes.c_void_p(d_in), ctypes.c_void_p(d_out),
                                            ctypes.c_void_p(stream))
        if r != OK:
            raise InfraError("milzma_decode_units_async: " + self.last_error())

    def decode_units_wait(self, n):
        """Returns (ctypes array of Result, kernel_ms, launches) of the batch in flight."""
        results = (Result * n)()
        if lib().milzma_decode_units_wait(self._h, results) != OK:
            raise InfraError("milzma_decode_units_wait: " + self.last_error())
        launches = ctypes.c_uint32()
        ms = lib().milzma_last_kernel_ms(self._h, ctypes.byref(launches))
        return results, ms, launches.value

    def crc_units(self, units, results, d_out, stream=0):
        """CRC-32 / CRC-64(XZ) of each unit's decoded (device-resident) output, computed on the GPU.
        Returns (list of crc32, list of crc64)."""
        n = len(units)
        c32 = (ctypes.c_uint32 * n)()
        c64 = (ctypes.c_uint64 * n)()
        r = lib().milzma_crc_units(self._h, units, n, ctypes.c_void_p(d_out), results, c32, c64, ctypes.c_void_p(stream))
        if r != OK:
            raise InfraError("milzma_crc_units: " + self.last_error())
        return list(c32), list(c64)

    def decode_units_host(self, units, h_in, out_bytes):
        """Host-resident variant: h_in bytes-like; returns (results, bytearray output)."""
        n = len(units)
        results = (Result * n)()
        pin, nin, keep = _as_buffer(h_in)
        out = (ctypes.c_char * max(out_bytes, 1))()
        r = lib().milzma_decode_units_host(self._h, units, n, pin, nin, ctypes.cast(out, ctypes.c_void_p),
                                           out_bytes, results)
        del keep
        if r != OK:
            raise InfraError("milzma_decode_units_host: " + self.last_error())
        return results, bytearray(out)[:out_bytes]

    # ---- whole-file API -------------------------------------------------------------------
    def lzma(self, data, options=None):
        out = _COutput()
        p, n, keep = _as_buffer(data)
        o = _c_options(options)
        lib().milzma_lzma_decompress(self._h, p, n, ctypes.byref(o), ctypes.byref(out))
        return Decoded(out)

    def lzma2(self, data):
        out = _COutput()
        p, n, keep = _as_buffer(data)
        lib().milzma_lzma2_decompress(self._h, p, n, ctypes.byref(out))
        return Decoded(out)

    def xz(self, data):
        out = _COutput()
        p, n, keep = _as_buffer(data)
        lib().milzma_xz_decompress(self._h, p, n, ctypes.byref(out))
        return Decoded(out)

    def _batch(self, fn, datas, options=None, with_options=False):
        n = len(datas)
        bufs = [_as_buffer(d) for d in datas]
        ptrs = (ctypes.c_void_p * n)(*[b[0] for b in bufs])
        lens = (ctypes.c_size_t * n)(*[b[1] for b in bufs])
        outs = (_COutput * n)()
        if with_options:
            o = _c_options(options)
            rc = fn(self._h, n, ptrs, lens, ctypes.byref(o), outs)
        else:
            rc = fn(self._h, n, ptrs, lens, outs)
        decs = [Decoded(outs[i]) for i in range(n)]   # (also hands the buffers of a failed call back)
        if rc != OK:                                  # an infrastructure failure of the call itself, as batch_wait reports it
            raise InfraError("batch call failed: " + self.last_error())
        return decs

    def lzma_batch(self, datas, options=None):
        return self._batch(lib().milzma_lzma_decompress_batch, datas, options, True)

    def lzma2_batch(self, datas):
        return self._batch(lib().milzma_lzma2_decompress_batch, datas)

    def xz_batch(self, datas):
        return self._batch(lib().milzma_xz_decompress_batch, datas)

    def batch_async(self, kind, datas, options=None):
        """milzma_{lzma,lzma2,xz}_decompress_batch_async: returns at once; batch_wait() gives the list of Decoded."""
        n = len(datas)
        bufs = [_as_buffer(d) for d in datas]
        ptrs = (ctypes.c_void_p * n)(*[b[0] for b in bufs])
        lens = (ctypes.c_size_t * n)(*[b[1] for b in bufs])
        outs = (_COutput * n)()
        L = lib()
        if kind == "lzma":
            o = _c_options(options)
            r = L.milzma_lzma_decompress_batch_async(self._h, n, ptrs, lens, ctypes.byref(o), outs)
        elif kind == "lzma2":
            r = L.milzma_lzma2_decompress_batch_async(self._h, n, ptrs, lens, outs)
        else:
            r = L.milzma_xz_decompress_batch_async(self._h, n, ptrs, lens, outs)
        if r != OK:
            raise InfraError("batch_async: " + self.last_error())
        self._inflight = (bufs, outs, n)

    def batch_wait(self):
        bufs, outs, n = self._inflight
        rc = lib().milzma_batch_wait(self._h)
        self._inflight = None
        decs = [Decoded(outs[i]) for i in range(n)]
        if rc != OK:
            raise InfraError("batch_wait: " + self.last_error())
        return decs


def partition(weights, parts, groups=None):
    """milzma_partition: part index of every item (longest-processing-time-first by weight; items sharing a
    non-zero group id stay together).  Needs no GPU."""
    n = len(weights)
    w = (ctypes.c_uint64 * max(n, 1))(*weights)
    g = (ctypes.c_uint32 * max(n, 1))(*groups) if groups is not None else None
    out = (ctypes.c_uint32 * max(n, 1))()
    if lib().milzma_partition(w, g, n, parts, out) != OK:
        raise InfraError("milzma_partition: bad arguments")
    return list(out)[:n]


class MultiContext:
    """milzma_multi: the GPUs of one node behind one handle (one context + one host worker thread per device).
    device_mask: bit d = HIP ordinal d, 0 = every visible device."""

    def __init__(self, device_mask=0):
        self._h = ctypes.c_void_p()
        L = lib()
        if L.milzma_multi_create(device_mask, ctypes.byref(self._h)) != OK:
            raise InfraError("milzma_multi_create failed: " + (L.milzma_multi_last_error(None) or b"").decode())
        n = L.milzma_multi_devices(self._h, None, 0)
        ords = (ctypes.c_int * n)()
        L.milzma_multi_devices(self._h, ords, n)
        self.devices = list(ords)

    def close(self):
        if self._h:
            lib().milzma_multi_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        return (lib().milzma_multi_last_error(self._h) or b"").decode()

    def kernel_ms(self, k=0xFFFFFFFF):
        launches = ctypes.c_uint32()
        return lib().milzma_multi_last_kernel_ms(self._h, k, ctypes.byref(launches)), launches.value

    def decode_units_host(self, units, h_in, out_bytes):
        n = len(units)
        results = (Result * n)()
        pin, nin, keep = _as_buffer(h_in)
        out = (ctypes.c_char * max(out_bytes, 1))()
        r = lib().milzma_multi_decode_units_host(self._h, units, n, pin, nin, ctypes.cast(out, ctypes.c_void_p), out_bytes, results)
        del keep
        if r != OK:
            raise InfraError("milzma_multi_decode_units_host: " + self.last_error())
        return results, bytearray(out)[:out_bytes]

    def decode_units(self, units, device_of, d_ins, d_outs):
        """units: ctypes array; device_of: device index per unit; d_ins / d_outs: device pointer (int) per device."""
        n = len(units)
        results = (Result * n)()
        dev = (ctypes.c_uint32 * max(n, 1))(*device_of)
        nd = len(self.devices)
        pin = (ctypes.c_void_p * nd)(*d_ins)
        pout = (ctypes.c_void_p * nd)(*d_outs)
        if lib().milzma_multi_decode_units(self._h, units, n, dev, pin, pout, results) != OK:
            raise InfraError("milzma_multi_decode_units: " + self.last_error())
        return results

    def decode_units_rooted(self, root, units, d_in, d_out):
        """milzma_multi_decode_units_rooted: input and output resident on device index `root`; the other devices get their shares
        device to device.  Returns (results, (scatter_ms, decode_ms, gather_ms))."""
        n = len(units)
        results = (Result * n)()
        if lib().milzma_multi_decode_units_rooted(self._h, root, units, n, ctypes.c_void_p(d_in), ctypes.c_void_p(d_out), results) != OK:
            raise InfraError("milzma_multi_decode_units_rooted: " + self.last_error())
        a, b, c = ctypes.c_float(), ctypes.c_float(), ctypes.c_float()
        lib().milzma_multi_last_transfer_ms(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return results, (a.value, b.value, c.value)

    def _batch(self, fn, datas, options=None, with_options=False):
        n = len(datas)
        bufs = [_as_buffer(d) for d in datas]
        ptrs = (ctypes.c_void_p * n)(*[b[0] for b in bufs])
        lens = (ctypes.c_size_t * n)(*[b[1] for b in bufs])
        outs = (_COutput * n)()
        if with_options:
            o = _c_options(options)
            rc = fn(self._h, n, ptrs, lens, ctypes.byref(o), outs)
        else:
            rc = fn(self._h, n, ptrs, lens, outs)
        decs = [Decoded(outs[i]) for i in range(n)]
        if rc != OK:   # (never a list of empty successes: the library fills every slot, and the call's failure is raised)
            raise InfraError("multi batch call failed: " + self.last_error())
        return decs

    def lzma_batch(self, datas, options=None):
        return self._batch(lib().milzma_multi_lzma_decompress_batch, datas, options, True)

    def lzma2_batch(self, datas):
        return self._batch(lib().milzma_multi_lzma2_decompress_batch, datas)

    def xz_batch(self, datas):
        return self._batch(lib().milzma_multi_xz_decompress_batch, datas)


class Streams:
    """A batch of push-mode .lzma decoders: n x lzma_rs::decompress::Stream (feature `stream`, src/decode/stream.rs) on one GPU
    (milzma_streams_*).  write({stream: bytes, ...}) is io::Write::write_all for each named stream -- the dict it returns maps a stream to
    the text of the io::Error of a failed write (empty dict: every write succeeded); finish() is Stream::finish for every stream: a list
    of Decoded."""

    def __init__(self, ctx, n, options=None, kind=KIND_RAW_LZMA):
        self.n = n
        self._h = ctypes.c_void_p()
        copts = None
        if options is not None:
            opts = options if isinstance(options, (list, tuple)) else [options] * n
            assert len(opts) == n
            copts = (_COptions * n)(*[_c_options(o) for o in opts])
        if lib().milzma_streams_open(ctx._h, kind, n, copts, ctypes.byref(self._h)) != OK:
            raise InfraError("milzma_streams_open: " + ctx.last_error())

    def write(self, pieces):
        items = [(i, b if isinstance(b, bytes) else bytes(b)) for i, b in pieces.items()]
        k = len(items)
        idx = (ctypes.c_uint32 * k)(*[i for i, _ in items])
        # (the bytes objects themselves are read, not copies of them: they stay alive in `items` for the call)
        ptrs = (ctypes.c_void_p * k)(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p).value if b else None for _, b in items])
        lens = (ctypes.c_size_t * k)(*[len(b) for _, b in items])
        status = (ctypes.c_int32 * k)()
        if lib().milzma_streams_write(self._h, k, idx, ptrs, lens, status) != OK:
            raise InfraError("milzma_streams_write: " + lib().milzma_streams_last_error(self._h).decode())
        return {items[j][0]: lib().milzma_streams_write_error(self._h, items[j][0]).decode() for j in range(k) if status[j] != OK}

    def taken(self, i):
        """bytes of the most recent write that stream i took (the sum of the Ok(n) of the crate's Stream::write under write_all)"""
        return lib().milzma_streams_write_taken(self._h, i)

    def output(self, i):
        """Stream::get_output (stream.rs:102-116) of stream i: the bytes its sink holds right now (every completed flush of the ring), or
        None after a failed write (the crate's state is gone)."""
        n, has = ctypes.c_uint64(), ctypes.c_int32()
        if lib().milzma_streams_output(self._h, i, 0, None, 0, ctypes.byref(n), ctypes.byref(has)) != OK:
            raise InfraError("milzma_streams_output: " + lib().milzma_streams_last_error(self._h).decode())
        if not has.value:
            return None
        buf = ctypes.create_string_buffer(max(1, n.value))
        if n.value and lib().milzma_streams_output(self._h, i, 0, buf, n.value, ctypes.byref(n), ctypes.byref(has)) != OK:
            raise InfraError("milzma_streams_output: " + lib().milzma_streams_last_error(self._h).decode())
        return buf.raw[:n.value]

    def finish(self):
        outs = (_COutput * self.n)()
        rc = lib().milzma_streams_finish(self._h, outs)
        decs = [Decoded(outs[i]) for i in range(self.n)]
        if rc != OK:
            raise InfraError("milzma_streams_finish: " + lib().milzma_streams_last_error(self._h).decode())
        return decs

    def close(self):
        if self._h:
            lib().milzma_streams_close(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decompress_reader(ctx, kind, data, bufsize, options=None):
    """The one-shot `lzma_decompress` / `lzma2_decompress` over an `io::BufRead` whose buffer holds `bufsize` bytes -- a BufReader over a
    socket or a large file -- the way integration/rust/src/lib.rs `run_fed` does it: every `fill_buf` view is written to a push-mode
    stream in READER mode and consumed; a write that fails (or reports WriteZero: the stream has ended) ends the loop, and finish says how
    much of the last view the decoder used.  Returns (Decoded, reader position afterwards)."""
    s = Streams(ctx, 1, options, kind=kind | STREAMS_AS_READER)
    pos = fed = 0
    while True:
        view = bytes(data[pos:pos + bufsize])      # fill_buf
        if not view:
            break
        if s.write({0: view}):
            break                                  # (not consumed: finish says how much of it belongs to the stream)
        pos += len(view)                           # consume
        fed += len(view)
    d = s.finish()[0]
    s.close()
    return d, pos + max(0, d.in_consumed - fed)


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("MILZMA_USE_LOCAL_RANK") else 0)
    return _default_ctx


# ---- the crate's function surface -------------------------------------------------------------
def _read_all(inp):
    if isinstance(inp, (bytes, bytearray, memoryview)):
        return bytes(inp), None, 0
    start = inp.tell() if inp.seekable() else None
    return inp.read(), inp, start


def _deliver(dec, output, inp, start):
    if isinstance(output, bytearray):
        output += dec.data
    else:
        output.write(dec.data)
    if inp is not None and start is not None:
        inp.seek(start + dec.in_consumed)
    err = dec.error()
    if err is not None:
        raise err


def lzma_decompress_with_options(input, output, options, ctx=None):
    data, inp, start = _read_all(input)
    _deliver((ctx or default_context()).lzma(data, options), output, inp, start)


def lzma_decompress(input, output, ctx=None):
    lzma_decompress_with_options(input, output, Options(), ctx)


def lzma2_decompress(input, output, ctx=None):
    data, inp, start = _read_all(input)
    _deliver((ctx or default_context()).lzma2(data), output, inp, start)


def xz_decompress(input, output, ctx=None):
    data, inp, start = _read_all(input)
    _deliver((ctx or default_context()).xz(data), output, inp, start)


# ---- host-only helpers (no GPU) ----------------------------------------------------------------
def lzma_read_header(data, options=None):
    """LzmaParams::read_header: returns (Unit, header_len) or raises the reference's error."""
    u = Unit()
    hl = ctypes.c_size_t()
    out = _COutput()
    p, n, keep = _as_buffer(data)
    o = _c_options(options)
    kind = lib().milzma_lzma_read_header(p, n, ctypes.byref(o), ctypes.byref(u), ctypes.byref(hl),
                                         ctypes.byref(out))
    if kind != OK:
        raise _ERRORS[kind](out.msg.decode())
    return u, hl.value


def xz_plan(data):
    """One LZMA2 Unit per block of a well-formed .xz file (offsets relative to the file) and the check id,
    or raises XzError when the file's Index cannot be used."""
    p, n, keep = _as_buffer(data)
    cnt, chk = ctypes.c_uint32(), ctypes.c_uint32()
    if lib().milzma_xz_plan(p, n, None, 0, ctypes.byref(cnt), ctypes.byref(chk)) == XZ_ERROR:
        raise XzError("xz error: the file's index cannot be used to plan its blocks")
    units = (Unit * max(cnt.value, 1))()
    if lib().milzma_xz_plan(p, n, units, cnt.value, ctypes.byref(cnt), ctypes.byref(chk)) != OK:
        raise XzError("xz error: the file's index cannot be used to plan its blocks")
    return [units[i] for i in range(cnt.value)], chk.value


def crc32(data):
    p, n, keep = _as_buffer(data)
    return lib().milzma_crc32(p, n)


def crc64(data):
    p, n, keep = _as_buffer(data)
    return lib().milzma_crc64(p, n)


def result_message(result, unit_kind):
    buf = ctypes.create_string_buffer(400)
    kind = lib().milzma_result_message(ctypes.byref(result), unit_kind, buf, 400)
    return kind, buf.value.decode()
