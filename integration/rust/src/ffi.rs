//! `extern "C"` declarations of include/milzma.h, one for one (struct layouts, constants, every exported
//! symbol).  tests/test_host_abi.py diffs the function names and the struct field lists against the header.
#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_float, c_int, c_void};

pub const MILZMA_ABI_VERSION: u32 = 6;

// error kinds: error::Error variants (src/error.rs:8-17)
pub const MILZMA_OK: c_int = 0;
pub const MILZMA_IO_ERROR: c_int = 1;
pub const MILZMA_HEADER_TOO_SHORT: c_int = 2;
pub const MILZMA_LZMA_ERROR: c_int = 3;
pub const MILZMA_XZ_ERROR: c_int = 4;
pub const MILZMA_INFRA_ERROR: c_int = 5;

pub const MILZMA_KIND_RAW_LZMA: u8 = 0;
pub const MILZMA_KIND_LZMA2: u8 = 1;
/// or-ed into `milzma_unit.kind` in a `MILZMA_DECODE_FEED` call: this unit's view ends where its stream ends
pub const MILZMA_KIND_LAST_VIEW: u8 = 0x80;
/// ... in a RESUME | FEED call: this parked unit stays parked (nothing new for it)
pub const MILZMA_KIND_HOLD: u8 = 0x40;
/// ... in a RESUME | FEED call: this unit is new and starts now, beside the units that resume
pub const MILZMA_KIND_START: u8 = 0x20;
/// ... in a FEED call, RAW units: an end marker that ends a view which is not the last does not end the unit (the crate's Partial mode)
pub const MILZMA_KIND_PARTIAL: u8 = 0x10;
pub const MILZMA_SIZE_UNKNOWN: u64 = u64::MAX;
pub const MILZMA_NO_LIMIT: u64 = u64::MAX;
pub const MILZMA_MAX_UNIT_BYTES: u64 = 0xFFFF_FF00;

// unpacked_size_mode: decompress::UnpackedSize (src/decode/options.rs:22-43)
pub const MILZMA_READ_FROM_HEADER: i32 = 0;
pub const MILZMA_READ_HEADER_BUT_USE_PROVIDED: i32 = 1;
pub const MILZMA_USE_PROVIDED: i32 = 2;

pub const MILZMA_ST_OK: u32 = 0;
pub const MILZMA_ST_OUT_FULL: u32 = 32;
/// `MILZMA_DECODE_FEED`: the unit stopped within 20 bytes of the end of its input view (`err_a == MILZMA_PARKED`)
pub const MILZMA_ST_NEED_INPUT: u32 = 37;
/// `milzma_result.err_a` of a unit that stopped for room and can be resumed (MILZMA_DECODE_RESUME)
pub const MILZMA_PARKED: u64 = 1;
pub const MILZMA_DECODE_GROW: u32 = 1;
pub const MILZMA_DECODE_RESUME: u32 = 2;
/// fed input: every unit's (in_off, in_len) is a view of a stream that goes on behind it (include/milzma.h)
pub const MILZMA_DECODE_FEED: u32 = 4;
/// or-ed into `milzma_streams_open`'s kind: finish hands over what the one-shot call over a reader would (include/milzma.h)
pub const MILZMA_STREAMS_AS_READER: u32 = 0x100;
pub const MILZMA_PATH_STREAMED: u32 = 1;
pub const MILZMA_PATH_TWO_PART_INPUT: u32 = 2;
pub const MILZMA_PATH_CLASSIC: u32 = 4;
pub const MILZMA_PATH_GROUPED: u32 = 8;

/// One independent serial decode job (one wavefront).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct milzma_unit {
    pub in_off: u64,
    pub in_len: u64,
    pub out_off: u64,
    pub out_cap: u64,
    pub unpacked_size: u64,
    pub memlimit: u64,
    pub dict_size: u32,
    pub lc: u8,
    pub lp: u8,
    pub pb: u8,
    pub kind: u8,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct milzma_result {
    pub status: u32,
    pub chunks: u32,
    pub out_len: u64,
    pub out_flushed: u64,
    pub in_consumed: u64,
    pub err_a: u64,
    pub err_b: u64,
}

/// decompress::Options (src/decode/options.rs:3-20); allow_incomplete is stream-API only (milzma_streams_*).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct milzma_options {
    pub unpacked_size_mode: i32,
    pub provided_is_some: i32,
    pub provided: u64,
    pub memlimit_is_some: i32,
    pub allow_incomplete: i32,
    pub memlimit: u64,
}

/// What a whole-file call did to the caller's reader and writer.
#[repr(C)]
pub struct milzma_output {
    pub data: *mut u8,
    pub len: usize,
    pub in_consumed: usize,
    pub kind: i32,
    pub msg: [c_char; 388],
}

#[repr(C)]
pub struct milzma_ctx {
    _opaque: [u8; 0],
}

/// n push-mode .lzma decoders on one GPU (lzma_rs::decompress::Stream for a batch)
#[repr(C)]
pub struct milzma_streams {
    _opaque: [u8; 0],
}

/// Several GPUs of one node behind one handle (one context + one host worker thread per device).
#[repr(C)]
pub struct milzma_multi {
    _opaque: [u8; 0],
}

extern "C" {
    pub fn milzma_abi_version() -> u32;
    pub fn milzma_create(device: c_int, out_ctx: *mut *mut milzma_ctx) -> c_int;
    pub fn milzma_destroy(ctx: *mut milzma_ctx);
    pub fn milzma_last_error(ctx: *const milzma_ctx) -> *const c_char;

    pub fn milzma_decode_units(
        ctx: *mut milzma_ctx,
        units: *const milzma_unit,
        n: u32,
        d_in: *const c_void,
        d_out: *mut c_void,
        results: *mut milzma_result,
        hip_stream: *mut c_void,
    ) -> c_int;
    pub fn milzma_decode_units_async(
        ctx: *mut milzma_ctx,
        units: *const milzma_unit,
        n: u32,
        d_in: *const c_void,
        d_out: *mut c_void,
        hip_stream: *mut c_void,
    ) -> c_int;
    pub fn milzma_decode_units_wait(ctx: *mut milzma_ctx, results: *mut milzma_result) -> c_int;
    /// growable output: MILZMA_DECODE_GROW parks units that run out of room, MILZMA_DECODE_RESUME continues them in larger slices
    pub fn milzma_decode_units_ex(
        ctx: *mut milzma_ctx,
        units: *const milzma_unit,
        n: u32,
        d_in: *const c_void,
        d_out: *mut c_void,
        results: *mut milzma_result,
        hip_stream: *mut c_void,
        flags: u32,
    ) -> c_int;
    pub fn milzma_move_units(
        ctx: *mut milzma_ctx,
        n: u32,
        d_src: *const c_void,
        src_off: *const u64,
        d_dst: *mut c_void,
        dst_off: *const u64,
        len: *const u64,
        hip_stream: *mut c_void,
    ) -> c_int;
    pub fn milzma_decode_units_host(
        ctx: *mut milzma_ctx,
        units: *const milzma_unit,
        n: u32,
        h_in: *const c_void,
        in_bytes: usize,
        h_out: *mut c_void,
        out_bytes: usize,
        results: *mut milzma_result,
    ) -> c_int;
    pub fn milzma_last_kernel_ms(ctx: *const milzma_ctx, launches: *mut u32) -> c_float;
    pub fn milzma_last_call_paths(ctx: *const milzma_ctx) -> u32;
    pub fn milzma_crc_units(
        ctx: *mut milzma_ctx,
        units: *const milzma_unit,
        n: u32,
        d_out: *const c_void,
        results: *const milzma_result,
        crc32: *mut u32,
        crc64: *mut u64,
        hip_stream: *mut c_void,
    ) -> c_int;
    pub fn milzma_result_message(r: *const milzma_result, unit_kind: u32, msg: *mut c_char, cap: usize) -> c_int;

    pub fn milzma_default_options(opt: *mut milzma_options);
    pub fn milzma_free(p: *mut c_void);
    pub fn milzma_pool_trim(keep_bytes: usize) -> usize;

    pub fn milzma_lzma_decompress(
        ctx: *mut milzma_ctx,
        input: *const u8,
        in_len: usize,
        opt: *const milzma_options,
        out: *mut milzma_output,
    ) -> c_int;
    pub fn milzma_lzma2_decompress(ctx: *mut milzma_ctx, input: *const u8, in_len: usize, out: *mut milzma_output) -> c_int;
    pub fn milzma_xz_decompress(ctx: *mut milzma_ctx, input: *const u8, in_len: usize, out: *mut milzma_output) -> c_int;
    pub fn milzma_lzma_decompress_batch(
        ctx: *mut milzma_ctx,
        n: u32,
        ins: *const *const u8,
        in_lens: *const usize,
        opt: *const milzma_options,
        outs: *mut milzma_output,
    ) -> c_int;
    pub fn milzma_lzma2_decompress_batch(
        ctx: *mut milzma_ctx,
        n: u32,
        ins: *const *const u8,
        in_lens: *const usize,
        outs: *mut milzma_output,
    ) -> c_int;
    pub fn milzma_xz_decompress_batch(
        ctx: *mut milzma_ctx,
        n: u32,
        ins: *const *const u8,
        in_lens: *const usize,
        outs: *mut milzma_output,
    ) -> c_int;

    pub fn milzma_lzma_read_header(
        input: *const u8,
        in_len: usize,
        opt: *const milzma_options,
        unit: *mut milzma_unit,
        header_len: *mut usize,
        out: *mut milzma_output,
    ) -> c_int;
    pub fn milzma_xz_plan(
        input: *const u8,
        in_len: usize,
        units: *mut milzma_unit,
        cap: u32,
        n_units: *mut u32,
        check_id: *mut u32,
    ) -> c_int;
    pub fn milzma_crc32(p: *const u8, n: usize) -> u32;
    pub fn milzma_crc64(p: *const u8, n: usize) -> u64;

    pub fn milzma_lzma_decompress_batch_async(
        ctx: *mut milzma_ctx,
        n: u32,
        ins: *const *const u8,
        in_lens: *const usize,
        opt: *const milzma_options,
        outs: *mut milzma_output,
    ) -> c_int;
    pub fn milzma_lzma2_decompress_batch_async(
        ctx: *mut milzma_ctx,
        n: u32,
        ins: *const *const u8,
        in_lens: *const usize,
        outs: *mut milzma_output,
    ) -> c_int;
    pub fn milzma_xz_decompress_batch_async(
        ctx: *mut milzma_ctx,
        n: u32,
        ins: *const *const u8,
        in_lens: *const usize,
        outs: *mut milzma_output,
    ) -> c_int;
    pub fn milzma_batch_wait(ctx: *mut milzma_ctx) -> c_int;

    // ---- several GPUs of one node ------------------------------------------------------------------
    pub fn milzma_multi_create(device_mask: u64, out: *mut *mut milzma_multi) -> c_int;
    pub fn milzma_multi_destroy(m: *mut milzma_multi);
    pub fn milzma_multi_devices(m: *const milzma_multi, ordinals: *mut c_int, cap: u32) -> u32;
    pub fn milzma_multi_last_error(m: *const milzma_multi) -> *const c_char;
    pub fn milzma_multi_last_kernel_ms(m: *const milzma_multi, k: u32, launches: *mut u32) -> c_float;
    pub fn milzma_partition(weights: *const u64, group: *const u32, n: u32, parts: u32, part_of: *mut u32) -> c_int;
    pub fn milzma_multi_decode_units_host(
        m: *mut milzma_multi,
        units: *const milzma_unit,
        n: u32,
        h_in: *const c_void,
        in_bytes: usize,
        h_out: *mut c_void,
        out_bytes: usize,
        results: *mut milzma_result,
    ) -> c_int;
    pub fn milzma_multi_decode_units(
        m: *mut milzma_multi,
        units: *const milzma_unit,
        n: u32,
        device_of: *const u32,
        d_in: *const *const c_void,
        d_out: *const *mut c_void,
        results: *mut milzma_result,
    ) -> c_int;
    /// one ingest point: input and output resident on device index `root`; shares travel device to device (xGMI)
    pub fn milzma_multi_decode_units_rooted(
        m: *mut milzma_multi,
        root: u32,
        units: *const milzma_unit,
        n: u32,
        d_in: *const c_void,
        d_out: *mut c_void,
        results: *mut milzma_result,
    ) -> c_int;
    pub fn milzma_multi_last_transfer_ms(m: *const milzma_multi, scatter_ms: *mut f32, decode_ms: *mut f32, gather_ms: *mut f32);
    pub fn milzma_multi_lzma_decompress_batch(
        m: *mut milzma_multi,
        n: u32,
        ins: *const *const u8,
        in_lens: *const usize,
        opt: *const milzma_options,
        outs: *mut milzma_output,
    ) -> c_int;
    pub fn milzma_multi_lzma2_decompress_batch(
        m: *mut milzma_multi,
        n: u32,
        ins: *const *const u8,
        in_lens: *const usize,
        outs: *mut milzma_output,
    ) -> c_int;
    pub fn milzma_multi_xz_decompress_batch(
        m: *mut milzma_multi,
        n: u32,
        ins: *const *const u8,
        in_lens: *const usize,
        outs: *mut milzma_output,
    ) -> c_int;
    // push-mode decoding: Stream (feature `stream`) for a batch of streams
    pub fn milzma_streams_open(ctx: *mut milzma_ctx, kind: u32, n: u32, options: *const milzma_options, out: *mut *mut milzma_streams) -> c_int;
    pub fn milzma_streams_write(
        s: *mut milzma_streams,
        k: u32,
        idx: *const u32,
        data: *const *const c_void,
        len: *const usize,
        status: *mut i32,
    ) -> c_int;
    pub fn milzma_streams_write_error(s: *const milzma_streams, stream: u32) -> *const c_char;
    pub fn milzma_streams_finish(s: *mut milzma_streams, outs: *mut milzma_output) -> c_int;
    pub fn milzma_streams_close(s: *mut milzma_streams);
    pub fn milzma_streams_last_error(s: *const milzma_streams) -> *const c_char;
    pub fn milzma_streams_write_taken(s: *const milzma_streams, stream: u32) -> u64;
    pub fn milzma_streams_output(
        s: *mut milzma_streams,
        stream: u32,
        offset: u64,
        dst: *mut c_void,
        cap: usize,
        sink_len: *mut u64,
        has_sink: *mut i32,
    ) -> c_int;
}
