//! The decode entry points of gendx/lzma-rs (src/lib.rs:44-105) with the same signatures, decoded on an MI355X
//! through libmilzma (include/milzma.h):
//!
//! ```ignore
//! pub fn lzma_decompress<R: io::BufRead, W: io::Write>(input: &mut R, output: &mut W) -> error::Result<()>
//! pub fn lzma_decompress_with_options<R, W>(input, output, options: &decompress::Options) -> error::Result<()>
//! pub fn lzma2_decompress<R, W>(input, output) -> error::Result<()>
//! pub fn xz_decompress<R, W>(input, output) -> error::Result<()>
//! ```
//!
//! `error` and `decompress` ARE the crate's own modules (re-exported from the `lzma-rs` dependency): the error type a
//! caller matches on and the options it builds are the same values whichever decoder runs.
//!
//! Semantics kept from the crate: on error the bytes the reference would already have written to `W` are
//! still written (ring flushes, LZMA2 dictionary resets); the reader is left after the last byte the
//! reference would have consumed (a known-size `.lzma` stops before trailing bytes, LZMA2 after its 0x00
//! status byte); the Display strings of `error::Error` are the reference's.
//!
//! How the generic `R: BufRead` meets a batch decoder (`run`): the bytes one `fill_buf` shows are decoded WITHOUT
//! consuming, and the library reports `in_consumed`.
//! * `in_consumed` < the bytes shown: the verdict (success or error) was reached inside the view and cannot depend on
//!   what follows it; exactly `in_consumed` bytes are consumed -- for a slice or a `Cursor` the reference's reader
//!   position to the byte, errors included.
//! * `in_consumed` == the bytes shown: the decoder ran to the end of the view -- which is the real end of a slice, but only
//!   a buffer boundary of a `BufReader` over a large file (an unknown-size `.lzma` stream is "finished" when the reader is
//!   at EOF with `code == 0`, src/decode/lzma.rs:446-455: a view cut there would be accepted short).  The view is
//!   consumed and the reader probed with another `fill_buf`: empty = that was the end, the verdict stands; otherwise
//!   `.lzma` and LZMA2 input is FED (round 5, `run_fed`): the views go to a push-mode stream as the reader shows them and the
//!   reader ends up exactly where the reference's would (right behind a stream of known size, at the failing byte of a
//!   damaged one).  Only `.xz` still reads everything and decodes again, leaving such a reader at ITS end (`io::BufRead`
//!   has no un-read; the container's Index sits at the end of the file).
//! `*_batch` functions take slices and have no such caveat.
//!
//! NOTE: written without a Rust toolchain (the build image has none); never compiled.  README.md: how to build and
//! test it where one exists.

pub mod ffi;

pub use lzma_rs::{decompress, error};

use std::ffi::CStr;
use std::io;
use std::ptr;
use std::rc::Rc;

fn infra(what: &str, msg: *const std::os::raw::c_char) -> error::Error {
    let msg = if msg.is_null() { "".into() } else { unsafe { CStr::from_ptr(msg) }.to_string_lossy() };
    error::Error::IoError(io::Error::new(io::ErrorKind::Other, format!("{}: {}", what, msg)))
}

/// A decoder bound to one GPU.  Creating it fails when no MI355X / HIP runtime is usable: there is no CPU
/// fallback (use the crate itself for that).
pub struct Context {
    raw: *mut ffi::milzma_ctx,
}

// The library serialises GPU use per context internally; a context may move between threads.
unsafe impl Send for Context {}

impl Context {
    /// `device`: HIP device ordinal.
    pub fn new(device: i32) -> error::Result<Context> {
        let mut raw = ptr::null_mut();
        let rc = unsafe { ffi::milzma_create(device, &mut raw) };
        if rc != ffi::MILZMA_OK {
            return Err(infra("milzma_create", unsafe { ffi::milzma_last_error(ptr::null()) }));
        }
        Ok(Context { raw })
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { ffi::milzma_destroy(self.raw) }
    }
}

/// The GPUs of one node behind one handle: one context and one host worker thread per device, a call's files
/// partitioned by size (`milzma_partition`), each device fed over its own PCIe link, all concurrently.
pub struct MultiContext {
    raw: *mut ffi::milzma_multi,
}

unsafe impl Send for MultiContext {}

impl MultiContext {
    /// `device_mask`: bit d = HIP ordinal d; 0 = every visible device.
    pub fn new(device_mask: u64) -> error::Result<MultiContext> {
        let mut raw = ptr::null_mut();
        let rc = unsafe { ffi::milzma_multi_create(device_mask, &mut raw) };
        if rc != ffi::MILZMA_OK {
            return Err(infra("milzma_multi_create", unsafe { ffi::milzma_multi_last_error(ptr::null()) }));
        }
        Ok(MultiContext { raw })
    }

    /// HIP ordinals of the devices behind this handle.
    pub fn devices(&self) -> Vec<i32> {
        let n = unsafe { ffi::milzma_multi_devices(self.raw, ptr::null_mut(), 0) };
        let mut v = vec![0i32; n as usize];
        unsafe { ffi::milzma_multi_devices(self.raw, v.as_mut_ptr(), n) };
        v
    }
}

impl Drop for MultiContext {
    fn drop(&mut self) {
        unsafe { ffi::milzma_multi_destroy(self.raw) }
    }
}

/// Unit-level calls on memory that already lives on the device (a pipeline that keeps files and output in HBM).
impl Context {
    /// `milzma_last_call_paths`: which way the most recent whole-file batch call on this context sent its files
    /// (`ffi::MILZMA_PATH_STREAMED | ..._TWO_PART_INPUT | ..._CLASSIC | ..._GROUPED`): a diagnostic, every path hands back the same bytes.
    pub fn last_call_paths(&self) -> u32 {
        unsafe { ffi::milzma_last_call_paths(self.raw) }
    }

    /// `milzma_decode_units_ex`: `units[i]` names a slice of `d_in` / `d_out` (device pointers of this context's device).
    /// With `ffi::MILZMA_DECODE_GROW` a unit that fills its slice before its stream ends comes back as
    /// `status == MILZMA_ST_OUT_FULL` with `err_a == MILZMA_PARKED` -- its decoder state stays in the context; give it a
    /// larger slice (`move_units` carries what it has written: that is its dictionary) and call again with
    /// `ffi::MILZMA_DECODE_RESUME`, the SAME units in the same order (the parked ones naming their new slices) and the previous
    /// call's results; only the parked units run.  Nothing is decoded twice.
    ///
    /// With `ffi::MILZMA_DECODE_FEED` -- the device-resident counterpart of `impl Write for Stream` (src/decode/stream.rs:223-283) -- a
    /// unit's `in_off / in_len` name a VIEW of its stream: the bytes that have arrived so far.  A unit that comes within 20 bytes of the
    /// view's end comes back as `status == MILZMA_ST_NEED_INPUT`, `err_a == MILZMA_PARKED`, `in_consumed` = bytes of this view it has
    /// used; resume it (`RESUME | FEED`) with a view that starts at that byte -- the unused tail in front of what has arrived since --
    /// and set `ffi::MILZMA_KIND_LAST_VIEW` in `kind` once its view ends where its stream ends (`Stream::finish`).
    ///
    /// # Safety
    /// `d_in` / `d_out` must be device allocations covering every slice the descriptors name, and the caller's own
    /// work on them must have completed (the library uses streams of its own unless `hip_stream` is given).
    pub unsafe fn decode_units_ex(
        &self,
        units: &[ffi::milzma_unit],
        d_in: *const std::os::raw::c_void,
        d_out: *mut std::os::raw::c_void,
        hip_stream: *mut std::os::raw::c_void,
        flags: u32,
        previous: Option<&[ffi::milzma_result]>,
    ) -> error::Result<Vec<ffi::milzma_result>> {
        // (`MILZMA_DECODE_RESUME` reads the previous call's results -- which units are parked -- out of the array it then writes)
        let mut results: Vec<ffi::milzma_result> = match previous {
            Some(p) if p.len() == units.len() => p.to_vec(),
            Some(_) => return Err(error::Error::IoError(std::io::Error::new(std::io::ErrorKind::InvalidInput, "previous results of another batch size"))),
            None => (0..units.len()).map(|_| std::mem::zeroed()).collect(),
        };
        let rc = ffi::milzma_decode_units_ex(self.raw, units.as_ptr(), units.len() as u32, d_in, d_out, results.as_mut_ptr(), hip_stream, flags);
        if rc != ffi::MILZMA_OK {
            return Err(infra("milzma_decode_units_ex", ffi::milzma_last_error(self.raw)));
        }
        Ok(results)
    }

    /// `milzma_move_units`: `len[i]` bytes from `d_src + src_off[i]` to `d_dst + dst_off[i]`, on the device.
    ///
    /// # Safety
    /// As for `decode_units_ex`; source and destination ranges must not overlap.
    pub unsafe fn move_units(
        &self,
        d_src: *const std::os::raw::c_void,
        src_off: &[u64],
        d_dst: *mut std::os::raw::c_void,
        dst_off: &[u64],
        len: &[u64],
        hip_stream: *mut std::os::raw::c_void,
    ) -> error::Result<()> {
        assert!(src_off.len() == len.len() && dst_off.len() == len.len());
        let rc = ffi::milzma_move_units(self.raw, len.len() as u32, d_src, src_off.as_ptr(), d_dst, dst_off.as_ptr(), len.as_ptr(), hip_stream);
        if rc != ffi::MILZMA_OK {
            return Err(infra("milzma_move_units", ffi::milzma_last_error(self.raw)));
        }
        Ok(())
    }
}

impl MultiContext {
    /// `milzma_multi_decode_units_rooted`: everything resident on device `root` (an index into `devices()`); the other
    /// devices receive their shares device to device and send their output back the same way.
    ///
    /// # Safety
    /// As for `Context::decode_units_ex`, with `d_in` / `d_out` on the root device.
    pub unsafe fn decode_units_rooted(
        &self,
        root: u32,
        units: &[ffi::milzma_unit],
        d_in: *const std::os::raw::c_void,
        d_out: *mut std::os::raw::c_void,
    ) -> error::Result<Vec<ffi::milzma_result>> {
        let mut results: Vec<ffi::milzma_result> = (0..units.len()).map(|_| std::mem::zeroed()).collect();
        let rc = ffi::milzma_multi_decode_units_rooted(self.raw, root, units.as_ptr(), units.len() as u32, d_in, d_out, results.as_mut_ptr());
        if rc != ffi::MILZMA_OK {
            return Err(infra("milzma_multi_decode_units_rooted", ffi::milzma_multi_last_error(self.raw)));
        }
        Ok(results)
    }

    /// Slowest device's (copy in, decode, copy back) of the last rooted call, milliseconds.
    pub fn last_transfer_ms(&self) -> (f32, f32, f32) {
        let (mut s, mut d, mut g) = (0f32, 0f32, 0f32);
        unsafe { ffi::milzma_multi_last_transfer_ms(self.raw, &mut s, &mut d, &mut g) };
        (s, d, g)
    }
}

/// Returns pooled result buffers to the system until at most `keep_bytes` stay pooled; the bytes still pooled.
/// (`Decoded::data` buffers are copied out of the pool by this shim, so a long-lived process may call this after a burst.)
pub fn pool_trim(keep_bytes: usize) -> usize {
    unsafe { ffi::milzma_pool_trim(keep_bytes) }
}

thread_local! {
    static DEFAULT_CTX: std::cell::RefCell<Option<Rc<Context>>> = std::cell::RefCell::new(None);
}

/// The calling thread's context on device `MILZMA_DEVICE` (default 0).  The `RefCell` is only borrowed to fetch the
/// handle: `f` -- which runs the caller's `Write` -- may itself decode on this thread.
fn with_default_ctx<T>(f: impl FnOnce(&Context) -> error::Result<T>) -> error::Result<T> {
    let ctx = DEFAULT_CTX.with(|slot| -> error::Result<Rc<Context>> {
        let mut slot = slot.borrow_mut();
        if slot.is_none() {
            let device = std::env::var("MILZMA_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
            *slot = Some(Rc::new(Context::new(device)?));
        }
        Ok(slot.as_ref().unwrap().clone())
    })?;
    f(&ctx)
}

fn c_options(o: &decompress::Options) -> ffi::milzma_options {
    use decompress::UnpackedSize::*;
    let (mode, provided) = match o.unpacked_size {
        ReadFromHeader => (ffi::MILZMA_READ_FROM_HEADER, None),
        ReadHeaderButUseProvided(x) => (ffi::MILZMA_READ_HEADER_BUT_USE_PROVIDED, x),
        UseProvided(x) => (ffi::MILZMA_USE_PROVIDED, x),
    };
    ffi::milzma_options {
        unpacked_size_mode: mode,
        provided_is_some: provided.is_some() as i32,
        provided: provided.unwrap_or(0),
        memlimit_is_some: o.memlimit.is_some() as i32,
        allow_incomplete: o.allow_incomplete as i32,
        memlimit: o.memlimit.unwrap_or(0) as u64,
    }
}

/// Runs `decode` on what `input` holds and hands the verdict to `output` / `input` the way the reference would
/// (the two cases of the module note).
fn run<R: io::BufRead, W: io::Write>(
    input: &mut R,
    output: &mut W,
    decode: impl Fn(&[u8], &mut ffi::milzma_output),
) -> error::Result<()> {
    let first = input.fill_buf()?.to_vec();
    let mut out = empty_output();
    decode(&first, &mut out);
    if out.in_consumed < first.len() {
        // decided inside the view: nothing after it can change the verdict
        return deliver(&mut out, input, false, output);
    }
    // the decoder reached the end of the view: the reader's end, or only its buffer's?
    input.consume(first.len());
    if input.fill_buf()?.is_empty() {
        return deliver(&mut out, input, true, output);
    }
    release(&mut out);
    let mut all = first;
    input.read_to_end(&mut all)?;
    let mut out = empty_output();
    decode(&all, &mut out);
    deliver(&mut out, input, true, output)
}

/// `run` for the two entry points whose decoder can be FED (`.lzma`, LZMA2; round 5): if the first view is not the whole input, the
/// views are written to a push-mode stream in READER mode as `fill_buf` shows them (`MILZMA_STREAMS_AS_READER`, include/milzma.h) --
/// nothing is read twice, and the reader is left exactly where the reference's would stand: right behind a stream of known size or an
/// LZMA2 end byte, at the failing byte of a damaged one.  What it costs: the FIRST view is decoded twice (by the one-shot call that
/// finds out whether the reader shows everything at once, then as the stream's first piece), and every later view is one upload, one
/// launch and one wait of its own on a batch of ONE stream -- with `BufReader`'s default 8 KiB that is 130 000 serial launches per GiB.
/// Give the reader a large buffer (`BufReader::with_capacity(64 << 20, file)`: a view per 64 MiB), or, for many files, read them and use
/// the `*_batch` calls: a single stream never fills the chip either way.  (`tests/test_gpu_reader.py` runs this loop --
/// `lzma_rs_amd.decompress_reader` -- for buffers of 1 byte to everything against the oracle.)
fn run_fed<R: io::BufRead, W: io::Write>(
    ctx: &Context,
    kind: u32,
    options: Option<&decompress::Options>,
    input: &mut R,
    output: &mut W,
    one_shot: impl Fn(&[u8], &mut ffi::milzma_output),
) -> error::Result<()> {
    // the common case first: a reader that shows everything at once is decoded by the one-shot call, as before
    let first = input.fill_buf()?.to_vec();
    let mut out = empty_output();
    one_shot(&first, &mut out);
    if out.in_consumed < first.len() {
        return deliver(&mut out, input, false, output);
    }
    // the decoder ran to the end of the view: the reader's end, or only its buffer's?
    input.consume(first.len());
    if input.fill_buf()?.is_empty() {
        return deliver(&mut out, input, true, output);
    }
    release(&mut out);
    // only its buffer's: from here on the views are fed, the first one (consumed already, kept in `first`) ahead of what the reader shows next
    let opts: Vec<decompress::Options> = options.map(|o| vec![*o]).unwrap_or_default();
    let mut s = Streams::open(ctx, kind | ffi::MILZMA_STREAMS_AS_READER, 1, &opts)?;
    let mut fed = first.len(); // bytes of the views consumed so far
    // (the one-shot call got through all of `first`: neither an end nor an error lies strictly inside it)
    let mut more = s.write(&[(0, &first[..])])?.pop().expect("one piece").is_ok();
    while more {
        let view = input.fill_buf()?;
        if view.is_empty() {
            break;
        }
        let n = view.len();
        let wrote = s.write(&[(0, view)])?.pop().expect("one piece");
        if wrote.is_err() {
            more = false; // the stream ended or failed inside (or right in front of) this view: finish says where; nothing of it is consumed yet
        } else {
            input.consume(n);
            fed += n;
        }
    }
    let d = s.finish().pop().expect("one stream");
    input.consume(d.in_consumed.saturating_sub(fed));
    let wrote = output.write_all(&d.data).and_then(|_| output.flush());
    wrote?;
    d.result
}

fn release(out: &mut ffi::milzma_output) {
    if !out.data.is_null() {
        unsafe { ffi::milzma_free(out.data as *mut _) };
        out.data = ptr::null_mut();
    }
}

/// Hands the library's verdict to the caller's writer / reader the way the reference would have.
fn deliver<R: io::BufRead, W: io::Write>(
    out: &mut ffi::milzma_output,
    input: &mut R,
    already_consumed: bool,
    output: &mut W,
) -> error::Result<()> {
    let data = if out.len == 0 { &[][..] } else { unsafe { std::slice::from_raw_parts(out.data, out.len) } };
    let wrote = output.write_all(data).and_then(|_| output.flush());
    release(out);
    if !already_consumed {
        input.consume(out.in_consumed);
    }
    wrote?; // a failing sink is Error::IoError, as for the reference's write_all / flush
    let msg = unsafe { CStr::from_ptr(out.msg.as_ptr()) }.to_string_lossy().into_owned();
    // msg is the full Display string ("lzma error: ..."); the variants carry the part after the prefix
    let tail = |p: &str| msg.strip_prefix(p).unwrap_or(&msg).to_string();
    match out.kind {
        ffi::MILZMA_OK => Ok(()),
        ffi::MILZMA_IO_ERROR => Err(error::Error::IoError(io::Error::new(io::ErrorKind::UnexpectedEof, tail("io error: ")))),
        ffi::MILZMA_HEADER_TOO_SHORT => {
            Err(error::Error::HeaderTooShort(io::Error::new(io::ErrorKind::UnexpectedEof, tail("header too short: "))))
        }
        ffi::MILZMA_LZMA_ERROR => Err(error::Error::LzmaError(tail("lzma error: "))),
        ffi::MILZMA_XZ_ERROR => Err(error::Error::XzError(tail("xz error: "))),
        _ => Err(error::Error::IoError(io::Error::new(io::ErrorKind::Other, msg))),
    }
}

fn empty_output() -> ffi::milzma_output {
    ffi::milzma_output { data: ptr::null_mut(), len: 0, in_consumed: 0, kind: 0, msg: [0; 388] }
}

/// Decompress LZMA data with default [`Options`](decompress/struct.Options.html) (src/lib.rs:44-49).
pub fn lzma_decompress<R: io::BufRead, W: io::Write>(input: &mut R, output: &mut W) -> error::Result<()> {
    lzma_decompress_with_options(input, output, &decompress::Options::default())
}

/// Decompress LZMA data with the provided options (src/lib.rs:52-60).
pub fn lzma_decompress_with_options<R: io::BufRead, W: io::Write>(
    input: &mut R,
    output: &mut W,
    options: &decompress::Options,
) -> error::Result<()> {
    let opt = c_options(options);
    with_default_ctx(|ctx| {
        // everything in the first view (a slice, a Cursor, a small file): the one-shot call; else piece by piece (`run_fed`)
        run_fed(ctx, ffi::MILZMA_KIND_RAW_LZMA as u32, Some(options), input, output, |bytes, out| unsafe {
            ffi::milzma_lzma_decompress(ctx.raw, bytes.as_ptr(), bytes.len(), &opt, out);
        })
    })
}

/// Decompress LZMA2 data with default options (src/lib.rs:83-88).
pub fn lzma2_decompress<R: io::BufRead, W: io::Write>(input: &mut R, output: &mut W) -> error::Result<()> {
    with_default_ctx(|ctx| {
        run_fed(ctx, ffi::MILZMA_KIND_LZMA2 as u32, None, input, output, |bytes, out| unsafe {
            ffi::milzma_lzma2_decompress(ctx.raw, bytes.as_ptr(), bytes.len(), out);
        })
    })
}

/// Decompress XZ data with default options (src/lib.rs:100-105).
pub fn xz_decompress<R: io::BufRead, W: io::Write>(input: &mut R, output: &mut W) -> error::Result<()> {
    with_default_ctx(|ctx| {
        run(input, output, |bytes, out| unsafe {
            ffi::milzma_xz_decompress(ctx.raw, bytes.as_ptr(), bytes.len(), out);
        })
    })
}

/// What one file of a batch produced: the bytes for its writer (also on error), how far its reader would
/// stand, and the verdict.
pub struct Decoded {
    pub data: Vec<u8>,
    pub in_consumed: usize,
    pub result: error::Result<()>,
}

/// `rc`: the batch call's own return code.  MILZMA_INFRA_ERROR there means the library could not run the batch (no
/// memory, a HIP failure): no file of it may pass as decoded, whatever its slot happens to hold.
fn collect(rc: i32, why: impl Fn() -> error::Error, outs: Vec<ffi::milzma_output>) -> Vec<Decoded> {
    outs.into_iter()
        .map(|mut o| {
            let mut data = Vec::new();
            let mut no_reader: &[u8] = &[];
            let mut result = deliver(&mut o, &mut no_reader, true, &mut data);
            if rc != ffi::MILZMA_OK && result.is_ok() {
                data.clear();
                result = Err(why());
            }
            Decoded { data, in_consumed: o.in_consumed, result }
        })
        .collect()
}

fn views(files: &[&[u8]]) -> (Vec<*const u8>, Vec<usize>, Vec<ffi::milzma_output>) {
    (
        files.iter().map(|f| f.as_ptr()).collect(),
        files.iter().map(|f| f.len()).collect(),
        (0..files.len()).map(|_| empty_output()).collect(),
    )
}

/// Many complete `.lzma` files in one launch (every stream is one wavefront): what the GPU is for.
pub fn lzma_decompress_batch(ctx: &Context, files: &[&[u8]], options: &decompress::Options) -> Vec<Decoded> {
    let (ptrs, lens, mut outs) = views(files);
    let opt = c_options(options);
    let rc = unsafe { ffi::milzma_lzma_decompress_batch(ctx.raw, files.len() as u32, ptrs.as_ptr(), lens.as_ptr(), &opt, outs.as_mut_ptr()) };
    collect(rc, || infra("milzma_lzma_decompress_batch", unsafe { ffi::milzma_last_error(ctx.raw) }), outs)
}

/// Many complete LZMA2 streams in one launch.
pub fn lzma2_decompress_batch(ctx: &Context, files: &[&[u8]]) -> Vec<Decoded> {
    let (ptrs, lens, mut outs) = views(files);
    let rc = unsafe { ffi::milzma_lzma2_decompress_batch(ctx.raw, files.len() as u32, ptrs.as_ptr(), lens.as_ptr(), outs.as_mut_ptr()) };
    collect(rc, || infra("milzma_lzma2_decompress_batch", unsafe { ffi::milzma_last_error(ctx.raw) }), outs)
}

/// Many complete `.xz` files in one launch (every block of every file is one wavefront).
pub fn xz_decompress_batch(ctx: &Context, files: &[&[u8]]) -> Vec<Decoded> {
    let (ptrs, lens, mut outs) = views(files);
    let rc = unsafe { ffi::milzma_xz_decompress_batch(ctx.raw, files.len() as u32, ptrs.as_ptr(), lens.as_ptr(), outs.as_mut_ptr()) };
    collect(rc, || infra("milzma_xz_decompress_batch", unsafe { ffi::milzma_last_error(ctx.raw) }), outs)
}

/// The same three over every GPU of a [`MultiContext`]: files partitioned by size, one launch per device, concurrently.
pub fn lzma_decompress_batch_multi(m: &MultiContext, files: &[&[u8]], options: &decompress::Options) -> Vec<Decoded> {
    let (ptrs, lens, mut outs) = views(files);
    let opt = c_options(options);
    let rc = unsafe { ffi::milzma_multi_lzma_decompress_batch(m.raw, files.len() as u32, ptrs.as_ptr(), lens.as_ptr(), &opt, outs.as_mut_ptr()) };
    collect(rc, || infra("milzma_multi_lzma_decompress_batch", unsafe { ffi::milzma_multi_last_error(m.raw) }), outs)
}

pub fn lzma2_decompress_batch_multi(m: &MultiContext, files: &[&[u8]]) -> Vec<Decoded> {
    let (ptrs, lens, mut outs) = views(files);
    let rc = unsafe { ffi::milzma_multi_lzma2_decompress_batch(m.raw, files.len() as u32, ptrs.as_ptr(), lens.as_ptr(), outs.as_mut_ptr()) };
    collect(rc, || infra("milzma_multi_lzma2_decompress_batch", unsafe { ffi::milzma_multi_last_error(m.raw) }), outs)
}

pub fn xz_decompress_batch_multi(m: &MultiContext, files: &[&[u8]]) -> Vec<Decoded> {
    let (ptrs, lens, mut outs) = views(files);
    let rc = unsafe { ffi::milzma_multi_xz_decompress_batch(m.raw, files.len() as u32, ptrs.as_ptr(), lens.as_ptr(), outs.as_mut_ptr()) };
    collect(rc, || infra("milzma_multi_xz_decompress_batch", unsafe { ffi::milzma_multi_last_error(m.raw) }), outs)
}

// ---- whole-file batches in two halves: keeping two calls in flight --------------------------------------------------

/// A whole-file batch running on a host thread of the library (`milzma_*_decompress_batch_async`), started by
/// [`Context::lzma_batch_begin`] / [`Context::lzma2_batch_begin`] / [`Context::xz_batch_begin`] and finished by
/// [`InFlight::wait`].  One per context at a time; with two contexts on a device the upload and the hand-over of one call run
/// under the decode kernel of the other (4096 x 1 MiB `.lzma` files per call: 16 GB/s against 12 for calls made one after the
/// other).  The files' bytes are borrowed until the batch is waited for (or dropped, which waits).
pub struct InFlight<'c, 'f> {
    ctx: &'c Context,
    outs: Vec<ffi::milzma_output>, // the library's thread fills these: the buffer must not move or go away before the wait
    what: &'static str,
    waited: bool,
    _files: std::marker::PhantomData<&'f [u8]>,
}

impl<'c, 'f> InFlight<'c, 'f> {
    /// Blocks until the batch is done; one [`Decoded`] per file, in order.
    pub fn wait(mut self) -> Vec<Decoded> {
        let rc = unsafe { ffi::milzma_batch_wait(self.ctx.raw) };
        self.waited = true;
        let outs = std::mem::take(&mut self.outs);
        let (ctx, what) = (self.ctx, self.what);
        collect(rc, || infra(what, unsafe { ffi::milzma_last_error(ctx.raw) }), outs)
    }
}

impl Drop for InFlight<'_, '_> {
    fn drop(&mut self) {
        if !self.waited {
            unsafe { ffi::milzma_batch_wait(self.ctx.raw) };
            for o in self.outs.iter_mut() {
                release(o);
            }
        }
    }
}

impl Context {
    fn began<'c, 'f>(&'c self, rc: i32, what: &'static str, outs: Vec<ffi::milzma_output>) -> error::Result<InFlight<'c, 'f>> {
        if rc != ffi::MILZMA_OK {
            // nothing was started (a batch already in flight on this context, bad arguments)
            return Err(infra(what, unsafe { ffi::milzma_last_error(self.raw) }));
        }
        Ok(InFlight { ctx: self, outs, what, waited: false, _files: std::marker::PhantomData })
    }

    /// [`lzma_decompress_batch`] in two halves.  (The library copies the pointer / length arrays and the options before it returns.)
    pub fn lzma_batch_begin<'c, 'f>(&'c self, files: &[&'f [u8]], options: &decompress::Options) -> error::Result<InFlight<'c, 'f>> {
        let (ptrs, lens, mut outs) = views(files);
        let opt = c_options(options);
        let rc = unsafe {
            ffi::milzma_lzma_decompress_batch_async(self.raw, files.len() as u32, ptrs.as_ptr(), lens.as_ptr(), &opt, outs.as_mut_ptr())
        };
        self.began(rc, "milzma_lzma_decompress_batch_async", outs)
    }

    /// [`lzma2_decompress_batch`] in two halves.
    pub fn lzma2_batch_begin<'c, 'f>(&'c self, files: &[&'f [u8]]) -> error::Result<InFlight<'c, 'f>> {
        let (ptrs, lens, mut outs) = views(files);
        let rc = unsafe { ffi::milzma_lzma2_decompress_batch_async(self.raw, files.len() as u32, ptrs.as_ptr(), lens.as_ptr(), outs.as_mut_ptr()) };
        self.began(rc, "milzma_lzma2_decompress_batch_async", outs)
    }

    /// [`xz_decompress_batch`] in two halves.
    pub fn xz_batch_begin<'c, 'f>(&'c self, files: &[&'f [u8]]) -> error::Result<InFlight<'c, 'f>> {
        let (ptrs, lens, mut outs) = views(files);
        let rc = unsafe { ffi::milzma_xz_decompress_batch_async(self.raw, files.len() as u32, ptrs.as_ptr(), lens.as_ptr(), outs.as_mut_ptr()) };
        self.began(rc, "milzma_xz_decompress_batch_async", outs)
    }
}

/// A sequence of `.xz` batches with two calls in flight, alternating between two contexts of one device; `sink(k, decoded)` gets
/// the batches back in submission order.  The pattern for `.lzma` / LZMA2 batches is the same with the other `*_batch_begin`.
pub fn xz_decompress_batches_pipelined<'f>(
    a: &Context,
    b: &Context,
    batches: &[Vec<&'f [u8]>],
    mut sink: impl FnMut(usize, Vec<Decoded>),
) -> error::Result<()> {
    let ctxs = [a, b];
    let mut pending: [Option<(usize, InFlight<'_, 'f>)>; 2] = [None, None];
    for (k, files) in batches.iter().enumerate() {
        let slot = k % 2;
        if let Some((idx, call)) = pending[slot].take() {
            sink(idx, call.wait());
        }
        pending[slot] = Some((k, ctxs[slot].xz_batch_begin(files)?)); // (on an error the calls still in flight are waited for by Drop)
    }
    let older = batches.len() % 2;
    for slot in [older, 1 - older] {
        if let Some((idx, call)) = pending[slot].take() {
            sink(idx, call.wait());
        }
    }
    Ok(())
}

/// `lzma_rs::decompress::Stream` (feature `stream`, src/decode/stream.rs) for a BATCH of streams on one GPU: the compressed `.lzma`
/// bytes of each stream are written piece by piece, `finish` hands every stream's verdict and output over.  One `write` call gives any
/// number of the streams another turn in ONE launch (fed input, `MILZMA_DECODE_FEED`: streams start when their header is complete,
/// resume where they parked, or sit the call out).  Verdicts, texts, bytes and the call that fails are the crate's, behind an end marker too
/// (include/milzma.h).
pub struct Streams {
    raw: *mut ffi::milzma_streams,
    n: usize,
}

impl Streams {
    /// n x `Stream::new_with_options` (stream.rs:88-101) -- `.lzma` files, header first; `options`: one per stream, or empty for
    /// `Options::default()`.
    pub fn new(ctx: &Context, n: usize, options: &[decompress::Options]) -> error::Result<Streams> {
        Self::open(ctx, ffi::MILZMA_KIND_RAW_LZMA as u32, n, options)
    }

    /// n raw LZMA2 streams fed the same way (the crate has no such type; include/milzma.h).
    pub fn new_lzma2(ctx: &Context, n: usize) -> error::Result<Streams> {
        Self::open(ctx, ffi::MILZMA_KIND_LZMA2 as u32, n, &[])
    }

    /// `kind`: `MILZMA_KIND_RAW_LZMA` / `MILZMA_KIND_LZMA2`, optionally `| MILZMA_STREAMS_AS_READER` (finish = the one-shot verdict).
    pub fn open(ctx: &Context, kind: u32, n: usize, options: &[decompress::Options]) -> error::Result<Streams> {
        assert!(options.is_empty() || options.len() == n);
        let copts: Vec<ffi::milzma_options> = options.iter().map(c_options).collect();
        let mut raw = ptr::null_mut();
        let rc = unsafe { ffi::milzma_streams_open(ctx.raw, kind, n as u32, if copts.is_empty() { ptr::null() } else { copts.as_ptr() }, &mut raw) };
        if rc != ffi::MILZMA_OK {
            return Err(infra("milzma_streams_open", unsafe { ffi::milzma_last_error(ctx.raw) }));
        }
        Ok(Streams { raw, n })
    }

    /// `io::Write::write_all` for each `(stream, bytes)` of `pieces` (a stream at most once per call): what comes back is one
    /// `io::Result` per piece, `Err` with the text the crate's `write_all` error displays (stream.rs:291-299, :343-347).
    pub fn write(&mut self, pieces: &[(usize, &[u8])]) -> error::Result<Vec<io::Result<()>>> {
        let idx: Vec<u32> = pieces.iter().map(|p| p.0 as u32).collect();
        let data: Vec<*const std::os::raw::c_void> = pieces.iter().map(|p| p.1.as_ptr() as *const _).collect();
        let len: Vec<usize> = pieces.iter().map(|p| p.1.len()).collect();
        let mut status = vec![0i32; pieces.len()];
        let rc = unsafe { ffi::milzma_streams_write(self.raw, pieces.len() as u32, idx.as_ptr(), data.as_ptr(), len.as_ptr(), status.as_mut_ptr()) };
        if rc != ffi::MILZMA_OK {
            return Err(infra("milzma_streams_write", unsafe { ffi::milzma_streams_last_error(self.raw) }));
        }
        Ok(status
            .iter()
            .zip(&idx)
            .map(|(&st, &i)| {
                if st == ffi::MILZMA_OK {
                    return Ok(());
                }
                let text = unsafe { CStr::from_ptr(ffi::milzma_streams_write_error(self.raw, i)) }.to_string_lossy().into_owned();
                let kind = if text == "failed to write whole buffer" { io::ErrorKind::WriteZero } else { io::ErrorKind::Other };
                Err(io::Error::new(kind, text))
            })
            .collect())
    }

    /// `Stream::finish` (stream.rs:119-150) for every stream; consumes the batch.
    pub fn finish(self) -> Vec<Decoded> {
        let mut outs: Vec<ffi::milzma_output> = (0..self.n).map(|_| empty_output()).collect();
        let rc = unsafe { ffi::milzma_streams_finish(self.raw, outs.as_mut_ptr()) };
        let raw = self.raw;
        collect(rc, || infra("milzma_streams_finish", unsafe { ffi::milzma_streams_last_error(raw) }), outs)
        // (Drop closes the batch)
    }
}

impl Drop for Streams {
    fn drop(&mut self) {
        unsafe { ffi::milzma_streams_close(self.raw) }
    }
}

/// ONE push-mode decoder with the crate's own shape -- `Stream::new(output) -> Self`, `io::Write`, `get_output`, `finish() -> Result<W>`
/// (src/decode/stream.rs:80-151) -- for code that is written against `lzma_rs::decompress::Stream<W>`.  (A batch of one stream leaves the
/// GPU idle: `Streams` is the form to use when many streams arrive at once.)
///
/// The sink is kept as the crate keeps it: every `write` hands `W` what the ring has flushed since (whole multiples of the dictionary
/// size, lzbuffer.rs:264-267), `finish` the rest; after a failed write the crate has dropped its state and the sink with it
/// (stream.rs:230) -- `get_output` answers `None` from then on here too.
pub struct Stream<W: io::Write> {
    /// `None`: the batch behind this stream could not be opened (no GPU, no memory).  `new` cannot fail in the crate, so the first `write`
    /// / `finish` reports it (`open_error`).
    inner: Option<Streams>,
    open_error: Option<String>,
    output: Option<W>,
    /// bytes of the sink that `output` has got already
    delivered: u64,
}

impl<W: io::Write> Stream<W> {
    /// `Stream::new` (stream.rs:86-88)
    pub fn new(output: W) -> Self {
        Self::new_with_options(&decompress::Options::default(), output)
    }

    /// `Stream::new_with_options` (stream.rs:93-99)
    pub fn new_with_options(options: &decompress::Options, output: W) -> Self {
        match with_default_ctx(|ctx| Streams::new(ctx, 1, std::slice::from_ref(options))) {
            Ok(inner) => Stream { inner: Some(inner), open_error: None, output: Some(output), delivered: 0 },
            Err(e) => Stream { inner: None, open_error: Some(e.to_string()), output: Some(output), delivered: 0 },
        }
    }

    /// `Stream::get_output` (stream.rs:102-107): the sink, holding everything the ring has flushed so far; `None` after a failed write.
    pub fn get_output(&self) -> Option<&W> {
        self.output.as_ref()
    }

    /// `Stream::get_output_mut` (stream.rs:110-115)
    pub fn get_output_mut(&mut self) -> Option<&mut W> {
        self.output.as_mut()
    }

    /// What the ring has flushed since the last call goes to `W` (`milzma_streams_output`); a stream whose state is gone drops `W`.
    fn sync_sink(&mut self) -> io::Result<()> {
        let inner = match self.inner.as_ref() {
            Some(s) => s,
            None => return Ok(()),
        };
        let (mut len, mut has) = (0u64, 0i32);
        let rc = unsafe { ffi::milzma_streams_output(inner.raw, 0, 0, ptr::null_mut(), 0, &mut len, &mut has) };
        if rc != ffi::MILZMA_OK {
            return Err(io::Error::new(io::ErrorKind::Other, infra("milzma_streams_output", unsafe { ffi::milzma_streams_last_error(inner.raw) }).to_string()));
        }
        if has == 0 {
            self.output = None;
            return Ok(());
        }
        if len > self.delivered {
            let mut buf = vec![0u8; (len - self.delivered) as usize];
            let rc = unsafe { ffi::milzma_streams_output(inner.raw, 0, self.delivered, buf.as_mut_ptr() as *mut _, buf.len(), &mut len, &mut has) };
            if rc != ffi::MILZMA_OK {
                return Err(io::Error::new(io::ErrorKind::Other, infra("milzma_streams_output", unsafe { ffi::milzma_streams_last_error(inner.raw) }).to_string()));
            }
            if let Some(w) = self.output.as_mut() {
                w.write_all(&buf)?;
            }
            self.delivered += buf.len() as u64;
        }
        Ok(())
    }

    /// Stream::finish (stream.rs:119-150): the sink with everything decoded, or the stream's error.
    pub fn finish(mut self) -> error::Result<W> {
        let inner = match self.inner.take() {
            Some(s) => s,
            None => return Err(error::Error::IoError(io::Error::new(io::ErrorKind::Other, self.open_error.take().unwrap_or_default()))),
        };
        let mut done = inner.finish();
        let d = done.pop().expect("one stream");
        d.result?;
        // (Ok: the state was there, and so is the sink)
        let mut output = self.output.take().expect("a stream that finishes well has its sink");
        output.write_all(&d.data[(self.delivered as usize).min(d.data.len())..])?;
        output.flush()?;
        Ok(output)
    }
}

impl<W: io::Write> io::Write for Stream<W> {
    /// `Stream::write` (stream.rs:227-325): `Ok(n)`, the bytes the stream took -- all of them, or, where the stream ends inside `data`
    /// (its declared size is reached), the ones in front of its end; `Ok(0)` from a stream that has ended or whose state is gone, which
    /// `write_all` turns into `ErrorKind::WriteZero` as it does for the crate.
    fn write(&mut self, data: &[u8]) -> io::Result<usize> {
        if data.is_empty() {
            return Ok(0);
        }
        let inner = match self.inner.as_mut() {
            Some(s) => s,
            None => return Err(io::Error::new(io::ErrorKind::Other, self.open_error.clone().unwrap_or_default())),
        };
        let mut r = inner
            .write(&[(0, data)])
            .map_err(|e| io::Error::new(io::ErrorKind::Other, e.to_string()))?;
        let taken = unsafe { ffi::milzma_streams_write_taken(inner.raw, 0) } as usize;
        let verdict = r.pop().expect("one piece");
        self.sync_sink()?;
        match verdict {
            Ok(()) => Ok(data.len()),
            Err(e) if e.kind() == io::ErrorKind::WriteZero => Ok(taken.min(data.len())),
            Err(e) => Err(e),
        }
    }

    /// `Stream::flush` (stream.rs:330-339): the sink's own flush (the ring is not flushed: that would corrupt the state)
    fn flush(&mut self) -> io::Result<()> {
        match self.output.as_mut() {
            Some(w) => w.flush(),
            None => Ok(()),
        }
    }
}

#[cfg(test)]
mod tests {
    //! Need an MI355X and libmilzma.so (README.md).  The streams are literal-only ones written by the crate's own encoder.
    use super::*;
    use std::io::{BufRead, BufReader, Cursor};

    fn known_size_stream(plain: &[u8]) -> Vec<u8> {
        let mut comp = Vec::new();
        let opts = lzma_rs::compress::Options { unpacked_size: lzma_rs::compress::UnpackedSize::WriteToHeader(Some(plain.len() as u64)) };
        lzma_rs::lzma_compress_with_options(&mut &plain[..], &mut comp, &opts).unwrap();
        comp
    }

    /// A known-size `.lzma` stream followed by other bytes in a `Cursor`: the reader must stand where the reference
    /// leaves it (the crate's own decoder is run beside ours for the position), the trailing bytes untouched.
    #[test]
    fn cursor_position_after_a_known_size_stream_with_trailing_bytes() {
        let plain = b"the reader stops where the stream stops".repeat(50);
        let mut file = known_size_stream(&plain);
        let stream_len = file.len();
        file.extend_from_slice(b"TRAILING-BYTES");
        let (mut ours, mut theirs) = (Cursor::new(&file[..]), Cursor::new(&file[..]));
        let (mut out_ours, mut out_theirs) = (Vec::new(), Vec::new());
        lzma_decompress(&mut ours, &mut out_ours).unwrap();
        lzma_rs::lzma_decompress(&mut theirs, &mut out_theirs).unwrap();
        assert_eq!(out_ours, plain);
        assert_eq!(out_ours, out_theirs);
        assert_eq!(ours.position(), theirs.position());
        assert!(ours.position() as usize <= stream_len);
        assert!(ours.fill_buf().unwrap().ends_with(b"TRAILING-BYTES"));
    }

    /// A `BufReader` whose buffer (64 bytes) is far smaller than the stream, unknown size (marker mode): the first view
    /// ends mid-stream; the verdict must come from the whole input.
    #[test]
    fn bufreader_over_a_file_larger_than_its_buffer() {
        let plain: Vec<u8> = (0..200_000u32).map(|i| (i * 7 % 251) as u8).collect();
        let mut comp = Vec::new();
        lzma_rs::lzma_compress(&mut &plain[..], &mut comp).unwrap();
        assert!(comp.len() > 4096);
        let mut rd = BufReader::with_capacity(64, Cursor::new(comp));
        let mut out = Vec::new();
        lzma_decompress(&mut rd, &mut out).unwrap();
        assert_eq!(out, plain);
        // and an error inside the first view leaves a slice reader exactly where the reference does
        let mut bad = known_size_stream(b"abcdefgh");
        bad[0] = 0xFF; // invalid properties byte
        let (mut ours, mut theirs) = (&bad[..], &bad[..]);
        let e1 = lzma_decompress(&mut ours, &mut Vec::new()).unwrap_err();
        let e2 = lzma_rs::lzma_decompress(&mut theirs, &mut Vec::new()).unwrap_err();
        assert_eq!(e1.to_string(), e2.to_string());
        assert_eq!(ours.len(), theirs.len());
    }

    /// Two calls in flight on two contexts: same bytes as the one-call form, batches back in order.
    #[test]
    fn two_batches_in_flight() {
        let plains: Vec<Vec<u8>> = (0..6u8).map(|k| vec![b'a' + k; 5000 + 100 * k as usize]).collect();
        let comps: Vec<Vec<u8>> = plains.iter().map(|p| { let mut c = Vec::new(); lzma_rs::xz_compress(&mut &p[..], &mut c).unwrap(); c }).collect();
        let batches: Vec<Vec<&[u8]>> = comps.chunks(2).map(|pair| pair.iter().map(|c| &c[..]).collect()).collect();
        let (a, b) = (Context::new(0).unwrap(), Context::new(0).unwrap());
        let mut seen = Vec::new();
        xz_decompress_batches_pipelined(&a, &b, &batches, |k, decoded| {
            for (j, d) in decoded.into_iter().enumerate() {
                d.result.unwrap();
                assert_eq!(d.data, plains[2 * k + j]);
            }
            seen.push(k);
        })
        .unwrap();
        assert_eq!(seen, vec![0, 1, 2]);
        // a second begin on a context with a batch in flight is refused, and the first still completes
        let first = a.xz_batch_begin(&batches[0]).unwrap();
        assert!(a.xz_batch_begin(&batches[1]).is_err());
        assert_eq!(first.wait().len(), 2);
    }
}
