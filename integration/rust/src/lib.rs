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
//! Semantics kept from the crate: on error the bytes the reference would already have written to `W` are
//! still written (ring flushes, LZMA2 dictionary resets); the reader is left after the last byte the
//! reference would have consumed (a known-size `.lzma` stops before trailing bytes, LZMA2 after its 0x00
//! status byte); the Display strings of `error::Error` are the reference's.
//!
//! How the generic `R: BufRead` meets a batch decoder (`run`): the bytes one `fill_buf` shows are decoded WITHOUT
//! consuming; the library reports `in_consumed`; exactly that many bytes are then `consume`d -- for a slice or a
//! `Cursor` this is the reference's behaviour to the byte.  A reader whose `fill_buf` cannot show the whole
//! input at once (a `BufReader` over a large file) is read to its end and decoded again; bytes past `in_consumed`
//! cannot be given back to such a reader (`io::BufRead` has no un-read), which is the one difference from the
//! streaming reference and only matters to callers that keep reading from the same reader after a known-size
//! stream.  `*_batch` functions take slices and have no such caveat.
//!
//! NOTE: written without a Rust toolchain (the build image has none); never compiled.

pub mod ffi;

use std::ffi::CStr;
use std::io;
use std::ptr;

/// Error handling: src/error.rs of the crate.
pub mod error {
    use std::fmt::Display;
    use std::{io, result};

    /// Library errors (src/error.rs:8-17).
    #[derive(Debug)]
    pub enum Error {
        /// I/O error.
        IoError(io::Error),
        /// Not enough bytes to complete header
        HeaderTooShort(io::Error),
        /// LZMA error.
        LzmaError(String),
        /// XZ error.
        XzError(String),
    }

    /// Library result alias.
    pub type Result<T> = result::Result<T, Error>;

    impl From<io::Error> for Error {
        fn from(e: io::Error) -> Error {
            Error::IoError(e)
        }
    }

    impl Display for Error {
        fn fmt(&self, fmt: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
            match self {
                Error::IoError(e) => write!(fmt, "io error: {}", e),
                Error::HeaderTooShort(e) => write!(fmt, "header too short: {}", e),
                Error::LzmaError(e) => write!(fmt, "lzma error: {}", e),
                Error::XzError(e) => write!(fmt, "xz error: {}", e),
            }
        }
    }

    impl std::error::Error for Error {
        fn source(&self) -> Option<&(dyn std::error::Error + 'static)> {
            match self {
                Error::IoError(e) | Error::HeaderTooShort(e) => Some(e),
                Error::LzmaError(_) | Error::XzError(_) => None,
            }
        }
    }
}

/// Decompression helpers: src/decode/options.rs of the crate.
pub mod decompress {
    /// Options to tweak decompression behavior (src/decode/options.rs:3-20).
    #[derive(Clone, Copy, Debug, PartialEq, Eq, Default)]
    pub struct Options {
        /// Whether the unpacked size is read from the header or provided.
        pub unpacked_size: UnpackedSize,
        /// Limit of the dictionary's dynamic size.
        pub memlimit: Option<usize>,
        /// Stream API only; no effect here (as in the crate's one-shot functions).
        pub allow_incomplete: bool,
    }

    /// Alternatives for defining the unpacked size (src/decode/options.rs:22-43).
    #[derive(Clone, Copy, Debug, PartialEq, Eq, Default)]
    pub enum UnpackedSize {
        /// 8 size bytes in the header; all ones = end-of-payload marker.
        #[default]
        ReadFromHeader,
        /// 8 size bytes in the header, read and ignored; the provided value is used.
        ReadHeaderButUseProvided(Option<u64>),
        /// No size bytes in the header; the provided value is used.
        UseProvided(Option<u64>),
    }
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
            let msg = unsafe { CStr::from_ptr(ffi::milzma_last_error(ptr::null())) };
            return Err(error::Error::IoError(io::Error::new(
                io::ErrorKind::Other,
                format!("milzma_create: {}", msg.to_string_lossy()),
            )));
        }
        Ok(Context { raw })
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { ffi::milzma_destroy(self.raw) }
    }
}

thread_local! {
    static DEFAULT_CTX: std::cell::RefCell<Option<Context>> = std::cell::RefCell::new(None);
}

fn with_default_ctx<T>(f: impl FnOnce(&Context) -> error::Result<T>) -> error::Result<T> {
    DEFAULT_CTX.with(|slot| {
        let mut slot = slot.borrow_mut();
        if slot.is_none() {
            let device = std::env::var("MILZMA_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
            *slot = Some(Context::new(device)?);
        }
        f(slot.as_ref().unwrap())
    })
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
        reserved: 0,
        memlimit: o.memlimit.unwrap_or(0) as u64,
    }
}

/// Runs `decode` on what `input` holds and hands the verdict to `output` / `input` the way the reference would.
///
/// First on the bytes one `fill_buf` shows, WITHOUT consuming: for a slice or a `Cursor` that is the whole input, and
/// afterwards exactly `in_consumed` bytes are consumed -- the reference's reader position.  Only if that attempt ends
/// in an error and the reader turns out to hold more (a `BufReader` over a file larger than its buffer), everything
/// is read and decoded again; such a reader is then left at its end (see the module note).
fn run<R: io::BufRead, W: io::Write>(
    input: &mut R,
    output: &mut W,
    decode: impl Fn(&[u8], &mut ffi::milzma_output),
) -> error::Result<()> {
    let first = input.fill_buf()?.to_vec();
    let mut out = empty_output();
    decode(&first, &mut out);
    if out.kind == ffi::MILZMA_OK {
        return deliver(&mut out, input, false, output);
    }
    // an error: truncation by the reader's buffer, or the stream's own?
    input.consume(first.len());
    let mut rest = Vec::new();
    input.read_to_end(&mut rest)?;
    if rest.is_empty() {
        return deliver(&mut out, input, true, output);
    }
    if !out.data.is_null() {
        unsafe { ffi::milzma_free(out.data as *mut _) };
    }
    let mut all = first;
    all.extend_from_slice(&rest);
    let mut out = empty_output();
    decode(&all, &mut out);
    deliver(&mut out, input, true, output)
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
    if !out.data.is_null() {
        unsafe { ffi::milzma_free(out.data as *mut _) };
        out.data = ptr::null_mut();
    }
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
        run(input, output, |bytes, out| unsafe {
            ffi::milzma_lzma_decompress(ctx.raw, bytes.as_ptr(), bytes.len(), &opt, out);
        })
    })
}

/// Decompress LZMA2 data with default options (src/lib.rs:83-88).
pub fn lzma2_decompress<R: io::BufRead, W: io::Write>(input: &mut R, output: &mut W) -> error::Result<()> {
    with_default_ctx(|ctx| {
        run(input, output, |bytes, out| unsafe {
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

fn collect(outs: Vec<ffi::milzma_output>) -> Vec<Decoded> {
    outs.into_iter()
        .map(|mut o| {
            let mut data = Vec::new();
            let mut no_reader: &[u8] = &[];
            let result = deliver(&mut o, &mut no_reader, true, &mut data);
            Decoded { data, in_consumed: o.in_consumed, result }
        })
        .collect()
}

/// Many complete `.lzma` files in one launch (every stream is one wavefront): what the GPU is for.
pub fn lzma_decompress_batch(ctx: &Context, files: &[&[u8]], options: &decompress::Options) -> Vec<Decoded> {
    let ptrs: Vec<*const u8> = files.iter().map(|f| f.as_ptr()).collect();
    let lens: Vec<usize> = files.iter().map(|f| f.len()).collect();
    let mut outs: Vec<ffi::milzma_output> = (0..files.len()).map(|_| empty_output()).collect();
    let opt = c_options(options);
    unsafe { ffi::milzma_lzma_decompress_batch(ctx.raw, files.len() as u32, ptrs.as_ptr(), lens.as_ptr(), &opt, outs.as_mut_ptr()) };
    collect(outs)
}

/// Many complete LZMA2 streams in one launch.
pub fn lzma2_decompress_batch(ctx: &Context, files: &[&[u8]]) -> Vec<Decoded> {
    let ptrs: Vec<*const u8> = files.iter().map(|f| f.as_ptr()).collect();
    let lens: Vec<usize> = files.iter().map(|f| f.len()).collect();
    let mut outs: Vec<ffi::milzma_output> = (0..files.len()).map(|_| empty_output()).collect();
    unsafe { ffi::milzma_lzma2_decompress_batch(ctx.raw, files.len() as u32, ptrs.as_ptr(), lens.as_ptr(), outs.as_mut_ptr()) };
    collect(outs)
}

/// Many complete `.xz` files in one launch (every block of every file is one wavefront).
pub fn xz_decompress_batch(ctx: &Context, files: &[&[u8]]) -> Vec<Decoded> {
    let ptrs: Vec<*const u8> = files.iter().map(|f| f.as_ptr()).collect();
    let lens: Vec<usize> = files.iter().map(|f| f.len()).collect();
    let mut outs: Vec<ffi::milzma_output> = (0..files.len()).map(|_| empty_output()).collect();
    unsafe { ffi::milzma_xz_decompress_batch(ctx.raw, files.len() as u32, ptrs.as_ptr(), lens.as_ptr(), outs.as_mut_ptr()) };
    collect(outs)
}
