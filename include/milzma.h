/*
 * milzma.h -- C ABI of the MI355X-native batched LZMA / LZMA2 / XZ decoder.
 *
 * Drop-in boundary for the decode hot path of gendx/lzma-rs (reference paths are
 * relative to the reference repository root):
 *
 *   src/decode/rangecoder.rs   RangeDecoder, BitTree, LenDecoder
 *   src/decode/lzma.rs:164-593 DecoderState (literal / match / rep state machine)
 *   src/decode/lzbuffer.rs     LzCircularBuffer / LzAccumBuffer (LZ77 window + output)
 *
 * The reference has no FFI layer; the seam this ABI replaces is
 * DecoderState::process(&mut LZB, &mut RangeDecoder) (src/decode/lzma.rs:255-261) as it is
 * called from LzmaDecoder::decompress (src/decode/lzma.rs:635-648) and
 * Lzma2Decoder::decompress (src/decode/lzma2.rs:52-82), plus the three public entry points
 * of src/lib.rs:44-105 for callers that want whole-file semantics.
 *
 * Plain pointers and sizes only; no torch / HIP types in any signature (a HIP stream is
 * passed as an opaque `void *`).  INTEGRATION.md shows the Rust `extern "C"` block a
 * maintainer of the crate would add to bind these.
 */
#ifndef MILZMA_H
#define MILZMA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MILZMA_ABI_VERSION 6

/* ---- error kinds: error::Error variants (src/error.rs:8-17) ---------------------------- */
enum {
  MILZMA_OK = 0,
  MILZMA_IO_ERROR = 1,         /* Error::IoError(io::Error)       "io error: ..."          */
  MILZMA_HEADER_TOO_SHORT = 2, /* Error::HeaderTooShort(io::Error) "header too short: ..."  */
  MILZMA_LZMA_ERROR = 3,       /* Error::LzmaError(String)        "lzma error: ..."        */
  MILZMA_XZ_ERROR = 4,         /* Error::XzError(String)          "xz error: ..."          */
  MILZMA_INFRA_ERROR = 5       /* not a reference error: HIP failure, bad argument, no GPU */
};

/* ---- decode units: one independent serial decode job = one wavefront -------------------- */

/* unit kinds */
#define MILZMA_KIND_RAW_LZMA 0u /* LzmaDecoder::decompress: input starts at the range coder's first byte */
#define MILZMA_KIND_LZMA2 1u    /* Lzma2Decoder::decompress: input starts at an LZMA2 status byte          */
#define MILZMA_KIND_LAST_VIEW 0x80u /* or-ed into `kind` in a MILZMA_DECODE_FEED call: this unit's view ends where its stream ends */
#define MILZMA_KIND_HOLD 0x40u  /* ... in a RESUME | FEED call: this parked unit stays parked (nothing new for it); its result is kept */
#define MILZMA_KIND_START 0x20u /* ... in a RESUME | FEED call: this unit is new -- it starts now, in the place i of the batch, beside the
                                   units that resume (continuous batching: the streams of a batch need not begin together) */

#define MILZMA_KIND_PARTIAL 0x10u /* ... in a MILZMA_DECODE_FEED call, RAW units: the crate's ProcessingMode::Partial at an END MARKER
                                     (src/decode/lzma.rs:493-495, :507-509): a marker that ends a view which is not the last (code == 0, no
                                     byte of the view behind it) does not end the unit -- the crate's loop merely leaves, so bytes that
                                     arrive later are decoded on from the marker's state (rep[0] = 0xFFFF_FFFF, the state after a match).
                                     The unit parks behind the marker (MILZMA_ST_NEED_INPUT, in_consumed = the whole view) and resumes like
                                     any fed unit; its LAST view ends it by the Finish-mode rules (lzma.rs:446-455, :513-521) */

#define MILZMA_SIZE_UNKNOWN UINT64_MAX /* unpacked_size: Option::None => end-of-stream marker mode */
#define MILZMA_NO_LIMIT UINT64_MAX     /* memlimit: Option::None                                    */

/* Largest input / output slice a single unit may span (positions are 32-bit on the device). */
#define MILZMA_MAX_UNIT_BYTES 0xFFFFFF00ull

typedef struct milzma_unit {
  uint64_t in_off;        /* offset of the unit's compressed bytes inside the input base        */
  uint64_t in_len;        /* bytes the unit's reader can see (its EOF)                           */
  uint64_t out_off;       /* offset of the unit's output slice inside the output base            */
  uint64_t out_cap;       /* bytes the unit may write                                            */
  uint64_t unpacked_size; /* RAW_LZMA: LzmaParams.unpacked_size (src/decode/lzma.rs:70-77)        */
  uint64_t memlimit;      /* RAW_LZMA: Options.memlimit (src/decode/options.rs:10-14)             */
  uint32_t dict_size;     /* RAW_LZMA: LzmaParams.dict_size                                       */
  uint8_t lc, lp, pb;     /* RAW_LZMA: LzmaProperties (src/decode/lzma.rs:43-58)                  */
  uint8_t kind;           /* MILZMA_KIND_*                                                        */
} milzma_unit;

/* Per-unit status: one code per error site on the hot path. `a`/`b`
 * are the integers the reference formats into its message. */
enum {
  MILZMA_ST_OK = 0,
  MILZMA_ST_RC_INIT = 1,          /* lzma.rs:643-644 / lzma2.rs:190-191: range decoder init hit EOF   */
  MILZMA_ST_INPUT_EOF = 2,        /* rangecoder.rs:64: normalize() hit EOF -> IoError(UnexpectedEof)    */
  MILZMA_ST_MATCH_DIST_DICT = 3,  /* lzbuffer.rs:241-245  "Match distance {a} is beyond dictionary size {b}" */
  MILZMA_ST_MATCH_DIST_OUT = 4,   /* lzbuffer.rs:246-251,98-105 "Match distance {a} is beyond output size {b}" */
  MILZMA_ST_LZ_DIST_DICT = 5,     /* lzbuffer.rs:274-278  "LZ distance {a} is beyond dictionary size {b}" */
  MILZMA_ST_LZ_DIST_OUT = 6,      /* lzbuffer.rs:279-285,125-133 "LZ distance {a} is beyond output size {b}" */
  MILZMA_ST_MEMLIMIT = 7,         /* lzbuffer.rs:210-217  "exceeded memory limit of {a}"                */
  MILZMA_ST_MARKER_TRAILING = 8,  /* lzma.rs:378-380      "Found end-of-stream marker but more bytes are available" */
  MILZMA_ST_SIZE_MISMATCH = 9,    /* lzma.rs:513-521      "Expected unpacked size of {a} but decompressed to {b}" */
  /* LZMA2 packet layer (src/decode/lzma2.rs), walked by the wavefront itself */
  MILZMA_ST_L2_STATUS_EOF = 16,     /* :60-62   "LZMA2 expected new status: {io}"                        */
  MILZMA_ST_L2_INVALID_STATUS = 17, /* :94-99   "LZMA2 invalid status {a}, must be 0, 1, 2 or >= 128"    */
  MILZMA_ST_L2_UNPACKED_EOF = 18,   /* :128-130,204-206 "LZMA2 expected unpacked size: {io}"             */
  MILZMA_ST_L2_PACKED_EOF = 19,     /* :133-135 "LZMA2 expected packed size: {io}"                       */
  MILZMA_ST_L2_PROPS_EOF = 20,      /* :153-155 "LZMA2 expected new properties: {io}"                    */
  MILZMA_ST_L2_PROPS_INVALID = 21,  /* :158-163 "LZMA2 invalid properties: {a} must be < 225"            */
  MILZMA_ST_L2_LCLP = 22,           /* :170-175 "LZMA2 invalid properties: lc + lp ({a} + {b}) must be <= 4" */
  MILZMA_ST_L2_STORED_EOF = 23,     /* :219-225 "LZMA2 expected {a} uncompressed bytes: {io}"            */
  /* not reference errors: conditions of this implementation, resolved by the host layer */
  MILZMA_ST_OUT_FULL = 32,   /* out_cap reached before the stream ended.  err_a == MILZMA_PARKED: the unit stopped in front
                                of the symbol that would not fit and its decoder state is parked in the context -- give it a
                                larger slice and call milzma_decode_units_ex(MILZMA_DECODE_RESUME); otherwise (err_a == 0):
                                grow the slice and decode the unit again from its first byte                              */
  MILZMA_ST_NEED_LCLP = 33,  /* unit needs a literal table for lc+lp = {a} larger than its launch class */
  MILZMA_ST_BAD_UNIT = 34,   /* descriptor rejected (slice > MILZMA_MAX_UNIT_BYTES, lc>8, lp>4, pb>4)   */
  MILZMA_ST_NEED_GENERIC = 35, /* props outside the fast kernel's specialisation: rerun in the generic one */
  MILZMA_ST_NEED_RERUN = 36,  /* internal to the whole-file calls' streamed launches (their input goes up in two parts): a unit read
                                 beyond the part that was in place: its output is void, it is decoded again                      */
  MILZMA_ST_NEED_INPUT = 37   /* MILZMA_DECODE_FEED: the unit stopped within 20 bytes of the end of its input VIEW (err_a ==
                                 MILZMA_PARKED, always); in_consumed = bytes of the view it has used.  Resume it with a view that
                                 starts at that byte                                                                            */
};

typedef struct milzma_result {
  uint32_t status;      /* MILZMA_ST_*                                                               */
  uint32_t chunks;      /* LZMA2: packets walked (diagnostic)                                          */
  uint64_t out_len;     /* bytes appended to the unit's output slice                                   */
  uint64_t out_flushed; /* of those, how many the reference's sink would have received at return:
                           all of them on success; on error only what had been flushed before
                           (ring wraps for RAW_LZMA, dictionary resets for LZMA2)                     */
  uint64_t in_consumed; /* reader position at return, relative to in_off                               */
  uint64_t err_a, err_b;
} milzma_result;

#define MILZMA_PARKED 1u /* milzma_result.err_a of a unit that stopped for room (status OUT_FULL) or input (NEED_INPUT) and can be resumed */

typedef struct milzma_ctx milzma_ctx;

/* Creates a context bound to HIP device `device` (ordinal). Fails (MILZMA_INFRA_ERROR) when no
 * GPU / HIP runtime is usable: there is no CPU fallback in this library. */
int milzma_create(int device, milzma_ctx **out_ctx);
void milzma_destroy(milzma_ctx *ctx);
/* Text of the last MILZMA_INFRA_ERROR on this context (or on creation when ctx == NULL). */
const char *milzma_last_error(const milzma_ctx *ctx);

/* The batch entry point: replaces n calls of DecoderState::process (src/decode/lzma.rs:255-261)
 * behind LzmaDecoder::decompress / Lzma2Decoder::decompress.
 *   units        host array of n descriptors
 *   d_in, d_out  DEVICE pointers: compressed input base / output base (HBM resident)
 *   results      host array of n results
 *   hip_stream   hipStream_t to launch on (NULL = default stream); the call returns after the
 *                results have been copied back (it synchronises that stream)
 * Returns MILZMA_OK or MILZMA_INFRA_ERROR; per-unit failures never fail the batch.
 * Addressing: a unit reads d_in[in_off, in_off + in_len) and owns d_out[out_off, out_off + out_cap); output slices
 * must not overlap.  The kernels fetch the input in aligned 64-byte (256-byte for the generic kernel) windows, so up
 * to 255 bytes before the first and after the last unit's input may be READ (never used): d_in must come from an
 * allocation that extends that far (any hipMalloc'd buffer does at the front; leave 256 bytes of slack at the end,
 * as milzma_decode_units_host does).  A RAW unit's dict_size below 4096 is raised to 4096 (lzma.rs:118-120). */
int milzma_decode_units(milzma_ctx *ctx, const milzma_unit *units, uint32_t n, const void *d_in,
                        void *d_out, milzma_result *results, void *hip_stream);

/* The same call in two halves, for callers that overlap a batch's decode with other work (the copies of the
 * next / previous batch on another stream, host-side parsing): _async enqueues the descriptor upload, the
 * kernel launches and the result download on `hip_stream` and returns without waiting for the GPU (`units`
 * may be freed right away; d_in / d_out must stay valid); _wait drains that stream, reruns the rare LZMA2 units
 * that switched property class, and fills `results` (n entries).  One batch in flight per context; use
 * one context per concurrent batch.  milzma_decode_units == _async followed by _wait. */
int milzma_decode_units_async(milzma_ctx *ctx, const milzma_unit *units, uint32_t n,
                              const void *d_in, void *d_out, void *hip_stream);
int milzma_decode_units_wait(milzma_ctx *ctx, milzma_result *results);

/* Growable output -- streams whose size is not known up front (every .lzma that liblzma / `xz --format=lzma` writes ends with a
 * marker and declares no size; the reference streams such output through its ring without a cap, src/decode/lzbuffer.rs:258-270,
 * src/decode/lzma.rs:441-455).  milzma_decode_units_ex is milzma_decode_units with `flags`:
 *   MILZMA_DECODE_GROW    a unit that fills its output slice before its stream ends is PARKED instead of failed: it stops at a
 *                         symbol boundary within 273 bytes of the slice's end (nothing of the next symbol looked at), its decoder
 *                         state stays in the context, its result is (MILZMA_ST_OUT_FULL, err_a = MILZMA_PARKED, out_len = bytes
 *                         produced so far, in_consumed = reader position).  A unit whose declared size fits its slice, or whose
 *                         limit is a memlimit, is never parked.  Units of every property set are parked this way since round 4 (lc + lp
 *                         >= 4 runs in the same kernel with its literal rows in a slab the context keeps beside the parked states;
 *                         such a unit's result carries bit 0x100 in err_b).  Only units the GENERIC kernel decodes -- the context was
 *                         created under MILZMA_KERNEL=generic / MILZMA_SPILL=generic, or the slab of a launch could not be
 *                         allocated -- are not parked: they report a plain OUT_FULL (err_a = 0) and must be decoded again from
 *                         the start in a larger slice.
 *   MILZMA_DECODE_RESUME  (implies GROW) continues the units the previous _ex call on this context parked: same n, same order;
 *                         `results` holds that call's results on entry, and only entries that say PARKED are touched.  For every
 *                         parked unit the descriptor names the NEW output slice (out_off / out_cap; at least 274 bytes more than
 *                         out_len) whose first out_len bytes hold the output so far -- the unit's dictionary; move them with
 *                         milzma_move_units or keep out_off and raise out_cap when the room behind the slice is free.  No byte is
 *                         decoded twice.  The input (d_in, in_off, in_len) must be what it was.  The context checks what it can
 *                         against what it recorded when it parked the units -- a unit that was not parked by the previous call, another
 *                         in_off / in_len / kind, an out_cap below the out_len it has produced: MILZMA_INFRA_ERROR, nothing launched,
 *                         the parked states stay -- and takes the launch class from its own record, not from `results`.
 *   MILZMA_DECODE_FEED    (implies GROW) fed input -- the reference's streaming front end (`impl Write for Stream`, src/decode/stream.rs:223-283:
 *                         the caller hands the compressed bytes over piece by piece; lzma.rs:435-524 `process_mode(Partial)` decodes
 *                         a symbol only while MAX_REQUIRED_INPUT = 20 bytes are at hand or a trial run shows that fewer suffice).
 *                         Every unit's (in_off, in_len) is a VIEW: the bytes of its stream that are on the device so far.  A unit
 *                         decodes what is COMPLETE in its view -- like the reference: every symbol while 20 bytes are left, then
 *                         the symbols of the tail as far as they fit -- and stops at
 *                         the symbol boundary in front of the first one that is not (an LZMA2 unit also: in front of a packet header
 *                         or a stored chunk that is not inside the view); fewer than 20 bytes are left unused then, and they belong
 *                         to that symbol or packet.  It is parked with (MILZMA_ST_NEED_INPUT, err_a = MILZMA_PARKED, in_consumed = bytes of THIS view it
 *                         has used, out_len = bytes produced so far).  EVERY unit a FEED call parks -- for input or, as under GROW,
 *                         for room (MILZMA_ST_OUT_FULL) -- is resumed (RESUME | FEED, or plain RESUME when no stream has more to
 *                         come) with a descriptor whose view starts at its first unused byte: the bytes from in_off + in_consumed
 *                         on, wherever the caller has them now in (the new) d_in -- the unused tail moved in front of the newly
 *                         arrived bytes of a ring, or simply a longer view of the same buffer.  A view may be of any length, also
 *                         empty; in_consumed of every result counts from the start of the view of its own call.  A unit whose view
 *                         ends where its stream ends carries MILZMA_KIND_LAST_VIEW in `kind`: it is decoded to its end like any
 *                         unit (a stream that ends early is the reference's UnexpectedEof there, reader position and delivered
 *                         bytes included); a call without the FEED flag treats every view as the last.  An end marker / declared
 *                         size / LZMA2 end byte met inside a view ends the unit as usual.  Asm-kernel classes only (a context under
 *                         MILZMA_KERNEL / MILZMA_SPILL = generic refuses the flag); LZMA2 units run with a literal-row slab (a
 *                         property switch can be followed without decoding again from a start that is no longer there).
 *                         A batch under FEED need not begin together: in a RESUME | FEED call a unit marked MILZMA_KIND_START starts
 *                         fresh in its place of the batch (which must hold no parked unit) beside the units that resume -- such a call
 *                         may even be the first of its batch --, and a parked unit marked MILZMA_KIND_HOLD is left parked.
 * Parked states live until the next decode call on the context that is not a RESUME. */
#define MILZMA_DECODE_GROW 1u
#define MILZMA_DECODE_RESUME 2u
#define MILZMA_DECODE_FEED 4u
int milzma_decode_units_ex(milzma_ctx *ctx, const milzma_unit *units, uint32_t n, const void *d_in,
                           void *d_out, milzma_result *results, void *hip_stream, uint32_t flags);

/* Device-side move of n byte ranges: d_dst[dst_off[i], +len[i]) = d_src[src_off[i], +len[i]) for all i in one launch (ranges of one
 * call must not overlap each other's destination).  What a caller of MILZMA_DECODE_RESUME uses to carry the parked units' output
 * into their larger slices; runs on hip_stream and returns when it has finished. */
int milzma_move_units(milzma_ctx *ctx, uint32_t n, const void *d_src, const uint64_t *src_off,
                      void *d_dst, const uint64_t *dst_off, const uint64_t *len, void *hip_stream);

/* Same, with host-resident input / output: the library stages both through device buffers it
 * owns (PCIe-inclusive path).  Every unit's slices are checked against in_bytes / out_bytes and
 * against each other (MILZMA_INFRA_ERROR, nothing decoded, if one lies outside or two outputs overlap). */
int milzma_decode_units_host(milzma_ctx *ctx, const milzma_unit *units, uint32_t n,
                             const void *h_in, size_t in_bytes, void *h_out, size_t out_bytes,
                             milzma_result *results);

/* Duration in ms of the decode kernels of the most recent milzma_decode_units* call, measured
 * with HIP events on the launch stream, and how many kernel launches that was. */
float milzma_last_kernel_ms(const milzma_ctx *ctx, uint32_t *launches);

/* Which way the most recent whole-file batch call on this context (milzma_{lzma,lzma2,xz}_decompress_batch) sent its files -- a bit mask;
 * for operators and tests ("did my batch take the streamed launch?"), never needed for correctness: every path hands back the same bytes. */
#define MILZMA_PATH_STREAMED 1u        /* one streamed launch: the waves wrote the output into the result buffers while they decoded */
#define MILZMA_PATH_TWO_PART_INPUT 2u  /* ... and its input went up in two parts (every unit's lead before the launch, the rest beside it) */
#define MILZMA_PATH_CLASSIC 4u         /* staged upload, decode launches, staged download */
#define MILZMA_PATH_GROUPED 8u         /* the call was cut into groups over contexts of the device (>= 8192 units, or MILZMA_LANES) */
uint32_t milzma_last_call_paths(const milzma_ctx *ctx);

/* Renders a unit result as the reference would: returns the error kind (MILZMA_OK ...
 * MILZMA_XZ_ERROR) and writes the full Display string (src/error.rs:28-36) into msg. */
int milzma_result_message(const milzma_result *res, uint32_t unit_kind, char *msg, size_t msg_cap);

/* ---- whole-file entry points: src/lib.rs:44-105 ------------------------------------------ */

/* decompress::UnpackedSize (src/decode/options.rs:22-43) */
enum {
  MILZMA_READ_FROM_HEADER = 0,
  MILZMA_READ_HEADER_BUT_USE_PROVIDED = 1,
  MILZMA_USE_PROVIDED = 2
};

/* decompress::Options (src/decode/options.rs:3-20); allow_incomplete is stream-API only (milzma_streams_*). */
typedef struct milzma_options {
  int32_t unpacked_size_mode;
  int32_t provided_is_some;
  uint64_t provided;
  int32_t memlimit_is_some;
  int32_t allow_incomplete; /* Options.allow_incomplete (options.rs:16-19): finish() of a stream whose input ends early is a success */
  uint64_t memlimit;
} milzma_options;

/* What the call did to the caller's reader and writer. `data` holds exactly the bytes the
 * reference would have written to its `W: io::Write` (also on error); free with milzma_free. */
typedef struct milzma_output {
  uint8_t *data;
  size_t len;
  size_t in_consumed; /* how far the reference would have advanced its `R: io::BufRead` */
  int32_t kind;       /* MILZMA_OK or the error kind */
  char msg[388];      /* Display string of the error */
} milzma_output;

void milzma_default_options(milzma_options *opt);
/* Returns an output buffer (milzma_output.data) to the library.  The library keeps a registry of the pointers it has handed out:
 * a pointer that is not in it (foreign, or freed already) is ignored WITHOUT being dereferenced.  Freed buffers rest in a
 * process-wide pool and are handed out again by later calls (their pages stay mapped: a batch call hands out thousands of buffers);
 * the pool holds at most what the caller had in use at once, and never more than MILZMA_POOL_BYTES (environment, default 8 GiB). */
void milzma_free(void *p);
/* Gives pooled output buffers back to the C allocator until at most keep_bytes rest in the pool (0: all of them), and forgets the
 * high-water mark that bounds what the pool keeps.  Returns the bytes still pooled.  Thread-safe; needs no context. */
size_t milzma_pool_trim(size_t keep_bytes);

/* lzma_decompress_with_options (src/lib.rs:52-60); opt == NULL => Options::default() */
int milzma_lzma_decompress(milzma_ctx *ctx, const uint8_t *in, size_t in_len,
                           const milzma_options *opt, milzma_output *out);
/* lzma2_decompress (src/lib.rs:83-88) */
int milzma_lzma2_decompress(milzma_ctx *ctx, const uint8_t *in, size_t in_len, milzma_output *out);
/* xz_decompress (src/lib.rs:100-105) */
int milzma_xz_decompress(milzma_ctx *ctx, const uint8_t *in, size_t in_len, milzma_output *out);

/* Many complete files in one go: every stream (.lzma) / every block of every file (.xz) becomes
 * one unit of a single launch. outs[i] is filled exactly as the single-file call would. */
int milzma_lzma_decompress_batch(milzma_ctx *ctx, uint32_t n, const uint8_t *const *ins,
                                 const size_t *in_lens, const milzma_options *opt,
                                 milzma_output *outs);
int milzma_lzma2_decompress_batch(milzma_ctx *ctx, uint32_t n, const uint8_t *const *ins,
                                  const size_t *in_lens, milzma_output *outs);
int milzma_xz_decompress_batch(milzma_ctx *ctx, uint32_t n, const uint8_t *const *ins,
                               const size_t *in_lens, milzma_output *outs);

/* The batch calls in two halves, so that a caller can keep two of them in flight (two contexts on one device: the
 * H2D of one call and the D2H + hand-over of the other overlap the decode kernel of whichever holds the GPU; a
 * 4096-stream kernel fills the chip, so two kernels never overlap).  _async copies the pointer / length arrays and
 * the options and returns at once; the files' bytes and `outs` must stay valid until milzma_batch_wait(ctx), which
 * returns what the synchronous call would have returned.  One batch in flight per context. */
int milzma_lzma_decompress_batch_async(milzma_ctx *ctx, uint32_t n, const uint8_t *const *ins,
                                       const size_t *in_lens, const milzma_options *opt,
                                       milzma_output *outs);
int milzma_lzma2_decompress_batch_async(milzma_ctx *ctx, uint32_t n, const uint8_t *const *ins,
                                        const size_t *in_lens, milzma_output *outs);
int milzma_xz_decompress_batch_async(milzma_ctx *ctx, uint32_t n, const uint8_t *const *ins,
                                     const size_t *in_lens, milzma_output *outs);
int milzma_batch_wait(milzma_ctx *ctx);

/* CRC-32 (ISO-HDLC) and CRC-64/XZ of what each unit decoded, computed on the GPU over the device-resident
 * output of a milzma_decode_units call with the same units / d_out / results: the digest step of
 * validate_block_check (src/decode/xz.rs:292-333) with the polynomials of src/xz/crc.rs:1-4, without a
 * host pass over the output.  crc32 / crc64: n host values each (either may be NULL); a unit whose status
 * is not OK gets 0. */
int milzma_crc_units(milzma_ctx *ctx, const milzma_unit *units, uint32_t n, const void *d_out,
                     const milzma_result *results, uint32_t *crc32, uint64_t *crc64, void *hip_stream);

/* ---- several GPUs of one node behind one handle --------------------------------------------------
 * Every public entry point of the reference builds a fresh decoder (src/lib.rs:44-105, src/decode/lzma2.rs:23-34):
 * streams, LZMA2 groups and XZ blocks are independent, so a batch shards across devices with no exchange between
 * them.  A milzma_multi owns one milzma_ctx and one host worker thread per device; the work of a call is
 * partitioned by compressed bytes (what a wavefront's time follows), each device's share travels over that
 * device's own PCIe link, is decoded there and comes back the same way, all devices concurrently.  Nothing is
 * funnelled through one GPU (no xGMI hop is needed for host-resident data); callers whose data already lives
 * in the devices' memory use milzma_multi_decode_units. */
typedef struct milzma_multi milzma_multi;

/* device_mask: bit d = HIP ordinal d; 0 = every visible device.  Fails (MILZMA_INFRA_ERROR, text from
 * milzma_multi_last_error(NULL)) if a named device is missing or is not a gfx950: no partial sets, no CPU fallback. */
int milzma_multi_create(uint64_t device_mask, milzma_multi **out);
void milzma_multi_destroy(milzma_multi *m);
/* number of devices; their HIP ordinals into ordinals[0..cap) when not NULL */
uint32_t milzma_multi_devices(const milzma_multi *m, int *ordinals, uint32_t cap);
const char *milzma_multi_last_error(const milzma_multi *m);
/* kernel ms / launches of device index k's share of the most recent call (max over devices when k == UINT32_MAX) */
float milzma_multi_last_kernel_ms(const milzma_multi *m, uint32_t k, uint32_t *launches);

/* The partition the multi entry points use; needs no GPU.  Item i (a unit, or a whole file) has weight weights[i]
 * (compressed bytes) and goes to part part_of[i] in [0, parts).  Longest-processing-time-first: items by falling
 * weight, each to the lightest part so far (ties: lowest part index; equal weights: lowest item index first), which
 * keeps the heaviest part within one item of the mean.  group (may be NULL): items with the same non-zero group id
 * stay together (the LZMA2 units of one stream, the blocks of one .xz file) and are placed as one item of their
 * summed weight.  Deterministic. */
int milzma_partition(const uint64_t *weights, const uint32_t *group, uint32_t n, uint32_t parts,
                     uint32_t *part_of);

/* milzma_decode_units_host over all devices of m: same arguments, same checks, same results. */
int milzma_multi_decode_units_host(milzma_multi *m, const milzma_unit *units, uint32_t n,
                                   const void *h_in, size_t in_bytes, void *h_out, size_t out_bytes,
                                   milzma_result *results);
/* milzma_decode_units for data that already lives on the devices: unit i runs on device index device_of[i] (an
 * index into the handle's device list) and addresses d_in[device_of[i]] / d_out[device_of[i]] (device pointers on
 * that device; slack rules as for milzma_decode_units).  One launch sequence per device, all devices concurrently;
 * returns when every device's results are back. */
int milzma_multi_decode_units(milzma_multi *m, const milzma_unit *units, uint32_t n,
                              const uint32_t *device_of, const void *const *d_in, void *const *d_out,
                              milzma_result *results);
/* One ingest point: the batch's compressed input is resident on ONE device of the handle (index `root` into its device list), the
 * output is wanted there too (d_in / d_out: device pointers on that device; descriptors and slack rules as for milzma_decode_units).
 * The units are partitioned over all devices by compressed bytes; the root's share is decoded in place, every other share is packed,
 * sent to its device with one device-to-device copy (the direct xGMI link where the GPUs have peer access; no host memory, no
 * collective) and decoded there; its output reaches the slices the descriptors name either from the decoding waves themselves (a
 * device with peer access to the root writes every unit's output there while it decodes) or by one device-to-device copy behind
 * the decode (MILZMA_ROOTED_STREAM=0 forces the copy).  All devices work concurrently.  The call works on streams of its own and does not order itself behind work the caller has queued on other
 * streams: d_in must be complete (and d_out free to be written) when it is made.  results[i]: as from milzma_decode_units.
 * milzma_multi_last_transfer_ms: the slowest device's copy in, the slowest
 * decode, and the slowest copy back plus the placement on the root, of the most recent such call (wall-clock ms). */
int milzma_multi_decode_units_rooted(milzma_multi *m, uint32_t root, const milzma_unit *units, uint32_t n,
                                     const void *d_in, void *d_out, milzma_result *results);
void milzma_multi_last_transfer_ms(const milzma_multi *m, float *scatter_ms, float *decode_ms, float *gather_ms);
/* The whole-file batch entry points over all devices: files are partitioned by size, outs[i] is exactly what the
 * single-device call (and the reference) produces for file i. */
int milzma_multi_lzma_decompress_batch(milzma_multi *m, uint32_t n, const uint8_t *const *ins,
                                       const size_t *in_lens, const milzma_options *opt,
                                       milzma_output *outs);
int milzma_multi_lzma2_decompress_batch(milzma_multi *m, uint32_t n, const uint8_t *const *ins,
                                        const size_t *in_lens, milzma_output *outs);
int milzma_multi_xz_decompress_batch(milzma_multi *m, uint32_t n, const uint8_t *const *ins,
                                     const size_t *in_lens, milzma_output *outs);

/* ---- host-side parsing, usable without a GPU (and tested without one) -------------------- */

/* LzmaParams::read_header (src/decode/lzma.rs:96-161): fills lc/lp/pb/dict_size/unpacked_size/
 * memlimit/kind of `unit` and sets *header_len to the bytes consumed. Returns the error kind and
 * message in `out` (kind/msg only) on failure. */
int milzma_lzma_read_header(const uint8_t *in, size_t in_len, const milzma_options *opt,
                            milzma_unit *unit, size_t *header_len, milzma_output *out);

/* Footer + Index of a well-formed .xz file (src/decode/xz.rs:35-110 read in the other direction) -> one
 * MILZMA_KIND_LZMA2 unit per block, for pipelines that keep files and output in device memory: in_off /
 * in_len locate the block's LZMA2 payload relative to the file's first byte, out_cap = unpacked_size = the
 * Index's uncompressed size, out_off packs the outputs from 0 (256-byte aligned).  *n_units = blocks in
 * the file (also when `units` is NULL or `cap` too small: MILZMA_INFRA_ERROR then); *check_id = the
 * stream's check type (0 none, 1 CRC32, 4 CRC64).  MILZMA_XZ_ERROR if the Index cannot be used (the
 * whole-file entry points then still give the reference's verdict).  Container integrity (header / index /
 * footer CRCs, the blocks' check values) is NOT verified here. */
int milzma_xz_plan(const uint8_t *in, size_t in_len, milzma_unit *units, uint32_t cap,
                   uint32_t *n_units, uint32_t *check_id);

/* CRC-32 (ISO-HDLC) and CRC-64/XZ as used by the XZ layer (src/xz/crc.rs:1-4). */
uint32_t milzma_crc32(const uint8_t *p, size_t n);
uint64_t milzma_crc64(const uint8_t *p, size_t n);

uint32_t milzma_abi_version(void);

/* ---- push-mode decoding: lzma_rs::decompress::Stream (feature `stream`, src/decode/stream.rs) for a batch of streams -----------------
 * The crate's Stream<W> is an io::Write: the compressed .lzma bytes are written to it piece by piece, finish() hands the sink back.
 * A milzma_streams is n of them over one GPU (a context of its own behind `ctx`'s device): a write call appends bytes to any of the
 * streams and runs ONE launch in which every stream that got bytes takes another turn (fed input, MILZMA_DECODE_FEED: streams whose
 * header has just become complete start in that launch, the others resume where they parked).  Output stays on the device until finish.
 *   milzma_streams_open    kind = MILZMA_KIND_RAW_LZMA: n x Stream::new_with_options (stream.rs:88-101) -- .lzma files, header first;
 *                          options: n entries (unpacked_size mode, memlimit, allow_incomplete) or NULL for Options::default().
 *                          kind = MILZMA_KIND_LZMA2: n raw LZMA2 streams fed the same way (the crate has no such type; what it gives a
 *                          binding is lzma2_decompress over a reader that shows its input piece by piece -- the verdict and the reader
 *                          position of Lzma2Decoder::decompress, src/decode/lzma2.rs:52-82 -- without reading ahead; options ignored).
 *                          A stream of either kind that has ENDED (declared size reached, LZMA2 end byte read) takes no more bytes:
 *                          WriteZero, and finish's in_consumed says where it ended.
 *   milzma_streams_write   io::Write::write_all (over Stream::write, stream.rs:223-326) for k of the streams: data[j] / len[j] go to
 *                          stream idx[j] (a stream at most once per call).  status[j] (optional): MILZMA_OK, or MILZMA_IO_ERROR when that
 *                          write_all returns Err -- milzma_streams_write_error(s, idx[j]) is the io::Error's text: a fatal header error
 *                          ("LZMA header invalid properties: 255 must be < 225"), a decode error in the crate's Debug form
 *                          (`LzmaError("LZ distance 5 is beyond output size 3")`, stream.rs:343-347), or "failed to write whole buffer"
 *                          (ErrorKind::WriteZero: bytes behind a stream whose declared size is reached; the stream itself is intact,
 *                          tests/lzma.rs:71-87).  After a failed write (other than WriteZero) the crate's Stream has no state left
 *                          (stream.rs:230): write() returns Ok(0), so every later write_all of bytes is WriteZero too, and finish() fails.
 *                          Returns MILZMA_INFRA_ERROR only for infrastructure failures (milzma_streams_last_error).
 *                          An END MARKER that ends a write's data does NOT end a .lzma stream (the crate's Partial-mode loop merely leaves
 *                          at ProcessingStatus::Finished, lzma.rs:493-495, :507-509): bytes written later are decoded on from the marker's
 *                          state -- rep[0] = 0xFFFF_FFFF and code == 0, so the next decision says "literal", a matched one whose match
 *                          byte lies 2^32 back: "Match distance 4294967296 is beyond dictionary size ..." from the write that brings the
 *                          20th byte behind the marker (with fewer at hand the crate's trial run fails and it waits), or from finish --
 *                          and a marker with bytes behind it in the same write is "Found end-of-stream marker but more bytes are available".
 *   milzma_streams_write_taken  how many bytes of the most recent write stream i TOOK -- the sum of the Ok(n) the crate's Stream::write
 *                          returns while write_all loops over it (stream.rs:324): all of them when the write succeeded; when it ended in
 *                          WriteZero, the bytes in front of the stream's end (0 for a stream that had ended or failed before).  For a
 *                          binding that implements io::Write::write, not only write_all.  "In front of the stream's end" is the crate's
 *                          reckoning: once a write has ended inside a symbol, the crate decodes through its 20-byte partial-input buffer
 *                          (lzma.rs:457-495), and when the stream then reaches its declared size the buffer has swallowed the 20 bytes
 *                          from the last symbol's first byte on -- bytes behind the stream's end among them: they count as taken, and a
 *                          write whose rest fits in there is NOT a WriteZero (the kernel notes where the last symbol began).
 *   milzma_streams_output  Stream::get_output (stream.rs:102-116) for one stream: *sink_len = the bytes its sink holds right now -- every
 *                          completed flush of the ring (whole multiples of the dictionary size, lzbuffer.rs:264-267; nothing before the
 *                          header is complete) --, *has_sink = 0 after a failed write (the crate answers None), and the sink's bytes
 *                          [offset, offset + cap) go to dst (a binding keeps the caller's `W` current by asking for what it has not
 *                          delivered yet).  Before milzma_streams_finish only.
 *   milzma_streams_finish  Stream::finish (stream.rs:119-150) for every stream: outs[i] as from milzma_lzma_decompress -- kind / msg of
 *                          the crate's Result (header incomplete: "lzma error: failed to read header"; input ends early: "io error:
 *                          failed to fill whole buffer", unless allow_incomplete, which hands over everything decoded so far; after a
 *                          failed write: "lzma error: can't finish stream because of previous write error"), data = what the sink holds.
 *                          A stream that stands behind an end marker gets the Finish-mode pass from that state (lzma.rs:446-455: done if no
 *                          size is expected; a provided size not reached decodes on into the marker's distance).
 *                          Once; afterwards only milzma_streams_close (which may also be called without finish).
 * A milzma_streams is used by one thread at a time (like a context); different ones are independent.  An infrastructure failure of a
 * write (MILZMA_INFRA_ERROR: a HIP error, no memory) leaves the batch unusable: close it.
 * The crate decodes a symbol as soon as 20 bytes are at hand OR a trial run shows it complete within fewer (lzma.rs:455-516); so does
 * this implementation -- the tail of every write's data is decoded as far as its symbols are complete (a second pass over the tail
 * from a saved state; symbol by symbol, a state saved in front of each, for streams with lc + lp >= 4, whose literal rows live in device
 * memory: decode_fast_asm.hip.h) -- and a failed write is the very write the crate fails.
 * A .lzma stream whose header asks for more literal rows (lc + lp) than the batch's slab keeps per stream -- 8 for large batches, every
 * legal value for small ones -- is decoded in a one-stream batch of its own behind the same calls: it costs the other streams nothing. */
typedef struct milzma_streams milzma_streams;
/* or-ed into milzma_streams_open's `kind`: READER mode -- the streams stand for the crate's one-shot lzma_decompress / lzma2_decompress over
 * a `BufRead` that shows its input piece by piece (a BufReader over a socket or a large file): the pieces are written as they are shown,
 * and milzma_streams_finish hands over what the ONE-SHOT call would -- a failed decode's own error with the bytes written before it,
 * not Stream::finish's "previous write error"; a header that never became complete as "header too short: ..."; an end marker that
 * ends one piece and is followed by another as "Found end-of-stream marker but more bytes are available" (the one-shot call asks the
 * reader itself whether it is at its end) -- with in_consumed the reader position of the whole stream.  A write that fails or reports WriteZero tells the caller to stop showing input and finish:
 * in_consumed minus the bytes of the pieces written BEFORE that one is how much of the last piece the reader is to consume
 * (integration/rust/src/lib.rs `run_fed`): every byte a parked stream has not used belongs to a symbol that is not complete yet, so
 * nothing behind a stream's end is ever taken with an earlier piece. */
#define MILZMA_STREAMS_AS_READER 0x100u
int milzma_streams_open(milzma_ctx *ctx, uint32_t kind, uint32_t n, const milzma_options *options, milzma_streams **out);
int milzma_streams_write(milzma_streams *s, uint32_t k, const uint32_t *idx, const void *const *data, const size_t *len, int32_t *status);
const char *milzma_streams_write_error(const milzma_streams *s, uint32_t stream);
int milzma_streams_finish(milzma_streams *s, milzma_output *outs);
void milzma_streams_close(milzma_streams *s);
const char *milzma_streams_last_error(const milzma_streams *s);
uint64_t milzma_streams_write_taken(const milzma_streams *s, uint32_t stream);
int milzma_streams_output(milzma_streams *s, uint32_t stream, uint64_t offset, void *dst, size_t cap, uint64_t *sink_len,
                          int32_t *has_sink);

#ifdef __cplusplus
}
#endif
#endif /* MILZMA_H */
