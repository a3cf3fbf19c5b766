/*
 * lzma_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the decode path of gendx/lzma-rs, written to
 * follow the reference's structure function by function so that it can stand
 * in for the crate (no Rust toolchain exists in this image) in every parity
 * check.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
 * leg may include, link or dlopen this.  The product library
 * (lzma_rs_amd/csrc) never does.
 *
 * Parity pin: checked against every fixture / inline known-answer vector of
 * the reference's own tests (tests/lzma.rs, tests/lzma2.rs, tests/xz.rs,
 * tests/files/) -- see tests/test_oracle_golden.py -- and differentially
 * against liblzma 5.2.5 (Python `lzma`) the way tests/lzma.rs:109-114 does.
 *
 * Each function cites the reference file:line it restates (paths relative to
 * the reference repository root).
 */
#ifndef LZMA_ORACLE_H
#define LZMA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error::Error variants, src/error.rs:8-17 */
enum {
  ORC_OK = 0,
  ORC_IO_ERROR = 1,         /* Error::IoError          "io error: ..."          */
  ORC_HEADER_TOO_SHORT = 2, /* Error::HeaderTooShort   "header too short: ..."  */
  ORC_LZMA_ERROR = 3,       /* Error::LzmaError        "lzma error: ..."        */
  ORC_XZ_ERROR = 4          /* Error::XzError          "xz error: ..."          */
};

/* decompress::UnpackedSize, src/decode/options.rs:22-43 */
enum {
  ORC_READ_FROM_HEADER = 0,
  ORC_READ_HEADER_BUT_USE_PROVIDED = 1,
  ORC_USE_PROVIDED = 2
};

/* decompress::Options, src/decode/options.rs:3-20 (allow_incomplete is
 * stream-API only and has no effect on the one-shot path). */
typedef struct orc_options {
  int unpacked_size_mode;      /* ORC_READ_FROM_HEADER / ... */
  int provided_is_some;        /* Option<u64> discriminant for the provided value */
  uint64_t provided;           /* the provided value when provided_is_some */
  int memlimit_is_some;        /* Option<usize> discriminant */
  uint64_t memlimit;
} orc_options;

/* Result of one decode call.  `msg` is the full Display string of the
 * error (src/error.rs:28-36), e.g. "lzma error: LZ distance 5 is beyond
 * output size 3".  `out`/`out_len` hold exactly the bytes the reference would
 * have pushed into its `W: io::Write` sink (also on error: whatever had been
 * flushed before the error).  `in_consumed` is the reader position at return.
 * The caller frees `out` with orc_free(). */
typedef struct orc_result {
  int kind;
  char msg[384];
  uint8_t *out;
  size_t out_len;
  size_t in_consumed;
} orc_result;

void orc_default_options(orc_options *o);
void orc_free(void *p);

/* src/lib.rs:44-60 */
int orc_lzma_decompress(const uint8_t *in, size_t in_len, const orc_options *opt,
                        orc_result *res);
/* src/lib.rs:83-88 */
int orc_lzma2_decompress(const uint8_t *in, size_t in_len, orc_result *res);
/* src/lib.rs:100-105 */
int orc_xz_decompress(const uint8_t *in, size_t in_len, orc_result *res);

/* Raw LZMA decode (feature raw_decoder: LzmaParams::new + LzmaDecoder::new +
 * decompress, src/decode/lzma.rs:83-93,607-648): no 13-byte header, the input
 * starts at the range coder's first byte. unpacked_is_some==0 => marker mode. */
int orc_lzma_raw_decompress(const uint8_t *in, size_t in_len, uint32_t lc, uint32_t lp,
                            uint32_t pb, uint32_t dict_size, int unpacked_is_some,
                            uint64_t unpacked_size, int memlimit_is_some, uint64_t memlimit,
                            orc_result *res);

/* CRCs used by the XZ layer (crate `crc` 3.0: CRC_32_ISO_HDLC, CRC_64_XZ;
 * src/xz/crc.rs:1-4). */
/* Tuning aid: while `hist64` (64 counters) is set, every LZ77 copy adds to [2*floor(log2(dist))] (matches) and
 * [2*floor(log2(dist)) + 1] (bytes).  NULL switches it off.  Not thread-safe. */
void orc_set_match_hist(uint64_t *hist64);

uint32_t orc_crc32(const uint8_t *p, size_t n);
uint64_t orc_crc64(const uint8_t *p, size_t n);

/* cpu_baseline helper for bench.py: decode `n` complete .lzma streams
 * (stream i = in_base[in_off[i] .. in_off[i]+in_len[i])) on `nthreads`
 * pthreads, one stream per thread at a time; returns total bytes produced and
 * stores an xor-fold of per-stream CRC32s in *check. Negative on any error. */
int64_t orc_bench_lzma_batch(const uint8_t *in_base, const uint64_t *in_off,
                             const uint64_t *in_len, uint32_t n, uint32_t nthreads,
                             uint32_t *check);

/* Stream (feature `stream`), src/decode/stream.rs: a push-mode .lzma decoder over a Vec<u8> sink, restated with the Partial mode of
 * DecoderState::process_mode (src/decode/lzma.rs:395-524: partial input buffer, try_process_next).
 *   orc_stream_new        Stream::new_with_options (stream.rs:88-101); allow_incomplete = Options.allow_incomplete
 *   orc_stream_write_all  io::Write::write_all over Stream::write (stream.rs:223-326): 0, or ORC_IO_ERROR with msg (384 bytes) = the
 *                         io::Error's Display text (no "io error: " prefix: it is not an error::Error)
 *   orc_stream_output     Stream::get_output (stream.rs:102-107): the sink so far; orc_stream_has_output: whether the crate answers
 *                         Some(..) (0 once a write has failed: the state was taken, stream.rs:230)
 *   orc_stream_finish     Stream::finish (stream.rs:119-150); frees the stream */
typedef struct orc_stream orc_stream;
orc_stream *orc_stream_new(const orc_options *opt, int allow_incomplete);
int orc_stream_write_all(orc_stream *s, const uint8_t *data, size_t len, char *msg);
size_t orc_stream_output(const orc_stream *s, const uint8_t **p);
int orc_stream_has_output(const orc_stream *s);
size_t orc_stream_last_taken(const orc_stream *s); /* bytes the last write_all handed to Stream::write with Ok(n) */
int orc_stream_finish(orc_stream *s, orc_result *res);

#ifdef __cplusplus
}
#endif
#endif
