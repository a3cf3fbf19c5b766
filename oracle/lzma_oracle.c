/*
 * lzma_oracle.c -- TEST INFRASTRUCTURE ONLY (see lzma_oracle.h).
 *
 * CPU restatement of gendx/lzma-rs's decode path.  It keeps the reference's
 * structure (one normalisation byte per bit, byte-at-a-time ring buffer, the
 * same error sites in the same order) so that status, message, bytes written
 * to the sink and reader position all match what the crate would produce.
 * It is the checker for the HIP path and the "port" CPU baseline of bench.py;
 * it is never the thing shipped.
 *
 * Citations are reference paths (relative to the reference repo root).
 */
#include "lzma_oracle.h"

#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* plumbing: reader (io::BufRead over a slice), sink (Vec<u8>), errors */
/* ------------------------------------------------------------------ */

typedef struct {
  const uint8_t *p;
  size_t pos;
  size_t end; /* current limit: slice end, or the end of an io::Read::take() window */
} reader_t;

typedef struct {
  uint8_t *data;
  size_t len, cap;
} sink_t;

typedef struct {
  int kind;
  char msg[384];
} err_t;

static const char *EOF_MSG = "failed to fill whole buffer"; /* io::ErrorKind::UnexpectedEof */

static int fail(err_t *e, int kind, const char *fmt, ...) {
  static const char *prefix[] = {"", "io error: ", "header too short: ", "lzma error: ",
                                 "xz error: "}; /* src/error.rs:28-36 */
  va_list ap;
  size_t n;
  e->kind = kind;
  n = (size_t)snprintf(e->msg, sizeof e->msg, "%s", prefix[kind]);
  va_start(ap, fmt);
  vsnprintf(e->msg + n, sizeof e->msg - n, fmt, ap);
  va_end(ap);
  return kind;
}

static int io_eof(err_t *e) { return fail(e, ORC_IO_ERROR, "%s", EOF_MSG); }

static void sink_write_all(sink_t *s, const uint8_t *p, size_t n) {
  if (n == 0) return;
  if (s->len + n > s->cap) {
    size_t nc = s->cap ? s->cap : 4096;
    while (nc < s->len + n) nc *= 2;
    s->data = (uint8_t *)realloc(s->data, nc);
    s->cap = nc;
  }
  memcpy(s->data + s->len, p, n);
  s->len += n;
}

/* byteorder::ReadBytesExt::read_u8 on a BufRead: 0 ok, 1 UnexpectedEof */
static int rd_u8(reader_t *r, uint8_t *v) {
  if (r->pos >= r->end) return 1;
  *v = r->p[r->pos++];
  return 0;
}

/* read_exact semantics: on a short read everything available is consumed. */
static int rd_exact(reader_t *r, uint8_t *dst, size_t n) {
  size_t avail = r->end - r->pos;
  if (avail < n) {
    r->pos = r->end;
    return 1;
  }
  if (dst) memcpy(dst, r->p + r->pos, n);
  r->pos += n;
  return 0;
}

static int rd_u16be(reader_t *r, uint32_t *v) {
  uint8_t b[2];
  if (rd_exact(r, b, 2)) return 1;
  *v = ((uint32_t)b[0] << 8) | b[1];
  return 0;
}
static int rd_u32be(reader_t *r, uint32_t *v) {
  uint8_t b[4];
  if (rd_exact(r, b, 4)) return 1;
  *v = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
  return 0;
}
static int rd_u32le(reader_t *r, uint32_t *v) {
  uint8_t b[4];
  if (rd_exact(r, b, 4)) return 1;
  *v = ((uint32_t)b[3] << 24) | ((uint32_t)b[2] << 16) | ((uint32_t)b[1] << 8) | b[0];
  return 0;
}
static int rd_u64le(reader_t *r, uint64_t *v) {
  uint8_t b[8];
  int i;
  if (rd_exact(r, b, 8)) return 1;
  *v = 0;
  for (i = 7; i >= 0; i--) *v = (*v << 8) | b[i];
  return 0;
}
/* decode::util::is_eof, src/decode/util.rs:9-12 */
static int rd_is_eof(const reader_t *r) { return r->pos >= r->end; }

/* ------------------------------------------------------------------ */
/* LzBuffer implementations, src/decode/lzbuffer.rs                    */
/* ------------------------------------------------------------------ */

typedef struct {
  int is_accum; /* 1: LzAccumBuffer (LZMA2), 0: LzCircularBuffer (.lzma) */
  sink_t *stream;
  uint8_t *buf;
  size_t buf_len, buf_cap; /* Vec<u8> len/capacity */
  size_t dict_size;        /* ring only */
  uint64_t memlimit;
  size_t cursor; /* ring only */
  uint64_t len;  /* total bytes through the buffer (since last accum reset) */
} lzbuf_t;

static void lzbuf_reserve(lzbuf_t *b, size_t n) {
  if (n > b->buf_cap) {
    size_t nc = b->buf_cap ? b->buf_cap : 4096;
    while (nc < n) nc *= 2;
    b->buf = (uint8_t *)realloc(b->buf, nc);
    b->buf_cap = nc;
  }
}

/* LzCircularBuffer::get, lzbuffer.rs:202-204 */
static uint8_t ring_get(const lzbuf_t *b, size_t index) {
  return index < b->buf_len ? b->buf[index] : 0;
}

/* LzCircularBuffer::set, lzbuffer.rs:206-221 */
static int ring_set(lzbuf_t *b, size_t index, uint8_t value, err_t *e) {
  size_t new_len = index + 1;
  if (b->buf_len < new_len) {
    if ((uint64_t)new_len <= b->memlimit) {
      lzbuf_reserve(b, new_len);
      memset(b->buf + b->buf_len, 0, new_len - b->buf_len);
      b->buf_len = new_len;
    } else {
      return fail(e, ORC_LZMA_ERROR, "exceeded memory limit of %llu",
                  (unsigned long long)b->memlimit);
    }
  }
  b->buf[index] = value;
  return 0;
}

/* LzBuffer::last_or, lzbuffer.rs:87-94 (accum), :232-238 (ring) */
static uint8_t lzbuf_last_or(const lzbuf_t *b, uint8_t lit) {
  if (b->is_accum) return b->buf_len == 0 ? lit : b->buf[b->buf_len - 1];
  if (b->len == 0) return lit;
  return ring_get(b, (b->dict_size + b->cursor - 1) % b->dict_size);
}

/* LzBuffer::last_n, lzbuffer.rs:96-106 (accum), :240-255 (ring) */
static int lzbuf_last_n(const lzbuf_t *b, uint64_t dist, uint8_t *out, err_t *e) {
  if (b->is_accum) {
    if (dist > b->buf_len)
      return fail(e, ORC_LZMA_ERROR, "Match distance %llu is beyond output size %llu",
                  (unsigned long long)dist, (unsigned long long)b->buf_len);
    *out = b->buf[b->buf_len - dist];
    return 0;
  }
  if (dist > b->dict_size)
    return fail(e, ORC_LZMA_ERROR, "Match distance %llu is beyond dictionary size %llu",
                (unsigned long long)dist, (unsigned long long)b->dict_size);
  if (dist > b->len)
    return fail(e, ORC_LZMA_ERROR, "Match distance %llu is beyond output size %llu",
                (unsigned long long)dist, (unsigned long long)b->len);
  *out = ring_get(b, (b->dict_size + b->cursor - (size_t)dist) % b->dict_size);
  return 0;
}

/* LzBuffer::append_literal, lzbuffer.rs:108-121 (accum), :257-270 (ring) */
static int lzbuf_append_literal(lzbuf_t *b, uint8_t lit, err_t *e) {
  if (b->is_accum) {
    uint64_t new_len = b->len + 1;
    if (new_len > b->memlimit)
      return fail(e, ORC_LZMA_ERROR, "exceeded memory limit of %llu",
                  (unsigned long long)b->memlimit);
    lzbuf_reserve(b, b->buf_len + 1);
    b->buf[b->buf_len++] = lit;
    b->len = new_len;
    return 0;
  }
  if (ring_set(b, b->cursor, lit, e)) return e->kind;
  b->cursor += 1;
  b->len += 1;
  if (b->cursor == b->dict_size) { /* flush the ring to the sink on wrap */
    sink_write_all(b->stream, b->buf, b->buf_len);
    b->cursor = 0;
  }
  return 0;
}

/* Workload statistics (tuning aid, not reference behaviour): matches by log2(distance) bucket and their bytes, filled
 * while orc_match_hist points somewhere (single-threaded use only). hist[2*k] = matches with 2^k <= dist < 2^(k+1),
 * hist[2*k+1] = bytes they copied. */
static uint64_t *orc_match_hist_ptr = NULL;
void orc_set_match_hist(uint64_t *hist64) { orc_match_hist_ptr = hist64; }

/* LzBuffer::append_lz, lzbuffer.rs:123-141 (accum), :272-297 (ring) */
static int lzbuf_append_lz(lzbuf_t *b, uint64_t len, uint64_t dist, err_t *e) {
  if (orc_match_hist_ptr && dist) {
    int k = 63 - __builtin_clzll(dist);
    if (k > 31) k = 31;
    orc_match_hist_ptr[2 * k] += 1;
    orc_match_hist_ptr[2 * k + 1] += len;
  }
  uint64_t i;
  if (b->is_accum) {
    size_t offset;
    if (dist > b->buf_len)
      return fail(e, ORC_LZMA_ERROR, "LZ distance %llu is beyond output size %llu",
                  (unsigned long long)dist, (unsigned long long)b->buf_len);
    offset = b->buf_len - (size_t)dist;
    lzbuf_reserve(b, b->buf_len + (size_t)len);
    for (i = 0; i < len; i++) { /* byte-serial: overlapping copies replicate */
      b->buf[b->buf_len++] = b->buf[offset++];
    }
    b->len += len;
    return 0;
  }
  if (dist > b->dict_size)
    return fail(e, ORC_LZMA_ERROR, "LZ distance %llu is beyond dictionary size %llu",
                (unsigned long long)dist, (unsigned long long)b->dict_size);
  if (dist > b->len)
    return fail(e, ORC_LZMA_ERROR, "LZ distance %llu is beyond output size %llu",
                (unsigned long long)dist, (unsigned long long)b->len);
  {
    size_t offset = (b->dict_size + b->cursor - (size_t)dist) % b->dict_size;
    for (i = 0; i < len; i++) {
      uint8_t x = ring_get(b, offset);
      if (lzbuf_append_literal(b, x, e)) return e->kind;
      offset += 1;
      if (offset == b->dict_size) offset = 0;
    }
  }
  return 0;
}

/* LzAccumBuffer::append_bytes, lzbuffer.rs:66-70 */
static void accum_append_bytes(lzbuf_t *b, const uint8_t *p, size_t n) {
  lzbuf_reserve(b, b->buf_len + n);
  memcpy(b->buf + b->buf_len, p, n);
  b->buf_len += n;
  b->len += n;
}

/* LzAccumBuffer::reset, lzbuffer.rs:72-78 */
static void accum_reset(lzbuf_t *b) {
  sink_write_all(b->stream, b->buf, b->buf_len);
  b->buf_len = 0;
  b->len = 0;
}

/* LzBuffer::finish, lzbuffer.rs:153-157 (accum), :309-315 (ring) */
static void lzbuf_finish(lzbuf_t *b) {
  if (b->is_accum) {
    sink_write_all(b->stream, b->buf, b->buf_len);
  } else if (b->cursor > 0) {
    sink_write_all(b->stream, b->buf, b->cursor);
  }
}

/* ------------------------------------------------------------------ */
/* Range decoder, src/decode/rangecoder.rs                             */
/* ------------------------------------------------------------------ */

typedef struct {
  reader_t *stream;
  uint32_t range, code;
} rc_t;

/* RangeDecoder::new, rangecoder.rs:20-30. Returns 1 on UnexpectedEof. */
static int rc_new(rc_t *rc, reader_t *stream) {
  uint8_t ignored;
  rc->stream = stream;
  rc->range = 0xFFFFFFFFu;
  rc->code = 0;
  if (rd_u8(stream, &ignored)) return 1; /* first byte is read and ignored */
  if (rd_u32be(stream, &rc->code)) return 1;
  return 0;
}

/* RangeDecoder::is_finished_ok, rangecoder.rs:49-52 */
static int rc_is_finished_ok(const rc_t *rc) { return rc->code == 0 && rd_is_eof(rc->stream); }

/* RangeDecoder::normalize, rangecoder.rs:59-69 (one byte at most). */
static inline int rc_normalize(rc_t *rc) {
  if (rc->range < 0x01000000u) {
    uint8_t b;
    rc->range <<= 8;
    if (rd_u8(rc->stream, &b)) return 1;
    rc->code = (rc->code << 8) ^ (uint32_t)b;
  }
  return 0;
}

/* RangeDecoder::get_bit, rangecoder.rs:71-82. bit<0 => UnexpectedEof */
static inline int rc_get_bit(rc_t *rc) {
  int bit;
  rc->range >>= 1;
  bit = rc->code >= rc->range;
  if (bit) rc->code -= rc->range;
  if (rc_normalize(rc)) return -1;
  return bit;
}

/* RangeDecoder::get, rangecoder.rs:84-90 */
static int rc_get(rc_t *rc, unsigned count, uint32_t *result) {
  uint32_t r = 0;
  unsigned i;
  for (i = 0; i < count; i++) {
    int bit = rc_get_bit(rc);
    if (bit < 0) return 1;
    r = (r << 1) ^ (uint32_t)bit;
  }
  *result = r;
  return 0;
}

/* RangeDecoder::decode_bit, rangecoder.rs:92-120 (update = true). <0 => EOF */
static inline int rc_decode_bit(rc_t *rc, uint16_t *prob) {
  uint32_t bound = (rc->range >> 11) * (uint32_t)*prob;
  if (rc->code < bound) {
    *prob = (uint16_t)(*prob + ((0x800u - *prob) >> 5));
    rc->range = bound;
    if (rc_normalize(rc)) return -1;
    return 0;
  } else {
    *prob = (uint16_t)(*prob - (*prob >> 5));
    rc->code -= bound;
    rc->range -= bound;
    if (rc_normalize(rc)) return -1;
    return 1;
  }
}

/* RangeDecoder::parse_bit_tree, rangecoder.rs:122-134 */
static int rc_parse_bit_tree(rc_t *rc, unsigned num_bits, uint16_t *probs, uint32_t *out) {
  uint32_t tmp = 1;
  unsigned i;
  for (i = 0; i < num_bits; i++) {
    int bit = rc_decode_bit(rc, &probs[tmp]);
    if (bit < 0) return 1;
    tmp = (tmp << 1) ^ (uint32_t)bit;
  }
  *out = tmp - (1u << num_bits);
  return 0;
}

/* RangeDecoder::parse_reverse_bit_tree, rangecoder.rs:136-151 */
static int rc_parse_reverse_bit_tree(rc_t *rc, unsigned num_bits, uint16_t *probs, size_t offset,
                                     uint32_t *out) {
  uint32_t result = 0;
  size_t tmp = 1;
  unsigned i;
  for (i = 0; i < num_bits; i++) {
    int bit = rc_decode_bit(rc, &probs[offset + tmp]);
    if (bit < 0) return 1;
    tmp = (tmp << 1) ^ (size_t)bit;
    result ^= (uint32_t)bit << i;
  }
  *out = result;
  return 0;
}

/* LenDecoder, rangecoder.rs:202-270 */
typedef struct {
  uint16_t choice, choice2;
  uint16_t low_coder[16][8];
  uint16_t mid_coder[16][8];
  uint16_t high_coder[256];
} lendec_t;

static void fill16(uint16_t *p, size_t n) {
  size_t i;
  for (i = 0; i < n; i++) p[i] = 0x400;
}

static void lendec_new(lendec_t *l) { fill16((uint16_t *)l, sizeof *l / 2); }

/* LenDecoder::decode, rangecoder.rs:256-269 */
static int lendec_decode(lendec_t *l, rc_t *rc, size_t pos_state, size_t *out) {
  uint32_t v;
  int bit = rc_decode_bit(rc, &l->choice);
  if (bit < 0) return 1;
  if (!bit) {
    if (rc_parse_bit_tree(rc, 3, l->low_coder[pos_state], &v)) return 1;
    *out = v;
    return 0;
  }
  bit = rc_decode_bit(rc, &l->choice2);
  if (bit < 0) return 1;
  if (!bit) {
    if (rc_parse_bit_tree(rc, 3, l->mid_coder[pos_state], &v)) return 1;
    *out = (size_t)v + 8;
    return 0;
  }
  if (rc_parse_bit_tree(rc, 8, l->high_coder, &v)) return 1;
  *out = (size_t)v + 16;
  return 0;
}

/* ------------------------------------------------------------------ */
/* DecoderState, src/decode/lzma.rs:164-593                            */
/* ------------------------------------------------------------------ */

typedef struct {
  uint32_t lc, lp, pb;
} props_t;

typedef struct {
  props_t lzma_props;
  int unpacked_is_some;
  uint64_t unpacked_size;
  uint16_t *literal_probs; /* Vec2D<u16>: (1 << (lc+lp)) rows x 0x300 */
  size_t literal_rows;
  uint16_t pos_slot_decoder[4][64];
  uint16_t align_decoder[16];
  uint16_t pos_decoders[115];
  uint16_t is_match[192];
  uint16_t is_rep[12];
  uint16_t is_rep_g0[12];
  uint16_t is_rep_g1[12];
  uint16_t is_rep_g2[12];
  uint16_t is_rep_0long[192];
  size_t state;
  uint64_t rep[4];
  lendec_t len_decoder;
  lendec_t rep_len_decoder;
} dstate_t;

static void dstate_fill_small(dstate_t *s) {
  fill16(&s->pos_slot_decoder[0][0], 4 * 64);
  fill16(s->align_decoder, 16);
  fill16(s->pos_decoders, 115);
  fill16(s->is_match, 192);
  fill16(s->is_rep, 12);
  fill16(s->is_rep_g0, 12);
  fill16(s->is_rep_g1, 12);
  fill16(s->is_rep_g2, 12);
  fill16(s->is_rep_0long, 192);
  s->state = 0;
  s->rep[0] = s->rep[1] = s->rep[2] = s->rep[3] = 0;
  lendec_new(&s->len_decoder);
  lendec_new(&s->rep_len_decoder);
}

/* DecoderState::new, lzma.rs:188-214 */
static void dstate_new(dstate_t *s, props_t props, int unpacked_is_some, uint64_t unpacked_size) {
  memset(s, 0, sizeof *s);
  s->lzma_props = props;
  s->unpacked_is_some = unpacked_is_some;
  s->unpacked_size = unpacked_size;
  s->literal_rows = (size_t)1 << (props.lc + props.lp);
  s->literal_probs = (uint16_t *)malloc(s->literal_rows * 0x300 * sizeof(uint16_t));
  fill16(s->literal_probs, s->literal_rows * 0x300);
  dstate_fill_small(s);
}

/* DecoderState::reset_state, lzma.rs:216-249 */
static void dstate_reset_state(dstate_t *s, props_t new_props) {
  if (s->lzma_props.lc + s->lzma_props.lp != new_props.lc + new_props.lp) {
    free(s->literal_probs);
    s->literal_rows = (size_t)1 << (new_props.lc + new_props.lp);
    s->literal_probs = (uint16_t *)malloc(s->literal_rows * 0x300 * sizeof(uint16_t));
  }
  fill16(s->literal_probs, s->literal_rows * 0x300);
  s->lzma_props = new_props;
  dstate_fill_small(s);
}

static void dstate_free(dstate_t *s) {
  free(s->literal_probs);
  s->literal_probs = NULL;
}

/* DecoderState::decode_literal, lzma.rs:526-561 */
static int dstate_decode_literal(dstate_t *s, lzbuf_t *output, rc_t *rc, uint8_t *byte, err_t *e) {
  size_t prev_byte = lzbuf_last_or(output, 0);
  size_t result = 1;
  size_t lit_state =
      ((size_t)(output->len & (((uint64_t)1 << s->lzma_props.lp) - 1)) << s->lzma_props.lc) +
      (prev_byte >> (8 - s->lzma_props.lc));
  uint16_t *probs = s->literal_probs + lit_state * 0x300;

  if (s->state >= 7) {
    uint8_t mb = 0;
    size_t match_byte;
    if (lzbuf_last_n(output, s->rep[0] + 1, &mb, e)) return e->kind;
    match_byte = mb;
    while (result < 0x100) {
      size_t match_bit = (match_byte >> 7) & 1;
      int bit;
      match_byte <<= 1;
      bit = rc_decode_bit(rc, &probs[((1 + match_bit) << 8) + result]);
      if (bit < 0) return io_eof(e);
      result = (result << 1) ^ (size_t)bit;
      if (match_bit != (size_t)bit) break;
    }
  }
  while (result < 0x100) {
    int bit = rc_decode_bit(rc, &probs[result]);
    if (bit < 0) return io_eof(e);
    result = (result << 1) ^ (size_t)bit;
  }
  *byte = (uint8_t)(result - 0x100);
  return 0;
}

/* DecoderState::decode_distance, lzma.rs:563-592 */
static int dstate_decode_distance(dstate_t *s, rc_t *rc, size_t length, uint64_t *out, err_t *e) {
  size_t len_state = length > 3 ? 3 : length;
  uint32_t v;
  size_t pos_slot, num_direct_bits;
  uint64_t result;

  if (rc_parse_bit_tree(rc, 6, s->pos_slot_decoder[len_state], &v)) return io_eof(e);
  pos_slot = v;
  if (pos_slot < 4) {
    *out = pos_slot;
    return 0;
  }
  num_direct_bits = (pos_slot >> 1) - 1;
  result = (uint64_t)(2 ^ (pos_slot & 1)) << num_direct_bits;
  if (pos_slot < 14) {
    if (rc_parse_reverse_bit_tree(rc, (unsigned)num_direct_bits, s->pos_decoders,
                                  (size_t)(result - pos_slot), &v))
      return io_eof(e);
    result += v;
  } else {
    if (rc_get(rc, (unsigned)(num_direct_bits - 4), &v)) return io_eof(e);
    result += (uint64_t)v << 4;
    if (rc_parse_reverse_bit_tree(rc, 4, s->align_decoder, 0, &v)) return io_eof(e);
    result += v;
  }
  *out = result;
  return 0;
}

enum { ST_CONTINUE = 0, ST_FINISHED = 1, ST_ERROR = 2 };

/* DecoderState::process_next_inner, lzma.rs:278-393 (update = true) */
static int dstate_process_next(dstate_t *s, lzbuf_t *output, rc_t *rc, err_t *e) {
  size_t pos_state = (size_t)(output->len & (((uint64_t)1 << s->lzma_props.pb) - 1));
  size_t len;
  int bit;

  bit = rc_decode_bit(rc, &s->is_match[(s->state << 4) + pos_state]);
  if (bit < 0) return io_eof(e), ST_ERROR;
  if (!bit) { /* literal */
    uint8_t byte = 0;
    if (dstate_decode_literal(s, output, rc, &byte, e)) return ST_ERROR;
    if (lzbuf_append_literal(output, byte, e)) return ST_ERROR;
    s->state = s->state < 4 ? 0 : (s->state < 10 ? s->state - 3 : s->state - 6);
    return ST_CONTINUE;
  }

  bit = rc_decode_bit(rc, &s->is_rep[s->state]);
  if (bit < 0) return io_eof(e), ST_ERROR;
  if (bit) { /* distance repeated from LRU */
    bit = rc_decode_bit(rc, &s->is_rep_g0[s->state]);
    if (bit < 0) return io_eof(e), ST_ERROR;
    if (!bit) {
      bit = rc_decode_bit(rc, &s->is_rep_0long[(s->state << 4) + pos_state]);
      if (bit < 0) return io_eof(e), ST_ERROR;
      if (!bit) { /* short rep: len 1 */
        s->state = s->state < 7 ? 9 : 11;
        if (lzbuf_append_lz(output, 1, s->rep[0] + 1, e)) return ST_ERROR;
        return ST_CONTINUE;
      }
    } else {
      size_t idx, i;
      uint64_t dist;
      bit = rc_decode_bit(rc, &s->is_rep_g1[s->state]);
      if (bit < 0) return io_eof(e), ST_ERROR;
      if (!bit) {
        idx = 1;
      } else {
        bit = rc_decode_bit(rc, &s->is_rep_g2[s->state]);
        if (bit < 0) return io_eof(e), ST_ERROR;
        idx = bit ? 3 : 2;
      }
      dist = s->rep[idx];
      for (i = idx; i > 0; i--) s->rep[i] = s->rep[i - 1];
      s->rep[0] = dist;
    }
    if (lendec_decode(&s->rep_len_decoder, rc, pos_state, &len)) return io_eof(e), ST_ERROR;
    s->state = s->state < 7 ? 8 : 11;
  } else { /* new distance */
    uint64_t rep_0 = 0;
    s->rep[3] = s->rep[2];
    s->rep[2] = s->rep[1];
    s->rep[1] = s->rep[0];
    if (lendec_decode(&s->len_decoder, rc, pos_state, &len)) return io_eof(e), ST_ERROR;
    s->state = s->state < 7 ? 7 : 10;
    if (dstate_decode_distance(s, rc, len, &rep_0, e)) return ST_ERROR;
    s->rep[0] = rep_0;
    if (s->rep[0] == 0xFFFFFFFFull) {
      if (rc_is_finished_ok(rc)) return ST_FINISHED;
      fail(e, ORC_LZMA_ERROR, "Found end-of-stream marker but more bytes are available");
      return ST_ERROR;
    }
  }
  len += 2;
  if (lzbuf_append_lz(output, len, s->rep[0] + 1, e)) return ST_ERROR;
  return ST_CONTINUE;
}

/* DecoderState::process -> process_mode(Finish), lzma.rs:255-261, 435-524 */
static int dstate_process(dstate_t *s, lzbuf_t *output, rc_t *rc, err_t *e) {
  for (;;) {
    int st;
    if (s->unpacked_is_some) {
      if (output->len >= s->unpacked_size) break;
    } else if (rc_is_finished_ok(rc)) {
      break;
    }
    st = dstate_process_next(s, output, rc, e);
    if (st == ST_ERROR) return e->kind;
    if (st == ST_FINISHED) break;
  }
  if (s->unpacked_is_some && s->unpacked_size != output->len)
    return fail(e, ORC_LZMA_ERROR, "Expected unpacked size of %llu but decompressed to %llu",
                (unsigned long long)s->unpacked_size, (unsigned long long)output->len);
  return 0;
}

/* ------------------------------------------------------------------ */
/* .lzma header + LzmaDecoder shell, src/decode/lzma.rs:96-161,597-648 */
/* ------------------------------------------------------------------ */

typedef struct {
  props_t properties;
  uint32_t dict_size;
  int unpacked_is_some;
  uint64_t unpacked_size;
} params_t;

/* LzmaParams::read_header, lzma.rs:96-161 */
static int read_header(reader_t *in, const orc_options *opt, params_t *out, err_t *e) {
  uint8_t props;
  uint32_t pb, lc, lp, dict;
  uint64_t u;
  if (rd_u8(in, &props)) return fail(e, ORC_HEADER_TOO_SHORT, "%s", EOF_MSG);
  pb = props;
  if (pb >= 225)
    return fail(e, ORC_LZMA_ERROR, "LZMA header invalid properties: %u must be < 225", pb);
  lc = pb % 9;
  pb /= 9;
  lp = pb % 5;
  pb /= 5;
  if (rd_u32le(in, &dict)) return fail(e, ORC_HEADER_TOO_SHORT, "%s", EOF_MSG);
  if (dict < 0x1000) dict = 0x1000;
  switch (opt->unpacked_size_mode) {
  case ORC_READ_FROM_HEADER:
    if (rd_u64le(in, &u)) return fail(e, ORC_HEADER_TOO_SHORT, "%s", EOF_MSG);
    out->unpacked_is_some = (u != 0xFFFFFFFFFFFFFFFFull);
    out->unpacked_size = u;
    break;
  case ORC_READ_HEADER_BUT_USE_PROVIDED:
    if (rd_u64le(in, &u)) return fail(e, ORC_HEADER_TOO_SHORT, "%s", EOF_MSG);
    out->unpacked_is_some = opt->provided_is_some;
    out->unpacked_size = opt->provided;
    break;
  default: /* UseProvided */
    out->unpacked_is_some = opt->provided_is_some;
    out->unpacked_size = opt->provided;
    break;
  }
  out->properties.lc = lc;
  out->properties.lp = lp;
  out->properties.pb = pb;
  out->dict_size = dict;
  return 0;
}

/* LzmaDecoder::new + decompress, lzma.rs:607-613,635-648 */
static int lzma_decoder_decompress(const params_t *params, int memlimit_is_some, uint64_t memlimit,
                                   reader_t *in, sink_t *out, err_t *e) {
  dstate_t st;
  lzbuf_t ring;
  rc_t rc;
  int r;
  dstate_new(&st, params->properties, params->unpacked_is_some, params->unpacked_size);
  memset(&ring, 0, sizeof ring);
  ring.is_accum = 0;
  ring.stream = out;
  ring.dict_size = params->dict_size;
  ring.memlimit = memlimit_is_some ? memlimit : UINT64_MAX;
  if (rc_new(&rc, in)) {
    dstate_free(&st);
    return fail(e, ORC_LZMA_ERROR, "LZMA stream too short: %s", EOF_MSG);
  }
  r = dstate_process(&st, &ring, &rc, e);
  if (!r) lzbuf_finish(&ring); /* on error the tail is dropped, flushed rings stay */
  free(ring.buf);
  dstate_free(&st);
  return r;
}

/* ------------------------------------------------------------------ */
/* LZMA2, src/decode/lzma2.rs                                          */
/* ------------------------------------------------------------------ */

/* Lzma2Decoder::parse_uncompressed, lzma2.rs:195-229 */
static int lzma2_parse_uncompressed(lzbuf_t *accum, reader_t *in, int reset_dict, err_t *e) {
  uint32_t us;
  size_t unpacked_size;
  if (rd_u16be(in, &us))
    return fail(e, ORC_LZMA_ERROR, "LZMA2 expected unpacked size: %s", EOF_MSG);
  unpacked_size = (size_t)us + 1;
  if (reset_dict) accum_reset(accum);
  if (in->end - in->pos < unpacked_size) {
    in->pos = in->end;
    return fail(e, ORC_LZMA_ERROR, "LZMA2 expected %llu uncompressed bytes: %s",
                (unsigned long long)unpacked_size, EOF_MSG);
  }
  accum_append_bytes(accum, in->p + in->pos, unpacked_size);
  in->pos += unpacked_size;
  return 0;
}

/* Lzma2Decoder::parse_lzma, lzma2.rs:84-193 */
static int lzma2_parse_lzma(dstate_t *st, lzbuf_t *accum, reader_t *in, uint8_t status, err_t *e) {
  int reset_dict, reset_state, reset_props;
  uint32_t v;
  uint64_t unpacked_size, packed_size;
  size_t saved_end;
  rc_t rc;
  int r;

  if ((status & 0x80) == 0)
    return fail(e, ORC_LZMA_ERROR, "LZMA2 invalid status %u, must be 0, 1, 2 or >= 128",
                (unsigned)status);
  switch ((status >> 5) & 0x3) {
  case 0: reset_dict = 0; reset_state = 0; reset_props = 0; break;
  case 1: reset_dict = 0; reset_state = 1; reset_props = 0; break;
  case 2: reset_dict = 0; reset_state = 1; reset_props = 1; break;
  default: reset_dict = 1; reset_state = 1; reset_props = 1; break;
  }
  if (rd_u16be(in, &v))
    return fail(e, ORC_LZMA_ERROR, "LZMA2 expected unpacked size: %s", EOF_MSG);
  unpacked_size = ((((uint64_t)(status & 0x1F)) << 16) | (uint64_t)v) + 1;
  if (rd_u16be(in, &v)) return fail(e, ORC_LZMA_ERROR, "LZMA2 expected packed size: %s", EOF_MSG);
  packed_size = (uint64_t)v + 1;

  if (reset_dict) accum_reset(accum);

  if (reset_state) {
    props_t new_props;
    if (reset_props) {
      uint8_t props;
      uint32_t pb, lc, lp;
      if (rd_u8(in, &props))
        return fail(e, ORC_LZMA_ERROR, "LZMA2 expected new properties: %s", EOF_MSG);
      pb = props;
      if (pb >= 225)
        return fail(e, ORC_LZMA_ERROR, "LZMA2 invalid properties: %u must be < 225", pb);
      lc = pb % 9;
      pb /= 9;
      lp = pb % 5;
      pb /= 5;
      if (lc + lp > 4)
        return fail(e, ORC_LZMA_ERROR,
                    "LZMA2 invalid properties: lc + lp (%u + %u) must be <= 4", lc, lp);
      new_props.lc = lc;
      new_props.lp = lp;
      new_props.pb = pb;
    } else {
      new_props = st->lzma_props;
    }
    dstate_reset_state(st, new_props);
  }

  st->unpacked_is_some = 1;
  st->unpacked_size = unpacked_size + accum->len;

  /* input.take(packed_size): a window on the reader; whatever the range
   * decoder leaves unread inside it is NOT skipped afterwards (lzma2.rs:189-192). */
  saved_end = in->end;
  if ((uint64_t)(in->end - in->pos) > packed_size) in->end = in->pos + (size_t)packed_size;
  if (rc_new(&rc, in)) {
    in->end = saved_end;
    return fail(e, ORC_LZMA_ERROR, "LZMA input too short: %s", EOF_MSG);
  }
  r = dstate_process(st, accum, &rc, e);
  in->end = saved_end;
  return r;
}

/* Lzma2Decoder::new + decompress, lzma2.rs:23-34,52-82 */
static int lzma2_decoder_decompress(reader_t *in, sink_t *out, err_t *e) {
  dstate_t st;
  lzbuf_t accum;
  props_t p0 = {0, 0, 0};
  int r = 0;
  dstate_new(&st, p0, 0, 0);
  memset(&accum, 0, sizeof accum);
  accum.is_accum = 1;
  accum.stream = out;
  accum.memlimit = UINT64_MAX;
  for (;;) {
    uint8_t status;
    if (rd_u8(in, &status)) {
      r = fail(e, ORC_LZMA_ERROR, "LZMA2 expected new status: %s", EOF_MSG);
      break;
    }
    if (status == 0) break;
    if (status == 1)
      r = lzma2_parse_uncompressed(&accum, in, 1, e);
    else if (status == 2)
      r = lzma2_parse_uncompressed(&accum, in, 0, e);
    else
      r = lzma2_parse_lzma(&st, &accum, in, status, e);
    if (r) break;
  }
  if (!r) lzbuf_finish(&accum);
  free(accum.buf);
  dstate_free(&st);
  return r;
}

/* ------------------------------------------------------------------ */
/* CRC-32 (ISO-HDLC) and CRC-64/XZ, src/xz/crc.rs                       */
/* ------------------------------------------------------------------ */

static uint32_t crc32_tab[256];
static uint64_t crc64_tab[256];
static pthread_once_t crc_once = PTHREAD_ONCE_INIT;

static void crc_init(void) {
  uint32_t i, j;
  for (i = 0; i < 256; i++) {
    uint32_t c = i;
    uint64_t d = i;
    for (j = 0; j < 8; j++) {
      c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
      d = (d & 1) ? (d >> 1) ^ 0xC96C5795D7870F42ull : d >> 1;
    }
    crc32_tab[i] = c;
    crc64_tab[i] = d;
  }
}

static uint32_t crc32_update(uint32_t c, const uint8_t *p, size_t n) {
  size_t i;
  pthread_once(&crc_once, crc_init);
  for (i = 0; i < n; i++) c = crc32_tab[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c;
}

uint32_t orc_crc32(const uint8_t *p, size_t n) { return ~crc32_update(0xFFFFFFFFu, p, n); }

uint64_t orc_crc64(const uint8_t *p, size_t n) {
  uint64_t c = ~(uint64_t)0;
  size_t i;
  pthread_once(&crc_once, crc_init);
  for (i = 0; i < n; i++) c = crc64_tab[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return ~c;
}

/* ------------------------------------------------------------------ */
/* XZ container, src/decode/xz.rs + src/xz/{mod,header,footer}.rs       */
/* ------------------------------------------------------------------ */

enum { CHECK_NONE = 0x00, CHECK_CRC32 = 0x01, CHECK_CRC64 = 0x04, CHECK_SHA256 = 0x0A };

static const char *check_name(int m) {
  switch (m) {
  case CHECK_NONE: return "None";
  case CHECK_CRC32: return "Crc32";
  case CHECK_CRC64: return "Crc64";
  default: return "Sha256";
  }
}

/* StreamFlags::parse, src/xz/mod.rs:15-31 + CheckMethod::try_from :54-66 */
static int stream_flags_parse(uint32_t field, int *check_method, err_t *e) {
  uint32_t b0 = (field >> 8) & 0xFF, b1 = field & 0xFF;
  if (b0 != 0x00) return fail(e, ORC_XZ_ERROR, "Invalid null byte in Stream Flags: %x", b0);
  if (b1 != CHECK_NONE && b1 != CHECK_CRC32 && b1 != CHECK_CRC64 && b1 != CHECK_SHA256)
    return fail(e, ORC_XZ_ERROR,
                "Invalid check method %x, expected one of [0x00, 0x01, 0x04, 0x0A]", b1);
  *check_method = (int)b1;
  return 0;
}

/* get_multibyte, src/decode/xz.rs:448-464. 0 ok, 1 eof, 2 invalid */
static int get_multibyte(reader_t *in, uint64_t *out) {
  uint64_t result = 0;
  int i;
  for (i = 0; i < 9; i++) {
    uint8_t byte = 0;
    if (rd_u8(in, &byte)) return 1;
    result ^= ((uint64_t)(byte & 0x7F)) << (i * 7);
    if ((byte & 0x80) == 0) {
      *out = result;
      return 0;
    }
  }
  return 2;
}

static int multibyte_err(int rc, err_t *e) {
  if (rc == 1) return io_eof(e);
  return fail(e, ORC_XZ_ERROR, "Invalid multi-byte encoding");
}

typedef struct {
  uint64_t unpadded_size, unpacked_size;
} record_t;

typedef struct {
  size_t num_filters;
  size_t props_len[4];
  int has_packed, has_unpacked;
  uint64_t packed_size, unpacked_size;
} block_header_t;

/* read_block_header, src/decode/xz.rs:356-446. `in` is limited to the header bytes. */
static int read_block_header(reader_t *in, uint64_t header_size, block_header_t *bh, err_t *e) {
  uint8_t flags;
  size_t num_filters, i;
  int rc;
  if (rd_u8(in, &flags)) return io_eof(e);
  num_filters = (size_t)(flags & 0x03) + 1;
  if ((flags & 0x3C) != 0)
    return fail(e, ORC_XZ_ERROR,
                "Invalid block flags %u, reserved bits (mask 0x3C) must be zero", (unsigned)flags);
  bh->has_packed = (flags & 0x40) != 0;
  bh->has_unpacked = (flags & 0x80) != 0;
  if (bh->has_packed && (rc = get_multibyte(in, &bh->packed_size))) return multibyte_err(rc, e);
  if (bh->has_unpacked && (rc = get_multibyte(in, &bh->unpacked_size)))
    return multibyte_err(rc, e);
  bh->num_filters = 0;
  for (i = 0; i < num_filters; i++) {
    uint64_t id, size_of_properties;
    if ((rc = get_multibyte(in, &id))) return multibyte_err(rc, e);
    if (id != 0x21)
      return fail(e, ORC_XZ_ERROR, "Unknown filter id %llu", (unsigned long long)id);
    if ((rc = get_multibyte(in, &size_of_properties))) return multibyte_err(rc, e);
    if (size_of_properties > header_size)
      return fail(e, ORC_XZ_ERROR,
                  "Size of filter properties exceeds block header size (%llu > %llu)",
                  (unsigned long long)size_of_properties, (unsigned long long)header_size);
    if (rd_exact(in, NULL, (size_t)size_of_properties))
      return fail(e, ORC_XZ_ERROR, "Could not read filter properties of size %llu: %s",
                  (unsigned long long)size_of_properties, EOF_MSG);
    bh->props_len[bh->num_filters++] = (size_t)size_of_properties;
  }
  /* util::flush_zero_padding, src/decode/util.rs:14-36 */
  while (in->pos < in->end) {
    if (in->p[in->pos] != 0)
      return fail(e, ORC_XZ_ERROR, "Invalid block header padding, must be null bytes");
    in->pos++;
  }
  return 0;
}

/* decode_filter, src/decode/xz.rs:335-354: returns bytes consumed via *consumed */
static int decode_filter(reader_t *in, sink_t *out, size_t props_len, size_t *consumed, err_t *e) {
  size_t start = in->pos;
  int r;
  if (props_len != 1) return fail(e, ORC_XZ_ERROR, "Invalid properties for filter Lzma2");
  r = lzma2_decoder_decompress(in, out, e);
  *consumed = in->pos - start;
  return r;
}

/* read_block, src/decode/xz.rs:196-290. block_start = position of the header_size byte. */
static int read_block(reader_t *in, size_t block_start, sink_t *output, int check_method,
                      record_t **records, size_t *nrecords, uint8_t header_size_byte, err_t *e) {
  uint64_t header_size = ((uint64_t)header_size_byte << 2) - 1;
  block_header_t bh = {0, {0, 0, 0, 0}, 0, 0, 0, 0};
  uint32_t crc, digest;
  size_t hdr_begin = in->pos, hdr_end, saved_end = in->end;
  sink_t tmpbuf = {0, 0, 0};
  size_t i, count, padding_size, unpacked_size;
  int r;

  /* header bytes go through take(header_size) + BufReader + CrcDigestRead */
  hdr_end = (uint64_t)(in->end - in->pos) > header_size ? in->pos + (size_t)header_size : in->end;
  in->end = hdr_end;
  r = read_block_header(in, header_size, &bh, e);
  in->end = saved_end;
  if (r) return r;
  in->pos = hdr_end; /* the BufReader pulled the whole window */
  digest = crc32_update(0xFFFFFFFFu, &header_size_byte, 1);
  digest = ~crc32_update(digest, in->p + hdr_begin, hdr_end - hdr_begin);

  if (rd_u32le(in, &crc)) return io_eof(e);
  if (crc != digest)
    return fail(e, ORC_XZ_ERROR, "Invalid header CRC32: expected 0x%08x but got 0x%08x", crc,
                digest);

  for (i = 0; i < bh.num_filters; i++) {
    if (i == 0) {
      size_t packed = 0;
      r = decode_filter(in, &tmpbuf, bh.props_len[0], &packed, e);
      if (r) goto done;
      if (bh.has_packed && (uint64_t)packed != bh.packed_size) {
        r = fail(e, ORC_XZ_ERROR, "Invalid compressed size: expected %llu but got %llu",
                 (unsigned long long)bh.packed_size, (unsigned long long)packed);
        goto done;
      }
    } else {
      sink_t newbuf = {0, 0, 0};
      reader_t sub;
      size_t packed = 0;
      sub.p = tmpbuf.data;
      sub.pos = 0;
      sub.end = tmpbuf.len;
      r = decode_filter(&sub, &newbuf, bh.props_len[i], &packed, e);
      free(tmpbuf.data);
      tmpbuf = newbuf;
      if (r) goto done;
    }
  }

  unpacked_size = tmpbuf.len;
  if (bh.has_unpacked && (uint64_t)unpacked_size != bh.unpacked_size) {
    r = fail(e, ORC_XZ_ERROR, "Invalid decompressed size: expected %llu but got %llu",
             (unsigned long long)bh.unpacked_size, (unsigned long long)unpacked_size);
    goto done;
  }

  count = in->pos - block_start;
  padding_size = ((count ^ 0x03) + 1) & 0x03;
  for (i = 0; i < padding_size; i++) {
    uint8_t byte = 0;
    if (rd_u8(in, &byte)) {
      r = io_eof(e);
      goto done;
    }
    if (byte != 0) {
      r = fail(e, ORC_XZ_ERROR, "Invalid block padding, must be null bytes");
      goto done;
    }
  }

  /* validate_block_check, src/decode/xz.rs:292-333 */
  switch (check_method) {
  case CHECK_NONE: break;
  case CHECK_CRC32: {
    uint32_t c, d;
    if (rd_u32le(in, &c)) {
      r = io_eof(e);
      goto done;
    }
    d = orc_crc32(tmpbuf.data, tmpbuf.len);
    if (c != d) {
      r = fail(e, ORC_XZ_ERROR, "Invalid block CRC32, expected 0x%08x but got 0x%08x", c, d);
      goto done;
    }
    break;
  }
  case CHECK_CRC64: {
    uint64_t c, d;
    if (rd_u64le(in, &c)) {
      r = io_eof(e);
      goto done;
    }
    d = orc_crc64(tmpbuf.data, tmpbuf.len);
    if (c != d) {
      r = fail(e, ORC_XZ_ERROR, "Invalid block CRC64, expected 0x%016llx but got 0x%016llx",
               (unsigned long long)c, (unsigned long long)d);
      goto done;
    }
    break;
  }
  default:
    r = fail(e, ORC_XZ_ERROR, "Unsupported SHA-256 checksum (not yet implemented)");
    goto done;
  }

  sink_write_all(output, tmpbuf.data, tmpbuf.len);
  *records = (record_t *)realloc(*records, (*nrecords + 1) * sizeof(record_t));
  (*records)[*nrecords].unpadded_size = (uint64_t)(in->pos - block_start - padding_size);
  (*records)[*nrecords].unpacked_size = (uint64_t)unpacked_size;
  *nrecords += 1;
done:
  free(tmpbuf.data);
  return r;
}

/* check_index, src/decode/xz.rs:96-171. index_start = position of the 0x00 indicator. */
static int check_index(reader_t *in, size_t index_start, const record_t *records, size_t nrecords,
                       err_t *e) {
  uint64_t num_records, v;
  size_t i, count, padding_size, digest_from = in->pos;
  uint32_t digest, crc;
  uint8_t tag = 0;
  int rc;
  if ((rc = get_multibyte(in, &num_records))) return multibyte_err(rc, e);
  if (num_records != (uint64_t)nrecords)
    return fail(e, ORC_XZ_ERROR, "Expected %llu records but got %llu records",
                (unsigned long long)num_records, (unsigned long long)nrecords);
  for (i = 0; i < nrecords; i++) {
    if ((rc = get_multibyte(in, &v))) return multibyte_err(rc, e);
    if (v != records[i].unpadded_size)
      return fail(e, ORC_XZ_ERROR,
                  "Invalid index for record %llu: unpadded size (%llu) does not match index (%llu)",
                  (unsigned long long)i, (unsigned long long)records[i].unpadded_size,
                  (unsigned long long)v);
    if ((rc = get_multibyte(in, &v))) return multibyte_err(rc, e);
    if (v != records[i].unpacked_size)
      return fail(e, ORC_XZ_ERROR,
                  "Invalid index for record %llu: unpacked size (%llu) does not match index (%llu)",
                  (unsigned long long)i, (unsigned long long)records[i].unpacked_size,
                  (unsigned long long)v);
  }
  count = in->pos - index_start;
  padding_size = ((count ^ 0x03) + 1) & 0x03;
  for (i = 0; i < padding_size; i++) {
    uint8_t byte = 0;
    if (rd_u8(in, &byte)) return io_eof(e);
    if (byte != 0) return fail(e, ORC_XZ_ERROR, "Invalid index padding, must be null bytes");
  }
  digest = crc32_update(0xFFFFFFFFu, &tag, 1);
  digest = ~crc32_update(digest, in->p + digest_from, in->pos - digest_from);
  if (rd_u32le(in, &crc)) return io_eof(e);
  if (crc != digest)
    return fail(e, ORC_XZ_ERROR, "Invalid index CRC32: expected 0x%08x but got 0x%08x", crc,
                digest);
  return 0;
}

/* xz::decode_stream, src/decode/xz.rs:18-94 (+ StreamHeader::parse, src/xz/header.rs:20-51) */
static int xz_decode_stream(reader_t *in, sink_t *output, err_t *e) {
  static const uint8_t XZ_MAGIC[6] = {0xFD, 0x37, 0x7A, 0x58, 0x5A, 0x00};
  uint8_t tag[6];
  uint32_t flags_field, crc, digest, backward_size;
  int check_method = 0, footer_check = 0;
  record_t *records = NULL;
  size_t nrecords = 0, index_size = 0;
  int r = 0;

  /* util::read_tag, src/decode/util.rs:3-7 */
  if (rd_exact(in, tag, 6)) return io_eof(e);
  if (memcmp(tag, XZ_MAGIC, 6) != 0)
    return fail(e, ORC_XZ_ERROR, "Invalid XZ magic, expected [253, 55, 122, 88, 90, 0]");
  {
    size_t from = in->pos;
    if (rd_u16be(in, &flags_field)) return io_eof(e);
    digest = orc_crc32(in->p + from, 2);
  }
  if (rd_u32le(in, &crc)) return io_eof(e);
  if (crc != digest)
    return fail(e, ORC_XZ_ERROR, "Invalid header CRC32: expected 0x%08x but got 0x%08x", crc,
                digest);
  if ((r = stream_flags_parse(flags_field, &check_method, e))) return r;

  for (;;) {
    size_t start = in->pos;
    uint8_t header_size;
    if (rd_u8(in, &header_size)) {
      r = io_eof(e);
      goto done;
    }
    if (header_size == 0) {
      r = check_index(in, start, records, nrecords, e);
      if (r) goto done;
      index_size = in->pos - start;
      break;
    }
    r = read_block(in, start, output, check_method, &records, &nrecords, header_size, e);
    if (r) goto done;
  }

  if (rd_u32le(in, &crc)) {
    r = io_eof(e);
    goto done;
  }
  {
    size_t from = in->pos;
    if (rd_u32le(in, &backward_size)) {
      r = io_eof(e);
      goto done;
    }
    if ((uint32_t)index_size != (uint32_t)((backward_size + 1u) << 2)) {
      r = fail(e, ORC_XZ_ERROR, "Invalid index size: expected %u but got %llu",
               (uint32_t)((backward_size + 1u) << 2), (unsigned long long)index_size);
      goto done;
    }
    if (rd_u16be(in, &flags_field)) {
      r = io_eof(e);
      goto done;
    }
    if ((r = stream_flags_parse(flags_field, &footer_check, e))) goto done;
    if (footer_check != check_method) {
      r = fail(e, ORC_XZ_ERROR,
               "Flags in header (StreamFlags { check_method: %s }) does not match footer "
               "(StreamFlags { check_method: %s })",
               check_name(check_method), check_name(footer_check));
      goto done;
    }
    digest = orc_crc32(in->p + from, in->pos - from);
  }
  if (crc != digest) {
    r = fail(e, ORC_XZ_ERROR, "Invalid footer CRC32: expected 0x%08x but got 0x%08x", crc, digest);
    goto done;
  }
  if (rd_exact(in, tag, 2)) {
    r = io_eof(e);
    goto done;
  }
  if (tag[0] != 0x59 || tag[1] != 0x5A) {
    r = fail(e, ORC_XZ_ERROR, "Invalid footer magic, expected [89, 90]");
    goto done;
  }
  if (!rd_is_eof(in)) {
    r = fail(e, ORC_XZ_ERROR, "Unexpected data after last XZ block");
    goto done;
  }
done:
  free(records);
  return r;
}

/* ------------------------------------------------------------------ */
/* public entry points                                                 */
/* ------------------------------------------------------------------ */

void orc_default_options(orc_options *o) { memset(o, 0, sizeof *o); }
void orc_free(void *p) { free(p); }

static int finish_result(orc_result *res, const err_t *e, sink_t *out, const reader_t *in) {
  res->kind = e->kind;
  memcpy(res->msg, e->msg, sizeof res->msg);
  res->out = out->data;
  res->out_len = out->len;
  res->in_consumed = in->pos;
  return e->kind;
}

int orc_lzma_decompress(const uint8_t *in, size_t in_len, const orc_options *opt, orc_result *res) {
  reader_t rd = {in, 0, in_len};
  sink_t out = {0, 0, 0};
  err_t e;
  params_t params;
  orc_options dflt;
  memset(&e, 0, sizeof e);
  if (!opt) {
    orc_default_options(&dflt);
    opt = &dflt;
  }
  if (!read_header(&rd, opt, &params, &e))
    lzma_decoder_decompress(&params, opt->memlimit_is_some, opt->memlimit, &rd, &out, &e);
  return finish_result(res, &e, &out, &rd);
}

int orc_lzma_raw_decompress(const uint8_t *in, size_t in_len, uint32_t lc, uint32_t lp, uint32_t pb,
                            uint32_t dict_size, int unpacked_is_some, uint64_t unpacked_size,
                            int memlimit_is_some, uint64_t memlimit, orc_result *res) {
  reader_t rd = {in, 0, in_len};
  sink_t out = {0, 0, 0};
  err_t e;
  params_t params;
  memset(&e, 0, sizeof e);
  params.properties.lc = lc;
  params.properties.lp = lp;
  params.properties.pb = pb;
  params.dict_size = dict_size;
  params.unpacked_is_some = unpacked_is_some;
  params.unpacked_size = unpacked_size;
  lzma_decoder_decompress(&params, memlimit_is_some, memlimit, &rd, &out, &e);
  return finish_result(res, &e, &out, &rd);
}

int orc_lzma2_decompress(const uint8_t *in, size_t in_len, orc_result *res) {
  reader_t rd = {in, 0, in_len};
  sink_t out = {0, 0, 0};
  err_t e;
  memset(&e, 0, sizeof e);
  lzma2_decoder_decompress(&rd, &out, &e);
  return finish_result(res, &e, &out, &rd);
}

int orc_xz_decompress(const uint8_t *in, size_t in_len, orc_result *res) {
  reader_t rd = {in, 0, in_len};
  sink_t out = {0, 0, 0};
  err_t e;
  memset(&e, 0, sizeof e);
  xz_decode_stream(&rd, &out, &e);
  return finish_result(res, &e, &out, &rd);
}

/* ------------------------------------------------------------------ */
/* Stream (feature `stream`), src/decode/stream.rs + the Partial mode   */
/* of DecoderState::process_mode, src/decode/lzma.rs:395-524            */
/* ------------------------------------------------------------------ */

/* RangeDecoder::decode_bit with update = false (rangecoder.rs:92-120): the probability is read, not written. <0 => EOF */
static inline int rc_peek_bit(rc_t *rc, uint16_t prob) {
  uint32_t bound = (rc->range >> 11) * (uint32_t)prob;
  if (rc->code < bound) {
    rc->range = bound;
    if (rc_normalize(rc)) return -1;
    return 0;
  }
  rc->code -= bound;
  rc->range -= bound;
  if (rc_normalize(rc)) return -1;
  return 1;
}

/* parse_bit_tree / parse_reverse_bit_tree with update = false (rangecoder.rs:122-151) */
static int rc_peek_bit_tree(rc_t *rc, unsigned num_bits, const uint16_t *probs, uint32_t *out) {
  uint32_t tmp = 1;
  unsigned i;
  for (i = 0; i < num_bits; i++) {
    int bit = rc_peek_bit(rc, probs[tmp]);
    if (bit < 0) return 1;
    tmp = (tmp << 1) ^ (uint32_t)bit;
  }
  *out = tmp - (1u << num_bits);
  return 0;
}
static int rc_peek_reverse_bit_tree(rc_t *rc, unsigned num_bits, const uint16_t *probs, size_t offset, uint32_t *out) {
  uint32_t result = 0, tmp = 1;
  unsigned i;
  for (i = 0; i < num_bits; i++) {
    int bit = rc_peek_bit(rc, probs[offset + tmp]);
    if (bit < 0) return 1;
    tmp = (tmp << 1) ^ (uint32_t)bit;
    result ^= (uint32_t)bit << i;
  }
  *out = result;
  return 0;
}

/* LenDecoder::decode with update = false (rangecoder.rs:256-269) */
static int lendec_peek(const lendec_t *l, rc_t *rc, size_t pos_state, size_t *out) {
  uint32_t v;
  int bit = rc_peek_bit(rc, l->choice);
  if (bit < 0) return 1;
  if (!bit) {
    if (rc_peek_bit_tree(rc, 3, l->low_coder[pos_state], &v)) return 1;
    *out = v;
    return 0;
  }
  bit = rc_peek_bit(rc, l->choice2);
  if (bit < 0) return 1;
  if (!bit) {
    if (rc_peek_bit_tree(rc, 3, l->mid_coder[pos_state], &v)) return 1;
    *out = (size_t)v + 8;
    return 0;
  }
  if (rc_peek_bit_tree(rc, 8, l->high_coder, &v)) return 1;
  *out = (size_t)v + 16;
  return 0;
}

/* DecoderState::try_process_next, lzma.rs:401-417: process_next_inner(update = false) (lzma.rs:278-393) on a copy of the range coder
 * over `buf` -- nothing of the decoder or the output changes; what can fail is a read beyond `buf` and the matched literal's
 * last_n (decode_literal, lzma.rs:543-545).  The marker test, the LRU / state updates and every append are inside `if update`.
 * Returns 0 = Ok, 1 = Err. */
static int dstate_try_process_next(const dstate_t *s, const lzbuf_t *output, const uint8_t *buf, size_t n, uint32_t range,
                                   uint32_t code) {
  reader_t temp = {buf, 0, n};
  rc_t rc;
  err_t scratch;
  size_t pos_state = (size_t)(output->len & (((uint64_t)1 << s->lzma_props.pb) - 1));
  size_t len;
  int bit;
  rc.stream = &temp;
  rc.range = range;
  rc.code = code;
  memset(&scratch, 0, sizeof scratch);
  bit = rc_peek_bit(&rc, s->is_match[(s->state << 4) + pos_state]);
  if (bit < 0) return 1;
  if (!bit) { /* decode_literal, lzma.rs:526-561, update = false */
    size_t prev_byte = lzbuf_last_or(output, 0);
    size_t result = 1;
    size_t lit_state = ((size_t)(output->len & (((uint64_t)1 << s->lzma_props.lp) - 1)) << s->lzma_props.lc) +
                       (prev_byte >> (8 - s->lzma_props.lc));
    const uint16_t *probs = s->literal_probs + lit_state * 0x300;
    if (s->state >= 7) {
      uint8_t mb = 0;
      size_t match_byte;
      if (lzbuf_last_n(output, s->rep[0] + 1, &mb, &scratch)) return 1;
      match_byte = mb;
      while (result < 0x100) {
        size_t match_bit = (match_byte >> 7) & 1;
        match_byte <<= 1;
        bit = rc_peek_bit(&rc, probs[((1 + match_bit) << 8) + result]);
        if (bit < 0) return 1;
        result = (result << 1) ^ (size_t)bit;
        if (match_bit != (size_t)bit) break;
      }
    }
    while (result < 0x100) {
      bit = rc_peek_bit(&rc, probs[result]);
      if (bit < 0) return 1;
      result = (result << 1) ^ (size_t)bit;
    }
    return 0;
  }
  bit = rc_peek_bit(&rc, s->is_rep[s->state]);
  if (bit < 0) return 1;
  if (bit) {
    bit = rc_peek_bit(&rc, s->is_rep_g0[s->state]);
    if (bit < 0) return 1;
    if (!bit) {
      bit = rc_peek_bit(&rc, s->is_rep_0long[(s->state << 4) + pos_state]);
      if (bit < 0) return 1;
      if (!bit) return 0; /* short rep */
    } else {
      bit = rc_peek_bit(&rc, s->is_rep_g1[s->state]);
      if (bit < 0) return 1;
      if (bit) {
        bit = rc_peek_bit(&rc, s->is_rep_g2[s->state]);
        if (bit < 0) return 1;
      }
    }
    if (lendec_peek(&s->rep_len_decoder, &rc, pos_state, &len)) return 1;
  } else { /* new distance: decode_distance, lzma.rs:563-592, update = false */
    size_t len_state, pos_slot, num_direct_bits;
    uint64_t result;
    uint32_t v;
    if (lendec_peek(&s->len_decoder, &rc, pos_state, &len)) return 1;
    len_state = len > 3 ? 3 : len;
    if (rc_peek_bit_tree(&rc, 6, s->pos_slot_decoder[len_state], &v)) return 1;
    pos_slot = v;
    if (pos_slot >= 4) {
      num_direct_bits = (pos_slot >> 1) - 1;
      result = (uint64_t)(2 ^ (pos_slot & 1)) << num_direct_bits;
      if (pos_slot < 14) {
        if (rc_peek_reverse_bit_tree(&rc, (unsigned)num_direct_bits, s->pos_decoders, (size_t)(result - pos_slot), &v)) return 1;
      } else {
        if (rc_get(&rc, (unsigned)(num_direct_bits - 4), &v)) return 1;
        if (rc_peek_reverse_bit_tree(&rc, 4, s->align_decoder, 0, &v)) return 1;
      }
    }
  }
  return 0;
}

#define ORC_MAX_REQUIRED_INPUT 20 /* lzma.rs:13 */
enum { ORC_MODE_PARTIAL = 0, ORC_MODE_FINISH = 1 };

/* DecoderState::process_mode, lzma.rs:435-524, with the partial input buffer of lzma.rs:166-168 (`partial`, `*partial_pos`).
 * Returns 0 or the error kind. */
static int dstate_process_mode(dstate_t *s, lzbuf_t *output, rc_t *rc, int mode, uint8_t *partial, size_t *partial_pos, err_t *e) {
  for (;;) {
    if (s->unpacked_is_some) {
      if (output->len >= s->unpacked_size) break;
    } else if (mode == ORC_MODE_PARTIAL ? (rd_is_eof(rc->stream) && *partial_pos == 0)
                                        : (rc_is_finished_ok(rc) && *partial_pos == 0)) {
      break;
    }
    if (*partial_pos > 0) {
      uint8_t tmp[ORC_MAX_REQUIRED_INPUT];
      reader_t tmp_reader;
      rc_t tmp_rc;
      size_t take, new_len;
      int st;
      /* read_partial_input_buf, lzma.rs:420-433: as much as there is, up to the buffer's end */
      take = rc->stream->end - rc->stream->pos;
      if (take > ORC_MAX_REQUIRED_INPUT - *partial_pos) take = ORC_MAX_REQUIRED_INPUT - *partial_pos;
      memcpy(partial + *partial_pos, rc->stream->p + rc->stream->pos, take);
      rc->stream->pos += take;
      *partial_pos += take;
      memcpy(tmp, partial, sizeof tmp);
      if (mode == ORC_MODE_PARTIAL && *partial_pos < ORC_MAX_REQUIRED_INPUT &&
          dstate_try_process_next(s, output, tmp, *partial_pos, rc->range, rc->code))
        return 0; /* need more data */
      tmp_reader.p = tmp;
      tmp_reader.pos = 0;
      tmp_reader.end = *partial_pos;
      tmp_rc.stream = &tmp_reader;
      tmp_rc.range = rc->range;
      tmp_rc.code = rc->code;
      st = dstate_process_next(s, output, &tmp_rc, e);
      if (st == ST_ERROR) return e->kind;
      rc->range = tmp_rc.range;
      rc->code = tmp_rc.code;
      new_len = *partial_pos - tmp_reader.pos;
      memcpy(partial, tmp + tmp_reader.pos, new_len);
      *partial_pos = new_len;
      if (st == ST_FINISHED) break;
    } else {
      const uint8_t *buf = rc->stream->p + rc->stream->pos;
      size_t n = rc->stream->end - rc->stream->pos;
      int st;
      if (mode == ORC_MODE_PARTIAL && n < ORC_MAX_REQUIRED_INPUT && dstate_try_process_next(s, output, buf, n, rc->range, rc->code)) {
        /* return self.read_partial_input_buf(rangecoder) */
        size_t take = n > ORC_MAX_REQUIRED_INPUT ? ORC_MAX_REQUIRED_INPUT : n;
        memcpy(partial, buf, take);
        rc->stream->pos += take;
        *partial_pos = take;
        return 0;
      }
      st = dstate_process_next(s, output, rc, e);
      if (st == ST_ERROR) return e->kind;
      if (st == ST_FINISHED) break;
    }
  }
  if (s->unpacked_is_some && mode == ORC_MODE_FINISH && s->unpacked_size != output->len)
    return fail(e, ORC_LZMA_ERROR, "Expected unpacked size of %llu but decompressed to %llu", (unsigned long long)s->unpacked_size,
                (unsigned long long)output->len);
  return 0;
}

#define ORC_MAX_TMP_LEN 18 /* stream.rs:9-24: MAX_HEADER_LEN (5 + 8) + START_BYTES (5) */

struct orc_stream {
  orc_options options;
  int allow_incomplete;
  uint8_t tmp[ORC_MAX_TMP_LEN]; /* Stream.tmp */
  size_t tmp_pos;
  int state; /* 0: State::Header(W), 1: State::Data(RunState), 2: None (a write failed, or finished) */
  sink_t sink; /* W = Vec<u8> */
  dstate_t decoder; /* RunState */
  uint32_t range, code;
  lzbuf_t output;
  uint8_t partial[ORC_MAX_REQUIRED_INPUT]; /* DecoderState.partial_input_buf */
  size_t partial_pos;
  size_t last_taken; /* test aid: the Ok(n) of Stream::write summed over the last write_all */
};

/* Stream::new_with_options, stream.rs:88-101 */
orc_stream *orc_stream_new(const orc_options *opt, int allow_incomplete) {
  orc_stream *s = (orc_stream *)calloc(1, sizeof *s);
  if (opt) s->options = *opt;
  s->allow_incomplete = allow_incomplete;
  return s;
}

/* Stream::read_header, stream.rs:153-189.  Returns the next state (0 Header / 1 Data), or -1 with *e set (fatal). */
static int stream_read_header(orc_stream *s, reader_t *input, err_t *e) {
  params_t params;
  err_t he;
  rc_t rc;
  memset(&he, 0, sizeof he);
  if (read_header(input, &s->options, &params, &he)) {
    if (he.kind == ORC_HEADER_TOO_SHORT) return 0; /* need more data, try again later */
    *e = he;
    return -1;
  }
  if (rc_new(&rc, input)) return 0; /* header read, range coder start not there yet: Header again (the decoder made here is dropped) */
  dstate_new(&s->decoder, params.properties, params.unpacked_is_some, params.unpacked_size);
  memset(&s->output, 0, sizeof s->output);
  s->output.is_accum = 0;
  s->output.stream = &s->sink;
  s->output.dict_size = params.dict_size;
  s->output.memlimit = s->options.memlimit_is_some ? s->options.memlimit : UINT64_MAX;
  s->range = rc.range;
  s->code = rc.code;
  s->partial_pos = 0;
  return 1;
}

/* Stream::read_data, stream.rs:192-207: process_stream; an error becomes io::Error::new(Other, format!("{:?}", error)) (stream.rs:343-347) */
static int stream_read_data(orc_stream *s, reader_t *input, err_t *e) {
  rc_t rc;
  err_t de;
  memset(&de, 0, sizeof de);
  rc.stream = input;
  rc.range = s->range;
  rc.code = s->code;
  if (dstate_process_mode(&s->decoder, &s->output, &rc, ORC_MODE_PARTIAL, s->partial, &s->partial_pos, &de)) {
    /* Debug of error::Error (derive): LzmaError("...") / XzError("...") / IoError(..) -- only LzmaError can come out of Partial mode on
     * a byte slice; the text after the Display prefix is the variant's String */
    const char *m = strchr(de.msg, ':');
    m = m ? m + 2 : de.msg;
    e->kind = ORC_IO_ERROR;
    snprintf(e->msg, sizeof e->msg, "%s(\"%.360s\")", de.kind == ORC_LZMA_ERROR ? "LzmaError" : de.kind == ORC_XZ_ERROR ? "XzError" : "IoError", m);
    return 1;
  }
  s->range = rc.range;
  s->code = rc.code;
  return 0;
}

/* <Stream as io::Write>::write, stream.rs:223-326.  *consumed = Ok(n); returns 0, or 1 with e = the io::Error (Display text in e->msg
 * WITHOUT a prefix: this is an io::Error, not an error::Error). */
static int stream_write(orc_stream *s, const uint8_t *data, size_t len, size_t *consumed, err_t *e) {
  reader_t input = {data, 0, len};
  if (s->state != 2) {
    int st = s->state;
    s->state = 2; /* self.state.take() */
    if (st == 0) {
      int res;
      if (s->tmp_pos > 0) {
        size_t take = len < ORC_MAX_TMP_LEN - s->tmp_pos ? len : ORC_MAX_TMP_LEN - s->tmp_pos;
        reader_t tmp_input;
        memcpy(s->tmp + s->tmp_pos, data, take);
        input.pos = take;
        s->tmp_pos += take;
        tmp_input.p = s->tmp;
        tmp_input.pos = 0;
        tmp_input.end = s->tmp_pos;
        res = stream_read_header(s, &tmp_input, e);
        if (res == 1) { /* discard the bytes up to position */
          size_t new_len = s->tmp_pos - tmp_input.pos;
          memmove(s->tmp, s->tmp + tmp_input.pos, new_len);
          s->tmp_pos = new_len;
        }
      } else {
        res = stream_read_header(s, &input, e);
      }
      if (res == 0) {
        if (s->tmp_pos == 0) { /* reset the cursor because we may have partial reads */
          size_t take = len < ORC_MAX_TMP_LEN ? len : ORC_MAX_TMP_LEN;
          memcpy(s->tmp, data, take);
          input.pos = take;
          s->tmp_pos = take;
        }
        st = 0;
      } else if (res == 1) {
        st = 1;
      } else { /* IoError(e) | HeaderTooShort(e) => e; LzmaError(s) | XzError(s) => io::Error::new(Other, s) */
        const char *m = strchr(e->msg, ':');
        char plain[384];
        snprintf(plain, sizeof plain, "%s", m ? m + 2 : e->msg);
        memcpy(e->msg, plain, sizeof plain);
        e->kind = ORC_IO_ERROR;
        return 1;
      }
    } else {
      if (s->tmp_pos > 0) {
        reader_t tmp_input = {s->tmp, 0, s->tmp_pos};
        if (stream_read_data(s, &tmp_input, e)) return 1;
        s->tmp_pos = 0;
      }
      if (stream_read_data(s, &input, e)) return 1;
      st = 1;
    }
    s->state = st;
  }
  *consumed = input.pos;
  return 0;
}

/* io::Write::write_all (std): write until everything is taken; Ok(0) from write is ErrorKind::WriteZero, "failed to write whole buffer".
 * Returns 0 or ORC_IO_ERROR; msg (384 bytes) = the io::Error's Display text. */
int orc_stream_write_all(orc_stream *s, const uint8_t *data, size_t len, char *msg) {
  err_t e;
  memset(&e, 0, sizeof e);
  s->last_taken = 0;
  while (len > 0) {
    size_t n = 0;
    if (stream_write(s, data, len, &n, &e)) {
      memcpy(msg, e.msg, sizeof e.msg);
      return ORC_IO_ERROR;
    }
    if (n == 0) {
      snprintf(msg, 384, "failed to write whole buffer");
      return ORC_IO_ERROR;
    }
    data += n;
    len -= n;
    s->last_taken += n;
  }
  msg[0] = 0;
  return 0;
}

/* how many bytes the last orc_stream_write_all got rid of before it returned (all of them on success) */
size_t orc_stream_last_taken(const orc_stream *s) { return s->last_taken; }

/* Stream::get_output, stream.rs:102-107: what the sink holds so far */
size_t orc_stream_output(const orc_stream *s, const uint8_t **p) {
  *p = s->sink.data;
  return s->sink.len;
}

/* ... is_some() of it: `self.state.as_ref().map(..)` -- None once a write has failed (the state was taken, stream.rs:230) */
int orc_stream_has_output(const orc_stream *s) { return s->state != 2; }

/* Stream::finish, stream.rs:119-150; frees the stream.  res->out = what the sink holds (Ok: the whole output; Err: the reference drops W). */
int orc_stream_finish(orc_stream *s, orc_result *res) {
  err_t e;
  reader_t none = {NULL, 0, 0};
  memset(&e, 0, sizeof e);
  if (s->state == 0) {
    if (s->tmp_pos > 0) fail(&e, ORC_LZMA_ERROR, "failed to read header");
  } else if (s->state == 1) {
    int r = 0;
    if (!s->allow_incomplete) { /* one last time with (what is left in tmp as) input: the end-of-stream checks */
      reader_t stream = {s->tmp, 0, s->tmp_pos};
      rc_t rc;
      rc.stream = &stream;
      rc.range = s->range;
      rc.code = s->code;
      r = dstate_process_mode(&s->decoder, &s->output, &rc, ORC_MODE_FINISH, s->partial, &s->partial_pos, &e);
    }
    if (!r) lzbuf_finish(&s->output);
  } else {
    fail(&e, ORC_LZMA_ERROR, "can't finish stream because of previous write error");
  }
  if (s->state == 1 || s->output.buf) {
    free(s->output.buf);
    dstate_free(&s->decoder);
  }
  finish_result(res, &e, &s->sink, &none);
  free(s);
  return res->kind;
}

/* ------------------------------------------------------------------ */
/* cpu_baseline helper                                                 */
/* ------------------------------------------------------------------ */

typedef struct {
  const uint8_t *in_base;
  const uint64_t *in_off, *in_len;
  uint32_t n;
  uint32_t *next; /* shared work counter */
  pthread_mutex_t *mu;
  int64_t bytes;
  uint32_t check;
  int failed;
} bench_arg_t;

static void *bench_worker(void *vp) {
  bench_arg_t *a = (bench_arg_t *)vp;
  sink_t out = {0, 0, 0}; /* reused across streams: one allocation per thread, like a caller
                             that hands the same Vec<u8> to lzma_decompress again and again */
  for (;;) {
    uint32_t i;
    reader_t rd;
    err_t e;
    params_t params;
    orc_options dflt;
    pthread_mutex_lock(a->mu);
    i = (*a->next)++;
    pthread_mutex_unlock(a->mu);
    if (i >= a->n) break;
    rd.p = a->in_base + a->in_off[i];
    rd.pos = 0;
    rd.end = (size_t)a->in_len[i];
    memset(&e, 0, sizeof e);
    orc_default_options(&dflt);
    out.len = 0;
    if (read_header(&rd, &dflt, &params, &e) ||
        lzma_decoder_decompress(&params, 0, 0, &rd, &out, &e))
      a->failed = 1;
    a->bytes += (int64_t)out.len;
    a->check ^= orc_crc32(out.data, out.len) + i;
  }
  free(out.data);
  return NULL;
}

int64_t orc_bench_lzma_batch(const uint8_t *in_base, const uint64_t *in_off, const uint64_t *in_len,
                             uint32_t n, uint32_t nthreads, uint32_t *check) {
  pthread_t th[256];
  bench_arg_t args[256];
  pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
  uint32_t next = 0, t;
  int64_t total = 0;
  uint32_t chk = 0;
  int failed = 0;
  if (nthreads == 0) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  for (t = 0; t < nthreads; t++) {
    bench_arg_t a = {in_base, in_off, in_len, n, &next, &mu, 0, 0, 0};
    args[t] = a;
    pthread_create(&th[t], NULL, bench_worker, &args[t]);
  }
  for (t = 0; t < nthreads; t++) {
    pthread_join(th[t], NULL);
    total += args[t].bytes;
    chk ^= args[t].check;
    failed |= args[t].failed;
  }
  if (check) *check = chk;
  return failed ? -1 : total;
}
