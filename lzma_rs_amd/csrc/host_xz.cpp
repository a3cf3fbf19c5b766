// host_xz.cpp -- the XZ container walk, the Index planner and the .xz batch call (see host_internal.h)
#include "host_internal.h"

using namespace milzma;
using namespace milzma::host;

// ------------------------------------------------------------------------------------------
// .xz: xz::decode_stream (src/decode/xz.rs:18-94) with the LZMA2 payload of each block decoded
// on the device.  The walk below is the reference's, statement for statement; what is new is
// that block payloads can be decoded ahead of the walk, all at once (see xz_batch).
// ------------------------------------------------------------------------------------------

MILZMA_HOST_NS_BEGIN

enum { CHECK_NONE = 0x00, CHECK_CRC32 = 0x01, CHECK_CRC64 = 0x04, CHECK_SHA256 = 0x0A };

const char* check_name(int m) {
  switch (m) {
    case CHECK_NONE: return "None";
    case CHECK_CRC32: return "Crc32";
    case CHECK_CRC64: return "Crc64";
    default: return "Sha256";
  }
}

// StreamFlags::parse (src/xz/mod.rs:15-31) + CheckMethod::try_from (:54-66)
int stream_flags_parse(uint32_t field, int* check, milzma_output* o) {
  const uint32_t b0 = (field >> 8) & 0xFF, b1 = field & 0xFF;
  if (b0 != 0) return out_fail(o, MILZMA_XZ_ERROR, "Invalid null byte in Stream Flags: %x", b0);
  if (b1 != CHECK_NONE && b1 != CHECK_CRC32 && b1 != CHECK_CRC64 && b1 != CHECK_SHA256)
    return out_fail(o, MILZMA_XZ_ERROR, "Invalid check method %x, expected one of [0x00, 0x01, 0x04, 0x0A]", b1);
  *check = int(b1);
  return MILZMA_OK;
}

// get_multibyte (src/decode/xz.rs:448-464): 0 ok, 1 eof, 2 invalid
int get_multibyte(Cursor& c, uint64_t* out) {
  uint64_t r = 0;
  for (int i = 0; i < 9; i++) {
    uint8_t b;
    if (!c.u8(&b)) return 1;
    r ^= uint64_t(b & 0x7F) << (i * 7);
    if ((b & 0x80) == 0) {
      *out = r;
      return 0;
    }
  }
  return 2;
}
int multibyte_err(int rc, milzma_output* o) {
  return rc == 1 ? out_io_eof(o) : out_fail(o, MILZMA_XZ_ERROR, "Invalid multi-byte encoding");
}

struct BlockHeader {
  size_t num_filters = 0;
  size_t props_len[4] = {0, 0, 0, 0};
  bool has_packed = false, has_unpacked = false;
  uint64_t packed = 0, unpacked = 0;
};

// read_block_header (src/decode/xz.rs:356-446); `c` is limited to the header bytes
int read_block_header(Cursor& c, uint64_t header_size, BlockHeader* bh, milzma_output* o) {
  uint8_t flags;
  if (!c.u8(&flags)) return out_io_eof(o);
  const size_t num_filters = size_t(flags & 3) + 1;
  if (flags & 0x3C)
    return out_fail(o, MILZMA_XZ_ERROR, "Invalid block flags %u, reserved bits (mask 0x3C) must be zero", unsigned(flags));
  bh->has_packed = (flags & 0x40) != 0;
  bh->has_unpacked = (flags & 0x80) != 0;
  int rc;
  if (bh->has_packed && (rc = get_multibyte(c, &bh->packed))) return multibyte_err(rc, o);
  if (bh->has_unpacked && (rc = get_multibyte(c, &bh->unpacked))) return multibyte_err(rc, o);
  for (size_t i = 0; i < num_filters; i++) {
    uint64_t id, psize;
    if ((rc = get_multibyte(c, &id))) return multibyte_err(rc, o);
    if (id != 0x21) return out_fail(o, MILZMA_XZ_ERROR, "Unknown filter id %" PRIu64, id);
    if ((rc = get_multibyte(c, &psize))) return multibyte_err(rc, o);
    if (psize > header_size)
      return out_fail(o, MILZMA_XZ_ERROR, "Size of filter properties exceeds block header size (%" PRIu64 " > %" PRIu64 ")",
                      psize, header_size);
    if (!c.exact(nullptr, size_t(psize)))
      return out_fail(o, MILZMA_XZ_ERROR, "Could not read filter properties of size %" PRIu64 ": %s", psize, kEofMsg);
    bh->props_len[bh->num_filters++] = size_t(psize);
  }
  while (c.pos < c.end) {  // util::flush_zero_padding (src/decode/util.rs:14-36)
    if (c.p[c.pos] != 0) return out_fail(o, MILZMA_XZ_ERROR, "Invalid block header padding, must be null bytes");
    c.pos++;
  }
  return MILZMA_OK;
}

// Result of decoding one LZMA2 payload (Lzma2Decoder::new().decompress, src/decode/xz.rs:350).
struct Payload {
  milzma_result res;
  const uint8_t* data = nullptr;  // res.out_len bytes (valid when res.status == OK)
  std::vector<uint8_t> own;       // backing store when decoded on demand
  size_t prefilled_at = SIZE_MAX; // the payload already sits at this offset of the file's output buffer (OutBuf::append)
  bool has_crc = false;           // crc32 / crc64 of data were computed on the GPU (milzma_crc_units' kernel)
  uint32_t crc32 = 0;
  uint64_t crc64 = 0;
};
// Decodes the LZMA2 stream that starts at in[0]; the reader's EOF is in_len.
using PayloadFn = std::function<bool(const uint8_t* in, size_t in_len, size_t cap_hint, Payload*)>;

struct Record {
  uint64_t unpadded, unpacked;
};

// read_block (src/decode/xz.rs:196-290); block_start = position of the header-size byte
// The file's output: a pooled buffer (out_alloc) that grows by moving and is handed to milzma_output as is.
struct OutBuf {
  uint8_t* p = nullptr;
  size_t n = 0, cap = 0;
  bool moved = false;  // the buffer was reallocated: whatever had been put beyond n beforehand is gone
  // hold_first: payloads may point INTO the first buffer (a streamed launch wrote them there): if the file outgrows it, it is kept
  // (in `keep`) until this object goes, so that those pointers stay good
  bool hold_first = false;
  uint8_t* keep = nullptr;
  OutBuf() = default;
  OutBuf(const OutBuf&) = delete;             // (owns its buffers)
  OutBuf& operator=(const OutBuf&) = delete;
  ~OutBuf() {
    milzma_free(p);
    milzma_free(keep);
  }
  bool reserve(size_t want) {
    if (want <= cap) return true;
    size_t c = std::max(want, cap + cap / 2);
    c = out_class(std::max<size_t>(c, 4096));
    uint8_t* q = out_alloc(c);
    if (!q) return false;
    if (n) memcpy(q, p, n);
    if (hold_first && !keep)
      keep = p;
    else
      milzma_free(p);
    moved = p != nullptr;
    p = q;
    cap = c;
    return true;
  }
  // the first allocation page-locked (a streamed launch writes into it from the device); growth moves to ordinary memory
  bool reserve_pinned(size_t want) {
    if (p) return reserve(want);
    const size_t c = out_class(std::max<size_t>(want, 4096));
    p = out_alloc(c, true);
    if (!p) return false;
    cap = c;
    return true;
  }
  // prefilled_at: the same bytes were put at that offset of this buffer beforehand (streamed xz batches copy every block's spans to
  // their place in the file's buffer while the kernel runs): if that is where the file stands, they are taken as they are
  bool append(const uint8_t* src, size_t len, size_t prefilled_at = SIZE_MAX) {
    if (prefilled_at == n && p && !moved && n + len <= cap) {
      n += len;
      return true;
    }
    // Bytes that are NOT at their place (a block decoded on demand, or one that is where the Index put it while the file stands
    // elsewhere: the Index lied about an earlier block) are copied in.  In the first buffer of a streamed batch that copy would run over
    // the places of the blocks behind it -- payloads the walk has yet to take, src itself perhaps: the file moves to a buffer of its own
    // first and the first one is kept (`keep`) for as long as payloads may point into it.
    if (hold_first && !keep && p) {
      const size_t c = out_class(std::max<size_t>(std::max(cap, n + len), 4096));
      uint8_t* q = out_alloc(c);
      if (!q) return false;
      if (n) memcpy(q, p, n);
      keep = p;
      p = q;
      cap = c;
      moved = true;
    } else if (!reserve(n + len)) {
      return false;
    }
    if (len) memcpy(p + n, src, len);
    n += len;
    return true;
  }
};

int read_block(milzma_ctx* ctx, Cursor& c, size_t block_start, OutBuf& output, int check,
               std::vector<Record>& records, uint8_t hsize_byte, const PayloadFn& decode, milzma_output* o) {
  const uint64_t header_size = (uint64_t(hsize_byte) << 2) - 1;
  BlockHeader bh;
  const size_t hdr_begin = c.pos, saved_end = c.end;
  const size_t hdr_end = uint64_t(c.end - c.pos) > header_size ? c.pos + size_t(header_size) : c.end;
  c.end = hdr_end;  // count_input.take(header_size) behind a BufReader + CrcDigestRead
  const int hr = read_block_header(c, header_size, &bh, o);
  c.end = saved_end;
  if (hr) return hr;
  c.pos = hdr_end;
  uint32_t digest = crc32_update(0xFFFFFFFFu, &hsize_byte, 1);
  digest = ~crc32_update(digest, c.p + hdr_begin, hdr_end - hdr_begin);
  uint32_t crc;
  if (!c.u32le(&crc)) return out_io_eof(o);
  if (crc != digest)
    return out_fail(o, MILZMA_XZ_ERROR, "Invalid header CRC32: expected 0x%08x but got 0x%08x", crc, digest);

  Payload cur;
  for (size_t i = 0; i < bh.num_filters; i++) {
    // decode_filter (src/decode/xz.rs:335-354)
    if (bh.props_len[i] != 1) return out_fail(o, MILZMA_XZ_ERROR, "Invalid properties for filter Lzma2");
    Payload next;
    const uint8_t* src = i == 0 ? c.p + c.pos : cur.data;
    const size_t src_len = i == 0 ? c.end - c.pos : size_t(cur.res.out_len);
    const size_t hint = (i == 0 && bh.has_unpacked) ? size_t(std::min<uint64_t>(bh.unpacked, MILZMA_MAX_UNIT_BYTES)) : 0;
    if (!decode(src, src_len, hint, &next)) return infra(ctx, o);
    if (next.res.status != MILZMA_ST_OK) {
      // (the LZMA2 decoder read from the file's reader: it stands where the payload's decode stopped -- round 4: was left at the
      //  payload's first byte, nothing compared the position of failed decodes)
      if (i == 0) c.pos += size_t(std::min<uint64_t>(next.res.in_consumed, c.end - c.pos));
      o->kind = milzma_result_message(&next.res, MILZMA_KIND_LZMA2, o->msg, sizeof o->msg);
      return o->kind;
    }
    if (i == 0) {
      const uint64_t packed = next.res.in_consumed;
      c.pos += size_t(packed);
      if (bh.has_packed && packed != bh.packed)
        return out_fail(o, MILZMA_XZ_ERROR, "Invalid compressed size: expected %" PRIu64 " but got %" PRIu64, bh.packed,
                        packed);
    }
    cur = std::move(next);
    if (!cur.own.empty()) cur.data = cur.own.data();
  }
  const uint64_t unpacked_size = cur.res.out_len;
  if (bh.has_unpacked && unpacked_size != bh.unpacked)
    return out_fail(o, MILZMA_XZ_ERROR, "Invalid decompressed size: expected %" PRIu64 " but got %" PRIu64, bh.unpacked,
                    unpacked_size);
  const size_t count = c.pos - block_start;
  const size_t padding = ((count ^ 3) + 1) & 3;
  for (size_t i = 0; i < padding; i++) {
    uint8_t b;
    if (!c.u8(&b)) return out_io_eof(o);
    if (b != 0) return out_fail(o, MILZMA_XZ_ERROR, "Invalid block padding, must be null bytes");
  }
  // validate_block_check (src/decode/xz.rs:292-333)
  switch (check) {
    case CHECK_NONE: break;
    case CHECK_CRC32: {
      uint32_t want;
      if (!c.u32le(&want)) return out_io_eof(o);
      const uint32_t got = cur.has_crc ? cur.crc32 : milzma_crc32(cur.data, size_t(unpacked_size));
      if (want != got) return out_fail(o, MILZMA_XZ_ERROR, "Invalid block CRC32, expected 0x%08x but got 0x%08x", want, got);
      break;
    }
    case CHECK_CRC64: {
      uint64_t want;
      if (!c.u64le(&want)) return out_io_eof(o);
      const uint64_t got = cur.has_crc ? cur.crc64 : milzma_crc64(cur.data, size_t(unpacked_size));
      if (want != got)
        return out_fail(o, MILZMA_XZ_ERROR, "Invalid block CRC64, expected 0x%016" PRIx64 " but got 0x%016" PRIx64, want, got);
      break;
    }
    default: return out_fail(o, MILZMA_XZ_ERROR, "Unsupported SHA-256 checksum (not yet implemented)");
  }
  if (!output.append(cur.data, size_t(unpacked_size), bh.num_filters == 1 ? cur.prefilled_at : SIZE_MAX))
    return out_fail(o, MILZMA_INFRA_ERROR, "out of memory");
  records.push_back(Record{uint64_t(c.pos - block_start - padding), unpacked_size});
  return MILZMA_OK;
}

// check_index (src/decode/xz.rs:96-171); index_start = position of the 0x00 indicator byte
int check_index(Cursor& c, size_t index_start, const std::vector<Record>& records, milzma_output* o) {
  const size_t digest_from = c.pos;
  uint64_t num, v;
  int rc;
  if ((rc = get_multibyte(c, &num))) return multibyte_err(rc, o);
  if (num != records.size())
    return out_fail(o, MILZMA_XZ_ERROR, "Expected %" PRIu64 " records but got %zu records", num, records.size());
  for (size_t i = 0; i < records.size(); i++) {
    if ((rc = get_multibyte(c, &v))) return multibyte_err(rc, o);
    if (v != records[i].unpadded)
      return out_fail(o, MILZMA_XZ_ERROR,
                      "Invalid index for record %zu: unpadded size (%" PRIu64 ") does not match index (%" PRIu64 ")", i,
                      records[i].unpadded, v);
    if ((rc = get_multibyte(c, &v))) return multibyte_err(rc, o);
    if (v != records[i].unpacked)
      return out_fail(o, MILZMA_XZ_ERROR,
                      "Invalid index for record %zu: unpacked size (%" PRIu64 ") does not match index (%" PRIu64 ")", i,
                      records[i].unpacked, v);
  }
  const size_t count = c.pos - index_start;
  const size_t padding = ((count ^ 3) + 1) & 3;
  for (size_t i = 0; i < padding; i++) {
    uint8_t b;
    if (!c.u8(&b)) return out_io_eof(o);
    if (b != 0) return out_fail(o, MILZMA_XZ_ERROR, "Invalid index padding, must be null bytes");
  }
  const uint8_t tag = 0;
  uint32_t digest = crc32_update(0xFFFFFFFFu, &tag, 1);
  digest = ~crc32_update(digest, c.p + digest_from, c.pos - digest_from);
  uint32_t crc;
  if (!c.u32le(&crc)) return out_io_eof(o);
  if (crc != digest) return out_fail(o, MILZMA_XZ_ERROR, "Invalid index CRC32: expected 0x%08x but got 0x%08x", crc, digest);
  return MILZMA_OK;
}

// xz::decode_stream (src/decode/xz.rs:18-94) + StreamHeader::parse (src/xz/header.rs:20-51)
int xz_walk(milzma_ctx* ctx, const uint8_t* in, size_t in_len, const PayloadFn& decode, milzma_output* o,
            size_t out_hint = 0, OutBuf* prefilled = nullptr) {
  static const uint8_t kMagic[6] = {0xFD, 0x37, 0x7A, 0x58, 0x5A, 0x00};
  out_reset(o);
  Cursor c{in, 0, in_len};
  OutBuf output;
  if (prefilled && prefilled->p) {  // (the file's buffer with its blocks' payloads already in place: see OutBuf::append)
    output.p = prefilled->p;
    output.cap = prefilled->cap;
    output.moved = prefilled->moved;
    output.hold_first = true;
    prefilled->p = nullptr;
    prefilled->cap = 0;
  }
  (void)output.reserve(std::max<size_t>(out_hint, 1));
  std::vector<Record> records;
  int r = MILZMA_OK;
  auto done = [&](int rr) {
    o->in_consumed = c.pos;
    if (!output.p && !output.reserve(1)) return out_fail(o, MILZMA_INFRA_ERROR, "out of memory");
    o->data = output.p;  // ownership moves to the caller (milzma_free)
    o->len = output.n;
    output.p = nullptr;
    return rr;
  };
  uint8_t tag[6];
  if (!c.exact(tag, 6)) return done(out_io_eof(o));
  if (memcmp(tag, kMagic, 6) != 0)
    return done(out_fail(o, MILZMA_XZ_ERROR, "Invalid XZ magic, expected [253, 55, 122, 88, 90, 0]"));
  uint32_t flags, crc, digest;
  {
    const size_t from = c.pos;
    if (!c.u16be(&flags)) return done(out_io_eof(o));
    digest = milzma_crc32(c.p + from, 2);
  }
  if (!c.u32le(&crc)) return done(out_io_eof(o));
  if (crc != digest)
    return done(out_fail(o, MILZMA_XZ_ERROR, "Invalid header CRC32: expected 0x%08x but got 0x%08x", crc, digest));
  int check = 0, footer_check = 0;
  if ((r = stream_flags_parse(flags, &check, o))) return done(r);

  size_t index_size = 0;
  for (;;) {
    const size_t start = c.pos;
    uint8_t hsize;
    if (!c.u8(&hsize)) return done(out_io_eof(o));
    if (hsize == 0) {
      if ((r = check_index(c, start, records, o))) return done(r);
      index_size = c.pos - start;
      break;
    }
    if ((r = read_block(ctx, c, start, output, check, records, hsize, decode, o))) return done(r);
  }
  if (!c.u32le(&crc)) return done(out_io_eof(o));
  {
    const size_t from = c.pos;
    uint32_t backward;
    if (!c.u32le(&backward)) return done(out_io_eof(o));
    const uint32_t expect = uint32_t((backward + 1u) << 2);
    if (uint32_t(index_size) != expect)
      return done(out_fail(o, MILZMA_XZ_ERROR, "Invalid index size: expected %u but got %zu", expect, index_size));
    if (!c.u16be(&flags)) return done(out_io_eof(o));
    if ((r = stream_flags_parse(flags, &footer_check, o))) return done(r);
    if (footer_check != check)
      return done(out_fail(o, MILZMA_XZ_ERROR,
                           "Flags in header (StreamFlags { check_method: %s }) does not match footer (StreamFlags { "
                           "check_method: %s })",
                           check_name(check), check_name(footer_check)));
    digest = milzma_crc32(c.p + from, c.pos - from);
  }
  if (crc != digest)
    return done(out_fail(o, MILZMA_XZ_ERROR, "Invalid footer CRC32: expected 0x%08x but got 0x%08x", crc, digest));
  if (!c.exact(tag, 2)) return done(out_io_eof(o));
  if (tag[0] != 0x59 || tag[1] != 0x5A) return done(out_fail(o, MILZMA_XZ_ERROR, "Invalid footer magic, expected [89, 90]"));
  if (!c.eof()) return done(out_fail(o, MILZMA_XZ_ERROR, "Unexpected data after last XZ block"));
  return done(MILZMA_OK);
}

// On-demand payload decode: one unit, reader limited only by the end of the file.
PayloadFn live_decoder(milzma_ctx* ctx) {
  return [ctx](const uint8_t* in, size_t in_len, size_t cap_hint, Payload* p) {
    milzma_unit u;
    memset(&u, 0, sizeof u);
    u.kind = MILZMA_KIND_LZMA2;
    SingleDecode sd;
    const size_t hint = cap_hint ? cap_hint + 64 : std::max<size_t>(1 << 16, in_len * 6);
    if (!decode_single(ctx, u, in, in_len, hint, &sd)) return false;
    p->res = sd.res;
    sd.out.resize(size_t(std::min<uint64_t>(sd.res.out_len, sd.out.size())));
    p->own = std::move(sd.out);
    p->data = p->own.data();
    return true;
  };
}


size_t check_size(int check) {
  switch (check) {
    case CHECK_CRC32: return 4;
    case CHECK_CRC64: return 8;
    case CHECK_SHA256: return 32;
    default: return 0;
  }
}

// Bytes of staging (input + output) the batch paths may plan ahead for: MILZMA_PLAN_BUDGET (bytes), else three quarters
// of the device memory that is free right now.
size_t plan_budget(milzma_ctx* ctx) {
  if (const char* e = env_get("MILZMA_PLAN_BUDGET")) return size_t(strtoull(e, nullptr, 0));
  size_t free_b = 0, total_b = 0;
  if (!ctx || hipSetDevice(ctx->device) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) return size_t(1) << 32;
  return (free_b / 4 * 3) / std::max(1u, ctx->budget_share) + ctx->in.cap + ctx->out.cap;
}

// Best-effort parse of footer + Index.  Any oddity => false (the exact walk then decodes on
// demand and reports whatever the reference would).
bool plan_from_index(const uint8_t* in, size_t n, std::vector<PlannedBlock>* blocks) {
  blocks->clear();
  if (n < 12 + 12 || (n & 3)) return false;
  if (in[n - 2] != 0x59 || in[n - 1] != 0x5A) return false;
  Cursor f{in, n - 12, n};
  uint32_t crc, backward, flags;
  if (!f.u32le(&crc) || !f.u32le(&backward) || !f.u16be(&flags)) return false;
  if (milzma_crc32(in + n - 8, 6) != crc) return false;
  if ((flags >> 8) != 0) return false;
  const int check = int(flags & 0xFF);
  if (check != CHECK_NONE && check != CHECK_CRC32 && check != CHECK_CRC64) return false;
  const uint64_t index_size = (uint64_t(backward) + 1) << 2;
  if (index_size + 24 > n) return false;
  const size_t index_start = n - 12 - size_t(index_size);
  Cursor c{in, index_start, n - 12};
  uint8_t tag;
  if (!c.u8(&tag) || tag != 0) return false;
  uint64_t num;
  if (get_multibyte(c, &num) || num > (n >> 2)) return false;
  size_t pos = 12;
  for (uint64_t i = 0; i < num; i++) {
    uint64_t unpadded, unpacked;
    if (get_multibyte(c, &unpadded) || get_multibyte(c, &unpacked)) return false;
    if (pos >= index_start || unpadded > index_start - pos) return false;
    const uint8_t hsize_byte = in[pos];
    if (hsize_byte == 0) return false;
    const size_t hsize = (size_t(hsize_byte) + 1) << 2;  // whole header incl. size byte and CRC32
    if (uint64_t(hsize) + check_size(check) > unpadded) return false;
    if (unpacked > MILZMA_MAX_UNIT_BYTES) return false;
    // An LZMA2 chunk is at least 11 bytes (6 header + 5 range-coder init) and yields at most 2 MiB: an Index that promises
    // more than that per payload byte is wrong, and believing it would reserve memory for it.
    const uint64_t payload = unpadded - hsize - check_size(check);
    if (unpacked > (payload / 11 + 1) * (uint64_t(2) << 20)) return false;
    blocks->push_back(PlannedBlock{pos + hsize, size_t(unpadded) - hsize - check_size(check), unpacked});
    pos += size_t((unpadded + 3) & ~uint64_t(3));
  }
  return pos == index_start;
}

MILZMA_HOST_NS_END

int milzma_xz_decompress_batch_impl(milzma_ctx* ctx, uint32_t n, const uint8_t* const* ins, const size_t* in_lens,
                                          milzma_output* outs) {
  if (!ctx) return MILZMA_INFRA_ERROR;
  ctx->last_paths = 0;
  // 1. plan: every block the Index of a file names becomes one LZMA2 unit of a single launch
  struct Ref {
    uint32_t file;
    size_t data_off;
    size_t blk_off;    // where the block's output starts in its file's output (sum of the Index's sizes of the blocks before it)
    size_t unpacked;   // the Index's size of the block's output
  };
  std::vector<milzma_unit> units;
  std::vector<Ref> refs;
  size_t in_total = 0, out_total = 0;
  std::vector<size_t> file_in_off(n, 0), file_out_hint(n, 0);
  std::vector<uint8_t> planned(n, 0);
  const size_t budget = plan_budget(ctx);
  for (uint32_t i = 0; i < n; i++) {
    std::vector<PlannedBlock> blocks;
    if (!plan_from_index(ins[i], in_lens[i], &blocks)) continue;
    size_t need = round_up(in_lens[i], 256);
    for (const auto& b : blocks) need += round_up(size_t(b.unpacked) + 16, 256);
    if (in_total + out_total + need > budget) continue;  // decoded on demand by the walk instead
    file_in_off[i] = in_total;
    for (const auto& b : blocks) {
      milzma_unit u;
      memset(&u, 0, sizeof u);
      u.kind = MILZMA_KIND_LZMA2;
      u.in_off = in_total + b.data_off;
      u.in_len = b.data_len;
      u.out_off = out_total;
      u.out_cap = round_up(size_t(b.unpacked) + 16, 256);
      out_total += size_t(u.out_cap);
      units.push_back(u);
      refs.push_back(Ref{i, b.data_off, file_out_hint[i], size_t(b.unpacked)});
      file_out_hint[i] += size_t(b.unpacked);
    }
    if (!blocks.empty()) {
      planned[i] = 1;
      in_total += round_up(in_lens[i], 256);
    }
  }
  // 2. one launch for all planned blocks.  Input and output are staged through page-locked buffers (PCIe at
  //    link speed); the blocks' CRC-32 / CRC-64 are computed on the GPU while the output is still there, so
  //    the host never has to read the decoded bytes except to hand them to the caller.
  const uint32_t nu = uint32_t(units.size());
  std::vector<milzma_result> res(nu);
  const uint8_t* hout = nullptr;
  const uint8_t* parts = nullptr;
  ChunkedCopy d2h;
  // Streamed form (many blocks of about one size -- the usual .xz: 1 .. 8 MiB blocks): ONE time-sliced launch whose waves write
  // their output to the page-locked host buffer themselves, span by span (kernels.h), while a host thread copies every span that
  // has arrived to its place in the FILE's output buffer (block offsets follow from the Index): when the kernel ends the files'
  // buffers are nearly complete and the walks below append without copying (OutBuf::append, prefilled_at).
  struct {
    size_t pitch = 0, span = 0;
    uint32_t spans = 0;
  } geo;
  std::vector<OutBuf> filebuf(n);
  StreamedSlot streamed_slot;
  bool streamed_done = false, streamed_direct = false;
  std::unordered_map<size_t, std::vector<uint8_t>> longer;   // blocks that came out LONGER than the Index says (their place holds only the Index's size)
  if (nu) {
    const char* const stream_env = env_get("MILZMA_STREAM");
    const bool off = stream_env && !strcmp(stream_env, "0");
    size_t max_cap = 0;
    for (const milzma_unit& u : units) max_cap = std::max(max_cap, size_t(u.out_cap));
    const size_t pitch = round_up(max_cap, 256);
    size_t min_units, min_bytes;
    bool ragged_ok = false;
    stream_minimum(&min_units, &min_bytes, &ragged_ok);
    if (ctx->use_fast && !off && nu >= min_units && out_total >= min_bytes && (ragged_ok || pitch * nu <= out_total + out_total / 4) &&
        in_total + pitch * nu <= budget && streamed_slot.try_take(ctx->device)) {
      size_t span = size_t(64) << 10;
      if (const char* e = env_get("MILZMA_SPAN")) span = std::max<size_t>(size_t(1) << 16, round_up(size_t(strtoull(e, nullptr, 0)), size_t(1) << 16));
      while ((pitch + span) / span + 1 > milzma_ctx::kMaxSpans) span *= 2;
      geo.pitch = pitch;
      geo.span = span;
      geo.spans = uint32_t((pitch + span + span - 1) / span);
      out_total = 0;
      for (milzma_unit& u : units) {
        u.out_off = out_total;
        out_total += pitch;
      }
    }
  }
  PinLease crc_parts_lease;   // (the walks of step 3 read the blocks' CRC parts out of pin_small until the call returns)
  if (nu) {
    // Decoding ahead is an optimisation: if its memory cannot be had (or anything else goes wrong here) the walk below
    // decodes every block on demand and each file still gets the reference's verdict.
    const auto ahead = [&]() -> bool {
      if (!hip_ok(ctx, hipSetDevice(ctx->device), "hipSetDevice")) return false;
      if (!crc_parts_lease.take(ctx->pin_small, "the .xz batch's CRC parts")) return false;
      const size_t parts_bytes = size_t(nu) * kCrcPartsBytes;
      if (!pin_reserve(ctx, ctx->pin_in, in_total) || !pin_reserve(ctx, ctx->pin_out, out_total) ||
          !pin_reserve(ctx, ctx->pin_small, parts_bytes) || !dev_reserve(ctx, ctx->in, in_total + 512) ||
          !dev_reserve(ctx, ctx->out, out_total + 512) || !dev_reserve(ctx, ctx->crc, parts_bytes))
        return false;
      uint8_t* hin = static_cast<uint8_t*>(ctx->pin_in.p);
      // the planned files, in groups: the gather of one group overlaps the transfer of the one before
      std::vector<uint32_t> pf;
      for (uint32_t i = 0; i < n; i++)
        if (planned[i]) pf.push_back(i);
      const size_t groups = std::min<size_t>(geo.spans ? 16 : 8, pf.size());
      std::vector<size_t> first(groups + 1), bounds(groups + 1);
      for (size_t g = 0; g <= groups; g++) {
        first[g] = pf.size() * g / groups;
        bounds[g] = g == groups ? in_total : file_in_off[pf[first[g]]];
      }
      const auto fill = [&](size_t g) {
        parallel_for(first[g + 1] - first[g], [&](size_t k0) {
          const uint32_t i = pf[first[g] + k0];
          memcpy(hin + file_in_off[i], ins[i], in_lens[i]);
        });
      };
      hipStream_t ws = work_stream(ctx);
      void* host_dev = nullptr;
      const bool stream_it = geo.spans && hipHostGetDevicePointer(&host_dev, ctx->pin_out.p, 0) == hipSuccess && ensure_progress(ctx);
      // The input goes up whole before the launch (gather of one group of files under the transfer of the one before).  The two-part
      // form the .lzma batches use (upload_leads / upload_rest) is there for MILZMA_TWO_PART=1: measured on the 16-core GPU boxes the
      // host's gather (~23 GB/s) is what both forms wait for, and with four blocks per file the second part came too late for the
      // decoders (profiles/r04_batch_api.txt).
      const char* const two_part_env = env_get("MILZMA_TWO_PART");
      const bool two_part = two_part_env && !strcmp(two_part_env, "1");
      if (!stream_it) (void)hipGetLastError();
      if (!(stream_it && two_part) && !staged_h2d(ctx, ctx->in.p, hin, bounds, fill)) return false;
      if (stream_it) {
        // the files' result buffers page-locked from the pool, every block written to its place by the waves themselves (kernels.h:
        // host_ptrs); without page-locked memory: ordinary buffers, filled from the staging buffer by a host thread
        std::atomic<int> alloc_failed{0};
        bool direct = pinned_results_wanted();
        if (direct) {
          parallel_for(n, [&](size_t i) {
            if (planned[i] && !filebuf[i].reserve_pinned(file_out_hint[i] + 512)) alloc_failed = 1;
          });
          if (alloc_failed) {
            for (OutBuf& b : filebuf) {
              milzma_free(b.p);
              b.p = nullptr;
              b.cap = 0;
            }
            alloc_failed = 0;
            direct = false;
          }
        }
        if (!direct)
          parallel_for(n, [&](size_t i) {
            if (planned[i] && !filebuf[i].reserve(std::max<size_t>(file_out_hint[i], 1))) alloc_failed = 1;
          });
        if (direct && !alloc_failed) {
          std::vector<uint64_t> ptrs(size_t(nu) * 2);   // (a block never writes beyond the size the Index gives it: the next block's place)
          for (uint32_t k = 0; k < nu; k++) {
            ptrs[2 * size_t(k)] = uint64_t(reinterpret_cast<uintptr_t>(filebuf[refs[k].file].p + refs[k].blk_off));
            ptrs[2 * size_t(k) + 1] = refs[k].unpacked;
          }
          if (!upload_host_ptrs(ctx, ptrs, ws)) alloc_failed = 1;
        }
        // the input in two parts (upload_leads / upload_rest): every block's first bytes before the launch, the files while it runs
        trace_mark(ctx, "streamed: leads");
        if (two_part && (alloc_failed || !upload_leads(ctx, units, [&](size_t k) { return ins[refs[k].file] + refs[k].data_off; }, ws))) {
          if (!staged_h2d(ctx, ctx->in.p, hin, bounds, fill)) return false;
          alloc_failed = 1;   // (falls through to the classic decode below)
        }
        if (!alloc_failed) {
          __atomic_store_n(&ctx->progress[milzma_ctx::kMaxSpans], 0u, __ATOMIC_RELEASE);
          ctx->stream_span = uint32_t(geo.span);
          ctx->stream_spans = direct ? 1 : geo.spans;   // (page-locked result buffers: nobody reads the counters mid-kernel -- host_files.cpp)
          ctx->stream_host = static_cast<uint8_t*>(host_dev);
          ctx->stream_ptrs = direct ? static_cast<const uint64_t*>(ctx->hostptrs.p) : nullptr;
          ctx->stream_in_host = two_part;
          trace_mark(ctx, "streamed: launch");
          const bool launched = milzma_decode_units_async_impl(ctx, units.data(), nu, ctx->in.p, ctx->out.p, ws, 0, nullptr) == MILZMA_OK;
          ctx->stream_span = ctx->stream_spans = 0;
          ctx->stream_host = nullptr;
          ctx->stream_ptrs = nullptr;
          ctx->stream_in_host = false;
          const bool rest = !two_part || upload_rest(ctx, hin, bounds, fill);
          trace_mark(ctx, "streamed: input complete");
          if (!rest) {
            if (launched) (void)milzma_decode_units_wait_impl(ctx, res.data());
            return false;
          }
          if (!launched) return false;
          if (ctx->stream_active) {
            std::atomic<bool> kernel_done{false};
            const uint8_t* pout = static_cast<const uint8_t*>(ctx->pin_out.p);
            std::thread consumer([&] {
              if (direct) return;   // (the waves put the blocks in place themselves)
              for (uint32_t sp = 0; sp < geo.spans; sp++) {
                while (__atomic_load_n(&ctx->progress[sp], __ATOMIC_ACQUIRE) < nu && !kernel_done.load(std::memory_order_acquire))
                  std::this_thread::sleep_for(std::chrono::microseconds(50));
                parallel_for(nu, [&](size_t k) {
                  const size_t phase = (k & 15u) * (geo.span >> 4), len = refs[k].unpacked;
                  const size_t lo = sp * geo.span > phase ? sp * geo.span - phase : 0, hi = std::min(len, (sp + 1) * geo.span - phase);
                  if (lo < hi) memcpy(filebuf[refs[k].file].p + refs[k].blk_off + lo, pout + size_t(units[k].out_off) + lo, hi - lo);
                });
              }
            });
            int wr;
            {
              JoinOnExit joined{consumer, kernel_done};
              wr = milzma_decode_units_wait_impl(ctx, res.data());
            }
            trace_mark(ctx, "streamed decode + placement: done");
            ctx->last_paths |= MILZMA_PATH_STREAMED | (two_part ? MILZMA_PATH_TWO_PART_INPUT : 0u);
            if (wr != MILZMA_OK) return false;
            streamed_done = true;
            streamed_direct = direct;
            // fetched whole from the device: the rare block LONGER than the Index says (the waves wrote no more than the Index's size to
            // its place), and a block that was decoded again in another launch class (that launch has no host destinations)
            std::vector<uint8_t> fetch(nu, 0);
            if (direct)
              for (uint32_t k = 0; k < nu; k++) fetch[k] = res[k].out_len > refs[k].unpacked;
            for (uint32_t k : ctx->promoted)
              if (k < nu) fetch[k] = 1;
            for (uint32_t k = 0; k < nu; k++)
              if (fetch[k] && res[k].status == MILZMA_ST_OK && res[k].out_len <= units[k].out_cap) {
                std::vector<uint8_t>& v = longer[k];
                v.resize(size_t(res[k].out_len));
                if (!v.empty() &&
                    !hip_ok(ctx, hipMemcpy(v.data(), static_cast<const uint8_t*>(ctx->out.p) + units[k].out_off, v.size(), hipMemcpyDeviceToHost),
                            "D2H block"))
                  return false;
              }
          } else if (milzma_decode_units_wait_impl(ctx, res.data()) != MILZMA_OK) {
            return false;
          } else {
            ctx->last_paths |= MILZMA_PATH_CLASSIC;
          }
        } else if (milzma_decode_units(ctx, units.data(), nu, ctx->in.p, ctx->out.p, res.data(), ws) != MILZMA_OK) {
          return false;
        } else {
          ctx->last_paths |= MILZMA_PATH_CLASSIC;
        }
      } else if (milzma_decode_units(ctx, units.data(), nu, ctx->in.p, ctx->out.p, res.data(), ws) != MILZMA_OK) {
        return false;
      } else {
        ctx->last_paths |= MILZMA_PATH_CLASSIC;
      }
      // (the decode leaves the units and the final results in ctx->units / ctx->results)
      if (streamed_done)   // the output is on the host already: only the blocks' CRC parts are still to come
        return hip_ok(ctx,
                      launch_crc_units(static_cast<const milzma_unit*>(ctx->units.p), nu, static_cast<const uint8_t*>(ctx->out.p),
                                       static_cast<const milzma_result*>(ctx->results.p), ctx->crc.p, ws),
                      "crc kernel launch") &&
               hip_ok(ctx, hipMemcpyAsync(ctx->pin_small.p, ctx->crc.p, parts_bytes, hipMemcpyDeviceToHost, ws), "D2H crc parts") &&
               hip_ok(ctx, hipStreamSynchronize(ws), "hipStreamSynchronize");
      return hip_ok(ctx,
                    launch_crc_units(static_cast<const milzma_unit*>(ctx->units.p), nu, static_cast<const uint8_t*>(ctx->out.p),
                                     static_cast<const milzma_result*>(ctx->results.p), ctx->crc.p, ws),
                    "crc kernel launch") &&
             hip_ok(ctx, hipMemcpyAsync(ctx->pin_small.p, ctx->crc.p, parts_bytes, hipMemcpyDeviceToHost, ws), "D2H crc parts") &&
             d2h.start_d2h(ctx, ctx->pin_out.p, ctx->out.p, out_total) &&  // in chunks: the walks below start on the first ones
             hip_ok(ctx, hipStreamSynchronize(ws), "hipStreamSynchronize");  // (the CRC parts; the output keeps coming)
    };
    if (ahead()) {
      hout = static_cast<const uint8_t*>(ctx->pin_out.p);
      parts = static_cast<const uint8_t*>(ctx->pin_small.p);
    }
  }
  // 3. the reference's walk per file (files in parallel on the host); a payload decoded ahead is used only
  //    if it is provably what an unlimited reader would have produced (clean status, consumed exactly the
  //    planned bytes, no take() window cut short by the planned end); everything else is decoded on demand
  //    (one GPU user at a time).
  std::vector<std::unordered_map<size_t, size_t>> by_off(n);
  for (size_t k = 0; k < refs.size(); k++) by_off[refs[k].file][refs[k].data_off] = k;
  std::vector<const uint8_t*> fb_base(n, nullptr);   // (the walks take the buffers over: their addresses, for the payloads inside them)
  for (uint32_t i = 0; i < n; i++) fb_base[i] = filebuf[i].p;
  const PayloadFn live_unlocked = live_decoder(ctx);
  const PayloadFn live = [&](const uint8_t* in, size_t in_len, size_t cap_hint, Payload* p) {
    std::lock_guard<std::mutex> lock(ctx->mu);
    // An on-demand decode launches on the null stream and writes ctx->out from offset 0 -- the buffer the chunked D2H of the
    // blocks decoded ahead is still reading on the (non-blocking) copy stream.  Let that copy finish first: from then on
    // every planned payload is on the host and ctx->out is free.
    if (hout && ctx->copy_stream && !hip_ok(ctx, hipStreamSynchronize(ctx->copy_stream), "hipStreamSynchronize")) return false;
    return live_unlocked(in, in_len, cap_hint, p);
  };
  parallel_for(n, [&](size_t i) {
    const uint8_t* base = ins[i];
    PayloadFn fn = [&, base, i](const uint8_t* in, size_t in_len, size_t cap_hint, Payload* p) {
      const auto& m = by_off[i];
      if (hout && in >= base && in < base + in_lens[i]) {
        const auto it = m.find(size_t(in - base));
        if (it != m.end()) {
          const size_t k = it->second;
          const milzma_result& r = res[k];
          if (r.status == MILZMA_ST_OK && r.in_consumed == units[k].in_len && !(r.chunks & 0x80000000u) &&
              r.out_len <= units[k].out_cap && (streamed_done || d2h.wait_until(size_t(units[k].out_off + r.out_len)))) {
            p->res = r;
            // streamed: the block sits at its place in the file's buffer (and, unless the waves wrote it there themselves, in the
            // staging buffer too); classic: in the staging buffer
            p->data = streamed_done ? fb_base[i] + refs[k].blk_off : hout + units[k].out_off;
            if (streamed_done) {
              const auto lit = longer.find(k);
              if (lit != longer.end()) {
                p->data = lit->second.data();              // (fetched whole: see above)
              } else if (r.out_len == refs[k].unpacked) {
                p->prefilled_at = refs[k].blk_off;
              } else if (r.out_len > refs[k].unpacked) {
                if (streamed_direct) return live(in, in_len, cap_hint, p);
                p->data = hout + units[k].out_off;         // (whole in the staging buffer)
              }
            }
            crc_fold(parts + k * kCrcPartsBytes, r.out_len, &p->crc32, &p->crc64);
            p->has_crc = true;
            return true;
          }
        }
      }
      return live(in, in_len, cap_hint, p);
    };
    xz_walk(ctx, ins[i], in_lens[i], fn, &outs[i], file_out_hint[i], streamed_done ? &filebuf[i] : nullptr);
  });
  return MILZMA_OK;
}

#ifdef MILZMA_TEST_HOOKS
// Test builds only (tests/san: this file under ASan + UBSan, no GPU): the XZ container walk -- header, blocks, index, footer, every
// check the reference makes -- with the caller's LZMA2 decoder standing in for the device.  fn returns a MILZMA_ST_* status and,
// for MILZMA_ST_OK, the payload's output (*out: malloc'd, taken over here) and how many input bytes it consumed.
typedef int (*milzma_test_lzma2_fn)(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len, size_t* consumed, void* user);
extern "C" int milzma_test_xz_walk(const uint8_t* in, size_t in_len, milzma_test_lzma2_fn fn, void* user, milzma_output* out) {
  try {
    const PayloadFn decode = [&](const uint8_t* p, size_t n, size_t, Payload* pl) {
      uint8_t* o = nullptr;
      size_t on = 0, used = 0;
      const int st = fn(p, n, &o, &on, &used, user);
      memset(&pl->res, 0, sizeof pl->res);
      pl->res.status = uint32_t(st);
      pl->res.out_len = pl->res.out_flushed = on;
      pl->res.in_consumed = used;
      if (o) pl->own.assign(o, o + on);
      free(o);
      pl->own.reserve(1);
      pl->data = pl->own.data();
      return true;
    };
    return xz_walk(nullptr, in, in_len, decode, out);
  } catch (const std::exception& e) {
    if (out) out_fail(out, MILZMA_INFRA_ERROR, "%s", e.what());
    return MILZMA_INFRA_ERROR;
  }
}
#endif

extern "C" int milzma_xz_decompress(milzma_ctx* ctx, const uint8_t* in, size_t in_len, milzma_output* out) {
  const uint8_t* ins[1] = {in};
  const size_t lens[1] = {in_len};
  const int r = milzma_xz_decompress_batch(ctx, 1, ins, lens, out);
  return r != MILZMA_OK ? r : out->kind;
}

