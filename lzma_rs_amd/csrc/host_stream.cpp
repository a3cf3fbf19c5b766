// host_stream.cpp -- push-mode .lzma decoding for a BATCH of streams: lzma_rs::decompress::Stream (feature `stream`,
// src/decode/stream.rs) on top of fed input (MILZMA_DECODE_FEED).  See host_internal.h for the map of the host side.
//
// One `milzma_streams` is n independent `Stream<Vec<u8>>` objects that share a context of their own (a batch of n places in its
// parking lot): a write call appends bytes to some of them and runs ONE resuming launch in which every stream that got bytes takes
// another turn -- streams whose header has just become complete start in that launch (MILZMA_KIND_START), streams that got nothing
// stay parked (MILZMA_KIND_HOLD).  What the reference does per stream and call, and where it is restated here:
//   * State::Header (stream.rs:228-303): bytes are kept until the .lzma header and the range coder's five start bytes are there; a
//     property byte >= 225 fails the write that delivers it;
//   * State::Data (stream.rs:305-316, lzma.rs:435-524 in ProcessingMode::Partial): symbols are decoded while at least
//     MAX_REQUIRED_INPUT = 20 bytes are at hand -- the kernel's FEED margin is that very number --, and with fewer as far as a TRIAL run
//     shows them complete; the kernel does the same with a second pass over the tail of every view (decode_fast_asm.hip.h).  So a write
//     fails in the very call the reference's fails in, and the bytes a stream has not used belong to a symbol that is not complete yet;
//   * an END MARKER that ends a write's data (lzma.rs:493-495, :507-509): Partial mode merely leaves its loop at `Finished`; the stream
//     stays in State::Data with the marker's state, and bytes written later are decoded on from it.  The units of .lzma streams carry
//     MILZMA_KIND_PARTIAL: they park behind such a marker like a unit that needs input (decode_fast_asm.hip.h), so a later write -- good
//     bytes, garbage, another stream -- gets the crate's verdict, and finish() its Finish-mode pass from that state;
//   * finish (stream.rs:119-150): the last view runs with MILZMA_KIND_LAST_VIEW; with allow_incomplete what stops in front of an
//     incomplete symbol is a success with everything decoded so far;
//   * get_output (stream.rs:102-116): the sink holds every completed flush of the ring (lzbuffer.rs:264-267): milzma_streams_output.
#include "host_internal.h"

using namespace milzma;
using namespace milzma::host;

namespace {

struct One {
  // SIDE: a .lzma stream with more literal rows (lc + lp) than the batch's slab has per stream lives in a one-stream batch of its own
  // (`side`: a context and a slab for itself); every call on it is passed on
  enum St : uint8_t { HEADER, DATA, DONE, FAILED, SIDE } st = HEADER;
  milzma_streams* side = nullptr;
  milzma_options opt;
  std::vector<uint8_t> pending;   // HEADER: the header bytes so far; DATA: payload that has arrived and is not consumed yet
  size_t hdr_len = 0;
  uint64_t consumed = 0;           // payload bytes the decoder has taken
  bool started = false;            // its unit has been launched (it owns an output slice)
  std::string write_err;           // text of the io::Error of the write that failed
  uint64_t taken = 0;              // bytes of the last write the stream took (what Stream::write's Ok(n) add up to, stream.rs:324)
  size_t in_tmp = 0;               // bytes (pending[5 .. 5 + in_tmp)) that the crate holds in Stream.tmp behind a header it read through that
                                   // buffer: its next write() call's first input (stream.rs:312-317); the unit has not started yet
  size_t view = 0;                 // feed_round: only this much of `pending` is the unit's view (0: all of it)
  // The stream's RESULT buffer (page-locked, from the pool: what finish hands over), filled by the decoding waves themselves while they
  // decode -- every turn's bytes go over PCIe under the kernel (kernels.h: host_ptrs), finish has nothing left to copy.
  uint8_t* hbuf = nullptr;
  size_t hcap = 0;
  // For the length of one write call: bytes that follow `pending` but still lie in the CALLER's buffer (a running stream's data go from
  // there straight into the upload buffer; only what the decoder leaves unused is kept)
  const uint8_t* ext = nullptr;
  size_t ext_len = 0;
  size_t held() const { return pending.size() + ext_len; }
  // reader mode (MILZMA_STREAMS_AS_READER): what the failed decode itself said, for finish to hand over instead of Stream::finish's
  // "previous write error"
  bool has_fail = false;
  milzma_result fail;              // a decode error (with the reader position of the whole stream in in_consumed)
  milzma_output header_fail;       // a fatal header error
};

// the io::Error a failed Stream::write returns for a decode error: io::Error::new(Other, format!("{:?}", error)) (stream.rs:343-347);
// Debug of error::Error::LzmaError(String) is LzmaError("...")
std::string debug_lzma_error(const milzma_result& r, uint32_t kind) {
  char msg[400];
  milzma_result_message(&r, kind, msg, sizeof msg);
  const char* m = strchr(msg, ':');
  return std::string("LzmaError(\"") + (m ? m + 2 : msg) + "\")";
}

}  // namespace

struct milzma_streams {
  milzma_ctx* ctx = nullptr;   // its own: the parking lot of a context belongs to one batch
  uint32_t n = 0;
  std::vector<One> s;
  std::vector<milzma_unit> units;
  std::vector<milzma_result> res;
  DevBuf out;                  // the streams' output slices
  size_t out_used = 0;
  bool finished = false;
  uint32_t kind = MILZMA_KIND_RAW_LZMA;   // MILZMA_KIND_RAW_LZMA: .lzma files (header first, as the crate's Stream); MILZMA_KIND_LZMA2: raw LZMA2 streams
  bool as_reader = false;      // finish hands over what the ONE-SHOT call would (lzma_decompress over a reader that shows its input piece by piece)
  // the crate's Partial mode at an end marker (MILZMA_KIND_PARTIAL): .lzma streams in Stream mode.  (Reader mode stands for the one-shot
  // call, whose range decoder asks the READER whether it is at its end; LZMA2 streams end at their end byte.)
  bool partial() const { return kind == MILZMA_KIND_RAW_LZMA && !as_reader; }
  bool deliver = false;        // the waves write every stream's output into its result buffer while they decode (One::hbuf)
};

MILZMA_HIDDEN int milzma_streams_open_impl(milzma_ctx* ctx, uint32_t kind, uint32_t n, const milzma_options* options, milzma_streams** out);
MILZMA_HIDDEN int milzma_streams_write_impl(milzma_streams* S, uint32_t k, const uint32_t* idx, const void* const* data, const size_t* len,
                                           int32_t* status);
MILZMA_HIDDEN int milzma_streams_finish_impl(milzma_streams* S, milzma_output* outs);
MILZMA_HIDDEN void milzma_streams_close_impl(milzma_streams* S);

MILZMA_HOST_NS_BEGIN

static bool parked(const milzma_result& r) { return is_parked_result(r); }

// New slices for the streams in `want` (stream -> capacity), everything that has produced output moved along: one fresh buffer, packed.
static bool regrow(milzma_streams* S, const std::vector<std::pair<uint32_t, uint64_t>>& want) {
  milzma_ctx* ctx = S->ctx;
  std::vector<uint64_t> cap(S->n, 0);
  for (uint32_t i = 0; i < S->n; i++)
    if (S->s[i].started) cap[i] = S->units[i].out_cap;
  for (const auto& w : want) cap[w.first] = std::max<uint64_t>(cap[w.first], std::min<uint64_t>(round_up(size_t(w.second), 256), MILZMA_MAX_UNIT_BYTES));
  size_t total = 0;
  std::vector<uint64_t> off(S->n, 0), so, dof, ln;
  for (uint32_t i = 0; i < S->n; i++) {
    off[i] = total;
    total += round_up(size_t(cap[i]), 256);
  }
  DevBuf fresh;
  if (!dev_reserve(ctx, fresh, total + 512)) return false;
  for (uint32_t i = 0; i < S->n; i++) {
    const uint64_t have = S->s[i].started ? std::min<uint64_t>(S->res[i].out_len, S->units[i].out_cap) : 0;
    if (have) {
      so.push_back(S->units[i].out_off);
      dof.push_back(off[i]);
      ln.push_back(have);
    }
  }
  if (!so.empty() && move_units_impl(ctx, uint32_t(so.size()), S->out.p, so.data(), fresh.p, dof.data(), ln.data(), work_stream(ctx)) != MILZMA_OK) {
    dev_release(fresh);
    return false;
  }
  dev_release(S->out);
  S->out = fresh;
  S->out_used = total;
  for (uint32_t i = 0; i < S->n; i++) {
    S->units[i].out_off = off[i];
    S->units[i].out_cap = cap[i];
  }
  if (S->deliver) {   // the result buffers follow the slices (what has been delivered moves along: rare -- the first guess is generous)
    std::vector<uint32_t> who;
    std::vector<size_t> sizes;
    for (const auto& w : want)
      if (S->s[w.first].hcap < cap[w.first]) {
        who.push_back(w.first);
        sizes.push_back(size_t(cap[w.first]));
      }
    std::vector<uint8_t*> fresh_bufs(who.size(), nullptr);
    const bool got = who.empty() || out_alloc_many(sizes.data(), who.size(), true, fresh_bufs.data());   // (the pool under one lock)
    if (got)
      parallel_for(who.size(), [&](size_t w) {
        One& o = S->s[who[w]];
        const size_t have = o.hbuf ? size_t(std::min<uint64_t>(S->res[who[w]].out_len, o.hcap)) : 0;
        if (have) memcpy(fresh_bufs[w], o.hbuf, have);
        if (o.hbuf) milzma_free(o.hbuf);
        o.hbuf = fresh_bufs[w];
        o.hcap = sizes[w];
      });
    else
      for (uint8_t* b : fresh_bufs)
        if (b) milzma_free(b);
    if (!got) {   // page-locked memory has run out: from here on finish copies (what was delivered so far is on the device too)
      S->deliver = false;
      for (One& o : S->s) {
        if (o.hbuf) milzma_free(o.hbuf);
        o.hbuf = nullptr;
        o.hcap = 0;
      }
    }
  }
  return true;
}

// One turn for the streams in `active` (state DATA): their pending bytes are their views; `last`: the views end where the streams end.
// Streams that run out of room get larger slices and go on until they need input (or end).  Updates pending / consumed / res.
static bool feed_round(milzma_streams* S, std::vector<uint32_t> active, bool last) {
  milzma_ctx* ctx = S->ctx;
  if (!hip_ok(ctx, hipSetDevice(ctx->device), "hipSetDevice")) return false;   // (the calling thread may have been on another device)
  // slices for the streams that start now
  {
    std::vector<std::pair<uint32_t, uint64_t>> want;
    for (uint32_t i : active)
      if (!S->s[i].started) {
        const milzma_unit& u = S->units[i];
        const uint64_t plausible = std::max<uint64_t>(uint64_t(1) << 20, uint64_t(S->s[i].held()) * 1024);
        // (no size declared: 8 x what has arrived -- 16 x where the waves deliver into result buffers, which a regrow would have to copy)
        uint64_t cap = std::max<uint64_t>(1 << 16, uint64_t(S->s[i].held()) * (S->deliver ? 16 : 8));
        if (u.unpacked_size != MILZMA_SIZE_UNKNOWN) cap = std::min<uint64_t>(u.unpacked_size, plausible) + 512;
        if (u.memlimit < u.dict_size) cap = std::min<uint64_t>(cap, u.memlimit + 512);   // (ends at memlimit bytes: lzbuffer.rs:206-217)
        want.emplace_back(i, cap);
      }
    if (!want.empty()) {
      for (const auto& w : want) {
        S->s[w.first].started = true;
        S->units[w.first].out_cap = 0;
        memset(&S->res[w.first], 0, sizeof(milzma_result));
      }
      if (!regrow(S, want)) return false;
      for (const auto& w : want) S->units[w.first].kind = uint8_t(S->kind | MILZMA_KIND_START);   // (marks this round only)
    }
  }
  for (int round = 0; !active.empty(); round++) {
    if (round > 64) {
      ctx->err = "a stream keeps asking for room";
      return false;
    }
    std::vector<uint8_t> is_active(S->n, 0);
    size_t in_total = 0;
    for (uint32_t i : active) {
      is_active[i] = 1;
      const size_t view = S->s[i].view ? std::min(S->s[i].view, S->s[i].held()) : S->s[i].held();
      S->units[i].in_off = in_total;
      S->units[i].in_len = view;
      in_total += round_up(view, 64) + 64;
    }
    trace_mark(ctx, "streams: round begins");
    // (with headroom when they must grow: the next call's views are a few bytes longer -- what the streams left unused comes in front of
    //  the new data --, and re-pinning 400 MB for them was 40 ms of a write call)
    const size_t need = in_total + 512;
    if ((ctx->pin_in.cap < need && !pin_reserve(ctx, ctx->pin_in, need + need / 4)) || (ctx->in.cap < need && !dev_reserve(ctx, ctx->in, need + need / 4)))
      return false;
    // in four pieces: the host threads gather piece g + 1 while piece g crosses the link (the gather is 4 ms of a write call at configs[1]'s
    // size, the copy 7: profiles/r06_streams.txt)
    if (!hip_ok(ctx, hipSetDevice(ctx->device), "hipSetDevice")) return false;
    const size_t pieces = active.size() >= 64 ? 4 : 1;
    for (size_t g = 0; g < pieces; g++) {
      const size_t a0 = active.size() * g / pieces, a1 = active.size() * (g + 1) / pieces;
      if (a1 == a0) continue;
      parallel_for(a1 - a0, [&](size_t a) {
        const uint32_t i = active[a0 + a];
        One& o = S->s[i];
        uint8_t* dst = static_cast<uint8_t*>(ctx->pin_in.p) + S->units[i].in_off;
        const size_t view = size_t(S->units[i].in_len), own = std::min(view, o.pending.size());
        if (own) memcpy(dst, o.pending.data(), own);
        if (view > own) memcpy(dst + own, o.ext, view - own);
      });
      const size_t lo = size_t(S->units[active[a0]].in_off), hi = a1 < active.size() ? size_t(S->units[active[a1]].in_off) : in_total;
      if (!hip_ok(ctx, hipMemcpyAsync(static_cast<uint8_t*>(ctx->in.p) + lo, static_cast<const uint8_t*>(ctx->pin_in.p) + lo, hi - lo,
                                      hipMemcpyHostToDevice, work_stream(ctx)), "H2D views")) {
        (void)hipStreamSynchronize(work_stream(ctx));   // (the pieces already queued read pin_in)
        return false;
      }
    }
    trace_mark(ctx, "streams: views gathered");
    for (uint32_t i = 0; i < S->n; i++) {
      uint8_t k = S->units[i].kind & (MILZMA_KIND_START | 0x0Fu);
      if (!is_active[i]) {
        k &= uint8_t(~MILZMA_KIND_START);
        if (parked(S->res[i])) k |= MILZMA_KIND_HOLD;
      } else if (last) {
        k |= MILZMA_KIND_LAST_VIEW;
      } else if (S->partial()) {
        k |= MILZMA_KIND_PARTIAL;
      }
      S->units[i].kind = k;
    }
    if (S->deliver) {   // every unit's result buffer and its size, for the waves (address 0: the output stays on the device)
      std::vector<uint64_t> ptrs(size_t(S->n) * 2, 0);
      for (uint32_t i = 0; i < S->n; i++) {
        ptrs[2 * size_t(i)] = uint64_t(reinterpret_cast<uintptr_t>(S->s[i].hbuf));
        ptrs[2 * size_t(i) + 1] = S->s[i].hcap;
      }
      if (!ensure_progress(ctx) || !upload_host_ptrs(ctx, ptrs, work_stream(ctx))) return false;
      // ONE span: nobody waits on the counters, a turn's bytes go out when the turn ends (the scheduler's quantum: 96 .. 192 KiB).  (Cutting
      // the turns at staggered 64 KiB spans like the whole-file calls' streamed launches was measured and is worse here -- 0.400 s per run
      // against 0.338, one launch of 128 ms among them: profiles/r06_streams.txt.)
      ctx->stream_span = 0x80000000u;
      ctx->stream_spans = 1;
      ctx->stream_host = nullptr;
      ctx->stream_ptrs = static_cast<const uint64_t*>(ctx->hostptrs.p);
      ctx->stream_in_host = false;
      ctx->stream_feed = true;
    }
    const int rc = milzma_decode_units_impl(ctx, S->units.data(), S->n, ctx->in.p, S->out.p, S->res.data(), work_stream(ctx),
                                            MILZMA_DECODE_RESUME | MILZMA_DECODE_FEED);
    ctx->stream_span = ctx->stream_spans = 0;
    ctx->stream_ptrs = nullptr;
    ctx->stream_feed = false;
    trace_mark(ctx, "streams: launch done");
    for (uint32_t i = 0; i < S->n; i++) S->units[i].kind &= 0x0Fu;
    if (rc != MILZMA_OK) return false;
    std::vector<uint32_t> again;
    std::vector<std::pair<uint32_t, uint64_t>> want;
    for (uint32_t i : active) {
      One& o = S->s[i];
      const milzma_result& r = S->res[i];
      const uint64_t took = std::min<uint64_t>(r.in_consumed, S->units[i].in_len);
      if (o.ext) {   // what the decoder has not used -- of the view (in the upload buffer) and behind it -- is kept from here on
        const uint8_t* view = static_cast<const uint8_t*>(ctx->pin_in.p) + S->units[i].in_off;
        const size_t in_view = size_t(S->units[i].in_len), own = o.pending.size();
        std::vector<uint8_t> rest(view + took, view + in_view);
        if (own > in_view) rest.insert(rest.end(), o.pending.begin() + ptrdiff_t(in_view), o.pending.end());
        const size_t ext_used = in_view > own ? in_view - own : 0;
        rest.insert(rest.end(), o.ext + ext_used, o.ext + o.ext_len);
        o.pending.swap(rest);
        o.ext = nullptr;
        o.ext_len = 0;
      } else {
        o.pending.erase(o.pending.begin(), o.pending.begin() + ptrdiff_t(took));
      }
      o.consumed += took;
      if (o.view) o.view -= size_t(took);   // (0 only with everything taken: nothing of the view is left for another round)
      if (parked(r) && r.status == MILZMA_ST_OUT_FULL) {   // more room: what its progress predicts for the bytes at hand, at least 2.5 x
        const long double rate = (long double)(r.out_len + 1) / (long double)std::max<uint64_t>(1, o.consumed);
        uint64_t cap = uint64_t(rate * (long double)(o.consumed + o.pending.size()) * 1.25L) + 65536;
        cap = std::max<uint64_t>(cap, S->units[i].out_cap * 5 / 2 + 4096);
        if (S->units[i].unpacked_size != MILZMA_SIZE_UNKNOWN && S->units[i].unpacked_size + 512 > r.out_len + 274)
          cap = std::min<uint64_t>(cap, S->units[i].unpacked_size + 512);
        want.emplace_back(i, cap);
        again.push_back(i);
      }
    }
    if (!want.empty() && !regrow(S, want)) return false;
    active.swap(again);
    trace_mark(ctx, "streams: round done");
  }
  return true;
}

MILZMA_HOST_NS_END

MILZMA_HIDDEN int milzma_streams_open_impl(milzma_ctx* ctx, uint32_t kind, uint32_t n, const milzma_options* options, milzma_streams** out) {
  if (!ctx || !out) return MILZMA_INFRA_ERROR;
  *out = nullptr;
  const bool as_reader = (kind & MILZMA_STREAMS_AS_READER) != 0;
  kind &= ~uint32_t(MILZMA_STREAMS_AS_READER);
  if (kind != MILZMA_KIND_RAW_LZMA && kind != MILZMA_KIND_LZMA2) {
    ctx->err = "push-mode streams: kind must be MILZMA_KIND_RAW_LZMA (.lzma files) or MILZMA_KIND_LZMA2";
    return MILZMA_INFRA_ERROR;
  }
  if (!(ctx->use_fast && ctx->fast_spill)) {
    ctx->err = "push-mode streams need the asm kernel's launch classes (MILZMA_KERNEL / MILZMA_SPILL = generic set)";
    return MILZMA_INFRA_ERROR;
  }
  milzma_ctx* own = nullptr;
  if (milzma_create(ctx->device, &own) != MILZMA_OK) {
    ctx->err = std::string("push-mode streams: ") + milzma_last_error(nullptr);
    return MILZMA_INFRA_ERROR;
  }
  auto* S = new milzma_streams();
  S->ctx = own;
  S->kind = kind;
  S->as_reader = as_reader;
  S->n = n;
  S->s.resize(n);
  S->units.resize(n);
  S->res.resize(n);
  milzma_options dflt;
  milzma_default_options(&dflt);
  for (uint32_t i = 0; i < n; i++) {
    S->s[i].opt = options ? options[i] : dflt;
    memset(&S->units[i], 0, sizeof(milzma_unit));
    memset(&S->res[i], 0, sizeof(milzma_result));
  }
  // Literal rows for streams that join later: the batch's slab has ONE stride, fixed by its first launch of the class -- as many rows per
  // stream as a quarter of the free memory allows: every legal lc + lp (12: 6 MiB per stream) for a small batch (256 MiB in all), 8 (384
  // KiB per stream) otherwise.  A .lzma stream that asks for more gets a one-stream batch of its own (One::SIDE), so its header byte
  // costs nobody else anything.
  size_t free_b = 0, total_b = 0;
  uint32_t lclp = 12;
  while (lclp > 8 && (size_t(1536) << lclp) * std::max<uint32_t>(n, 1) > (size_t(256) << 20)) lclp--;
  if (hipSetDevice(own->device) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess)
    while (lclp > 4 && (size_t(1536) << lclp) * std::max<uint32_t>(n, 1) > free_b / 4) lclp--;
  own->slab_min_lclp = lclp;
  // Large batches: the waves deliver.  (A small batch's launches are over before a PCIe store has crossed the link; its few copies at
  // finish cost nothing.  MILZMA_PINNED_OUT=0 turns page-locked result buffers off altogether.)
  S->deliver = n >= 64 && pinned_results_wanted();
  *out = S;
  return MILZMA_OK;
}

// A stream of its own for stream i (One::SIDE): everything written so far goes there first.  status / write_err as from a write.
static bool to_side(milzma_streams* S, uint32_t i, size_t before, int32_t* st) {
  One& o = S->s[i];
  milzma_streams* side = nullptr;
  if (milzma_streams_open_impl(S->ctx, S->kind | (S->as_reader ? MILZMA_STREAMS_AS_READER : 0u), 1, &o.opt, &side) != MILZMA_OK) return false;
  const uint32_t zero = 0;
  if (o.ext) {   // (everything in one piece for the batch of its own)
    o.pending.insert(o.pending.end(), o.ext, o.ext + o.ext_len);
    o.ext = nullptr;
    o.ext_len = 0;
  }
  const void* data = o.pending.data();
  const size_t len = o.pending.size();
  int32_t s1 = MILZMA_OK;
  if (milzma_streams_write_impl(side, 1, &zero, &data, &len, &s1) != MILZMA_OK) {
    S->ctx->err = side->ctx->err;
    milzma_streams_close_impl(side);
    return false;
  }
  o.side = side;
  o.st = One::SIDE;
  o.pending.clear();
  o.pending.shrink_to_fit();
  o.write_err = side->s[0].write_err;
  o.taken = side->s[0].taken > before ? side->s[0].taken - before : 0;   // (of THIS write's bytes: `before` were buffered by earlier writes)
  *st = s1;
  return true;
}

// One CURSOR of a write: the bytes one call of the crate's Stream::write hands to process_stream as its `input` (stream.rs:305-319) -- the
// written data, or, in front of it, what a header read through Stream.tmp had left in that buffer (stream.rs:312-317: a call of its
// own).  The unit's view is what it holds un-decoded (the bytes of an incomplete symbol: the crate's partial-input buffer; for a unit that
// starts, the range coder's five start bytes) + the cursor's `data` bytes; `view` > 0: only that much of what is pending (the rest
// belongs to the next cursor).
struct WriteCursor {
  uint32_t j, i;
  size_t data, view;
};

// The cursors in `cur` take their turn (one feed_round), then every stream's verdict: its state, st[j] / write_err, and refused[j] = how
// many of the cursor's bytes the crate's write() would NOT have taken (> 0: write_all's ErrorKind::WriteZero).  A stream that reaches its
// declared size takes nothing more -- but what the crate had moved into its partial-input buffer by then is taken (lzma.rs:420-433,
// :457-495): on that path (entered where a cursor ends inside a symbol, so: whenever a cursor begins with such bytes at hand) every
// iteration first fills the buffer to 20 bytes, counted from the symbol's first byte; off it, the reader stands at the stream's last byte.
static bool run_cursors(milzma_streams* S, const std::vector<WriteCursor>& cur, std::vector<int32_t>& st, std::vector<uint64_t>* refused) {
  if (cur.empty()) return true;
  std::vector<uint32_t> active;
  std::vector<size_t> total(cur.size()), begin(cur.size()), behind(cur.size());
  std::vector<uint8_t> partial_path(cur.size());
  for (size_t c = 0; c < cur.size(); c++) {
    One& o = S->s[cur[c].i];
    total[c] = cur[c].view ? std::min(cur[c].view, o.held()) : o.held();
    behind[c] = o.held() - total[c];
    begin[c] = total[c] - std::min(cur[c].data, total[c]);
    partial_path[c] = S->partial() && o.started && begin[c] > 0;   // (a unit that starts holds the five start bytes, not a symbol's)
    o.view = cur[c].view;
    active.push_back(cur[c].i);
  }
  const bool ok = feed_round(S, active, false);
  for (const WriteCursor& c : cur) S->s[c.i].view = 0;
  if (!ok) return false;
  for (size_t c = 0; c < cur.size(); c++) {
    const uint32_t i = cur[c].i, j = cur[c].j;
    One& o = S->s[i];
    const milzma_result& r = S->res[i];
    if (parked(r)) continue;   // wants more input (or stands behind an end marker, MILZMA_KIND_PARTIAL): everything was taken
    if (r.status == MILZMA_ST_OK || r.status == MILZMA_ST_SIZE_MISMATCH) {
      // the stream has ended (its declared size is reached -- exactly, or overshot by its last match; reader mode / LZMA2: its end marker
      // / end byte has been read with nothing behind it): Partial mode leaves its loop; "Expected unpacked size ..." is a check of
      // finish() (lzma.rs:513-521, Finish mode only)
      o.st = One::DONE;
      const size_t left = o.pending.size() - std::min(behind[c], o.pending.size());   // bytes of the view the decoder has not used
      const size_t end = total[c] - std::min(left, total[c]);                          // the reader's position in the view
      size_t slurped = end;
      if (partial_path[c]) {
        const size_t top = end - std::min<size_t>(end, r.chunks);   // where the last symbol began (the kernel notes it: decode_fast_asm.hip.h, near_step)
        slurped = std::min<size_t>(top + 20, total[c]);
      }
      if (refused) (*refused)[j] += total[c] - std::min(total[c], std::max(slurped, begin[c]));
    } else {
      o.write_err = debug_lzma_error(r, S->kind);
      o.fail = r;
      o.has_fail = true;
      o.st = One::FAILED;
      st[j] = MILZMA_IO_ERROR;
    }
  }
  return true;
}

MILZMA_HIDDEN int milzma_streams_write_impl(milzma_streams* S, uint32_t k, const uint32_t* idx, const void* const* data, const size_t* len,
                                           int32_t* status) {
  if (!S) return MILZMA_INFRA_ERROR;
  milzma_ctx* ctx = S->ctx;
  ctx->err.clear();
  if (S->finished) {
    ctx->err = "the streams have been finished";
    return MILZMA_INFRA_ERROR;
  }
  if (k && (!idx || !data || !len)) {
    ctx->err = "null arguments";
    return MILZMA_INFRA_ERROR;
  }
  std::vector<uint8_t> seen(S->n, 0);
  for (uint32_t j = 0; j < k; j++) {
    if (idx[j] >= S->n || seen[idx[j]] || (len[j] && !data[j])) {
      ctx->err = "stream " + std::to_string(idx[j]) + ": not a stream of this batch, named twice in one call, or null data";
      return MILZMA_INFRA_ERROR;
    }
    seen[idx[j]] = 1;
  }
  struct DropExt {   // (whatever way the call ends: no stream keeps a pointer into the caller's buffers)
    milzma_streams* S;
    uint32_t k;
    const uint32_t* idx;
    ~DropExt() {
      for (uint32_t j = 0; j < k; j++) {
        One& o = S->s[idx[j]];
        if (o.ext) {   // (not taken over by a round: an infrastructure failure on the way -- the batch is closed by its owner; keep the bytes anyway)
          try {
            o.pending.insert(o.pending.end(), o.ext, o.ext + o.ext_len);
          } catch (const std::exception&) {
          }
          o.ext = nullptr;
          o.ext_len = 0;
        }
      }
    }
  } drop_ext{S, k, idx};
  trace_mark(ctx, "streams: write begins");
  std::vector<int32_t> st(k, MILZMA_OK);
  std::vector<uint64_t> refused(k, 0);   // bytes of the write the stream does not take (> 0: ErrorKind::WriteZero)
  std::vector<size_t> before(k, 0);      // bytes the stream held when the call began
  bool side_failed = false;              // an infrastructure failure of a stream that lives in a batch of its own
  // the bytes first, every stream's on its own (streams in Header / Data state keep them), on the host threads
  std::atomic<int> no_memory{0};
  parallel_for(k, [&](size_t j) {
    One& o = S->s[idx[j]];
    before[j] = o.pending.size();
    if (len[j] && o.st == One::DATA && o.started && o.in_tmp == 0) {   // a running stream: its data stay where they are until the upload
      o.ext = static_cast<const uint8_t*>(data[j]);
      o.ext_len = len[j];
      return;
    }
    if (len[j] >= 64 && o.st == One::HEADER && o.pending.empty()) {
      // a stream's first bytes, header and all in one piece: only what the header (<= 13 bytes) and the range coder's start (5) can take is
      // kept here, the rest stays with the caller until the upload
      const size_t head = S->kind == MILZMA_KIND_LZMA2 ? 0 : 18;
      const uint8_t* p = static_cast<const uint8_t*>(data[j]);
      try {
        o.pending.assign(p, p + head);
      } catch (const std::exception&) {
        no_memory.store(1);
        return;
      }
      o.ext = p + head;
      o.ext_len = len[j] - head;
      return;
    }
    if (len[j] && (o.st == One::HEADER || o.st == One::DATA)) {   // (a SIDE stream's bytes are passed on below)
      const uint8_t* p = static_cast<const uint8_t*>(data[j]);
      try {   // (an exception must not leave a host thread)
        o.pending.insert(o.pending.end(), p, p + len[j]);
      } catch (const std::exception&) {
        no_memory.store(1);
      }
    }
  });
  if (no_memory.load()) {   // some streams have taken their bytes, some have not: the batch cannot go on
    S->finished = true;
    ctx->err = "out of memory while buffering the written bytes: the streams are closed";
    if (status)
      for (uint32_t j = 0; j < k; j++) status[j] = MILZMA_INFRA_ERROR;
    return MILZMA_INFRA_ERROR;
  }
  // What the crate's write() calls do with the data, cursor by cursor.  `first`: what Stream.tmp holds from a header that was read through it
  // (a cursor of its own, in front of the data); `then`: the data.
  std::vector<WriteCursor> first, then;
  for (uint32_t j = 0; j < k; j++) {
    const uint32_t i = idx[j];
    One& o = S->s[i];
    o.write_err.clear();
    o.taken = 0;
    if (len[j] == 0) continue;   // (write_all of nothing calls nobody: std::io::Write::write_all)
    switch (o.st) {
      case One::FAILED:
        // Stream.state is None (taken by the write that failed, stream.rs:230): write() does nothing and returns Ok(input.position()) =
        // Ok(0) (stream.rs:324), which write_all turns into ErrorKind::WriteZero -- a dead stream refuses its bytes, it does not swallow them
        refused[j] = len[j];
        break;
      case One::SIDE: {
        const uint32_t zero = 0;
        if (milzma_streams_write_impl(o.side, 1, &zero, &data[j], &len[j], &st[j]) != MILZMA_OK) {
          ctx->err = o.side->ctx->err;
          side_failed = true;
        }
        o.write_err = o.side->s[0].write_err;
        refused[j] = len[j] - std::min<uint64_t>(len[j], o.side->s[0].taken);
        if (st[j] == MILZMA_IO_ERROR && refused[j] > 0 && o.write_err == "failed to write whole buffer") st[j] = MILZMA_OK;   // (said again below)
        break;
      }
      case One::DONE:
        if (S->as_reader && S->kind == MILZMA_KIND_RAW_LZMA && S->units[i].unpacked_size == MILZMA_SIZE_UNKNOWN && S->res[i].status == MILZMA_ST_OK) {
          // reader mode, a stream that ended with its END MARKER where a piece ended -- and the reader shows more: the one-shot call asks
          // the reader itself whether it is at its end (is_finished_ok, rangecoder.rs:49-52) and fails (lzma.rs:378-380)
          milzma_result r = S->res[i];
          r.status = MILZMA_ST_MARKER_TRAILING;
          r.out_flushed = r.out_len / S->units[i].dict_size * S->units[i].dict_size;   // (what the ring had flushed: lzbuffer.rs:264-267)
          o.fail = r;
          o.has_fail = true;
          o.write_err = debug_lzma_error(r, S->kind);
          o.st = One::FAILED;
          st[j] = MILZMA_IO_ERROR;
          break;
        }
        // the declared size is reached (an LZMA2 stream: its end byte is read; reader mode: the stream has ended): write() takes nothing,
        // write_all reports ErrorKind::WriteZero (lzma.rs:441-445; tests/lzma.rs:71-87).  (A .lzma stream in Stream mode is never DONE
        // by its end marker: it is parked behind it and decodes on, MILZMA_KIND_PARTIAL.)
        refused[j] = len[j];
        break;
      case One::HEADER: {
        if (S->kind == MILZMA_KIND_LZMA2) {   // no header: the stream begins with its first packet
          milzma_unit u;
          memset(&u, 0, sizeof u);
          u.kind = MILZMA_KIND_LZMA2;
          S->units[i] = u;
          o.st = One::DATA;
          then.push_back({j, i, o.held(), 0});
          break;
        }
        // The crate reads the header from the written data itself -- or, once an earlier write has left bytes in Stream.tmp, from THAT
        // buffer, filled up to its 18 bytes (stream.rs:233-270): what it then holds behind the header and the range coder's five start
        // bytes (up to 8 bytes, behind a five-byte header: UnpackedSize::UseProvided) stays there and is the next write() call's first input.
        const bool via_tmp = before[j] > 0;
        const size_t shown = via_tmp ? std::min<size_t>(o.pending.size(), 18) : o.pending.size();
        milzma_unit u;
        size_t hl = 0;
        milzma_output ho;
        const int hr = milzma_lzma_read_header(o.pending.data(), shown, &o.opt, &u, &hl, &ho);
        if (hr == MILZMA_HEADER_TOO_SHORT) break;   // need more data, try again later (stream.rs:185)
        if (hr != MILZMA_OK) {                      // fatal: LzmaError(s) => io::Error::new(Other, s) (stream.rs:291-299)
          const char* m = strchr(ho.msg, ':');
          o.write_err = m ? m + 2 : ho.msg;
          o.header_fail = ho;
          o.st = One::FAILED;
          st[j] = MILZMA_IO_ERROR;
          break;
        }
        if (shown - hl < 5) break;                  // RangeDecoder::new needs five bytes: Header again (stream.rs:176-180)
        if (uint32_t(u.lc) + u.lp > 3u && uint32_t(u.lc) + u.lp > (ctx->slab_live ? ctx->slab_lclp : ctx->slab_min_lclp) && S->n > 1) {
          if (!to_side(S, i, before[j], &st[j])) side_failed = true;   // more literal rows than the batch's slab has per stream
          refused[j] = len[j] - std::min<uint64_t>(len[j], o.taken);
          if (st[j] == MILZMA_IO_ERROR && refused[j] > 0 && o.write_err == "failed to write whole buffer") st[j] = MILZMA_OK;   // (said again below)
          break;
        }
        o.hdr_len = hl;
        o.pending.erase(o.pending.begin(), o.pending.begin() + ptrdiff_t(hl));
        u.out_off = u.out_cap = 0;
        S->units[i] = u;
        o.st = One::DATA;
        const size_t in_tmp = via_tmp ? shown - hl - 5 : 0;      // bytes Stream.tmp keeps behind the header and the five start bytes
        const size_t rest = o.held() - 5 - in_tmp;               // what write_all hands to the next write() call
        if (in_tmp == 0) {
          then.push_back({j, i, rest, 0});
        } else if (rest == 0) {
          o.in_tmp = in_tmp;   // nothing is decoded yet: Stream.tmp is looked at by the next write (stream.rs:312-317), or by finish
        } else {
          first.push_back({j, i, in_tmp, 5 + in_tmp});
          then.push_back({j, i, rest, 0});
        }
        break;
      }
      case One::DATA:
        if (o.in_tmp) {   // Stream.tmp first (a header read through it had left bytes there): a write() call of its own
          first.push_back({j, i, o.in_tmp, 5 + o.in_tmp});
          o.in_tmp = 0;
        }
        then.push_back({j, i, len[j], 0});
        break;
    }
  }
  bool ok = !side_failed && run_cursors(S, first, st, nullptr);   // (what Stream.tmp held and the stream did not take is dropped: stream.rs:316)
  if (ok) {
    std::vector<WriteCursor> go;
    for (const WriteCursor& c : then) {
      One& o = S->s[c.i];
      if (o.st == One::DONE) refused[c.j] = c.data;   // it ended inside Stream.tmp's bytes: read_data returns at once, nothing is taken
      if (o.st == One::DATA) go.push_back(c);         // (else: failed there)
    }
    ok = run_cursors(S, go, st, &refused);
  }
  if (!ok) {
    if (status)
      for (uint32_t j = 0; j < k; j++) status[j] = MILZMA_INFRA_ERROR;
    return MILZMA_INFRA_ERROR;
  }
  for (uint32_t j = 0; j < k; j++) {
    One& o = S->s[idx[j]];
    if (st[j] == MILZMA_OK && refused[j] > 0) {
      o.write_err = "failed to write whole buffer";
      st[j] = MILZMA_IO_ERROR;
    }
    o.taken = len[j] - std::min<uint64_t>(len[j], refused[j]);
  }
  if (status)
    for (uint32_t j = 0; j < k; j++) status[j] = st[j];
  trace_mark(ctx, "streams: write done");
  return MILZMA_OK;
}

MILZMA_HIDDEN int milzma_streams_finish_impl(milzma_streams* S, milzma_output* outs) {
  if (!S || !outs) return MILZMA_INFRA_ERROR;
  milzma_ctx* ctx = S->ctx;
  ctx->err.clear();
  if (S->finished) {
    ctx->err = "the streams have been finished";
    return MILZMA_INFRA_ERROR;
  }
  S->finished = true;
  for (uint32_t i = 0; i < S->n; i++) out_reset(&outs[i]);
  std::vector<uint32_t> active;
  for (uint32_t i = 0; i < S->n; i++)
    if (S->s[i].st == One::DATA) active.push_back(i);
  // (with allow_incomplete the reference skips the last pass altogether -- what it has is everything a trial run found complete; here
  //  that is what the last view decodes before it stops in front of the incomplete symbol)
  if (!active.empty() && !feed_round(S, active, true)) {
    for (uint32_t i = 0; i < S->n; i++) infra(ctx, &outs[i]);
    return MILZMA_INFRA_ERROR;
  }
  int worst = MILZMA_OK;
  // The streams' result buffers have been filled by the decoding waves, turn by turn, under the kernels of the write calls
  // (milzma_streams::deliver): nothing is left to copy.  Where that is off (small batches, no page-locked memory): every stream's bytes go
  // from its slice into a buffer from the pool, all copies queued on the work stream, one wait.
  const bool pinned = S->n >= 64 && pinned_results_wanted();
  std::vector<milzma_result> fin(S->n);
  std::vector<uint8_t> has(S->n, 0), delivered(S->n, 0);
  if (!hip_ok(ctx, hipSetDevice(ctx->device), "hipSetDevice")) {
    for (uint32_t i = 0; i < S->n; i++) infra(ctx, &outs[i]);
    return MILZMA_INFRA_ERROR;
  }
  parallel_for(S->n, [&](size_t ii) {   // (taking thousands of buffers from the pool -- or pinning them, the first time -- on the host threads)
    const uint32_t i = uint32_t(ii);
    One& o = S->s[i];
    milzma_output* out = &outs[i];
    if (o.st == One::SIDE) return;   // (below: a batch of its own)
    if (o.st == One::HEADER) {
      if (S->as_reader) return;   // (done below, one after the other: it uses the context)
      if (!o.pending.empty()) out_fail(out, MILZMA_LZMA_ERROR, "failed to read header");   // stream.rs:122-128
      return;
    }
    if (o.st == One::FAILED && !(S->as_reader && o.has_fail)) {
      if (S->as_reader) {   // the header's own error, reader position included
        *out = o.header_fail;
        out->data = nullptr;
        out->len = 0;
        return;
      }
      out_fail(out, MILZMA_LZMA_ERROR, "can't finish stream because of previous write error");   // stream.rs:144-148
      return;
    }
    milzma_result r = o.st == One::FAILED ? o.fail : S->res[i];
    const bool incomplete_ok =
        o.opt.allow_incomplete && (r.status == MILZMA_ST_INPUT_EOF || r.status == MILZMA_ST_MATCH_DIST_DICT || r.status == MILZMA_ST_MATCH_DIST_OUT ||
                                   r.status == MILZMA_ST_SIZE_MISMATCH);
    if (incomplete_ok) {   // stops in front of a symbol a trial run would not get through (lzma.rs:401-417), no end-of-stream checks: Ok
      r.status = MILZMA_ST_OK;
      r.out_flushed = r.out_len;
    }
    const size_t visible = size_t(std::min<uint64_t>(r.out_flushed, S->units[i].out_cap));
    if (S->deliver && o.hbuf && visible <= o.hcap) {   // the waves have filled it while they decoded: handed over as it is
      out->data = o.hbuf;
      o.hbuf = nullptr;
      o.hcap = 0;
      delivered[i] = 1;
    } else {
      out->data = out_alloc(visible, pinned && visible >= 4096);
    }
    if (!out->data) {
      out_fail(out, MILZMA_INFRA_ERROR, "out of memory");
      return;
    }
    out->len = visible;
    fin[i] = r;
    has[i] = 1;
  });
  if (S->as_reader)   // a stream that never got past its header: the one-shot call on the few bytes there are ("header too short: ...",
    for (uint32_t i = 0; i < S->n; i++)   //  "LZMA stream too short: ...") -- on the context, so not on the host threads above
      if (S->s[i].st == One::HEADER) {
        One& o = S->s[i];
        if (S->kind == MILZMA_KIND_LZMA2)
          milzma_lzma2_decompress_impl(ctx, o.pending.data(), o.pending.size(), &outs[i]);
        else
          milzma_lzma_decompress_impl(ctx, o.pending.data(), o.pending.size(), &o.opt, &outs[i]);
      }
  for (uint32_t i = 0; i < S->n; i++)
    if (S->s[i].st == One::SIDE && milzma_streams_finish_impl(S->s[i].side, &outs[i]) != MILZMA_OK) {
      ctx->err = S->s[i].side->ctx->err;
      worst = MILZMA_INFRA_ERROR;
    }
  bool copies_ok = true;
  for (uint32_t i = 0; i < S->n && copies_ok; i++)
    if (has[i] && outs[i].len && !delivered[i])
      copies_ok = hip_ok(ctx, hipMemcpyAsync(outs[i].data, static_cast<const uint8_t*>(S->out.p) + S->units[i].out_off, outs[i].len, hipMemcpyDeviceToHost,
                                             work_stream(ctx)),
                         "D2H output");
  copies_ok = hip_ok(ctx, hipStreamSynchronize(work_stream(ctx)), "hipStreamSynchronize") && copies_ok;
  for (uint32_t i = 0; i < S->n; i++) {
    if (!has[i]) {
      if (outs[i].kind == MILZMA_INFRA_ERROR) worst = MILZMA_INFRA_ERROR;
      continue;
    }
    milzma_output* out = &outs[i];
    if (!copies_ok) {
      milzma_free(out->data);
      out->data = nullptr;
      out->len = 0;
      infra(ctx, out);
      worst = MILZMA_INFRA_ERROR;
      continue;
    }
    out->in_consumed = S->s[i].hdr_len + size_t(S->s[i].consumed);   // the whole stream's reader position
    out->kind = milzma_result_message(&fin[i], S->kind, out->msg, sizeof out->msg);
  }
  return worst;
}

// Stream::get_output (stream.rs:102-116): what the sink of stream i holds right now.  State::Header: the sink as it was given -- nothing;
// State::Data: what the ring has flushed (LzCircularBuffer::append_literal, lzbuffer.rs:264-267: the whole buffer every time the cursor
// reaches dict_size; an LZMA2 stream's accumulating buffer: everything in front of its last dictionary reset); after a failed write the
// crate's state is None and get_output answers None: *has_sink = 0.
MILZMA_HIDDEN int milzma_streams_output_impl(milzma_streams* S, uint32_t stream, uint64_t offset, void* dst, size_t cap, uint64_t* sink_len,
                                            int32_t* has_sink) {
  if (!S || stream >= S->n) return MILZMA_INFRA_ERROR;
  milzma_ctx* ctx = S->ctx;
  ctx->err.clear();
  if (S->finished) {
    ctx->err = "the streams have been finished";
    return MILZMA_INFRA_ERROR;
  }
  const One& o = S->s[stream];
  if (o.st == One::SIDE) {
    const int rc = milzma_streams_output_impl(o.side, 0, offset, dst, cap, sink_len, has_sink);
    if (rc != MILZMA_OK) ctx->err = o.side->ctx->err;
    return rc;
  }
  uint64_t len = 0;
  if (o.st != One::HEADER && o.st != One::FAILED && o.started) {
    const milzma_result& r = S->res[stream];
    if (S->kind == MILZMA_KIND_RAW_LZMA)   // the ring's flushes (finish() brings the rest)
      len = r.out_len / std::max<uint32_t>(S->units[stream].dict_size, 0x1000u) * std::max<uint32_t>(S->units[stream].dict_size, 0x1000u);
    else   // (a parked unit's result carries its dict_base there; one that has ended, everything)
      len = r.out_flushed;
    len = std::min<uint64_t>(len, S->units[stream].out_cap);
  }
  if (has_sink) *has_sink = o.st == One::FAILED ? 0 : 1;
  if (sink_len) *sink_len = o.st == One::FAILED ? 0 : len;
  if (o.st == One::FAILED || !dst || cap == 0 || offset >= len) return MILZMA_OK;
  const size_t take = size_t(std::min<uint64_t>(cap, len - offset));
  if (!hip_ok(ctx, hipSetDevice(ctx->device), "hipSetDevice") ||
      !hip_ok(ctx, hipMemcpyAsync(dst, static_cast<const uint8_t*>(S->out.p) + S->units[stream].out_off + offset, take, hipMemcpyDeviceToHost, work_stream(ctx)),
              "D2H sink") ||
      !hip_ok(ctx, hipStreamSynchronize(work_stream(ctx)), "hipStreamSynchronize"))
    return MILZMA_INFRA_ERROR;
  return MILZMA_OK;
}

MILZMA_HIDDEN void milzma_streams_close_impl(milzma_streams* S) {
  if (!S) return;
  for (One& o : S->s) {
    if (o.side) {
      milzma_streams_close_impl(o.side);
      o.side = nullptr;
    }
    if (o.hbuf) {
      milzma_free(o.hbuf);
      o.hbuf = nullptr;
    }
  }
  if (S->ctx) {
    (void)hipSetDevice(S->ctx->device);
    dev_release(S->out);
    milzma_destroy(S->ctx);
  }
  delete S;
}

MILZMA_HIDDEN const char* milzma_streams_write_error_impl(const milzma_streams* S, uint32_t stream) {
  if (!S || stream >= S->n) return "";
  return S->s[stream].write_err.c_str();   // (a SIDE stream's text is copied over by every write)
}

MILZMA_HIDDEN uint64_t milzma_streams_write_taken_impl(const milzma_streams* S, uint32_t stream) {
  return S && stream < S->n ? S->s[stream].taken : 0;
}

MILZMA_HIDDEN const char* milzma_streams_last_error_impl(const milzma_streams* S) { return S && S->ctx ? S->ctx->err.c_str() : "no streams"; }
