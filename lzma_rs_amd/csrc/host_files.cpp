// host_files.cpp -- the whole-file entry points of .lzma and LZMA2 (see host_internal.h for the map of the host side)
#include "host_internal.h"

using namespace milzma;
using namespace milzma::host;

MILZMA_HOST_NS_BEGIN


void out_reset(milzma_output* o) {
  o->data = nullptr;
  o->len = 0;
  o->in_consumed = 0;
  o->kind = MILZMA_OK;
  o->msg[0] = 0;
}

int out_fail(milzma_output* o, int kind, const char* fmt, ...) {
  o->kind = kind;
  const int n = snprintf(o->msg, sizeof o->msg, "%s", kPrefix[kind]);
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(o->msg + n, sizeof o->msg - size_t(n), fmt, ap);
  va_end(ap);
  return kind;
}

int out_io_eof(milzma_output* o) { return out_fail(o, MILZMA_IO_ERROR, "%s", kEofMsg); }

bool out_set_data(milzma_output* o, const uint8_t* p, size_t n) {
  o->data = out_alloc(n);
  if (!o->data) return false;
  if (n) memcpy(o->data, p, n);
  o->len = n;
  return true;
}

int infra(milzma_ctx* ctx, milzma_output* o) {
  return out_fail(o, MILZMA_INFRA_ERROR, "%s", ctx ? ctx->err.c_str() : "no context");
}

size_t plan_budget(milzma_ctx* ctx);


inline bool is_parked(const milzma_result& r) { return r.status == MILZMA_ST_OUT_FULL && r.err_a == MILZMA_PARKED; }

// The next slice size for a unit that ran out of room: what its progress so far predicts for the whole stream (output per input
// byte x the input that is left) plus an eighth, at least twice and at most sixteen times what it had.
size_t grown_cap(const milzma_unit& u, const milzma_result& r) {
  const long double rate = (long double)(r.out_len + 1) / (long double)std::max<uint64_t>(1, r.in_consumed);
  const long double est = rate * (long double)u.in_len * 1.125L + 65536.0L;
  uint64_t cap = est > 1e18L ? UINT64_MAX / 2 : uint64_t(est);
  cap = std::max<uint64_t>(cap, 2 * u.out_cap + 4096);
  cap = std::min<uint64_t>(cap, 16 * u.out_cap + (uint64_t(1) << 20));
  return size_t(std::min<uint64_t>(round_up(size_t(cap), 256), MILZMA_MAX_UNIT_BYTES));
}

// Gives every unit of `parked` (indices into units / res: status PARKED) a larger slice in a FRESH output buffer, packed from offset
// 0 in list order, and moves what it has produced there (its dictionary); ctx->out becomes that buffer.  Nothing may still be
// reading the old one.  The descriptors are updated; the caller resumes the units with MILZMA_DECODE_RESUME.
bool regrow_parked(milzma_ctx* ctx, std::vector<milzma_unit>& units, const std::vector<milzma_result>& res,
                   const std::vector<uint32_t>& parked, hipStream_t ws, size_t* out_bytes) {
  std::vector<uint64_t> so(parked.size()), dof(parked.size()), ln(parked.size());
  std::vector<size_t> cap(parked.size());
  size_t total = 0;
  for (size_t j = 0; j < parked.size(); j++) {
    const uint32_t k = parked[j];
    cap[j] = grown_cap(units[k], res[k]);
    so[j] = units[k].out_off;
    dof[j] = total;
    ln[j] = std::min<uint64_t>(res[k].out_len, units[k].out_cap);
    total += cap[j];
  }
  DevBuf nb;
  if (!dev_reserve(ctx, nb, total + 512)) return false;
  if (move_units_impl(ctx, uint32_t(parked.size()), ctx->out.p, so.data(), nb.p, dof.data(), ln.data(), ws) != MILZMA_OK) {
    dev_release(nb);
    return false;
  }
  dev_release(ctx->out);
  ctx->out = nb;
  for (size_t j = 0; j < parked.size(); j++) {
    units[parked[j]].out_off = dof[j];
    units[parked[j]].out_cap = cap[j];
  }
  *out_bytes = total;
  return true;
}

// the per-unit host destinations of a streamed launch (kernels.h: host_ptrs) -> ctx->hostptrs
bool upload_host_ptrs(milzma_ctx* ctx, const std::vector<uint64_t>& ptrs, hipStream_t ws) {
  const size_t bytes = ptrs.size() * sizeof(uint64_t);
  return dev_reserve(ctx, ctx->hostptrs, bytes) && hip_ok(ctx, hipMemcpyAsync(ctx->hostptrs.p, ptrs.data(), bytes, hipMemcpyHostToDevice, ws), "H2D pointers") &&
         hip_ok(ctx, hipStreamSynchronize(ws), "hipStreamSynchronize");
}

// One streamed launch per device at a time: its persistent waves take the whole chip for the length of the call, so a second one
// (another context with a batch in flight: the *_batch_async pairs) would only fight it for the SIMDs -- that call runs the classic
// way instead (its copies ride under the first one's kernel: 2 x 4096 files in flight measured 11.9 GB/s streamed + streamed
// against 13.6 classic + classic, profiles/r04_batch_api.txt).
std::atomic<int> g_streamed_in_flight[64];

// streamed launches are for batches it pays for: at least this many units and output bytes, of about one size (one pitch for all
// slices: a ragged batch would reserve the largest unit's room for every unit).  MILZMA_STREAM_MIN="units,bytes[,1]": tests send small
// batches down the path; the third field lifts the one-size condition too (fuzzers: batches of anything).
void stream_minimum(size_t* units, size_t* bytes, bool* ragged_ok) {
  // (read at every call, not once per process: a test that sets it after the process's first batch call used to be ignored silently --
  //  the suite's streamed tests then ran the classic path; milzma_last_call_paths is what they assert on now)
  size_t mu = 256, mb = size_t(256) << 20;
  bool any = false;
  if (const char* e = env_get("MILZMA_STREAM_MIN")) {
    char* end = nullptr;
    mu = size_t(strtoull(e, &end, 0));
    if (end && *end == ',') {
      mb = size_t(strtoull(end + 1, &end, 0));
      if (end && *end == ',') any = strtoull(end + 1, nullptr, 0) != 0;
    }
  }
  *units = mu;
  *bytes = mb;
  if (ragged_ok) *ragged_ok = any;
}

bool pinned_results_wanted() {   // (read at every call: the tests flip it between batches)
  const char* e = env_get("MILZMA_PINNED_OUT");
  return !(e && !strcmp(e, "0"));
}




bool decode_single(milzma_ctx* ctx, milzma_unit u, const uint8_t* in, size_t in_len, size_t cap_hint, SingleDecode* sd) {
  if (!ctx) return false;
  size_t cap = std::max<size_t>(cap_hint, 4096);
  if (!hip_ok(ctx, hipSetDevice(ctx->device), "hipSetDevice") || !dev_reserve(ctx, ctx->in, in_len + 512)) return false;
  if (in_len && !hip_ok(ctx, hipMemcpy(ctx->in.p, in, in_len, hipMemcpyHostToDevice), "H2D input")) return false;
  std::vector<milzma_unit> units(1);
  std::vector<milzma_result> res(1);
  const std::vector<uint32_t> one{0};
  for (;;) {
    cap = std::min<size_t>(round_up(cap, 256), MILZMA_MAX_UNIT_BYTES);
    u.in_off = 0;
    u.in_len = in_len;
    u.out_off = 0;
    u.out_cap = cap;
    units[0] = u;
    if (!dev_reserve(ctx, ctx->out, cap + 512)) return false;
    uint32_t flags = MILZMA_DECODE_GROW;
    for (;;) {
      if (milzma_decode_units_impl(ctx, units.data(), 1, ctx->in.p, ctx->out.p, res.data(), work_stream(ctx), flags) != MILZMA_OK) return false;
      if (!is_parked(res[0]) || units[0].out_cap >= MILZMA_MAX_UNIT_BYTES) break;
      size_t bytes = 0;
      if (!regrow_parked(ctx, units, res, one, work_stream(ctx), &bytes)) return false;
      flags = MILZMA_DECODE_RESUME;
    }
    sd->res = res[0];
    cap = size_t(units[0].out_cap);
    if (sd->res.status == MILZMA_ST_OUT_FULL && !is_parked(sd->res) && cap < MILZMA_MAX_UNIT_BYTES) {
      cap = cap * 4;  // (not resumable: again from the first byte)
      continue;
    }
    if (is_parked(sd->res)) sd->res.err_a = 0;  // (at the largest slice there is: an ordinary OUT_FULL for whoever renders it)
    const size_t got = size_t(std::min<uint64_t>(sd->res.out_len, cap));  // only what was decoded travels back
    try {
      sd->out.resize(got);
    } catch (const std::bad_alloc&) {
      ctx->err = "out of host memory for a decoded stream";
      return false;
    }
    if (got && !hip_ok(ctx, hipMemcpy(sd->out.data(), ctx->out.p, got, hipMemcpyDeviceToHost), "D2H output")) return false;
    return true;
  }
}

MILZMA_HOST_NS_END

// ------------------------------------------------------------------------------------------
// .lzma: LzmaParams::read_header (src/decode/lzma.rs:96-161)
// ------------------------------------------------------------------------------------------

extern "C" int milzma_lzma_read_header(const uint8_t* in, size_t in_len, const milzma_options* opt, milzma_unit* unit,
                                       size_t* header_len, milzma_output* out) {
  milzma_options dflt;
  milzma_default_options(&dflt);
  if (!opt) opt = &dflt;
  milzma_output scratch;
  if (!out) out = &scratch;
  out_reset(out);
  Cursor c{in, 0, in_len};
  // (on failure the reader stands where the reference's stands: behind the bytes its read calls took -- all there were, for a short one)
  const auto fail_at = [&](int kind, const char* fmt, auto... a) {
    out->in_consumed = c.pos;
    return out_fail(out, kind, fmt, a...);
  };
  uint8_t props;
  if (!c.u8(&props)) return fail_at(MILZMA_HEADER_TOO_SHORT, "%s", kEofMsg);
  uint32_t pb = props;
  if (pb >= 225) return fail_at(MILZMA_LZMA_ERROR, "LZMA header invalid properties: %u must be < 225", pb);
  const uint32_t lc = pb % 9;
  pb /= 9;
  const uint32_t lp = pb % 5;
  pb /= 5;
  uint32_t dict;
  if (!c.u32le(&dict)) return fail_at(MILZMA_HEADER_TOO_SHORT, "%s", kEofMsg);
  if (dict < 0x1000) dict = 0x1000;
  uint64_t unpacked = MILZMA_SIZE_UNKNOWN;
  switch (opt->unpacked_size_mode) {
    case MILZMA_READ_FROM_HEADER: {
      uint64_t v;
      if (!c.u64le(&v)) return fail_at(MILZMA_HEADER_TOO_SHORT, "%s", kEofMsg);
      unpacked = v;  // 0xFFFF_FFFF_FFFF_FFFF == marker mode == MILZMA_SIZE_UNKNOWN
      break;
    }
    case MILZMA_READ_HEADER_BUT_USE_PROVIDED: {
      uint64_t v;
      if (!c.u64le(&v)) return fail_at(MILZMA_HEADER_TOO_SHORT, "%s", kEofMsg);
      unpacked = opt->provided_is_some ? opt->provided : MILZMA_SIZE_UNKNOWN;
      break;
    }
    default: unpacked = opt->provided_is_some ? opt->provided : MILZMA_SIZE_UNKNOWN; break;
  }
  memset(unit, 0, sizeof *unit);
  unit->kind = MILZMA_KIND_RAW_LZMA;
  unit->lc = uint8_t(lc);
  unit->lp = uint8_t(lp);
  unit->pb = uint8_t(pb);
  unit->dict_size = dict;
  unit->unpacked_size = unpacked;
  unit->memlimit = opt->memlimit_is_some ? opt->memlimit : MILZMA_NO_LIMIT;
  if (header_len) *header_len = c.pos;
  return MILZMA_OK;
}

MILZMA_HOST_NS_BEGIN

// Slice size to try first for a RAW unit.  The declared size comes from the (untrusted) header: it is only believed up to
// what the payload could plausibly expand to; a stream that really is denser goes through the OUT_FULL regrow rounds.
// A memlimit below the dictionary size ends the stream at memlimit bytes (lzbuffer.rs:206-217).
size_t lzma_cap_hint(const milzma_unit& u, size_t payload_len) {
  const uint64_t plausible = std::max<uint64_t>(uint64_t(1) << 20, uint64_t(payload_len) * 1024);
  uint64_t cap = std::max<uint64_t>(1 << 16, uint64_t(payload_len) * 6);
  if (u.unpacked_size != MILZMA_SIZE_UNKNOWN) cap = std::min<uint64_t>(u.unpacked_size, plausible) + 288;  // + one overshooting match
  if (u.memlimit < uint64_t(u.dict_size)) cap = std::min<uint64_t>(cap, u.memlimit + 288);
  return size_t(std::min<uint64_t>(cap, MILZMA_MAX_UNIT_BYTES - 512));
}

// Turns a finished RAW/LZMA2 unit into what the caller's writer / reader saw.
int finish_stream(const milzma_result& r, uint32_t kind, const uint8_t* slice, size_t slice_len, size_t header_len,
                  milzma_output* out) {
  out->in_consumed = header_len + size_t(r.in_consumed);
  const size_t visible = size_t(std::min<uint64_t>(r.out_flushed, slice_len));
  if (!out_set_data(out, slice, visible)) return out_fail(out, MILZMA_INFRA_ERROR, "out of memory");
  out->kind = milzma_result_message(&r, kind, out->msg, sizeof out->msg);
  return out->kind;
}

MILZMA_HOST_NS_END

int milzma_lzma_decompress_impl(milzma_ctx* ctx, const uint8_t* in, size_t in_len, const milzma_options* opt,
                                      milzma_output* out) {
  milzma_unit u;
  size_t hl = 0;
  const int hr = milzma_lzma_read_header(in, in_len, opt, &u, &hl, out);
  if (hr != MILZMA_OK) return hr;
  SingleDecode sd;
  if (!decode_single(ctx, u, in + hl, in_len - hl, lzma_cap_hint(u, in_len - hl), &sd)) return infra(ctx, out);
  return finish_stream(sd.res, MILZMA_KIND_RAW_LZMA, sd.out.data(), sd.out.size(), hl, out);
}

int milzma_lzma2_decompress_impl(milzma_ctx* ctx, const uint8_t* in, size_t in_len, milzma_output* out) {
  out_reset(out);
  milzma_unit u;
  memset(&u, 0, sizeof u);
  u.kind = MILZMA_KIND_LZMA2;
  SingleDecode sd;
  if (!decode_single(ctx, u, in, in_len, std::max<size_t>(1 << 16, in_len * 6), &sd)) return infra(ctx, out);
  return finish_stream(sd.res, MILZMA_KIND_LZMA2, sd.out.data(), sd.out.size(), 0, out);
}

// Batch driver for RAW / LZMA2 streams: one launch for all, stragglers (OUT_FULL) one by one.
MILZMA_HOST_NS_BEGIN

int stream_batch(milzma_ctx* ctx, uint32_t n, const uint8_t* const* ins, const size_t* in_lens, const milzma_options* opt,
                 bool lzma2, milzma_output* outs) {
  ctx->last_paths = 0;
  if (ctx) trace_mark(ctx, "batch: call begins");
  std::vector<milzma_unit> units;
  std::vector<uint32_t> owner;  // unit -> stream
  std::vector<size_t> hdr(n, 0);
  std::vector<uint32_t> alone;  // streams decoded one at a time
  const size_t budget = plan_budget(ctx);
  const auto single = [&](uint32_t i) {
    if (lzma2)
      milzma_lzma2_decompress(ctx, ins[i], in_lens[i], &outs[i]);
    else
      milzma_lzma_decompress(ctx, ins[i], in_lens[i], opt, &outs[i]);
  };
  size_t in_total = 0, out_total = 0;
  for (uint32_t i = 0; i < n; i++) {
    out_reset(&outs[i]);
    milzma_unit u;
    if (lzma2) {
      memset(&u, 0, sizeof u);
      u.kind = MILZMA_KIND_LZMA2;
    } else if (milzma_lzma_read_header(ins[i], in_lens[i], opt, &u, &hdr[i], &outs[i]) != MILZMA_OK) {
      continue;
    }
    const size_t payload = in_lens[i] - hdr[i];
    if (payload > MILZMA_MAX_UNIT_BYTES) {
      out_fail(&outs[i], MILZMA_INFRA_ERROR, "stream larger than MILZMA_MAX_UNIT_BYTES");
      continue;
    }
    u.in_off = in_total;
    u.in_len = payload;
    u.out_off = out_total;
    u.out_cap = std::min<size_t>(round_up(lzma2 ? std::max<size_t>(1 << 16, payload * 6) : lzma_cap_hint(u, payload), 256),
                                 MILZMA_MAX_UNIT_BYTES);
    if (in_total + out_total + round_up(payload, 256) + u.out_cap > budget) {  // on its own, after the batch
      alone.push_back(i);
      continue;
    }
    in_total += round_up(payload, 256);
    out_total += u.out_cap;
    units.push_back(u);
    owner.push_back(i);
  }
  const auto finish_alone = [&]() {
    for (uint32_t i : alone) single(i);
    return MILZMA_OK;
  };
  if (units.empty()) return finish_alone();
  // page-locked staging (PCIe at link speed), filled and emptied by several host threads
  auto fail_all = [&]() {
    for (uint32_t i : owner) infra(ctx, &outs[i]);
    for (uint32_t i : alone) single(i);  // (decoded, or given their own infrastructure error: never left as an empty success)
    return MILZMA_INFRA_ERROR;
  };
  if (!ctx) return fail_all();
  trace_mark(ctx, "batch: headers read, units planned");
  if (!hip_ok(ctx, hipSetDevice(ctx->device), "hipSetDevice")) return fail_all();
  // Streamed round 0 (below) for batches it pays for: many files of about one size, all in the fast kernel's class.  Their output
  // slices then sit at ONE pitch (span cuts are computed from the unit's index and the pitch alone).
  struct {
    size_t pitch = 0, span = 0;
    uint32_t spans = 0;
  } geo;
  StreamedSlot streamed_slot;
  {
    const char* const stream_env = env_get("MILZMA_STREAM");
    const bool off = stream_env && !strcmp(stream_env, "0");
    size_t max_cap = 0;
    size_t min_units, min_bytes;
    bool ragged_ok = false;
    stream_minimum(&min_units, &min_bytes, &ragged_ok);
    bool all_fast = ctx->use_fast && !off && units.size() >= min_units && out_total >= min_bytes;
    for (const milzma_unit& u : units) {
      max_cap = std::max(max_cap, size_t(u.out_cap));
      all_fast = all_fast && classify(ctx, u) == kFast;
    }
    const size_t pitch = round_up(max_cap, 256);
    if (all_fast && (ragged_ok || pitch * units.size() <= out_total + out_total / 4) && in_total + pitch * units.size() <= budget &&
        streamed_slot.try_take(ctx->device)) {
      size_t span = size_t(64) << 10;
      if (const char* e = env_get("MILZMA_SPAN")) span = std::max<size_t>(size_t(1) << 12, round_up(size_t(strtoull(e, nullptr, 0)), size_t(1) << 12));
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
  // (+ 512: the kernels fetch whole aligned windows, and a streamed launch reads this buffer itself)
  if (!pin_reserve(ctx, ctx->pin_in, in_total + 512) || !pin_reserve(ctx, ctx->pin_out, out_total) ||
      !dev_reserve(ctx, ctx->in, in_total + 512) || !dev_reserve(ctx, ctx->out, out_total + 512)) {
    for (uint32_t i : owner) single(i);  // the batch's staging cannot be had: one stream at a time
    return finish_alone();
  }
  uint8_t* hin = static_cast<uint8_t*>(ctx->pin_in.p);
  const void* d_input = ctx->in.p;  // where the decode calls find the compressed bytes: the device copy, or (streamed) the host buffer itself
  const auto upload = [&]() {
    // eight groups of streams: the gather of one group overlaps the transfer of the one before
    const size_t groups = std::min<size_t>(8, units.size());
    std::vector<size_t> first(groups + 1), bounds(groups + 1);
    for (size_t g = 0; g <= groups; g++) {
      first[g] = units.size() * g / groups;
      bounds[g] = g == groups ? in_total : size_t(units[first[g]].in_off);
    }
    return staged_h2d(ctx, ctx->in.p, hin, bounds, [&](size_t g) {
      parallel_for(first[g + 1] - first[g], [&](size_t k0) {
        const size_t k = first[g] + k0;
        memcpy(hin + units[k].in_off, ins[owner[k]] + hdr[owner[k]], size_t(units[k].in_len));
      });
    });
  };
  if (!geo.spans && !upload()) return fail_all();
  const uint32_t kind = lzma2 ? MILZMA_KIND_LZMA2 : MILZMA_KIND_RAW_LZMA;
  // Rounds.  A unit whose guessed output slice was too small (unknown-size streams: every .lzma that liblzma writes) is PARKED at
  // the end of its slice by the decode kernel, given a larger slice -- what it has produced moves there on the device -- and
  // RESUMED: no byte is decoded twice (the reference streams such output through its ring, lzbuffer.rs:258-270).  Every round
  // hands over the units that finished in it.  What cannot be parked (the generic kernel's units: lc + lp > 4) comes back with a
  // plain OUT_FULL and is decoded again afterwards with four times the room; the input stays on the device throughout.
  const uint32_t nu = uint32_t(units.size());
  std::vector<milzma_result> res(nu);
  std::vector<uint32_t> active(nu), restart;
  for (uint32_t k = 0; k < nu; k++) active[k] = k;
  size_t out_bytes = out_total;
  const auto give_up = [&](const std::vector<uint32_t>& list) {
    for (uint32_t k : list) infra(ctx, &outs[owner[k]]);
  };
  bool first = true;
  if (geo.spans) {
    // Round 0, streamed: one time-sliced launch whose waves write their output to the page-locked host buffer themselves, span by
    // span, while they decode (kernels.h); this thread waits for the kernel, a second one hands every span of every file over to the
    // caller's buffers as the span counters come in.  When the kernel ends, all that is left is the last span's hand-over.
    // The files' result buffers come page-locked from the pool: the waves write every span straight into the buffer the caller will
    // get (kernels.h: host_ptrs) and the host copies nothing.  If page-locked memory cannot be had, ordinary buffers are filled from
    // the page-locked staging buffer by a host thread, span by span.
    HeldBufs held;
    held.v.assign(nu, nullptr);
    std::vector<uint8_t*>& bufs = held.v;
    std::atomic<int> alloc_failed{0};
    bool direct = pinned_results_wanted();
    std::vector<size_t> caps(nu);
    for (uint32_t k = 0; k < nu; k++) caps[k] = size_t(units[k].out_cap);
    if (direct && !out_alloc_many(caps.data(), nu, true, bufs.data())) {
      held.drop();
      direct = false;
    }
    if (!direct && !out_alloc_many(caps.data(), nu, false, bufs.data())) alloc_failed = 1;
    // The input goes up in two parts (upload_leads / upload_rest above): the leads before the launch, everything while it runs.
    void* host_dev = nullptr;
    bool ok = !alloc_failed && ensure_progress(ctx);
    if (ok && direct) {
      std::vector<uint64_t> ptrs(size_t(nu) * 2);
      for (uint32_t k = 0; k < nu; k++) {
        ptrs[2 * size_t(k)] = uint64_t(reinterpret_cast<uintptr_t>(bufs[k]));
        ptrs[2 * size_t(k) + 1] = units[k].out_cap;
      }
      ok = upload_host_ptrs(ctx, ptrs, work_stream(ctx));
    } else if (ok) {
      ok = pin_reserve(ctx, ctx->pin_out, out_total) && hipHostGetDevicePointer(&host_dev, ctx->pin_out.p, 0) == hipSuccess;
    }
    if (!ok) (void)hipGetLastError();
    bool input_up = false;
    if (ok) {
      trace_mark(ctx, "streamed: result buffers and pointer table ready");
      ok = upload_leads(ctx, units, [&](size_t k) { return ins[owner[k]] + hdr[owner[k]]; }, work_stream(ctx));
      trace_mark(ctx, "streamed: leads up");
    }
    if (ok) {
      __atomic_store_n(&ctx->progress[milzma_ctx::kMaxSpans], 0u, __ATOMIC_RELEASE);
      ctx->stream_span = uint32_t(geo.span);
      // (Page-locked result buffers: nobody looks at the span counters before the kernel has ended, so only the first span is announced --
      //  an announcement is a release fence at system scope, the XCD's L2 written back, and 4096 waves x 16 spans of them were most of what
      //  the delivery cost the kernel: 234 -> ... ms, profiles/r06_batch_api.txt.  The turns are cut at every span as before.)
      ctx->stream_spans = direct ? 1 : geo.spans;
      ctx->stream_host = static_cast<uint8_t*>(host_dev);
      ctx->stream_ptrs = direct ? static_cast<const uint64_t*>(ctx->hostptrs.p) : nullptr;
      ctx->stream_in_host = true;
      trace_mark(ctx, "streamed: launch");
      ok = milzma_decode_units_async_impl(ctx, units.data(), nu, ctx->in.p, ctx->out.p, work_stream(ctx), MILZMA_DECODE_GROW, nullptr) == MILZMA_OK;
      ctx->stream_span = ctx->stream_spans = 0;
      ctx->stream_host = nullptr;
      ctx->stream_ptrs = nullptr;
      ctx->stream_in_host = false;
      // the whole input, in sixteen pieces, whatever became of the launch (the classic rounds want it too)
      const size_t pieces = std::min<size_t>(16, nu);
      std::vector<size_t> first_u(pieces + 1), bounds(pieces + 1);
      for (size_t g = 0; g <= pieces; g++) {
        first_u[g] = nu * g / pieces;
        bounds[g] = g == pieces ? in_total : size_t(units[first_u[g]].in_off);
      }
      input_up = upload_rest(ctx, hin, bounds, [&](size_t g) {
        parallel_for(first_u[g + 1] - first_u[g], [&](size_t k0) {
          const size_t k = first_u[g] + k0;
          memcpy(hin + units[k].in_off, ins[owner[k]] + hdr[owner[k]], size_t(units[k].in_len));
        });
      });
      trace_mark(ctx, "streamed: input complete");
      if (!input_up) {
        if (ok) (void)milzma_decode_units_wait_impl(ctx, res.data());
        held.drop();
        give_up(active);
        finish_alone();
        return MILZMA_INFRA_ERROR;
      }
    }
    if (ok && ctx->stream_active) {
      std::atomic<bool> kernel_done{false};
      const uint8_t* hout = static_cast<const uint8_t*>(ctx->pin_out.p);
      std::thread consumer([&] {
        if (direct) return;   // (the waves fill the result buffers themselves)
        for (uint32_t sp = 0; sp < geo.spans; sp++) {
          while (__atomic_load_n(&ctx->progress[sp], __ATOMIC_ACQUIRE) < nu && !kernel_done.load(std::memory_order_acquire))
            std::this_thread::sleep_for(std::chrono::microseconds(50));
          parallel_for(nu, [&](size_t k) {
            const size_t phase = (k & 15u) * (geo.span >> 4), cap = size_t(units[k].out_cap);
            const size_t lo = sp * geo.span > phase ? sp * geo.span - phase : 0, hi = std::min(cap, (sp + 1) * geo.span - phase);
            if (lo < hi) memcpy(bufs[k] + lo, hout + size_t(units[k].out_off) + lo, hi - lo);
          });
        }
      });
      int wr;
      {
        JoinOnExit joined{consumer, kernel_done};
        wr = milzma_decode_units_wait_impl(ctx, res.data());
      }
      trace_mark(ctx, "streamed decode + hand-over: done");
      ctx->last_paths |= MILZMA_PATH_STREAMED | MILZMA_PATH_TWO_PART_INPUT;
      if (wr != MILZMA_OK) {
        held.drop();
        give_up(active);
        finish_alone();
        return MILZMA_INFRA_ERROR;
      }
      // (a unit that ran again in another class did so in a launch of its own, without host destinations: its bytes are on the device)
      for (uint32_t k : ctx->promoted) {
        const size_t got = size_t(std::min<uint64_t>(res[k].out_len, units[k].out_cap));
        if (k < nu && bufs[k] && got &&
            !hip_ok(ctx, hipMemcpy(bufs[k], static_cast<const uint8_t*>(ctx->out.p) + units[k].out_off, got, hipMemcpyDeviceToHost), "D2H output")) {
          held.drop();
          give_up(active);
          finish_alone();
          return MILZMA_INFRA_ERROR;
        }
      }
      std::vector<uint32_t> parked;
      for (uint32_t k = 0; k < nu; k++) {
        const milzma_result& r = res[k];
        const bool more_room = units[k].out_cap < MILZMA_MAX_UNIT_BYTES;
        if (is_parked(r) && more_room) {
          parked.push_back(k);
        } else if ((r.status == MILZMA_ST_OUT_FULL && !is_parked(r) && more_room) || r.status == MILZMA_ST_NEED_RERUN) {
          restart.push_back(k);   // (NEED_RERUN: it outran the second part of the upload; its slice is big enough, more does not hurt)
        } else {
          milzma_output* o = &outs[owner[k]];
          milzma_result rr = r;
          if (is_parked(rr)) rr.err_a = 0;
          o->in_consumed = hdr[owner[k]] + size_t(rr.in_consumed);
          o->data = bufs[k];
          o->len = size_t(std::min<uint64_t>(rr.out_flushed, units[k].out_cap));
          o->kind = milzma_result_message(&rr, kind, o->msg, sizeof o->msg);
          bufs[k] = nullptr;
        }
      }
      held.drop();
      size_t ob = 0;
      if (!parked.empty() && !regrow_parked(ctx, units, res, parked, work_stream(ctx), &ob)) {
        give_up(parked);
        give_up(restart);
        finish_alone();
        return MILZMA_INFRA_ERROR;
      }
      if (!parked.empty()) out_bytes = ob;
      active.swap(parked);
      first = false;
    } else {
      // not to be had (no mapped memory, or the launch could not be time-sliced): the batch in flight, if any, is collected and the
      // classic rounds below do the work -- nothing has been handed over yet
      if (ok) (void)milzma_decode_units_wait_impl(ctx, res.data());
      held.drop();
      if (!input_up && !upload()) return fail_all();  // (whatever part of the input went up: all of it now)
    }
  }
  for (; !active.empty(); first = false) {
    std::vector<uint32_t> parked;
    {
      if (milzma_decode_units_impl(ctx, units.data(), nu, d_input, ctx->out.p, res.data(), work_stream(ctx),
                                   first ? MILZMA_DECODE_GROW : MILZMA_DECODE_RESUME) != MILZMA_OK) {
        give_up(active);
        give_up(restart);
        finish_alone();
        return MILZMA_INFRA_ERROR;
      }
      trace_mark(ctx, "decode: done");
      ctx->last_paths |= MILZMA_PATH_CLASSIC;
      // What finished travels back packed (an unknown-size stream's slice is a guess several times its output: the link should not
      // carry the slack): the move kernel gathers the finished outputs into a second device buffer, that one comes back in chunks
      // and a stream is handed over as soon as its bytes have arrived.  Where the slices are (nearly) full they go as they are.
      std::vector<uint32_t> fin;
      std::vector<uint64_t> so, dof, ln;
      size_t packed = 0, slack = 0;
      for (uint32_t k : active) {
        const milzma_result& r = res[k];
        if (is_parked(r) && units[k].out_cap < MILZMA_MAX_UNIT_BYTES) {
          parked.push_back(k);
          continue;
        }
        if (r.status == MILZMA_ST_OUT_FULL && !is_parked(r) && units[k].out_cap < MILZMA_MAX_UNIT_BYTES) {
          restart.push_back(k);
          continue;
        }
        const uint64_t visible = std::min<uint64_t>(r.out_flushed, units[k].out_cap);
        fin.push_back(k);
        so.push_back(units[k].out_off);
        dof.push_back(packed);
        ln.push_back(visible);
        packed += round_up(size_t(visible), 256);
        slack += size_t(units[k].out_cap);
      }
      ChunkedCopy d2h;
      const bool pack = !fin.empty() && slack > packed + packed / 8 + (size_t(1) << 20);
      const uint8_t* hout = nullptr;
      bool ok = true;
      if (pack) {
        ok = dev_reserve(ctx, ctx->pack, packed + 512) && pin_reserve(ctx, ctx->pin_out, packed) &&
             move_units_impl(ctx, uint32_t(fin.size()), ctx->out.p, so.data(), ctx->pack.p, dof.data(), ln.data(), work_stream(ctx)) == MILZMA_OK &&
             d2h.start_d2h(ctx, ctx->pin_out.p, ctx->pack.p, packed);
      } else if (!fin.empty()) {
        ok = pin_reserve(ctx, ctx->pin_out, out_bytes) && d2h.start_d2h(ctx, ctx->pin_out.p, ctx->out.p, out_bytes);
      }
      if (!ok) {
        give_up(fin);
        give_up(parked);
        give_up(restart);
        finish_alone();
        return MILZMA_INFRA_ERROR;
      }
      hout = static_cast<const uint8_t*>(ctx->pin_out.p);
      parallel_for(fin.size(), [&](size_t j) {
        const uint32_t k = fin[j], i = owner[k];
        const size_t off = pack ? size_t(dof[j]) : size_t(units[k].out_off);
        if (!d2h.wait_until(off + size_t(ln[j]))) {
          out_fail(&outs[i], MILZMA_INFRA_ERROR, "D2H output failed");
          return;
        }
        milzma_result r = res[k];
        if (is_parked(r)) r.err_a = 0;  // (at the largest slice there is: an ordinary OUT_FULL)
        finish_stream(r, kind, hout + off, size_t(ln[j]), hdr[i], &outs[i]);
      });
      trace_mark(ctx, "download + hand-over: done");
    }  // (the chunked copy has drained here: nothing reads ctx->out any more)
    if (!parked.empty() && !regrow_parked(ctx, units, res, parked, work_stream(ctx), &out_bytes)) {
      give_up(parked);
      give_up(restart);
      finish_alone();
      return MILZMA_INFRA_ERROR;
    }
    active.swap(parked);
  }
  // the units that could not be parked: again from their first byte, together, with four times the room (rounds as before)
  while (!restart.empty()) {
    std::vector<milzma_unit> sub(restart.size());
    size_t bytes = 0;
    for (size_t j = 0; j < restart.size(); j++) {
      milzma_unit& u = units[restart[j]];
      u.out_cap = std::min<uint64_t>(round_up(size_t(u.out_cap) * 4, 256), MILZMA_MAX_UNIT_BYTES);
      sub[j] = u;
      sub[j].out_off = bytes;
      bytes += size_t(sub[j].out_cap);
    }
    std::vector<milzma_result> r(sub.size());
    ChunkedCopy d2h;
    if (!pin_reserve(ctx, ctx->pin_out, bytes) || !dev_reserve(ctx, ctx->out, bytes + 512) ||
        milzma_decode_units_impl(ctx, sub.data(), uint32_t(sub.size()), d_input, ctx->out.p, r.data(), work_stream(ctx), 0) != MILZMA_OK ||
        !d2h.start_d2h(ctx, ctx->pin_out.p, ctx->out.p, bytes)) {
      give_up(restart);
      finish_alone();
      return MILZMA_INFRA_ERROR;
    }
    const uint8_t* hout = static_cast<const uint8_t*>(ctx->pin_out.p);
    std::vector<uint32_t> next;
    std::vector<uint8_t> again(sub.size(), 0);
    parallel_for(sub.size(), [&](size_t j) {
      if (r[j].status == MILZMA_ST_OUT_FULL && sub[j].out_cap < MILZMA_MAX_UNIT_BYTES) {
        again[j] = 1;
        return;
      }
      const uint32_t i = owner[restart[j]];
      if (!d2h.wait_until(size_t(sub[j].out_off + sub[j].out_cap))) {
        out_fail(&outs[i], MILZMA_INFRA_ERROR, "D2H output failed");
        return;
      }
      finish_stream(r[j], kind, hout + sub[j].out_off, size_t(sub[j].out_cap), hdr[i], &outs[i]);
    });
    for (size_t j = 0; j < sub.size(); j++)
      if (again[j]) next.push_back(restart[j]);
    restart.swap(next);
  }
  return finish_alone();
}

MILZMA_HOST_NS_END

int milzma_lzma_decompress_batch_impl(milzma_ctx* ctx, uint32_t n, const uint8_t* const* ins, const size_t* in_lens,
                                            const milzma_options* opt, milzma_output* outs) {
  return stream_batch(ctx, n, ins, in_lens, opt, false, outs);
}

int milzma_lzma2_decompress_batch_impl(milzma_ctx* ctx, uint32_t n, const uint8_t* const* ins, const size_t* in_lens,
                                             milzma_output* outs) {
  return stream_batch(ctx, n, ins, in_lens, nullptr, true, outs);
}

