// tests/san/fake_kernels.cpp -- TEST INFRASTRUCTURE.  Stand-ins for the kernel launchers of lzma_rs_amd/csrc/kernels.h in the sanitizer
// builds of the host side (tests/san/Makefile: pipeline_asan / pipeline_tsan; see fake_hip.cpp).  The CPU oracle decodes a unit where the
// GPU kernels would; what these stand-ins keep of the real kernels is their CONTRACT with the host code:
//   * one milzma_result per unit (status, out_len, out_flushed, in_consumed), written when the launch's task runs on its stream;
//   * growable output: a unit that does not fit its slice is parked (OUT_FULL, err_a = MILZMA_PARKED, out_len = what is in the slice) if the
//     launch is a growing one, and continues -- here: is decoded again into the larger slice -- when an order entry with bit 31 names it;
//   * the literal-row slab of class kFastSpill (round 5): a launch's units find their rows initialised (every probability 0x400: launch_slab_init) at
//     d_slab + unit * slab_bytes, a unit that parks leaves its trained rows there (here: a signature naming the unit and the stride), and a resumed
//     unit must find them where it left them -- a host that re-derives the stride from the resumed subset, or wipes the slab for another launch of
//     the batch, is caught (status 0xFFFF: the harness reports it);
//   * streamed launches: every unit's output also goes to its host destination (host_ptrs {address, limit} or host_out + out_off), then
//     the span counters move; a launch with an in_ready word waits for it before it reads beyond the units' leads (here: before it reads).
// Every error site of the hot path comes back with the status (and the integers) the real kernels report for it: the oracle's message is
// parsed back (the inverse of milzma_result_message), so the harness compares EVERY result with the oracle, failed payloads included.
// (A message that is none of them becomes status 0xFFFF: "unknown status" -- an infrastructure error the harness reports.)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "kernels.h"
#include "lzma_oracle.h"

void fake_hip_enqueue(hipStream_t stream, std::function<void()> fn);
bool fake_hip_launch_fails();
extern "C" uint32_t milzma_crc32(const uint8_t* p, size_t n);
extern "C" uint64_t milzma_crc64(const uint8_t* p, size_t n);

namespace {

struct Streamed {
  uint32_t span = 0, n_spans = 0;
  uint32_t* progress = nullptr;
  uint8_t* host_out = nullptr;
  uint32_t* in_ready = nullptr;
  const uint64_t* host_ptrs = nullptr;
};

// lc + lp of the first LZMA chunk of an LZMA2 stream that brings properties (control >= 0xC0: lzma2.rs:153-175), 0 if there is none in front
uint32_t first_lclp(const uint8_t* in, size_t n) {
  size_t pos = 0;
  while (pos < n) {
    const uint8_t c = in[pos];
    if (c == 0) return 0;
    if (c < 0x80) {  // stored chunk: 2 size bytes + data
      if (pos + 3 > n) return 0;
      pos += 3 + ((size_t(in[pos + 1]) << 8) | in[pos + 2]) + 1;
      continue;
    }
    if (c >= 0xC0 && pos + 5 < n) {
      const uint32_t props = in[pos + 5];
      if (props >= 225) return 0;
      return (props % 9) + ((props / 9) % 5);
    }
    return 0;
  }
  return 0;
}

constexpr uint64_t kRowsMagic = 0x524f57535f4f4b21ull;   // what a parked unit's rows look like in the slab: {magic, unit, stride}

// Fed input (MILZMA_DECODE_FEED): what a real parked unit keeps in registers -- everything it has consumed so far -- the stand-in keeps as
// BYTES, per parking lot and unit: a view that is not the last is consumed up to 31 bytes before its end and the unit parks (NEED_INPUT),
// the last view is decoded by the oracle behind everything kept.  What this exercises is the host side: the flag on its way to the
// launch, the descriptors as uploaded (MILZMA_KIND_LAST_VIEW), the previous results uploaded before a resuming launch, the parked-unit
// record and its checks.
std::mutex g_fed_mu;
std::map<std::pair<const void*, uint32_t>, std::vector<uint8_t>> g_fed;

void decode_one(const milzma_unit& u_in, uint32_t uidx, const uint8_t* d_in, uint8_t* d_out, milzma_result* res, bool grow, const Streamed* st,
                const uint8_t* d_slab, uint32_t slab_bytes, bool resume, bool feed = false, const void* lot = nullptr) {
  const bool has_slab = d_slab != nullptr;
  milzma_unit u = u_in;
  milzma_result r;
  memset(&r, 0, sizeof r);
  const bool last_view = feed && (u.kind & MILZMA_KIND_LAST_VIEW);
  if (feed) u.kind &= uint8_t(~MILZMA_KIND_LAST_VIEW);
  std::vector<uint8_t> kept;   // (what the unit consumed in earlier views)
  const bool was_fed = resume && (res[uidx].err_b & 0x200u) && res[uidx].err_a == MILZMA_PARKED;
  {
    std::lock_guard<std::mutex> g(g_fed_mu);
    auto it = g_fed.find({lot, uidx});
    if (was_fed && it != g_fed.end()) kept = it->second;
    if (it != g_fed.end() && !was_fed) g_fed.erase(it);
  }
  uint64_t* rows = has_slab && slab_bytes >= 64 ? reinterpret_cast<uint64_t*>(const_cast<uint8_t*>(d_slab) + size_t(uidx) * slab_bytes) : nullptr;
  if (rows) {  // the rows must be what the kernel's contract says: fresh for a first launch, this unit's own for a resumed one
    const uint64_t fresh = 0x0400040004000400ull;
    const bool ok = resume ? rows[0] == kRowsMagic && rows[1] == uidx && rows[2] == slab_bytes && rows[slab_bytes / 8 - 1] == kRowsMagic
                           : rows[0] == fresh && rows[1] == fresh && rows[2] == fresh && rows[slab_bytes / 8 - 1] == fresh;
    if (!ok) {
      fprintf(stderr, "fake kernel: unit %u finds %s literal rows in the slab (stride %u)\n", uidx, resume ? "someone else's / wiped" : "uninitialised", slab_bytes);
      r.status = 0xFFFFu;
      res[uidx] = r;
      return;
    }
  }
  orc_result o;
  memset(&o, 0, sizeof o);
  const uint8_t* in = d_in + u.in_off;
  if (feed && !last_view) {   // parks within 32 bytes of the view's end; its rows stay in the slab
    const uint64_t take = u.in_len > 31 ? u.in_len - 31 : 0;
    kept.insert(kept.end(), in, in + take);
    {
      std::lock_guard<std::mutex> g(g_fed_mu);
      g_fed[{lot, uidx}] = kept;
    }
    r.status = MILZMA_ST_NEED_INPUT;
    r.err_a = MILZMA_PARKED;
    r.err_b = (has_slab ? 0x100u : 0u) | 0x200u;
    r.in_consumed = take;
    if (rows) {
      rows[0] = kRowsMagic;
      rows[1] = uidx;
      rows[2] = slab_bytes;
      rows[slab_bytes / 8 - 1] = kRowsMagic;
    }
    res[uidx] = r;
    return;
  }
  std::vector<uint8_t> joined;
  if (!kept.empty()) {   // the last view behind everything consumed before: the stream from its first byte
    joined = kept;
    joined.insert(joined.end(), in, in + u.in_len);
    in = joined.data();
    u.in_len = joined.size();
  }
  // like the fast kernel without a literal-row slab: an LZMA2 unit whose chunk asks for lc + lp = 4 is sent back for another launch class
  if (u.kind == MILZMA_KIND_LZMA2 && !has_slab && first_lclp(in, size_t(u.in_len)) == 4) {
    r.status = MILZMA_ST_NEED_GENERIC;
    if (st && st->progress)
      for (uint32_t s = 0; s < st->n_spans; s++) __atomic_fetch_add(&st->progress[s], 1u, __ATOMIC_RELEASE);
    res[uidx] = r;
    return;
  }
  int kind;
  if (u.kind == MILZMA_KIND_LZMA2)
    kind = orc_lzma2_decompress(in, size_t(u.in_len), &o);
  else
    kind = orc_lzma_raw_decompress(in, size_t(u.in_len), u.lc, u.lp, u.pb, u.dict_size, u.unpacked_size != MILZMA_SIZE_UNKNOWN, u.unpacked_size,
                                   u.memlimit != MILZMA_NO_LIMIT, u.memlimit, &o);
  uint8_t* out = d_out + u.out_off;
  size_t visible = o.out_len;
  if (kind == ORC_OK && o.out_len > u.out_cap) {  // does not fit: parked in front of the symbol that would not (here: a little before the end)
    r.status = MILZMA_ST_OUT_FULL;
    r.err_a = grow ? MILZMA_PARKED : 0;
    r.err_b = (grow && has_slab ? 0x100u : 0u) | (grow && feed ? 0x200u : 0u);   // (the launch class it resumes in, as the real kernel reports it; 0x200: on a re-based view)
    if (grow && rows) {
      rows[0] = kRowsMagic;
      rows[1] = uidx;
      rows[2] = slab_bytes;
      rows[slab_bytes / 8 - 1] = kRowsMagic;
    }
    visible = size_t(u.out_cap > 300 ? u.out_cap - 300 : 0);
    r.out_len = r.out_flushed = visible;
    r.in_consumed = 0;
  } else {
    visible = size_t(o.out_len < u.out_cap ? o.out_len : u.out_cap);
    r.out_len = r.out_flushed = visible;
    r.in_consumed = o.in_consumed >= kept.size() ? o.in_consumed - kept.size() : 0;   // (counted from the start of THIS view)
    // the oracle's message back to the status (and the integers) the kernels report for that error site: milzma_result_message's inverse
    unsigned long long a = 0, b = 0;
    const char* m = strchr(o.msg, ':');
    m = m ? m + 2 : o.msg;
    r.status = 0xFFFFu;
    if (kind == ORC_OK) r.status = MILZMA_ST_OK;
    else if (kind == ORC_IO_ERROR && strstr(m, "failed to fill whole buffer")) r.status = MILZMA_ST_INPUT_EOF;
    else if (strstr(m, "too short: failed to fill whole buffer")) r.status = MILZMA_ST_RC_INIT;
    else if (sscanf(m, "Match distance %llu is beyond dictionary size %llu", &a, &b) == 2) r.status = MILZMA_ST_MATCH_DIST_DICT;
    else if (sscanf(m, "Match distance %llu is beyond output size %llu", &a, &b) == 2) r.status = MILZMA_ST_MATCH_DIST_OUT;
    else if (sscanf(m, "LZ distance %llu is beyond dictionary size %llu", &a, &b) == 2) r.status = MILZMA_ST_LZ_DIST_DICT;
    else if (sscanf(m, "LZ distance %llu is beyond output size %llu", &a, &b) == 2) r.status = MILZMA_ST_LZ_DIST_OUT;
    else if (sscanf(m, "exceeded memory limit of %llu", &a) == 1) r.status = MILZMA_ST_MEMLIMIT;
    else if (strstr(m, "Found end-of-stream marker but more bytes are available")) r.status = MILZMA_ST_MARKER_TRAILING;
    else if (sscanf(m, "Expected unpacked size of %llu but decompressed to %llu", &a, &b) == 2) r.status = MILZMA_ST_SIZE_MISMATCH;
    else if (strstr(m, "LZMA2 expected new status")) r.status = MILZMA_ST_L2_STATUS_EOF;
    else if (sscanf(m, "LZMA2 invalid status %llu", &a) == 1) r.status = MILZMA_ST_L2_INVALID_STATUS;
    else if (strstr(m, "LZMA2 expected unpacked size")) r.status = MILZMA_ST_L2_UNPACKED_EOF;
    else if (strstr(m, "LZMA2 expected packed size")) r.status = MILZMA_ST_L2_PACKED_EOF;
    else if (strstr(m, "LZMA2 expected new properties")) r.status = MILZMA_ST_L2_PROPS_EOF;
    else if (sscanf(m, "LZMA2 invalid properties: lc + lp (%llu + %llu)", &a, &b) == 2) r.status = MILZMA_ST_L2_LCLP;
    else if (sscanf(m, "LZMA2 invalid properties: %llu must be", &a) == 1) r.status = MILZMA_ST_L2_PROPS_INVALID;
    else if (sscanf(m, "LZMA2 expected %llu uncompressed bytes", &a) == 1) r.status = MILZMA_ST_L2_STORED_EOF;
    r.err_a = a;
    r.err_b = b;
  }
  if (visible) memcpy(out, o.out, visible);
  if (st && st->progress) {  // the streamed way out: the unit's own destination, then the counters
    uint8_t* host = st->host_out ? st->host_out + u.out_off : nullptr;
    size_t end = visible;
    if (st->host_ptrs) {
      host = reinterpret_cast<uint8_t*>(uintptr_t(st->host_ptrs[2 * size_t(uidx)]));
      const uint64_t lim = st->host_ptrs[2 * size_t(uidx) + 1];
      if (end > lim) end = size_t(lim);
    }
    if (host && end) memcpy(host, out, end);
    for (uint32_t s = 0; s < st->n_spans; s++) __atomic_fetch_add(&st->progress[s], 1u, __ATOMIC_RELEASE);
  }
  orc_free(o.out);
  res[uidx] = r;
}

void run_units(const milzma_unit* d_units, const uint32_t* d_order, uint32_t n, const uint8_t* d_in, uint8_t* d_out, milzma_result* d_results,
               bool grow, Streamed st, const uint8_t* d_slab, uint32_t slab_bytes = 0, bool feed = false, const void* lot = nullptr) {
  if (st.in_ready)
    while (__atomic_load_n(st.in_ready, __ATOMIC_ACQUIRE) == 0) std::this_thread::sleep_for(std::chrono::microseconds(100));
  // a few host threads stand for the chip: units really finish in any order and at the same time
  const unsigned t = n < 4 ? 1u : 4u;
  std::vector<std::thread> th;
  const auto body = [&](unsigned k) {
    for (uint32_t i = k; i < n; i += t) {
      const uint32_t uidx = d_order[i] & 0x7FFFFFFFu;
      decode_one(d_units[uidx], uidx, d_in, d_out, d_results, grow, st.progress ? &st : nullptr, d_slab, slab_bytes, (d_order[i] & 0x80000000u) != 0,
                 feed, lot);
    }
  };
  for (unsigned k = 1; k < t; k++) th.emplace_back(body, k);
  body(0);
  for (auto& x : th) x.join();
}

}  // namespace

namespace milzma {

hipError_t launch_generic(LitClass, const milzma_unit* d_units, const uint32_t* d_order, uint32_t n, const uint8_t* d_in, uint8_t* d_out,
                          milzma_result* d_results, uint16_t*, uint32_t, hipStream_t stream) {
  if (fake_hip_launch_fails()) return hipErrorLaunchFailure;
  static const uint8_t generic_has_its_own_tables = 0;   // (any lc + lp: never sent back for another class)
  fake_hip_enqueue(stream, [=] { run_units(d_units, d_order, n, d_in, d_out, d_results, false, Streamed(), &generic_has_its_own_tables); });
  return hipSuccess;
}
hipError_t launch_fast(const milzma_unit* d_units, const uint32_t* d_order, uint32_t n, const uint8_t* d_in, uint8_t* d_out, milzma_result* d_results,
                       hipStream_t stream, uint32_t, uint32_t*, const uint8_t* d_slab, uint32_t slab_bytes) {
  if (fake_hip_launch_fails()) return hipErrorLaunchFailure;
  fake_hip_enqueue(stream, [=] { run_units(d_units, d_order, n, d_in, d_out, d_results, false, Streamed(), d_slab, slab_bytes); });
  return hipSuccess;
}
uint32_t fast_resident_blocks(uint32_t) { return 16; }   // (a small chip: launches of more units than that take the time-sliced form)
size_t slice_ctx_bytes() { return 64; }
size_t slice_queue_bytes(uint32_t cap) { return size_t(cap) * 8 + 64; }
hipError_t launch_fast_sliced(const milzma_unit* d_units, const uint32_t* d_order, uint32_t n, const uint8_t* d_in, uint8_t* d_out,
                              milzma_result* d_results, hipStream_t stream, uint32_t, uint32_t*, void*, uint32_t, uint32_t, bool, void* d_ctxmem, bool grow, uint32_t feed,
                              uint32_t span_bytes, uint32_t n_spans, uint32_t* progress, uint8_t* host_out, uint32_t* in_ready,
                              const uint64_t* host_ptrs, const uint8_t* d_slab, uint32_t slab_bytes) {
  if (fake_hip_launch_fails()) return hipErrorLaunchFailure;
  Streamed st;
  st.span = span_bytes;
  st.n_spans = n_spans;
  st.progress = progress;
  st.host_out = host_out;
  st.in_ready = in_ready;
  st.host_ptrs = host_ptrs;
  fake_hip_enqueue(stream, [=] { run_units(d_units, d_order, n, d_in, d_out, d_results, grow, st, d_slab, slab_bytes, feed, d_ctxmem); });
  return hipSuccess;
}
uint32_t stream_lead_bytes(uint32_t in_len) {
  const uint32_t q4 = in_len >> 2;
  return q4 > 4096u ? q4 : 4096u;   // (the real kernels: a quarter, at least 128 KiB -- small here, so that small files have a second part)
}
hipError_t launch_slab_init(uint8_t* d_slab, uint32_t slab_bytes, const uint32_t* d_order, uint32_t n, hipStream_t stream) {
  if (fake_hip_launch_fails()) return hipErrorLaunchFailure;
  fake_hip_enqueue(stream, [=] {
    for (uint32_t i = 0; i < n; i++) {
      uint16_t* p = reinterpret_cast<uint16_t*>(d_slab + size_t(d_order[i] & 0x7FFFFFFFu) * slab_bytes);
      for (uint32_t k = 0; k < slab_bytes / 2; k++) p[k] = 0x0400;
    }
  });
  return hipSuccess;
}
hipError_t launch_move_units(const uint8_t* d_src, uint8_t* d_dst, const uint64_t* d_offs, uint32_t n, hipStream_t stream) {
  if (fake_hip_launch_fails()) return hipErrorLaunchFailure;
  fake_hip_enqueue(stream, [=] {
    for (uint32_t i = 0; i < n; i++)
      if (d_offs[2 * size_t(n) + i]) memmove(d_dst + d_offs[size_t(n) + i], d_src + d_offs[i], size_t(d_offs[2 * size_t(n) + i]));
  });
  return hipSuccess;
}
// crc_units.hip.h: 64 chunks per unit, the partial CRCs of each, folded by host.cpp's crc_fold
hipError_t launch_crc_units(const milzma_unit* d_units, uint32_t n, const uint8_t* d_out, const milzma_result* d_results, void* d_parts,
                            hipStream_t stream) {
  if (fake_hip_launch_fails()) return hipErrorLaunchFailure;
  fake_hip_enqueue(stream, [=] {
    uint8_t* parts = static_cast<uint8_t*>(d_parts);
    for (uint32_t u = 0; u < n; u++) {
      uint8_t* p = parts + size_t(u) * kCrcPartsBytes;
      uint32_t* c32 = reinterpret_cast<uint32_t*>(p);
      uint64_t* c64 = reinterpret_cast<uint64_t*>(p + 64 * 4);
      uint32_t valid = d_results[u].status == MILZMA_ST_OK;
      const uint64_t len = d_results[u].out_len;
      uint64_t chunk = ((len + 63) / 64 + 15) & ~uint64_t(15);
      if (chunk == 0) chunk = 16;
      const uint8_t* base = d_out + d_units[u].out_off;
      for (uint32_t l = 0; l < 64; l++) {
        const uint64_t begin = uint64_t(l) * chunk, end = begin + chunk < len ? begin + chunk : len;
        const bool some = valid && begin < end;
        c32[l] = some ? milzma_crc32(base + begin, size_t(end - begin)) : 0;
        c64[l] = some ? milzma_crc64(base + begin, size_t(end - begin)) : 0;
      }
      const uint32_t ch = uint32_t(chunk);
      memcpy(p + 64 * 4 + 64 * 8, &ch, 4);
      memcpy(p + 64 * 4 + 64 * 8 + 4, &valid, 4);
    }
  });
  return hipSuccess;
}

}  // namespace milzma
