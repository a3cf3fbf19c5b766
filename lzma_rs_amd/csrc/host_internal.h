// host_internal.h -- what the translation units of the library's host side share (internal: nothing here is part of the C ABI).
//
//   host.cpp        contexts, environment switches, device / page-locked buffers, the result-buffer pool, the launch classes and the
//                   unit-level decode calls (async / wait / grow / resume), result messages, CRC folding, milzma_move_units
//   host_files.cpp  the whole-file entry points of .lzma and LZMA2: header parsing, single files, the batch calls with their streamed
//                   launch (two-part upload, span hand-over, park / regrow / resume rounds)
//   host_xz.cpp     the XZ container: xz::decode_stream restated, the Index planner, the .xz batch call, milzma_xz_plan
//   host_api.cpp    the exported entry points' wrappers (no C++ exception crosses the ABI), calls cut into groups over lanes, the
//                   asynchronous halves
//   host_multi.cpp  the GPUs of one node behind one handle: partition planner, per-device workers, the one-ingest-point entry
//   host_stream.cpp push-mode .lzma decoding for a batch of streams (the crate's Stream) over fed input
//
// Round 5 split what was one 3 900-line translation unit along the seams the fault-injection harness exercises;
// behaviour is unchanged.  Everything internal lives in namespace milzma::host with hidden visibility: libmilzma.so exports the
// functions include/milzma.h declares and nothing else.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cinttypes>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "kernels.h"
#include "milzma.h"

#define MILZMA_HOST_NS_BEGIN namespace milzma { namespace host __attribute__((visibility("hidden"))) {
#define MILZMA_HOST_NS_END } }
#define MILZMA_HIDDEN __attribute__((visibility("hidden")))

MILZMA_HOST_NS_BEGIN
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};
struct PinBuf {  // page-locked host staging (hipHostMalloc): PCIe copies run at link speed from / to it
  void* p = nullptr;
  size_t cap = 0;
  std::atomic<const char*> holder{nullptr};   // who is using its contents right now (PinLease)
};
// A page-locked buffer of the context belongs to ONE activity at a time.  Round 4's data race -- an on-demand decode inside a batched XZ
// walk staged its move list in the buffer the other files' walks were still reading their blocks' CRC parts from -- was two activities
// sharing one PinBuf without either knowing.  Whoever keeps data in such a buffer across calls that may re-enter the library takes a lease
// and says who it is; a second taker gets `false` (the caller turns that into an infrastructure error), and in the sanitizer harness's
// builds (MILZMA_OWNERSHIP_CHECKS) the process aborts with both names.
struct PinLease {
  PinBuf* b = nullptr;
  bool take(PinBuf& buf, const char* who) {
    const char* none = nullptr;
    if (!buf.holder.compare_exchange_strong(none, who)) {
#ifdef MILZMA_OWNERSHIP_CHECKS
      fprintf(stderr, "milzma: page-locked buffer wanted by '%s' is held by '%s'\n", who, none);
      abort();
#endif
      return false;
    }
    b = &buf;
    return true;
  }
  void release() {
    if (b) b->holder.store(nullptr);
    b = nullptr;
  }
  ~PinLease() { release(); }
};
extern thread_local std::string g_create_error;   // what milzma_last_error(nullptr) returns: the calling thread's last failed create
inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// a unit the time-sliced kernel parked: for room (MILZMA_DECODE_GROW) or for input (MILZMA_DECODE_FEED)
inline bool is_parked_result(const milzma_result& r) {
  return (r.status == MILZMA_ST_OUT_FULL || r.status == MILZMA_ST_NEED_INPUT) && r.err_a == MILZMA_PARKED;
}
MILZMA_HOST_NS_END

using milzma::host::DevBuf;
using milzma::host::PinBuf;

struct UploadTurn {  // whose upload may use the PCIe link now: groups of one call go up in order
  std::mutex mu;
  std::condition_variable cv;
  uint32_t next = 0;
};

struct milzma_ctx {
  int device = 0;
  std::string err;
  DevBuf units, order, results, scratch, in, out, pack, crc, flags, slice_q, slice_ctx, hostptrs;  // pack: finished outputs gathered for the download  // slice_*: queue and parked states of time-sliced launches  // flags: 64 words, one per launch in flight (last-block flags)
  PinBuf pin_in, pin_out, pin_small, pin_lead, pin_moves;  // pin_lead: the units' first bytes, gathered for a streamed launch
  // pin_moves: move lists (milzma_move_units) -- a buffer of their own: an on-demand decode inside a batched XZ walk may regrow a parked
  // unit while other files' walks still read the blocks' CRC parts out of pin_small (ThreadSanitizer found them sharing it)
  std::mutex mu;  // serialises GPU use by the worker threads of the batched XZ walk
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // milzma_decode_units_async: what is in flight until milzma_decode_units_wait
  std::vector<hipEvent_t> ev_pool;      // pairs (start, stop), one per kernel launch of the batch in flight
  uint32_t ev_used = 0;
  bool pending = false;
  uint32_t pend_n = 0;
  hipStream_t pend_stream = nullptr;
  const uint8_t* pend_in = nullptr;
  uint8_t* pend_out = nullptr;
  std::vector<milzma_unit> pend_units;  // (the caller's array need not outlive the call)
  uint32_t pend_flags = 0;              // MILZMA_DECODE_* of the batch in flight
  // units of the last batch that were decoded AGAIN in another launch class (an LZMA2 chunk switched to properties outside its
  // class's reach): a streamed launch's host destinations hold only what the FIRST launch wrote -- whoever streamed fetches these
  std::vector<uint32_t> promoted;
  // growable output (milzma_decode_units_ex): the last GROW / RESUME call left units parked in slice_ctx (indexed by unit: the next
  // call may resume them as long as it is a RESUME with the same n); any other decode call on the context gives the parking lot up
  bool parked_valid = false;
  uint32_t parked_n = 0;
  // what the GROW / RESUME call that parked them recorded of the parked units (a RESUME is checked against it: the launch class and
  // the bytes produced so far are the context's knowledge, not the caller's)
  struct ParkRec {
    uint64_t in_off = 0, in_len = 0, out_len = 0;
    uint8_t parked = 0, spill = 0, kind = 0;
    uint8_t input = 0;   // parked for input (MILZMA_ST_NEED_INPUT), not for room
    uint8_t fed = 0;     // ... by a call with MILZMA_DECODE_FEED: its view is expected to change
  };
  std::vector<ParkRec> park_rec;
  std::vector<milzma_unit> feed_units;   // MILZMA_DECODE_FEED: the descriptors as the device gets them (with MILZMA_KIND_LAST_VIEW)
  // The literal-row slab of class kFastSpill lives in `scratch`, indexed by unit with ONE stride for the whole batch.  While units of a
  // GROW batch are (or may still get) parked their trained rows exist only there: slab_live pins the stride (slab_lclp) and the
  // allocation until the parking lot is given up -- a RESUME launch or a promotion launch sees only a subset of the units and must
  // neither re-derive the stride from it nor wipe the other units' rows.
  bool slab_live = false;
  uint32_t slab_lclp = 0;
  uint32_t slab_min_lclp = 0;   // rows per unit (as lc + lp) the slab is made for at least: a batch whose members start later (host_stream.cpp)
  // Streamed launches (the whole-file calls' progressive download): a caller that sets stream_span / stream_spans before the async
  // half asks for the batch's ONE fast launch to run time-sliced with span counters (kernels.h); stream_active says it happened.
  // progress: kMaxSpans counters in mapped host memory, written by the device, polled by SpanPump.
  static constexpr uint32_t kMaxSpans = 256;
  uint32_t* progress = nullptr;
  uint32_t* progress_dev = nullptr;
  uint32_t stream_span = 0, stream_spans = 0;
  uint8_t* stream_host = nullptr;   // where the waves of a streamed launch write their output: pin_out, as the device sees it
  const uint64_t* stream_ptrs = nullptr;  // ... or, per unit, the caller's own page-locked result buffer (device array in `hostptrs`)
  bool stream_in_host = false;      // ... and its input is read from host memory that is still being filled (progress[kMaxSpans] = ready)
  bool stream_active = false;
  bool stream_feed = false;         // ... for EVERY time-sliced launch of the call, resuming ones too (push-mode streams: nobody waits on the span
                                    // counters, the waves just deliver what they decode into the units' result buffers, host_stream.cpp)
  PinBuf pin_results;
  hipStream_t copy_stream = nullptr;    // chunked staging copies of the whole-file batch entry points
  hipStream_t work_stream = nullptr;    // decode launches of the whole-file / host-buffer entry points: the context's own
                                        // stream, so that two contexts with a batch in flight each do not wait for each
                                        // other's kernels whenever one of them drains "its" stream
  float last_ms = 0.f;
  uint32_t last_launches = 0;
  uint32_t last_paths = 0;              // MILZMA_PATH_* of the most recent whole-file batch call (milzma_last_call_paths)
  // milzma_*_decompress_batch_async: the whole-file batch running on its own host thread until milzma_batch_wait
  std::thread batch_thread;
  bool batch_pending = false;
  int batch_rc = 0;
  // Lanes: further contexts on the same device.  A whole-file call with enough files is cut into groups that run one per lane,
  // concurrently (grouped_batch).  While a context works as a lane: its uploads wait for their turn (group order, so that the
  // first group's kernel starts after 1/G of the upload, not after all of it) and its planning budget is its share of the device.
  std::vector<milzma_ctx*> lanes;
  struct UploadTurn* turn = nullptr;
  uint32_t turn_no = 0;
  bool turn_done = true;
  uint32_t budget_share = 1;
  // MILZMA_KERNEL=generic (A/B runs, tests) turns the lane-resident-model kernel off: everything runs in the generic one.
  bool use_fast = true;
  bool fast_spill = true;    // MILZMA_SPILL=generic: lc + lp > 4 in the generic kernel (round 3's path) instead of the asm loop's HBM variant
  int slice_mode = 0;        // MILZMA_SLICE: 0 auto (launches that are not a whole number of chip-fulls), 1 always, -1 never ("0"),
                             // 2 always and every unit parked at every quantum even if nobody waits (tests)
  uint32_t slice_quantum = 128u << 10;  // MILZMA_QUANTUM: output bytes per turn of a time-sliced launch
  int order_mode = 0;    // MILZMA_ORDER: 0 sorted by input length (default), 1 stride, 2 shuffle (tuning)
  uint32_t lds_pad = 0;  // MILZMA_LDS_PAD: bytes of unused dynamic LDS per block of the fast kernel (occupancy experiments)
};

// the unit-level and whole-file calls behind the exported wrappers (host_api.cpp catches what they throw)
MILZMA_HIDDEN int milzma_decode_units_async_impl(milzma_ctx* ctx, const milzma_unit* units, uint32_t n, const void* d_in, void* d_out, void* hip_stream,
                                               uint32_t flags = 0, const milzma_result* prev = nullptr);
MILZMA_HIDDEN int milzma_decode_units_wait_impl(milzma_ctx* ctx, milzma_result* results);
MILZMA_HIDDEN int milzma_decode_units_impl(milzma_ctx* ctx, const milzma_unit* units, uint32_t n, const void* d_in, void* d_out, milzma_result* results,
                                         void* hip_stream, uint32_t flags = 0);
MILZMA_HIDDEN int milzma_decode_units_host_impl(milzma_ctx* ctx, const milzma_unit* units, uint32_t n, const void* h_in, size_t in_bytes, void* h_out,
                                              size_t out_bytes, milzma_result* results);
MILZMA_HIDDEN int move_units_impl(milzma_ctx* ctx, uint32_t n, const void* d_src, const uint64_t* src_off, void* d_dst, const uint64_t* dst_off,
                                const uint64_t* len, hipStream_t stream);
MILZMA_HIDDEN int milzma_lzma_decompress_impl(milzma_ctx* ctx, const uint8_t* in, size_t in_len, const milzma_options* opt, milzma_output* out);
MILZMA_HIDDEN int milzma_lzma2_decompress_impl(milzma_ctx* ctx, const uint8_t* in, size_t in_len, milzma_output* out);
MILZMA_HIDDEN int milzma_lzma_decompress_batch_impl(milzma_ctx* ctx, uint32_t n, const uint8_t* const* ins, const size_t* in_lens, const milzma_options* opt,
                                                  milzma_output* outs);
MILZMA_HIDDEN int milzma_lzma2_decompress_batch_impl(milzma_ctx* ctx, uint32_t n, const uint8_t* const* ins, const size_t* in_lens, milzma_output* outs);
MILZMA_HIDDEN int milzma_xz_decompress_batch_impl(milzma_ctx* ctx, uint32_t n, const uint8_t* const* ins, const size_t* in_lens, milzma_output* outs);

MILZMA_HOST_NS_BEGIN

// ---- host.cpp ------------------------------------------------------------------------------------------------------------------
const char* env_get(const char* name);      // getenv for a name of kEnvSwitches (aborts on any other)
bool hip_ok(milzma_ctx* ctx, hipError_t e, const char* what);
hipStream_t work_stream(milzma_ctx* ctx);   // the stream the library's own (host-buffer) entry points launch on
void trace_mark(milzma_ctx* ctx, const char* what);
void turn_acquire(milzma_ctx* ctx);
void turn_release(milzma_ctx* ctx);
bool dev_reserve(milzma_ctx* ctx, DevBuf& b, size_t bytes);
void dev_release(DevBuf& b);
bool pin_reserve(milzma_ctx* ctx, PinBuf& b, size_t bytes);
void pin_release(PinBuf& b);
unsigned host_threads();
size_t out_class(size_t n);
uint8_t* out_alloc(size_t n, bool pinned = false);
size_t out_live_buffers();
size_t out_pooled_buffers();
bool out_alloc_many(const size_t* n, size_t count, bool pinned, uint8_t** out);   // (thousands at once: one lock, not two per buffer)
LitClass classify(const milzma_ctx* ctx, const milzma_unit& u);
bool ensure_progress(milzma_ctx* ctx);
extern const char* const kEofMsg;
extern const char* const kPrefix[6];
uint32_t crc32_update(uint32_t c, const uint8_t* p, size_t n);
void crc_fold(const uint8_t* parts, uint64_t len, uint32_t* crc32, uint64_t* crc64);

// runs fn(i) for i in [0, n) on up to host_threads() threads
template <class F>
void parallel_for(size_t n, F fn) {
  const unsigned t = unsigned(std::min<size_t>(host_threads(), n));
  if (t <= 1) {
    for (size_t i = 0; i < n; i++) fn(i);
    return;
  }
  // (a thread that cannot be started -- std::system_error -- must not take the process down through the vector's destructor while
  //  its siblings run: its stride is done here, the ones that did start are joined)
  std::vector<std::thread> pool;
  try {
    pool.reserve(t);
  } catch (const std::bad_alloc&) {
    for (size_t i = 0; i < n; i++) fn(i);
    return;
  }
  for (unsigned k = 0; k < t; k++) {
    try {
      pool.emplace_back([=] {
        for (size_t i = k; i < n; i += t) fn(i);
      });
    } catch (const std::exception&) {
      for (size_t i = k; i < n; i += t) fn(i);
    }
  }
  for (auto& th : pool) th.join();
}

// result buffers taken from the pool and not yet handed to the caller: back to the pool when the scope is left, however it is left
// (drop() nulls what it frees: a pooled buffer may be somebody else's a moment later)
struct HeldBufs {
  std::vector<uint8_t*> v;
  void drop() {
    for (uint8_t*& b : v) {
      if (b) milzma_free(b);
      b = nullptr;
    }
  }
  ~HeldBufs() { drop(); }
};

// joins a helper thread when the scope is left, however it is left (a joinable std::thread's destructor is std::terminate)
struct JoinOnExit {
  std::thread& th;
  std::atomic<bool>& stop;
  ~JoinOnExit() {
    stop.store(true, std::memory_order_release);
    if (th.joinable()) th.join();
  }
};

// A large device -> pinned-host copy cut in chunks with an event behind each, so that host threads can start on the
// front of the buffer while the back is still crossing PCIe (and the mirror image for host -> device).
struct ChunkedCopy {
  static constexpr size_t kChunk = size_t(64) << 20;
  milzma_ctx* ctx = nullptr;
  std::vector<hipEvent_t> ev;
  bool ok = true;

  bool stream_ready(milzma_ctx* c) {
    ctx = c;
    if (!ctx->copy_stream && !hip_ok(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking), "hipStreamCreate"))
      return false;
    return true;
  }
  // device [0, bytes) -> host, asynchronously; wait_until(end) blocks until [0, end) has arrived
  bool start_d2h(milzma_ctx* c, void* host, const void* dev, size_t bytes) {
    if (!stream_ready(c)) return ok = false;
    for (size_t o = 0; o < bytes; o += kChunk) {
      const size_t n = std::min(kChunk, bytes - o);
      hipEvent_t e = nullptr;
      if (!hip_ok(ctx, hipMemcpyAsync(static_cast<uint8_t*>(host) + o, static_cast<const uint8_t*>(dev) + o, n, hipMemcpyDeviceToHost,
                                      ctx->copy_stream),
                  "D2H output") ||
          !hip_ok(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate") ||
          !hip_ok(ctx, hipEventRecord(e, ctx->copy_stream), "hipEventRecord")) {
        if (e) (void)hipEventDestroy(e);
        (void)hipStreamSynchronize(ctx->copy_stream);
        return ok = false;
      }
      ev.push_back(e);
    }
    return true;
  }
  bool wait_until(size_t end) const {  // callable from several threads
    if (!ok) return false;
    if (end == 0 || ev.empty()) return true;
    const size_t k = std::min((end - 1) / kChunk, ev.size() - 1);
    return hipEventSynchronize(ev[k]) == hipSuccess;
  }
  ~ChunkedCopy() {
    if (ctx && ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  }
};

// Host -> device staging in groups: `fill(g)` writes group g's bytes [lo, hi) of the pinned buffer (on the host threads),
// then that range is sent; the next group is filled while this one crosses PCIe.  bounds: groups + 1 ascending offsets.
template <class F>
bool staged_h2d(milzma_ctx* ctx, void* dev, const void* host, const std::vector<size_t>& bounds, F fill) {
  ChunkedCopy cc;
  if (!cc.stream_ready(ctx)) return false;
  struct Turn {
    milzma_ctx* c;
    ~Turn() { turn_release(c); }
  } turn{ctx};
  trace_mark(ctx, "upload: start");
  if (bounds.size() > 1) fill(0);  // (the first gather does not need the link)
  turn_acquire(ctx);
  trace_mark(ctx, "upload: has the turn");
  for (size_t g = 0; g + 1 < bounds.size(); g++) {
    if (g) fill(g);
    const size_t lo = bounds[g], hi = bounds[g + 1];
    if (hi > lo && !hip_ok(ctx,
                           hipMemcpyAsync(static_cast<uint8_t*>(dev) + lo, static_cast<const uint8_t*>(host) + lo, hi - lo,
                                          hipMemcpyHostToDevice, ctx->copy_stream),
                           "H2D input"))
      return false;
  }
  const bool ok = hip_ok(ctx, hipStreamSynchronize(ctx->copy_stream), "hipStreamSynchronize");
  trace_mark(ctx, "upload: done");
  return ok;
}

// ---- host_files.cpp ------------------------------------------------------------------------------------------------------------
struct Cursor {  // io::BufRead over a slice
  const uint8_t* p;
  size_t pos, end;
  bool u8(uint8_t* v) {
    if (pos >= end) return false;
    *v = p[pos++];
    return true;
  }
  bool exact(uint8_t* dst, size_t n) {  // read_exact: a short read consumes what there is
    if (end - pos < n) {
      pos = end;
      return false;
    }
    if (dst) memcpy(dst, p + pos, n);
    pos += n;
    return true;
  }
  bool u16be(uint32_t* v) {
    uint8_t b[2];
    if (!exact(b, 2)) return false;
    *v = (uint32_t(b[0]) << 8) | b[1];
    return true;
  }
  bool u32le(uint32_t* v) {
    uint8_t b[4];
    if (!exact(b, 4)) return false;
    *v = uint32_t(b[0]) | (uint32_t(b[1]) << 8) | (uint32_t(b[2]) << 16) | (uint32_t(b[3]) << 24);
    return true;
  }
  bool u64le(uint64_t* v) {
    uint8_t b[8];
    if (!exact(b, 8)) return false;
    *v = 0;
    for (int i = 7; i >= 0; i--) *v = (*v << 8) | b[i];
    return true;
  }
  bool eof() const { return pos >= end; }
};

void out_reset(milzma_output* o);
int out_fail(milzma_output* o, int kind, const char* fmt, ...);
int out_io_eof(milzma_output* o);
bool out_set_data(milzma_output* o, const uint8_t* p, size_t n);
int infra(milzma_ctx* ctx, milzma_output* o);
int finish_stream(const milzma_result& r, uint32_t kind, const uint8_t* slice, size_t slice_len, size_t header_len, milzma_output* out);
bool upload_host_ptrs(milzma_ctx* ctx, const std::vector<uint64_t>& ptrs, hipStream_t ws);
void stream_minimum(size_t* units, size_t* bytes, bool* ragged_ok = nullptr);
bool pinned_results_wanted();
extern std::atomic<int> g_streamed_in_flight[64];
struct StreamedSlot {
  int dev = -1;
  bool try_take(int device) {
    if (device < 0 || device >= 64) return false;
    if (g_streamed_in_flight[device].fetch_add(1) != 0) {
      g_streamed_in_flight[device].fetch_sub(1);
      return false;
    }
    dev = device;
    return true;
  }
  ~StreamedSlot() {
    if (dev >= 0) g_streamed_in_flight[dev].fetch_sub(1);
  }
};

// Two-part upload for streamed launches.  A decode kernel needs the FIRST bytes of every unit when it starts and the rest only as
// fast as it decodes (6 GB/s for the whole chip, against 50 on the link), so:
//   begin:  the first stream_lead_bytes of every unit (its "lead") are gathered into one page-locked block, go up with one copy and
//           are put in place by one move kernel -- a few ms, then the kernel can be launched with in_ready = 0;
//   finish: while it runs, the complete input is gathered into the page-locked input buffer and sent in large consecutive pieces
//           (the copy engines work beside the kernel).  The pieces overwrite the leads with the bytes they already hold, which is
//           harmless; when the last piece has landed the ready word is set.  A wave that would come within a turn's reach of the end
//           of its lead before that waits (kernels.h: in_ready) -- a safety net, not the normal course.
// src(k): where unit k's input bytes are in the caller's memory.
template <class Src>
bool upload_leads(milzma_ctx* ctx, const std::vector<milzma_unit>& units, Src src, hipStream_t ws) {
  const uint32_t nu = uint32_t(units.size());
  std::vector<uint64_t> so(nu), dof(nu), ln(nu);
  size_t total = 0;
  for (uint32_t k = 0; k < nu; k++) {
    ln[k] = std::min<uint64_t>(units[k].in_len, stream_lead_bytes(uint32_t(std::min<uint64_t>(units[k].in_len, 0xFFFFFF00u))));
    so[k] = total;
    dof[k] = units[k].in_off;
    total += round_up(size_t(ln[k]), 256);
  }
  if (!pin_reserve(ctx, ctx->pin_lead, total) || !dev_reserve(ctx, ctx->pack, total + 512)) return false;
  uint8_t* h = static_cast<uint8_t*>(ctx->pin_lead.p);
  // in four pieces: the host threads gather piece g + 1 while piece g crosses the link (round 6: the one copy behind the whole gather was
  // 6 of this step's 16 ms at configs[1]'s size)
  const uint32_t pieces = nu >= 64 ? 4 : 1;
  for (uint32_t g = 0; g < pieces; g++) {
    const uint32_t k0 = uint32_t(uint64_t(nu) * g / pieces), k1 = uint32_t(uint64_t(nu) * (g + 1) / pieces);
    if (k1 == k0) continue;
    parallel_for(k1 - k0, [&](size_t k) { memcpy(h + so[k0 + k], src(k0 + k), size_t(ln[k0 + k])); });
    const size_t lo = size_t(so[k0]), hi = k1 < nu ? size_t(so[k1]) : total;
    if (!hip_ok(ctx, hipMemcpyAsync(static_cast<uint8_t*>(ctx->pack.p) + lo, h + lo, hi - lo, hipMemcpyHostToDevice, ws), "H2D leads")) {
      (void)hipStreamSynchronize(ws);   // (the pieces already queued read pin_lead)
      return false;
    }
  }
  return move_units_impl(ctx, nu, ctx->pack.p, so.data(), ctx->in.p, dof.data(), ln.data(), ws) == MILZMA_OK;
}

// bounds: ascending offsets into the input buffer (pieces); fill(g) gathers piece g's bytes [bounds[g], bounds[g + 1]) into hin
template <class F>
bool upload_rest(milzma_ctx* ctx, uint8_t* hin, const std::vector<size_t>& bounds, F fill) {
  ChunkedCopy cc;
  bool ok = cc.stream_ready(ctx);
  for (size_t g = 0; ok && g + 1 < bounds.size(); g++) {
    fill(g);
    const size_t lo = bounds[g], hi = bounds[g + 1];
    if (hi > lo)
      ok = hip_ok(ctx, hipMemcpyAsync(static_cast<uint8_t*>(ctx->in.p) + lo, hin + lo, hi - lo, hipMemcpyHostToDevice, ctx->copy_stream), "H2D input");
  }
  // (drained whatever became of the pieces: the ones that were queued write ctx->in, which whoever runs next -- the classic rounds, a file
  //  decoded on its own -- is about to use; found by ThreadSanitizer under fault injection)
  const std::string why = ctx->err;
  const bool drained = hipStreamSynchronize(ctx->copy_stream) == hipSuccess;
  if (!ok)
    ctx->err = why;
  else if (!drained)
    ok = hip_ok(ctx, hipErrorUnknown, "hipStreamSynchronize");
  // (ready also when a copy failed: the waves must not wait for ever -- the caller fails the call)
  __atomic_store_n(&ctx->progress[milzma_ctx::kMaxSpans], 1u, __ATOMIC_RELEASE);
  return ok;
}

// One unit through the device with host buffers.  Its output slice grows while the stream needs more room: a unit of the fast
// kernels is parked at the end of its slice and resumed in a larger one (nothing is decoded twice); a unit of the generic kernel
// (lc + lp > 4) reports a plain OUT_FULL and starts over with four times the room.  `cap_hint` is the first slice size to try.
struct SingleDecode {
  milzma_result res;
  std::vector<uint8_t> out;  // the unit's output slice (res.out_len bytes valid, capped by size)
};
bool decode_single(milzma_ctx* ctx, milzma_unit u, const uint8_t* in, size_t in_len, size_t cap_hint, SingleDecode* sd);

// ---- host_xz.cpp ---------------------------------------------------------------------------------------------------------------
// ---- batching: find the blocks of well-formed files up front through the Index ------------
struct PlannedBlock {
  size_t data_off;    // first byte of the block's LZMA2 payload within the file
  size_t data_len;    // payload bytes according to the Index (unpadded - header - check)
  uint64_t unpacked;  // uncompressed size according to the Index
};
size_t plan_budget(milzma_ctx* ctx);
bool plan_from_index(const uint8_t* in, size_t n, std::vector<PlannedBlock>* blocks);

MILZMA_HOST_NS_END
