// host.cpp -- host side of the C ABI (include/milzma.h).
//
// What runs here is what the reference runs *around* its hot loop: header and container
// parsing (LzmaParams::read_header, src/decode/lzma.rs:96-161; xz::decode_stream,
// src/decode/xz.rs), option handling (src/decode/options.rs), error rendering (src/error.rs)
// and -- new -- turning many streams / blocks into one batch of wavefront-sized decode units.
// All decoding happens in the HIP kernels; there is no CPU decode path in this library.
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

using namespace milzma;

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};
struct PinBuf {  // page-locked host staging (hipHostMalloc): PCIe copies run at link speed from / to it
  void* p = nullptr;
  size_t cap = 0;
};

thread_local std::string g_create_error;

// ---- environment switches: every MILZMA_* variable the library reads, in ONE table (tests/test_host_abi.py holds README.md against it).
// "when": create = read once when a context (or the pool / the handle) is made; call = read at every call that could use it, so that a
// test or an operator can flip it between calls of one process (round 4's knobs were cached in function-local statics: whichever call
// came first in a process fixed them for good, and a test that set one later ran another path than it thought).
struct EnvSwitch {
  const char* name;
  const char* when;
  const char* what;
};
constexpr EnvSwitch kEnvSwitches[] = {
    {"MILZMA_KERNEL", "create", "generic: every unit in the generic LDS-model kernel (a test double since round 4)"},
    {"MILZMA_SPILL", "create", "generic: lc + lp >= 4 in the generic kernel instead of the asm loop's HBM variant"},
    {"MILZMA_SLICE", "create", "0: never time-slice a launch; 1: always; 2: always, and park every unit at every quantum (tests)"},
    {"MILZMA_QUANTUM", "create", "output bytes per turn of a time-sliced launch (default 128 KiB)"},
    {"MILZMA_ORDER", "create", "tuning: stride / shuffle instead of longest-input-first inside a launch"},
    {"MILZMA_LDS_PAD", "create", "tuning: bytes of unused dynamic LDS per block (occupancy experiments)"},
    {"MILZMA_POOL_BYTES", "create", "bytes of result buffers kept behind milzma_free (default 8 GiB)"},
    {"MILZMA_HOST_THREADS", "call", "host threads of the whole-file batch entry points (default min(16, cores))"},
    {"MILZMA_MULTI_REPLICAS", "create", "testing aid: k contexts per device behind a milzma_multi"},
    {"MILZMA_TRACE", "create", "phase marks of the whole-file batch calls on stderr"},
    {"MILZMA_PLAN_BUDGET", "call", "bytes of staging the batch entry points may plan ahead for (default 3/4 of free device memory)"},
    {"MILZMA_STREAM", "call", "0: no streamed launches"},
    {"MILZMA_STREAM_MIN", "call", "units[,bytes[,1]] from which a batch is streamed (default 256 units and 256 MiB of output; third field: also ragged batches)"},
    {"MILZMA_SPAN", "call", "bytes per output span of a streamed launch (default 64 KiB)"},
    {"MILZMA_PINNED_OUT", "call", "0: pageable result buffers (a host thread copies spans out of a staging buffer)"},
    {"MILZMA_TWO_PART", "call", "1: .xz batches upload their input in two parts like .lzma batches"},
    {"MILZMA_ROOTED_STREAM", "call", "0: the one-ingest-point entry brings output home by a copy behind the decode"},
    {"MILZMA_ROOTED_PEER", "call", "0: the one-ingest-point entry treats peer access to the root as denied (tests: the fallback when a device cannot reach the root's memory)"},
    {"MILZMA_LANES", "call", "contexts a large whole-file call is spread over (default 2; 3-4 need GPU_MAX_HW_QUEUES >= 2 x lanes + 1)"},
    {"MILZMA_NO_GROUPS", "call", "a whole-file call is never cut into groups"},
};
const char* env_get(const char* name) {
  bool known = false;
  for (const EnvSwitch& e : kEnvSwitches) known = known || !strcmp(e.name, name);
  if (!known) {  // (a switch outside the table is a programming error: loud in every build)
    fprintf(stderr, "milzma: environment switch %s is not in kEnvSwitches\n", name);
    abort();
  }
  return getenv(name);
}

}  // namespace

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
  };
  std::vector<ParkRec> park_rec;
  // The literal-row slab of class kFastSpill lives in `scratch`, indexed by unit with ONE stride for the whole batch.  While units of a
  // GROW batch are (or may still get) parked their trained rows exist only there: slab_live pins the stride (slab_lclp) and the
  // allocation until the parking lot is given up -- a RESUME launch or a promotion launch sees only a subset of the units and must
  // neither re-derive the stride from it nor wipe the other units' rows.
  bool slab_live = false;
  uint32_t slab_lclp = 0;
  // Streamed launches (the whole-file calls' progressive download): a caller that sets stream_span / stream_spans before the async
  // half asks for the batch's ONE fast launch to run time-sliced with span counters (kernels.h); stream_active says it happened.
  // progress: kMaxSpans counters in mapped host memory, written by the device, polled by SpanPump.
  static constexpr uint32_t kMaxSpans = 64;
  uint32_t* progress = nullptr;
  uint32_t* progress_dev = nullptr;
  uint32_t stream_span = 0, stream_spans = 0;
  uint8_t* stream_host = nullptr;   // where the waves of a streamed launch write their output: pin_out, as the device sees it
  const uint64_t* stream_ptrs = nullptr;  // ... or, per unit, the caller's own page-locked result buffer (device array in `hostptrs`)
  bool stream_in_host = false;      // ... and its input is read from host memory that is still being filled (progress[kMaxSpans] = ready)
  bool stream_active = false;
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

namespace {

bool hip_ok(milzma_ctx* ctx, hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  char buf[256];
  snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  if (ctx)
    ctx->err = buf;
  else
    g_create_error = buf;
  return false;
}

// the stream the library's own (host-buffer) entry points launch on; the legacy stream if it cannot be created
hipStream_t work_stream(milzma_ctx* ctx) {
  if (!ctx->work_stream && hipStreamCreateWithFlags(&ctx->work_stream, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    ctx->work_stream = nullptr;
  }
  return ctx->work_stream;
}

// MILZMA_TRACE=1: wall-clock marks of the whole-file batch phases on stderr (tuning)
void trace_mark(milzma_ctx* ctx, const char* what) {
  static const bool on = env_get("MILZMA_TRACE") != nullptr;
  if (!on) return;
  using namespace std::chrono;
  static const steady_clock::time_point t0 = steady_clock::now();
  fprintf(stderr, "[milzma %p group %u] %8.1f ms  %s\n", static_cast<void*>(ctx), ctx->turn ? ctx->turn_no : 0u,
          duration<double, std::milli>(steady_clock::now() - t0).count(), what);
}

// (no-ops unless the context is working as a lane of a grouped call)
void turn_acquire(milzma_ctx* ctx) {
  if (!ctx->turn || ctx->turn_done) return;
  std::unique_lock<std::mutex> lock(ctx->turn->mu);
  ctx->turn->cv.wait(lock, [&] { return ctx->turn->next >= ctx->turn_no; });
}
void turn_release(milzma_ctx* ctx) {
  if (!ctx->turn || ctx->turn_done) return;
  ctx->turn_done = true;
  {
    std::lock_guard<std::mutex> lock(ctx->turn->mu);
    ctx->turn->next = std::max(ctx->turn->next, ctx->turn_no + 1);
  }
  ctx->turn->cv.notify_all();
}

bool dev_reserve(milzma_ctx* ctx, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return true;
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
  size_t want = std::max(bytes, size_t(1) << 16);
  want = (want + 4095) & ~size_t(4095);
  if (!hip_ok(ctx, hipMalloc(&b.p, want), "hipMalloc")) return false;
  b.cap = want;
  return true;
}

void dev_release(DevBuf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

bool pin_reserve(milzma_ctx* ctx, PinBuf& b, size_t bytes) {
  if (bytes <= b.cap) return true;
  if (b.p) (void)hipHostFree(b.p);
  b.p = nullptr;
  b.cap = 0;
  const size_t want = (std::max(bytes, size_t(1) << 16) + 4095) & ~size_t(4095);
  if (!hip_ok(ctx, hipHostMalloc(&b.p, want, hipHostMallocDefault), "hipHostMalloc")) return false;
  b.cap = want;
  return true;
}

void pin_release(PinBuf& b) {
  if (b.p) (void)hipHostFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

unsigned host_threads() {
  if (const char* e = env_get("MILZMA_HOST_THREADS")) return unsigned(std::max(1, atoi(e)));
  const unsigned hw = std::thread::hardware_concurrency();
  return std::min(16u, std::max(1u, hw));
}

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

}  // namespace

extern "C" uint32_t milzma_abi_version(void) { return MILZMA_ABI_VERSION; }

extern "C" int milzma_create(int device, milzma_ctx** out_ctx) {
  if (!out_ctx) return MILZMA_INFRA_ERROR;
  *out_ctx = nullptr;
  int count = 0;
  if (!hip_ok(nullptr, hipGetDeviceCount(&count), "hipGetDeviceCount")) return MILZMA_INFRA_ERROR;
  if (count <= 0 || device < 0 || device >= count) {
    g_create_error = "no usable HIP device (this library has no CPU decode path)";
    return MILZMA_INFRA_ERROR;
  }
  if (!hip_ok(nullptr, hipSetDevice(device), "hipSetDevice")) return MILZMA_INFRA_ERROR;
  hipDeviceProp_t prop;
  if (!hip_ok(nullptr, hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) return MILZMA_INFRA_ERROR;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_create_error = std::string("device is ") + prop.gcnArchName + ", the kernels are built for gfx950 only";
    return MILZMA_INFRA_ERROR;
  }
  auto* ctx = new milzma_ctx();
  ctx->device = device;
  if (const char* k = env_get("MILZMA_KERNEL")) {
    ctx->use_fast = strcmp(k, "generic") != 0;
  }
  if (const char* k = env_get("MILZMA_SPILL")) ctx->fast_spill = strcmp(k, "generic") != 0;
  if (const char* k = env_get("MILZMA_SLICE")) ctx->slice_mode = !strcmp(k, "2") ? 2 : !strcmp(k, "1") ? 1 : !strcmp(k, "0") ? -1 : 0;
  if (const char* k = env_get("MILZMA_QUANTUM")) ctx->slice_quantum = std::max<uint32_t>(1u, uint32_t(strtoul(k, nullptr, 0)));
  if (const char* k = env_get("MILZMA_ORDER")) ctx->order_mode = !strcmp(k, "stride") ? 1 : !strcmp(k, "shuffle") ? 2 : 0;
  if (const char* k = env_get("MILZMA_LDS_PAD")) {
    ctx->lds_pad = uint32_t(strtoul(k, nullptr, 0));
  }
  if (!hip_ok(nullptr, hipEventCreate(&ctx->ev0), "hipEventCreate") ||
      !hip_ok(nullptr, hipEventCreate(&ctx->ev1), "hipEventCreate")) {
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);   // (the first one may exist: fault injection found it left behind)
    delete ctx;
    return MILZMA_INFRA_ERROR;
  }
  *out_ctx = ctx;
  return MILZMA_OK;
}

extern "C" void milzma_destroy(milzma_ctx* ctx) {
  if (!ctx) return;
  if (ctx->batch_thread.joinable()) ctx->batch_thread.join();
  for (milzma_ctx* lane : ctx->lanes) milzma_destroy(lane);
  ctx->lanes.clear();
  (void)hipSetDevice(ctx->device);
  dev_release(ctx->units);
  dev_release(ctx->order);
  dev_release(ctx->results);
  dev_release(ctx->scratch);
  dev_release(ctx->in);
  dev_release(ctx->out);
  dev_release(ctx->pack);
  dev_release(ctx->hostptrs);
  dev_release(ctx->crc);
  dev_release(ctx->flags);
  dev_release(ctx->slice_q);
  dev_release(ctx->slice_ctx);
  pin_release(ctx->pin_in);
  pin_release(ctx->pin_out);
  pin_release(ctx->pin_small);
  pin_release(ctx->pin_lead);
  pin_release(ctx->pin_moves);
  for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->work_stream) (void)hipStreamDestroy(ctx->work_stream);
  pin_release(ctx->pin_results);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->progress) (void)hipHostFree(ctx->progress);
  delete ctx;
}

extern "C" const char* milzma_last_error(const milzma_ctx* ctx) {
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

extern "C" uint32_t milzma_last_call_paths(const milzma_ctx* ctx) { return ctx ? ctx->last_paths : 0u; }

extern "C" float milzma_last_kernel_ms(const milzma_ctx* ctx, uint32_t* launches) {
  if (launches) *launches = ctx ? ctx->last_launches : 0;
  return ctx ? ctx->last_ms : 0.f;
}

// ---- output buffers: a pool behind out_set_data / milzma_free --------------------------------------------------------
// A batch call hands back thousands of MiB-sized buffers.  Fresh from malloc each is its own mmap: a million page faults per
// 4 GiB call (and as many munmaps when the caller frees them), all serialised on the process's mmap lock -- a third of
// the call's host time.  Buffers freed with milzma_free are kept by size class and handed out again with their pages already
// mapped.  What the pool may hold: never more than the caller had handed out at once (the high-water mark of live bytes, so a
// process that decodes 64 MiB at a time keeps 64 MiB), never more than MILZMA_POOL_BYTES (default 8 GiB); milzma_pool_trim
// gives memory back on request.  Every pointer handed out is registered: milzma_free looks a pointer up instead of reading the
// bytes in front of it, so a foreign pointer (or one freed twice) is recognised without being dereferenced.
namespace {

struct OutHdr {
  uint64_t cap;
  uint64_t pinned;  // 1: page-locked (hipHostMalloc), a streamed launch writes it from the device; (also keeps the payload 16-byte aligned)
};

void hdr_free(OutHdr* h) {
  if (h->pinned)
    (void)hipHostFree(h);
  else
    free(h);
}

struct OutPool {
  std::mutex mu;
  std::map<size_t, std::vector<OutHdr*>> free_by_cap;   // ordered: a request takes the smallest class that holds it
  std::map<size_t, std::vector<OutHdr*>> free_pinned;   // the same for page-locked buffers (never handed out for ordinary requests)
  std::unordered_set<const void*> live;                  // payload pointers handed out and not yet freed
  size_t held = 0, live_bytes = 0, peak_live = 0, limit = size_t(8) << 30;
  OutPool() {
    if (const char* e = env_get("MILZMA_POOL_BYTES")) limit = size_t(strtoull(e, nullptr, 0));
  }
  ~OutPool() {
    for (auto& kv : free_by_cap)
      for (OutHdr* h : kv.second) free(h);
    // (page-locked buffers still pooled at exit are left to the process's end: the HIP runtime may be gone already)
  }
  // (mu held) frees pooled buffers, largest classes first, until at most `keep` bytes rest in the pool
  void trim_locked(size_t keep) {
    for (auto* m : {&free_pinned, &free_by_cap})
      for (auto it = m->end(); held > keep && it != m->begin();) {
        --it;
        while (held > keep && !it->second.empty()) {
          OutHdr* h = it->second.back();
          it->second.pop_back();
          held -= size_t(h->cap);
          hdr_free(h);
        }
        if (it->second.empty()) it = m->erase(it);
      }
  }
};
OutPool& out_pool() {
  static OutPool p;
  return p;
}

size_t out_class(size_t n) {  // capacity class: powers of two up to 64 KiB, multiples of 64 KiB above
  if (n <= 4096) return 4096;
  if (n <= (size_t(1) << 16)) {
    size_t c = 4096;
    while (c < n) c <<= 1;
    return c;
  }
  return (n + 0xFFFF) & ~size_t(0xFFFF);
}

// pinned: page-locked memory a streamed launch can write from the device (hipHostMalloc, portable: any device of the node); such
// buffers are handed out to the caller like any other and come back through milzma_free into a pool of their own
uint8_t* out_alloc(size_t n, bool pinned = false) {
  const size_t cap = out_class(n);
  OutPool& p = out_pool();
  OutHdr* h = nullptr;
  try {
    {
      std::lock_guard<std::mutex> lock(p.mu);
      // best fit: the smallest pooled class that holds the request, as long as it wastes at most half of itself (a workload of
      // varied sizes reuses what it has instead of filling the pool with classes that never match exactly)
      auto& m = pinned ? p.free_pinned : p.free_by_cap;
      auto it = m.lower_bound(cap);
      while (it != m.end() && it->second.empty()) it = m.erase(it);
      if (it != m.end() && it->first <= std::max(cap * 2, cap + (size_t(1) << 16))) {
        h = it->second.back();
        it->second.pop_back();
        p.held -= size_t(h->cap);
      }
    }
    if (!h) {  // (allocated outside the lock: pinning a MiB takes its time, and thousands are wanted at once)
      if (pinned) {
        void* q = nullptr;
        if (hipHostMalloc(&q, sizeof(OutHdr) + cap, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
          (void)hipGetLastError();
          return nullptr;
        }
        h = static_cast<OutHdr*>(q);
      } else {
        h = static_cast<OutHdr*>(malloc(sizeof(OutHdr) + cap));
        if (!h) return nullptr;
      }
      h->cap = cap;
      h->pinned = pinned ? 1 : 0;
    }
    std::lock_guard<std::mutex> lock(p.mu);
    p.live.insert(h + 1);
    p.live_bytes += size_t(h->cap);
    p.peak_live = std::max(p.peak_live, p.live_bytes);
  } catch (const std::bad_alloc&) {  // (the registry could not grow: the buffer is not handed out)
    if (h) hdr_free(h);
    return nullptr;
  }
  return reinterpret_cast<uint8_t*>(h + 1);
}

}  // namespace

extern "C" void milzma_free(void* ptr) {
  if (!ptr) return;
  OutPool& p = out_pool();
  OutHdr* h = nullptr;
  {
    std::lock_guard<std::mutex> lock(p.mu);
    const auto it = p.live.find(ptr);
    if (it == p.live.end()) return;  // not handed out by this library, or freed already: never dereferenced, left alone
    p.live.erase(it);
    h = static_cast<OutHdr*>(ptr) - 1;
    p.live_bytes -= size_t(h->cap);
    if (p.held + h->cap <= std::min(p.limit, p.peak_live)) {
      try {
        (h->pinned ? p.free_pinned : p.free_by_cap)[size_t(h->cap)].push_back(h);
        p.held += size_t(h->cap);
        return;
      } catch (const std::bad_alloc&) {
      }
    }
  }
  hdr_free(h);
}

extern "C" size_t milzma_pool_trim(size_t keep_bytes) {
  OutPool& p = out_pool();
  std::lock_guard<std::mutex> lock(p.mu);
  p.trim_locked(keep_bytes);
  p.peak_live = p.live_bytes;  // (the high-water mark starts over: the pool refills only as far as later calls go)
  return p.held;
}

extern "C" void milzma_default_options(milzma_options* opt) {
  if (opt) memset(opt, 0, sizeof *opt);
}

// ------------------------------------------------------------------------------------------
// the batch entry point
// ------------------------------------------------------------------------------------------

namespace {

// HBM scratch the spill class (lc + lp > 4) may hold for its literal tables: the slab is sized by the launch's largest
// lc + lp (1.5 KiB << lclp per block: 384 KiB at lc 8, 6 MiB at lc + lp = 12), and a launch takes as many units as fit
// (round 2 launched 32 at a time with 6 MiB each whatever their properties: 0.06 GB/s on 4096 streams).
constexpr size_t kSpillSlabBytes = size_t(24) << 30;

LitClass classify(const milzma_ctx* ctx, const milzma_unit& u) {
  // LZMA2 units start in the cheapest class; NEED_GENERIC / NEED_LCLP promote them when a chunk
  // switches to properties that class is not built for.
  if (u.kind != MILZMA_KIND_RAW_LZMA) return ctx->use_fast ? kFast : kLitLds3;
  const uint32_t lclp = uint32_t(u.lc) + u.lp;
  if (ctx->use_fast && u.pb <= 4 && lclp <= 3) return kFast;
  if (ctx->use_fast && ctx->fast_spill && u.pb <= 4 && u.lc <= 8 && u.lp <= 4) return kFastSpill;   // lc + lp >= 4: the loop's HBM variant
  if (lclp <= 3) return kLitLds3;
  if (lclp <= 4) return kLitLds4;
  return kLitSpill;
}

// the span counters (+ the input-ready word behind them) of streamed launches: host memory the device can reach
bool ensure_progress(milzma_ctx* ctx) {
  if (ctx->progress) return true;
  void* hp = nullptr;
  void* dp = nullptr;
  if (hipHostMalloc(&hp, (milzma_ctx::kMaxSpans + 16) * sizeof(uint32_t), hipHostMallocMapped) == hipSuccess &&
      hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
    ctx->progress = static_cast<uint32_t*>(hp);
    ctx->progress_dev = static_cast<uint32_t*>(dp);
    memset(hp, 0, (milzma_ctx::kMaxSpans + 16) * sizeof(uint32_t));
    return true;
  }
  (void)hipGetLastError();
  if (hp) (void)hipHostFree(hp);
  return false;
}

// Launches `order` (unit indices) in class `cls`; kernel time is accumulated into ctx.
bool launch_class(milzma_ctx* ctx, LitClass cls, const std::vector<uint32_t>& order, uint32_t order_base,
                  const uint8_t* d_in, uint8_t* d_out, hipStream_t stream, bool grow = false, bool resume = false) {
  if (order.empty()) return true;
  auto* d_units = static_cast<const milzma_unit*>(ctx->units.p);
  auto* d_order = static_cast<const uint32_t*>(ctx->order.p) + order_base;
  auto* d_results = static_cast<milzma_result*>(ctx->results.p);
  const uint32_t n = uint32_t(order.size());
  uint32_t step = n, spill_lclp = 0;
  const bool is_fast = cls == kFast || cls == kFastSpill;
  size_t slab_bytes = 0;
  if (cls == kFastSpill) {
    // The literal rows of every unit of the BATCH in one slab (indexed by unit, like the results: a unit keeps its rows across the
    // turns of a time-sliced launch and across a park / resume), every probability of THIS launch's units 0x400 before it starts.  Not
    // to be had (half of the free memory at most): the generic kernel's spill class takes the units, chunk by chunk.
    uint32_t need = 4;   // (at least what an LZMA2 unit can switch to)
    for (uint32_t i : order) need = std::max<uint32_t>(need, uint32_t(ctx->pend_units[i].lc) + ctx->pend_units[i].lp);
    need = std::min<uint32_t>(need, 12);
    if (ctx->slab_live) {
      // units of this batch are parked (or may be, by an earlier launch of this very call) with their rows in the slab: the stride and the
      // allocation are the ones the batch's first launch of this class chose -- never derived from the subset this launch sees
      spill_lclp = ctx->slab_lclp;
      slab_bytes = spill_bytes_per_block(spill_lclp);
      if (need > spill_lclp || slab_bytes * ctx->pend_n > ctx->scratch.cap) {
        ctx->err = "the literal-row slab of the parked units does not fit this launch";
        return false;
      }
    } else {
      spill_lclp = need;
      slab_bytes = spill_bytes_per_block(spill_lclp);
      const size_t total = slab_bytes * ctx->pend_n;
      size_t free_b = 0, total_b = 0;
      const bool fits = slab_bytes <= 0xFFFFFFFFu && total <= kSpillSlabBytes * 4 &&
                        (total <= ctx->scratch.cap || (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total <= free_b / 2 + ctx->scratch.cap));
      const std::string keep = ctx->err;
      if (resume || !fits || !dev_reserve(ctx, ctx->scratch, total)) {
        ctx->err = keep;
        if (resume) {   // (units parked in this class without a live slab: the context was used for something else in between)
          ctx->err = "no literal-row slab for the parked units";
          return false;
        }
        return launch_class(ctx, kLitSpill, order, order_base, d_in, d_out, stream);
      }
      ctx->slab_lclp = spill_lclp;
      ctx->slab_live = grow;   // (only a growing batch parks units beyond its launches)
    }
    if (!resume && !hip_ok(ctx, launch_slab_init(static_cast<uint8_t*>(ctx->scratch.p), uint32_t(slab_bytes), d_order, n, stream), "slab init"))
      return false;
  }
  if (cls == kLitSpill) {
    if (ctx->slab_live) {   // (its table would go where the parked units' rows are)
      ctx->err = "the generic spill class cannot run while parked units keep their literal rows in the scratch slab";
      return false;
    }
    for (uint32_t i : order) spill_lclp = std::max<uint32_t>(spill_lclp, uint32_t(ctx->pend_units[i].lc) + ctx->pend_units[i].lp);
    spill_lclp = std::min<uint32_t>(spill_lclp, 12);
    size_t free_b = 0, total_b = 0;
    size_t budget = kSpillSlabBytes;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, std::max(free_b / 2 + ctx->scratch.cap, spill_bytes_per_block(spill_lclp)));
    step = uint32_t(std::max<size_t>(1, std::min<size_t>(n, budget / spill_bytes_per_block(spill_lclp))));
    if (!dev_reserve(ctx, ctx->scratch, spill_bytes_per_block(spill_lclp) * step)) return false;
  }
  for (uint32_t i = 0; i < n; i += step) {
    const uint32_t m = std::min(step, n - i);
    while (ctx->ev_pool.size() < size_t(ctx->ev_used) * 2 + 2) {
      hipEvent_t e = nullptr;
      if (!hip_ok(ctx, hipEventCreate(&e), "hipEventCreate")) return false;
      ctx->ev_pool.push_back(e);
    }
    hipEvent_t e0 = ctx->ev_pool[size_t(ctx->ev_used) * 2], e1 = ctx->ev_pool[size_t(ctx->ev_used) * 2 + 1];
    if (!hip_ok(ctx, hipEventRecord(e0, stream), "hipEventRecord")) return false;
    // Time-sliced form for launches that would leave slots idle in their last round (kernels.h); the parked states need memory
    // (20-34 KB per unit): if that cannot be had the ordinary launch does the job.
    bool sliced = false;
    uint32_t cap = 0;
    if (is_fast) {
      const uint32_t resident = fast_resident_blocks(ctx->lds_pad);
      // (growable output is a feature of the time-sliced kernel: it is the one that can park a unit)
      const bool want_stream = ctx->stream_span != 0 && (ctx->stream_host != nullptr || ctx->stream_ptrs != nullptr) && m == ctx->pend_n && !resume;  // (the launch is the whole batch: the counters reach n)
      sliced = grow || want_stream || ctx->slice_mode > 0 || (ctx->slice_mode == 0 && m > resident && m % resident != 0);
      if (sliced) {
        uint64_t entries = m, longest = 0;
        const uint64_t least = std::max<uint32_t>(1u, ctx->slice_quantum / 4u * 3u);  // (a turn is 0.75 .. 1.5 quanta)
        for (uint32_t k = 0; k < m; k++) {
          const uint64_t cap_k = ctx->pend_units[order[i + k]].out_cap;
          entries += cap_k / least + 2;
          longest = std::max(longest, cap_k);
        }
        const size_t ctx_bytes = slice_ctx_bytes() * ctx->pend_n;  // (indexed by unit, not by launch position)
        size_t free_b = 0, total_b = 0;
        const std::string keep = ctx->err;
        // Not worth it / not to be had: units that all end within their first turn are never parked (the hardware's own block dispatch
        // does as well for them, without a parking lot of 20-34 KB per unit); a parking lot beyond a quarter of the free memory.
        if ((ctx->slice_mode == 0 && longest <= least && !grow && !want_stream) || entries > 0x7FFFFFF0ull ||
            (ctx_bytes > ctx->slice_ctx.cap && hipMemGetInfo(&free_b, &total_b) == hipSuccess && ctx_bytes > (free_b + ctx->slice_ctx.cap) / 4) ||
            !dev_reserve(ctx, ctx->slice_q, slice_queue_bytes(uint32_t(entries))) || !dev_reserve(ctx, ctx->slice_ctx, ctx_bytes)) {
          sliced = false;
          if (resume) {  // (the parked states cannot be reached without it)
            if (ctx->err.empty() || ctx->err == keep) ctx->err = "no memory for the time-sliced launch that resumes parked units";
            return false;
          }
          ctx->err = keep;
        }
        cap = uint32_t(entries);
        if (sliced && want_stream) {
          if (ensure_progress(ctx)) {
            memset(ctx->progress, 0, milzma_ctx::kMaxSpans * sizeof(uint32_t));
            ctx->stream_active = true;
          }
        }
        // A streamed launch whose input goes up in two parts waits for the ready word before it reads beyond the units' leads; a launch
        // that could not be made a streamed one would read what is not there yet (its results would be thrown away -- the caller falls
        // back to the classic rounds -- but it would run on garbage beside the upload): not launched at all.
        if (want_stream && ctx->stream_in_host && !ctx->stream_active) {
          ctx->err = "the streamed launch could not be set up";
          return false;
        }
      }
    }
    const hipError_t le = sliced
                              ? launch_fast_sliced(d_units, d_order + i, m, d_in, d_out, d_results, stream, ctx->lds_pad,
                                                   static_cast<uint32_t*>(ctx->flags.p) + (ctx->ev_used & 63u), ctx->slice_q.p, cap,
                                                   ctx->slice_quantum, ctx->slice_mode > 1, ctx->slice_ctx.p, grow, ctx->stream_span, ctx->stream_spans,
                                                   ctx->stream_active ? ctx->progress_dev : nullptr, ctx->stream_active ? ctx->stream_host : nullptr,
                                                   ctx->stream_active && ctx->stream_in_host ? ctx->progress_dev + milzma_ctx::kMaxSpans : nullptr,
                                                   ctx->stream_active ? ctx->stream_ptrs : nullptr,
                                                   cls == kFastSpill ? static_cast<const uint8_t*>(ctx->scratch.p) : nullptr, uint32_t(slab_bytes))
                          : is_fast
                              ? launch_fast(d_units, d_order + i, m, d_in, d_out, d_results, stream, ctx->lds_pad,
                                            static_cast<uint32_t*>(ctx->flags.p) + (ctx->ev_used & 63u),
                                            cls == kFastSpill ? static_cast<const uint8_t*>(ctx->scratch.p) : nullptr, uint32_t(slab_bytes))
                              : launch_generic(cls, d_units, d_order + i, m, d_in, d_out, d_results,
                                                        static_cast<uint16_t*>(ctx->scratch.p), spill_lclp, stream);
    if (!hip_ok(ctx, le, "kernel launch")) return false;
    if (!hip_ok(ctx, hipEventRecord(e1, stream), "hipEventRecord")) return false;
    ctx->ev_used++;  // (timed in collect_kernel_ms once the stream has drained: nothing here waits for the GPU)
  }
  return true;
}

// kernel time of the launches enqueued since ev_used was last reset; the stream must have been synchronised
bool collect_kernel_ms(milzma_ctx* ctx) {
  for (uint32_t k = 0; k < ctx->ev_used; k++) {
    float ms = 0.f;
    if (!hip_ok(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[size_t(k) * 2], ctx->ev_pool[size_t(k) * 2 + 1]), "hipEventElapsedTime"))
      return false;
    ctx->last_ms += ms;
    ctx->last_launches++;
  }
  ctx->ev_used = 0;
  return true;
}

}  // namespace

// Enqueue: descriptor upload, one launch per class, result download into a page-locked buffer -- all on `stream`,
// nothing waits for the GPU.
static int milzma_decode_units_async_impl(milzma_ctx* ctx, const milzma_unit* units, uint32_t n, const void* d_in,
                                          void* d_out, void* hip_stream, uint32_t flags = 0, const milzma_result* prev = nullptr) {
  if (!ctx) return MILZMA_INFRA_ERROR;
  const bool resume = (flags & MILZMA_DECODE_RESUME) != 0;
  const bool grow = resume || (flags & MILZMA_DECODE_GROW) != 0;
  if (resume && (!ctx->parked_valid || ctx->parked_n != n || !prev)) {
    ctx->err = "MILZMA_DECODE_RESUME: this context holds no parked units of a batch of that size";
    return MILZMA_INFRA_ERROR;
  }
  if (!resume) {  // (whatever was parked here is given up: the parking lot and the slab serve this batch now)
    ctx->parked_valid = false;
    ctx->slab_live = false;
  }
  if (ctx->pending) {
    ctx->err = "a batch is already in flight on this context: call milzma_decode_units_wait first";
    return MILZMA_INFRA_ERROR;
  }
  if (n && !units) {
    ctx->err = "null units";
    return MILZMA_INFRA_ERROR;
  }
  ctx->last_ms = 0.f;
  ctx->last_launches = 0;
  ctx->ev_used = 0;
  ctx->pend_n = n;
  ctx->pend_flags = (grow ? MILZMA_DECODE_GROW : 0u) | (resume ? MILZMA_DECODE_RESUME : 0u);
  ctx->stream_active = false;
  ctx->promoted.clear();
  ctx->pending = true;
  if (n == 0) return MILZMA_OK;
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  // Copies and kernels may already be queued when a later step fails: they still reference pend_units, the pinned
  // result buffer and the device buffers, so the stream is drained before the batch is declared gone.
  const auto fail = [&]() {
    const std::string why = ctx->err;
    // (a streamed launch already enqueued may have persistent waves spinning on the input-ready word, which only the second upload would
    //  set -- and this call is not going to get there: released first, or the drain below never returns.  What those waves decode is
    //  thrown away with the call.)
    if (ctx->stream_in_host && ctx->progress) __atomic_store_n(&ctx->progress[milzma_ctx::kMaxSpans], 1u, __ATOMIC_RELEASE);
    (void)hipStreamSynchronize(stream);
    ctx->err = why;
    ctx->ev_used = 0;
    ctx->pending = false;
    return MILZMA_INFRA_ERROR;
  };
  if (!hip_ok(ctx, hipSetDevice(ctx->device), "hipSetDevice")) {
    ctx->pending = false;
    return MILZMA_INFRA_ERROR;
  }
  ctx->pend_stream = stream;
  ctx->pend_in = static_cast<const uint8_t*>(d_in);
  ctx->pend_out = static_cast<uint8_t*>(d_out);
  // LzmaParams::read_header raises a dictionary below 4 KiB to 4 KiB (lzma.rs:118-120); a RAW unit built by hand gets
  // the same floor (the kernels divide by dict_size).
  ctx->pend_units.assign(units, units + n);
  for (milzma_unit& u : ctx->pend_units)
    if (u.kind == MILZMA_KIND_RAW_LZMA && u.dict_size < 0x1000u) u.dict_size = 0x1000u;
  units = ctx->pend_units.data();

  // Partition by launch class; inside a class longest input first, so that the hardware's
  // in-order block dispatch behaves like longest-processing-time-first scheduling.
  std::vector<uint32_t> order[kNumLitClasses];
  if (resume) {  // only what the previous call parked, each unit in the class the CONTEXT knows it was parked in
    for (uint32_t i = 0; i < n; i++)
      if (prev[i].status == MILZMA_ST_OUT_FULL && prev[i].err_a == MILZMA_PARKED) {
        const milzma_ctx::ParkRec* rec = i < ctx->park_rec.size() ? &ctx->park_rec[i] : nullptr;
        const char* why = !rec || !rec->parked                                                        ? "was not parked by the previous call"
                          : units[i].in_off != rec->in_off || units[i].in_len != rec->in_len || units[i].kind != rec->kind ? "names another input than the one it was parked with"
                          : units[i].out_cap < rec->out_len                                           ? "has a slice smaller than the output it has produced"
                                                                                                      : nullptr;
        if (why) {  // (nothing is launched: a descriptor that does not fit the parked state would write outside its slice)
          ctx->err = "MILZMA_DECODE_RESUME: unit " + std::to_string(i) + " " + why;
          ctx->pending = false;
          return MILZMA_INFRA_ERROR;
        }
        order[rec->spill ? kFastSpill : kFast].push_back(i);
      }
  } else {
    for (uint32_t i = 0; i < n; i++) order[classify(ctx, units[i])].push_back(i);
  }
  std::vector<uint32_t> flat;
  flat.reserve(n);
  uint32_t base[kNumLitClasses];
  for (int c = 0; c < kNumLitClasses; c++) {
    std::stable_sort(order[c].begin(), order[c].end(),
                     [&](uint32_t a, uint32_t b) { return units[a].in_len > units[b].in_len; });
    // MILZMA_ORDER (tuning): "stride" deals the sorted units out so that any 16 consecutive blocks (about a CU's worth)
    // hold the whole range of sizes instead of 16 neighbours of the sorted list; "shuffle": a fixed pseudo-random order
    if (ctx->order_mode && order[c].size() > 32) {
      std::vector<uint32_t>& o = order[c];
      const size_t m = o.size();
      std::vector<uint32_t> p(m);
      if (ctx->order_mode == 1) {
        const size_t g = (m + 15) / 16;
        size_t k = 0;
        for (size_t r = 0; r < g; r++)
          for (size_t j = r; j < m; j += g) p[k++] = o[j];
        // (p lists, for every residue r, the units r, r + g, r + 2g ...: 16 units spread over the whole sorted list)
      } else {
        p = o;
        uint64_t x = 0x9E3779B97F4A7C15ull;
        for (size_t i = m - 1; i > 0; i--) {
          x ^= x << 13;
          x ^= x >> 7;
          x ^= x << 17;
          std::swap(p[i], p[size_t(x % (i + 1))]);
        }
      }
      o.swap(p);
    }
    base[c] = uint32_t(flat.size());
    flat.insert(flat.end(), order[c].begin(), order[c].end());
  }

  if (!dev_reserve(ctx, ctx->units, size_t(n) * sizeof(milzma_unit)) ||
      !dev_reserve(ctx, ctx->order, size_t(n) * 2 * sizeof(uint32_t)) ||
      !dev_reserve(ctx, ctx->results, size_t(n) * sizeof(milzma_result)) || !dev_reserve(ctx, ctx->flags, 64 * sizeof(uint32_t)) ||
      !pin_reserve(ctx, ctx->pin_results, size_t(n) * (sizeof(milzma_result) + sizeof(uint32_t))))
    return fail();
  // (the order array is staged in page-locked memory behind the results so that its upload is asynchronous too)
  uint32_t* h_order = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(ctx->pin_results.p) + size_t(n) * sizeof(milzma_result));
  for (size_t k = 0; k < flat.size(); k++) h_order[k] = flat[k] | (resume ? 0x80000000u : 0u);  // (bit 31: resume from the parked state)
  if (!hip_ok(ctx, hipMemcpyAsync(ctx->units.p, units, size_t(n) * sizeof(milzma_unit), hipMemcpyHostToDevice, stream),
              "H2D units") ||
      (!flat.empty() &&
       !hip_ok(ctx, hipMemcpyAsync(ctx->order.p, h_order, flat.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream), "H2D order")))
    return fail();
  if (resume) {  // the units that are not resumed keep the results they have
    memcpy(ctx->pin_results.p, prev, size_t(n) * sizeof(milzma_result));
    if (!hip_ok(ctx, hipMemcpyAsync(ctx->results.p, ctx->pin_results.p, size_t(n) * sizeof(milzma_result), hipMemcpyHostToDevice, stream),
                "H2D results"))
      return fail();
  }

  for (int c = 0; c < kNumLitClasses; c++)
    if (!launch_class(ctx, LitClass(c), order[c], base[c], ctx->pend_in, ctx->pend_out, stream, grow, resume)) return fail();

  // (The results are fetched by the wait half, after the kernels: a copy queued behind a running kernel parks a DMA queue on
  //  that kernel's completion, and an unrelated upload of another context that lands on the same queue then waits for the whole
  //  kernel -- measured: 200 ms per grouped call, profiles/r03_batch_api.txt.)
  return MILZMA_OK;
}

// Wait: drain the stream, time the launches, rerun promoted LZMA2 units, hand the results over.
static int milzma_decode_units_wait_impl(milzma_ctx* ctx, milzma_result* results) {
  if (!ctx) return MILZMA_INFRA_ERROR;
  if (!ctx->pending) {
    ctx->err = "no batch in flight on this context";
    return MILZMA_INFRA_ERROR;
  }
  const uint32_t n = ctx->pend_n;
  if (n && !results) {  // (the batch stays in flight: the caller can still wait for it with a real buffer)
    ctx->err = "null results";
    return MILZMA_INFRA_ERROR;
  }
  ctx->pending = false;
  if (n == 0) return MILZMA_OK;
  hipStream_t stream = ctx->pend_stream;
  // Whatever fails from here on: nothing of this batch may still be running when the caller is told (a kernel of a promotion round, a
  // copy into the caller's results) -- it would write memory the caller is free to release (fault injection found a launch outliving
  // its failed call).  The device is drained first, the error text kept.
  const auto bail = [&]() {
    const std::string why = ctx->err;
    (void)hipStreamSynchronize(stream);
    (void)hipGetLastError();
    ctx->err = why;
    ctx->ev_used = 0;
    return MILZMA_INFRA_ERROR;
  };
  if (!hip_ok(ctx, hipSetDevice(ctx->device), "hipSetDevice") || !hip_ok(ctx, hipStreamSynchronize(stream), "hipStreamSynchronize") ||
      !collect_kernel_ms(ctx) ||
      !hip_ok(ctx, hipMemcpyAsync(ctx->pin_results.p, ctx->results.p, size_t(n) * sizeof(milzma_result), hipMemcpyDeviceToHost, stream),
              "D2H results") ||
      !hip_ok(ctx, hipStreamSynchronize(stream), "hipStreamSynchronize"))
    return bail();
  memcpy(results, ctx->pin_results.p, size_t(n) * sizeof(milzma_result));

  // Promotions: LZMA2 units whose chunks switched to properties outside their class's reach run
  // again, from the start, in the next class up (fast -> fast with a literal-row slab, which covers every LZMA2-legal property set;
  // with MILZMA_KERNEL=generic: generic/LDS3 -> generic/LDS4).
  for (int round = 0; round < 2; round++) {
    std::vector<uint32_t> again;
    LitClass next = ctx->use_fast && ctx->fast_spill ? kFastSpill : kLitLds3;
    for (uint32_t i = 0; i < n; i++) {
      if (round == 0 && results[i].status == MILZMA_ST_NEED_GENERIC) again.push_back(i);
      if (round == 1 && results[i].status == MILZMA_ST_NEED_LCLP && results[i].err_a <= 4) again.push_back(i);
    }
    if (round == 1) next = kLitLds4;
    if (again.empty()) continue;
    ctx->promoted.insert(ctx->promoted.end(), again.begin(), again.end());
    if (!hip_ok(ctx,
                hipMemcpyAsync(static_cast<uint32_t*>(ctx->order.p) + n, again.data(), again.size() * sizeof(uint32_t),
                               hipMemcpyHostToDevice, stream),
                "H2D order") ||
        !hip_ok(ctx, hipStreamSynchronize(stream), "hipStreamSynchronize"))  // (`again` is pageable and about to go away)
      return bail();
    if (!launch_class(ctx, next, again, n, ctx->pend_in, ctx->pend_out, stream, (ctx->pend_flags & MILZMA_DECODE_GROW) != 0)) return bail();
    if (!hip_ok(ctx,
                hipMemcpyAsync(results, ctx->results.p, size_t(n) * sizeof(milzma_result), hipMemcpyDeviceToHost, stream),
                "D2H results") ||
        !hip_ok(ctx, hipStreamSynchronize(stream), "hipStreamSynchronize") || !collect_kernel_ms(ctx))
      return bail();
  }
  if (ctx->pend_flags & MILZMA_DECODE_GROW) {
    bool any = false;
    for (uint32_t i = 0; i < n && !any; i++) any = results[i].status == MILZMA_ST_OUT_FULL && results[i].err_a == MILZMA_PARKED;
    ctx->parked_valid = any;
    ctx->parked_n = n;
    if (!any) ctx->slab_live = false;
    ctx->park_rec.assign(any ? n : 0, milzma_ctx::ParkRec());
    for (uint32_t i = 0; any && i < n; i++)
      if (results[i].status == MILZMA_ST_OUT_FULL && results[i].err_a == MILZMA_PARKED) {
        milzma_ctx::ParkRec& r = ctx->park_rec[i];
        r.parked = 1;
        r.spill = (results[i].err_b & 0x100) ? 1 : 0;
        r.kind = uint8_t(ctx->pend_units[i].kind);
        r.in_off = ctx->pend_units[i].in_off;
        r.in_len = ctx->pend_units[i].in_len;
        r.out_len = results[i].out_len;
      }
  }
  return MILZMA_OK;
}

static int milzma_decode_units_impl(milzma_ctx* ctx, const milzma_unit* units, uint32_t n, const void* d_in,
                                   void* d_out, milzma_result* results, void* hip_stream, uint32_t flags = 0) {
  if (!ctx) return MILZMA_INFRA_ERROR;
  if (n && !results) {
    ctx->err = "null units/results";
    return MILZMA_INFRA_ERROR;
  }
  std::vector<milzma_result> prev;  // (a RESUME reads the previous results and then writes the same array)
  if ((flags & MILZMA_DECODE_RESUME) && n) prev.assign(results, results + n);
  const int r = milzma_decode_units_async_impl(ctx, units, n, d_in, d_out, hip_stream, flags, prev.empty() ? nullptr : prev.data());
  return r != MILZMA_OK ? r : milzma_decode_units_wait_impl(ctx, results);
}

static int milzma_decode_units_host_impl(milzma_ctx* ctx, const milzma_unit* units, uint32_t n, const void* h_in,
                                        size_t in_bytes, void* h_out, size_t out_bytes, milzma_result* results) {
  if (!ctx) return MILZMA_INFRA_ERROR;
  // the kernels address input and output through the descriptors alone: a slice outside the buffers the caller
  // described, or two output slices that overlap, would be out-of-bounds device accesses
  {
    std::vector<std::pair<uint64_t, uint64_t>> spans;
    spans.reserve(n);
    for (uint32_t i = 0; i < n; i++) {
      const milzma_unit& u = units[i];
      if (u.in_off > in_bytes || u.in_len > in_bytes - u.in_off || u.out_off > out_bytes || u.out_cap > out_bytes - u.out_off) {
        ctx->err = "unit " + std::to_string(i) + ": input or output slice outside the buffers";
        return MILZMA_INFRA_ERROR;
      }
      if (u.out_cap) spans.emplace_back(u.out_off, u.out_off + u.out_cap);
    }
    std::sort(spans.begin(), spans.end());
    for (size_t k = 1; k < spans.size(); k++)
      if (spans[k].first < spans[k - 1].second) {
        ctx->err = "overlapping output slices";
        return MILZMA_INFRA_ERROR;
      }
  }
  if (!hip_ok(ctx, hipSetDevice(ctx->device), "hipSetDevice")) return MILZMA_INFRA_ERROR;
  if (!dev_reserve(ctx, ctx->in, in_bytes + 512) || !dev_reserve(ctx, ctx->out, out_bytes + 512)) return MILZMA_INFRA_ERROR;
  if (in_bytes && !hip_ok(ctx, hipMemcpy(ctx->in.p, h_in, in_bytes, hipMemcpyHostToDevice), "H2D input"))
    return MILZMA_INFRA_ERROR;
  const int r = milzma_decode_units(ctx, units, n, ctx->in.p, ctx->out.p, results, work_stream(ctx));
  if (r != MILZMA_OK) return r;
  if (out_bytes && !hip_ok(ctx, hipMemcpy(h_out, ctx->out.p, out_bytes, hipMemcpyDeviceToHost), "D2H output"))
    return MILZMA_INFRA_ERROR;
  return MILZMA_OK;
}

// ------------------------------------------------------------------------------------------
// error rendering: src/error.rs:28-36 prefixes + the message of each hot-path error site
// ------------------------------------------------------------------------------------------

namespace {

const char* const kEofMsg = "failed to fill whole buffer";  // io::ErrorKind::UnexpectedEof
const char* const kPrefix[] = {"", "io error: ", "header too short: ", "lzma error: ", "xz error: ", "milzma: "};

int render(char* msg, size_t cap, int kind, const char* fmt, ...) {
  if (msg && cap) {
    const int n = snprintf(msg, cap, "%s", kPrefix[kind]);
    va_list ap;
    va_start(ap, fmt);
    if (n >= 0 && size_t(n) < cap) vsnprintf(msg + n, cap - size_t(n), fmt, ap);
    va_end(ap);
  }
  return kind;
}

}  // namespace

extern "C" int milzma_result_message(const milzma_result* r, uint32_t unit_kind, char* msg, size_t cap) {
  const unsigned long long a = r->err_a, b = r->err_b;
  switch (r->status) {
    case MILZMA_ST_OK:
      if (msg && cap) msg[0] = 0;
      return MILZMA_OK;
    case MILZMA_ST_RC_INIT:
      return render(msg, cap, MILZMA_LZMA_ERROR,
                    unit_kind == MILZMA_KIND_LZMA2 ? "LZMA input too short: %s" : "LZMA stream too short: %s", kEofMsg);
    case MILZMA_ST_INPUT_EOF: return render(msg, cap, MILZMA_IO_ERROR, "%s", kEofMsg);
    case MILZMA_ST_MATCH_DIST_DICT:
      return render(msg, cap, MILZMA_LZMA_ERROR, "Match distance %llu is beyond dictionary size %llu", a, b);
    case MILZMA_ST_MATCH_DIST_OUT:
      return render(msg, cap, MILZMA_LZMA_ERROR, "Match distance %llu is beyond output size %llu", a, b);
    case MILZMA_ST_LZ_DIST_DICT:
      return render(msg, cap, MILZMA_LZMA_ERROR, "LZ distance %llu is beyond dictionary size %llu", a, b);
    case MILZMA_ST_LZ_DIST_OUT:
      return render(msg, cap, MILZMA_LZMA_ERROR, "LZ distance %llu is beyond output size %llu", a, b);
    case MILZMA_ST_MEMLIMIT: return render(msg, cap, MILZMA_LZMA_ERROR, "exceeded memory limit of %llu", a);
    case MILZMA_ST_MARKER_TRAILING:
      return render(msg, cap, MILZMA_LZMA_ERROR, "Found end-of-stream marker but more bytes are available");
    case MILZMA_ST_SIZE_MISMATCH:
      return render(msg, cap, MILZMA_LZMA_ERROR, "Expected unpacked size of %llu but decompressed to %llu", a, b);
    case MILZMA_ST_L2_STATUS_EOF: return render(msg, cap, MILZMA_LZMA_ERROR, "LZMA2 expected new status: %s", kEofMsg);
    case MILZMA_ST_L2_INVALID_STATUS:
      return render(msg, cap, MILZMA_LZMA_ERROR, "LZMA2 invalid status %llu, must be 0, 1, 2 or >= 128", a);
    case MILZMA_ST_L2_UNPACKED_EOF:
      return render(msg, cap, MILZMA_LZMA_ERROR, "LZMA2 expected unpacked size: %s", kEofMsg);
    case MILZMA_ST_L2_PACKED_EOF: return render(msg, cap, MILZMA_LZMA_ERROR, "LZMA2 expected packed size: %s", kEofMsg);
    case MILZMA_ST_L2_PROPS_EOF:
      return render(msg, cap, MILZMA_LZMA_ERROR, "LZMA2 expected new properties: %s", kEofMsg);
    case MILZMA_ST_L2_PROPS_INVALID:
      return render(msg, cap, MILZMA_LZMA_ERROR, "LZMA2 invalid properties: %llu must be < 225", a);
    case MILZMA_ST_L2_LCLP:
      return render(msg, cap, MILZMA_LZMA_ERROR, "LZMA2 invalid properties: lc + lp (%llu + %llu) must be <= 4", a, b);
    case MILZMA_ST_L2_STORED_EOF:
      return render(msg, cap, MILZMA_LZMA_ERROR, "LZMA2 expected %llu uncompressed bytes: %s", a, kEofMsg);
    case MILZMA_ST_OUT_FULL: return render(msg, cap, MILZMA_INFRA_ERROR, "output slice too small");
    case MILZMA_ST_NEED_LCLP: return render(msg, cap, MILZMA_INFRA_ERROR, "literal table class too small for lc+lp=%llu", a);
    case MILZMA_ST_BAD_UNIT: return render(msg, cap, MILZMA_INFRA_ERROR, "bad unit descriptor");
    case MILZMA_ST_NEED_GENERIC: return render(msg, cap, MILZMA_INFRA_ERROR, "properties outside the fast kernel's class");
    case MILZMA_ST_NEED_RERUN: return render(msg, cap, MILZMA_INFRA_ERROR, "unit outran its input upload");
    default: return render(msg, cap, MILZMA_INFRA_ERROR, "unknown status %u", r->status);
  }
}

// ------------------------------------------------------------------------------------------
// CRC-32 (ISO-HDLC) / CRC-64 (XZ), slicing-by-8 (src/xz/crc.rs:1-4 names the polynomials)
// ------------------------------------------------------------------------------------------

namespace {

struct CrcTables {
  uint32_t t32[8][256];
  uint64_t t64[8][256];
  CrcTables() {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      uint64_t d = i;
      for (int j = 0; j < 8; j++) {
        c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        d = (d & 1) ? (d >> 1) ^ 0xC96C5795D7870F42ull : d >> 1;
      }
      t32[0][i] = c;
      t64[0][i] = d;
    }
    for (uint32_t i = 0; i < 256; i++)
      for (int k = 1; k < 8; k++) {
        t32[k][i] = (t32[k - 1][i] >> 8) ^ t32[0][t32[k - 1][i] & 0xFF];
        t64[k][i] = (t64[k - 1][i] >> 8) ^ t64[0][t64[k - 1][i] & 0xFF];
      }
  }
};
const CrcTables& crc_tables() {
  static const CrcTables t;
  return t;
}

uint32_t crc32_update(uint32_t c, const uint8_t* p, size_t n) {
  const auto& T = crc_tables().t32;
  while (n >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = T[7][lo & 0xFF] ^ T[6][(lo >> 8) & 0xFF] ^ T[5][(lo >> 16) & 0xFF] ^ T[4][lo >> 24] ^ T[3][hi & 0xFF] ^
        T[2][(hi >> 8) & 0xFF] ^ T[1][(hi >> 16) & 0xFF] ^ T[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = T[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return c;
}

}  // namespace

extern "C" uint32_t milzma_crc32(const uint8_t* p, size_t n) { return ~crc32_update(0xFFFFFFFFu, p, n); }

extern "C" uint64_t milzma_crc64(const uint8_t* p, size_t n) {
  const auto& T = crc_tables().t64;
  uint64_t c = ~uint64_t(0);
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    w ^= c;
    c = T[7][w & 0xFF] ^ T[6][(w >> 8) & 0xFF] ^ T[5][(w >> 16) & 0xFF] ^ T[4][(w >> 24) & 0xFF] ^
        T[3][(w >> 32) & 0xFF] ^ T[2][(w >> 40) & 0xFF] ^ T[1][(w >> 48) & 0xFF] ^ T[0][w >> 56];
    p += 8;
    n -= 8;
  }
  while (n--) c = T[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return ~c;
}

// ---- folding the per-chunk CRCs the GPU computes (crc_units.hip.h) -------------------------------
// crc(A || B) = crc(A) * x^(8 |B|) mod P  xor  crc(B) for CRCs whose init and xorout are both all ones
// (true for CRC-32/ISO-HDLC and CRC-64/XZ); products are taken in the reflected representation, where
// the top bit stands for x^0.
namespace {

template <class T>
struct Gf2 {
  T poly, top;
  T x2n[64];  // x^(2^k) mod P
  Gf2(T poly_, T top_) : poly(poly_), top(top_) {
    x2n[0] = top >> 1;  // x^1
    for (int k = 1; k < 64; k++) x2n[k] = mul(x2n[k - 1], x2n[k - 1]);
  }
  T mul(T a, T b) const {
    T m = top, p = 0;
    for (;;) {
      if (a & m) {
        p ^= b;
        if ((a & (m - 1)) == 0) break;
      }
      m >>= 1;
      b = (b & 1) ? (b >> 1) ^ poly : b >> 1;
    }
    return p;
  }
  T xpow8(uint64_t nbytes) const {  // x^(8 * nbytes) mod P
    T p = top;
    for (int k = 3; nbytes; nbytes >>= 1, k++)
      if (nbytes & 1) p = mul(x2n[k & 63], p);
    return p;
  }
};
const Gf2<uint32_t>& gf32() {
  static const Gf2<uint32_t> g(0xEDB88320u, 0x80000000u);
  return g;
}
const Gf2<uint64_t>& gf64() {
  static const Gf2<uint64_t> g(0xC96C5795D7870F42ull, uint64_t(1) << 63);
  return g;
}

// parts: the kCrcPartsBytes record of one unit; len = the unit's out_len
void crc_fold(const uint8_t* parts, uint64_t len, uint32_t* crc32, uint64_t* crc64) {
  const uint32_t* c32 = reinterpret_cast<const uint32_t*>(parts);
  const uint64_t* c64 = reinterpret_cast<const uint64_t*>(parts + 64 * 4);
  uint32_t chunk;
  memcpy(&chunk, parts + 64 * 4 + 64 * 8, 4);
  const uint32_t s32 = gf32().xpow8(chunk);
  const uint64_t s64 = gf64().xpow8(chunk);
  uint32_t a32 = 0;
  uint64_t a64 = 0;
  uint64_t done = 0;
  for (int l = 0; l < 64 && done < len; l++) {
    const uint64_t n = std::min<uint64_t>(chunk, len - done);
    if (n == chunk) {
      a32 = gf32().mul(s32, a32) ^ c32[l];
      a64 = gf64().mul(s64, a64) ^ c64[l];
    } else {
      a32 = gf32().mul(gf32().xpow8(n), a32) ^ c32[l];
      a64 = gf64().mul(gf64().xpow8(n), a64) ^ c64[l];
    }
    done += n;
  }
  *crc32 = a32;
  *crc64 = a64;
}

}  // namespace

static int crc_units_impl(milzma_ctx* ctx, const milzma_unit* units, uint32_t n, const void* d_out, const milzma_result* results,
                          uint32_t* crc32, uint64_t* crc64, void* hip_stream);

extern "C" int milzma_crc_units(milzma_ctx* ctx, const milzma_unit* units, uint32_t n, const void* d_out,
                                const milzma_result* results, uint32_t* crc32, uint64_t* crc64, void* hip_stream) {
  if (ctx) ctx->err.clear();
  try {
    return crc_units_impl(ctx, units, n, d_out, results, crc32, crc64, hip_stream);
  } catch (const std::exception& e) {   // (std::bad_alloc from the staging vector: never across the C ABI)
    if (ctx) ctx->err = std::string("host exception: ") + e.what();
    return MILZMA_INFRA_ERROR;
  }
}

static int crc_units_impl(milzma_ctx* ctx, const milzma_unit* units, uint32_t n, const void* d_out, const milzma_result* results,
                          uint32_t* crc32, uint64_t* crc64, void* hip_stream) {
  if (!ctx) return MILZMA_INFRA_ERROR;
  if (n == 0) return MILZMA_OK;
  if (!units || !results) {
    ctx->err = "null units/results";
    return MILZMA_INFRA_ERROR;
  }
  if (!hip_ok(ctx, hipSetDevice(ctx->device), "hipSetDevice")) return MILZMA_INFRA_ERROR;
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  std::vector<uint8_t> parts(size_t(n) * kCrcPartsBytes);
  if (!dev_reserve(ctx, ctx->units, size_t(n) * sizeof(milzma_unit)) ||
      !dev_reserve(ctx, ctx->results, size_t(n) * sizeof(milzma_result)) ||
      !dev_reserve(ctx, ctx->crc, parts.size()))
    return MILZMA_INFRA_ERROR;
  if (!hip_ok(ctx, hipMemcpyAsync(ctx->units.p, units, size_t(n) * sizeof(milzma_unit), hipMemcpyHostToDevice, stream),
              "H2D units") ||
      !hip_ok(ctx, hipMemcpyAsync(ctx->results.p, results, size_t(n) * sizeof(milzma_result), hipMemcpyHostToDevice, stream),
              "H2D results") ||
      !hip_ok(ctx,
              launch_crc_units(static_cast<const milzma_unit*>(ctx->units.p), n, static_cast<const uint8_t*>(d_out),
                               static_cast<const milzma_result*>(ctx->results.p), ctx->crc.p, stream),
              "crc kernel launch") ||
      !hip_ok(ctx, hipMemcpyAsync(parts.data(), ctx->crc.p, parts.size(), hipMemcpyDeviceToHost, stream), "D2H crc parts") ||
      !hip_ok(ctx, hipStreamSynchronize(stream), "hipStreamSynchronize")) {
    const std::string why = ctx->err;   // (what was queued still reads the caller's arrays and writes `parts`: drained before either goes)
    (void)hipStreamSynchronize(stream);
    ctx->err = why;
    return MILZMA_INFRA_ERROR;
  }
  for (uint32_t i = 0; i < n; i++) {
    uint32_t a = 0;
    uint64_t b = 0;
    if (results[i].status == MILZMA_ST_OK) crc_fold(parts.data() + size_t(i) * kCrcPartsBytes, results[i].out_len, &a, &b);
    if (crc32) crc32[i] = a;
    if (crc64) crc64[i] = b;
  }
  return MILZMA_OK;
}

// ------------------------------------------------------------------------------------------
// helpers shared by the whole-file entry points
// ------------------------------------------------------------------------------------------

// d_dst[dst_off[i], +len[i]) = d_src[src_off[i], +len[i]) on the device, one launch (milzma_move_units)
static int move_units_impl(milzma_ctx* ctx, uint32_t n, const void* d_src, const uint64_t* src_off, void* d_dst, const uint64_t* dst_off,
                           const uint64_t* len, hipStream_t stream) {
  if (!ctx) return MILZMA_INFRA_ERROR;
  if (ctx->pending) {  // (the move list shares a device buffer with the batch's order array)
    ctx->err = "a batch is in flight on this context: call milzma_decode_units_wait first";
    return MILZMA_INFRA_ERROR;
  }
  if (n == 0) return MILZMA_OK;
  if (!d_src || !d_dst || !src_off || !dst_off || !len) {
    ctx->err = "null argument";
    return MILZMA_INFRA_ERROR;
  }
  const size_t bytes = size_t(n) * 3 * sizeof(uint64_t);
  if (!hip_ok(ctx, hipSetDevice(ctx->device), "hipSetDevice") || !dev_reserve(ctx, ctx->order, std::max(bytes, ctx->order.cap)) ||
      !pin_reserve(ctx, ctx->pin_moves, bytes))
    return MILZMA_INFRA_ERROR;
  uint64_t* h = static_cast<uint64_t*>(ctx->pin_moves.p);
  memcpy(h, src_off, size_t(n) * 8);
  memcpy(h + n, dst_off, size_t(n) * 8);
  memcpy(h + 2 * size_t(n), len, size_t(n) * 8);
  if (!hip_ok(ctx, hipMemcpyAsync(ctx->order.p, h, bytes, hipMemcpyHostToDevice, stream), "H2D move list") ||
      !hip_ok(ctx, launch_move_units(static_cast<const uint8_t*>(d_src), static_cast<uint8_t*>(d_dst), static_cast<const uint64_t*>(ctx->order.p), n, stream),
              "move kernel launch") ||
      !hip_ok(ctx, hipStreamSynchronize(stream), "hipStreamSynchronize")) {
    const std::string why = ctx->err;   // (a queued move reads and writes the caller's buffers: drained before the caller is told)
    (void)hipStreamSynchronize(stream);
    ctx->err = why;
    return MILZMA_INFRA_ERROR;
  }
  return MILZMA_OK;
}

namespace {

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

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
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

// streamed launches are for batches it pays for: at least this many units and output bytes, of about one size (one pitch for all
// slices: a ragged batch would reserve the largest unit's room for every unit).  MILZMA_STREAM_MIN="units,bytes[,1]": tests send small
// batches down the path; the third field lifts the one-size condition too (fuzzers: batches of anything).
void stream_minimum(size_t* units, size_t* bytes, bool* ragged_ok = nullptr) {
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
  parallel_for(nu, [&](size_t k) { memcpy(h + so[k], src(k), size_t(ln[k])); });
  return hip_ok(ctx, hipMemcpyAsync(ctx->pack.p, h, total, hipMemcpyHostToDevice, ws), "H2D leads") &&
         move_units_impl(ctx, nu, ctx->pack.p, so.data(), ctx->in.p, dof.data(), ln.data(), ws) == MILZMA_OK;
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

}  // namespace

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

namespace {

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

}  // namespace

static int milzma_lzma_decompress_impl(milzma_ctx* ctx, const uint8_t* in, size_t in_len, const milzma_options* opt,
                                      milzma_output* out) {
  milzma_unit u;
  size_t hl = 0;
  const int hr = milzma_lzma_read_header(in, in_len, opt, &u, &hl, out);
  if (hr != MILZMA_OK) return hr;
  SingleDecode sd;
  if (!decode_single(ctx, u, in + hl, in_len - hl, lzma_cap_hint(u, in_len - hl), &sd)) return infra(ctx, out);
  return finish_stream(sd.res, MILZMA_KIND_RAW_LZMA, sd.out.data(), sd.out.size(), hl, out);
}

static int milzma_lzma2_decompress_impl(milzma_ctx* ctx, const uint8_t* in, size_t in_len, milzma_output* out) {
  out_reset(out);
  milzma_unit u;
  memset(&u, 0, sizeof u);
  u.kind = MILZMA_KIND_LZMA2;
  SingleDecode sd;
  if (!decode_single(ctx, u, in, in_len, std::max<size_t>(1 << 16, in_len * 6), &sd)) return infra(ctx, out);
  return finish_stream(sd.res, MILZMA_KIND_LZMA2, sd.out.data(), sd.out.size(), 0, out);
}

// Batch driver for RAW / LZMA2 streams: one launch for all, stragglers (OUT_FULL) one by one.
namespace {

int stream_batch(milzma_ctx* ctx, uint32_t n, const uint8_t* const* ins, const size_t* in_lens, const milzma_options* opt,
                 bool lzma2, milzma_output* outs) {
  ctx->last_paths = 0;
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
    if (direct) {
      parallel_for(nu, [&](size_t k) {
        bufs[k] = out_alloc(size_t(units[k].out_cap), true);
        if (!bufs[k]) alloc_failed = 1;
      });
      if (alloc_failed) {
        held.drop();
        alloc_failed = 0;
        direct = false;
      }
    }
    if (!direct)
      parallel_for(nu, [&](size_t k) {
        bufs[k] = out_alloc(size_t(units[k].out_cap));
        if (!bufs[k]) alloc_failed = 1;
      });
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
      trace_mark(ctx, "streamed: leads");
      ok = upload_leads(ctx, units, [&](size_t k) { return ins[owner[k]] + hdr[owner[k]]; }, work_stream(ctx));
    }
    if (ok) {
      __atomic_store_n(&ctx->progress[milzma_ctx::kMaxSpans], 0u, __ATOMIC_RELEASE);
      ctx->stream_span = uint32_t(geo.span);
      ctx->stream_spans = geo.spans;
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

}  // namespace

static int milzma_lzma_decompress_batch_impl(milzma_ctx* ctx, uint32_t n, const uint8_t* const* ins, const size_t* in_lens,
                                            const milzma_options* opt, milzma_output* outs) {
  return stream_batch(ctx, n, ins, in_lens, opt, false, outs);
}

static int milzma_lzma2_decompress_batch_impl(milzma_ctx* ctx, uint32_t n, const uint8_t* const* ins, const size_t* in_lens,
                                             milzma_output* outs) {
  return stream_batch(ctx, n, ins, in_lens, nullptr, true, outs);
}

// ------------------------------------------------------------------------------------------
// .xz: xz::decode_stream (src/decode/xz.rs:18-94) with the LZMA2 payload of each block decoded
// on the device.  The walk below is the reference's, statement for statement; what is new is
// that block payloads can be decoded ahead of the walk, all at once (see xz_batch).
// ------------------------------------------------------------------------------------------

namespace {

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

// ---- batching: find the blocks of well-formed files up front through the Index ------------
struct PlannedBlock {
  size_t data_off;    // first byte of the block's LZMA2 payload within the file
  size_t data_len;    // payload bytes according to the Index (unpadded - header - check)
  uint64_t unpacked;  // uncompressed size according to the Index
};

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

}  // namespace

static int milzma_xz_decompress_batch_impl(milzma_ctx* ctx, uint32_t n, const uint8_t* const* ins, const size_t* in_lens,
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
  if (nu) {
    // Decoding ahead is an optimisation: if its memory cannot be had (or anything else goes wrong here) the walk below
    // decodes every block on demand and each file still gets the reference's verdict.
    const auto ahead = [&]() -> bool {
      if (!hip_ok(ctx, hipSetDevice(ctx->device), "hipSetDevice")) return false;
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
          ctx->stream_spans = geo.spans;
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

// (every public call starts with an empty error text: what milzma_last_error returns afterwards belongs to THIS call)
static inline void begin_call(milzma_ctx* ctx) {
  if (ctx) ctx->err.clear();
}

// A host exception (std::bad_alloc) in the middle of a unit-level call: copies and kernels may already be queued and still read the
// descriptors, the staging and the caller's buffers -- the device is drained before the batch is declared gone and the caller told.
static int unit_call_threw(milzma_ctx* ctx, const std::exception& e) {
  if (ctx) {
    if (ctx->pending) {
      if (ctx->progress) __atomic_store_n(&ctx->progress[milzma_ctx::kMaxSpans], 1u, __ATOMIC_RELEASE);  // (waves waiting for a second upload: see fail())
      (void)hipSetDevice(ctx->device);
      (void)hipDeviceSynchronize();
    }
    ctx->pending = false;
    ctx->ev_used = 0;
    ctx->err = std::string("host exception: ") + e.what();
  }
  return MILZMA_INFRA_ERROR;
}

extern "C" int milzma_decode_units(milzma_ctx* ctx, const milzma_unit* units, uint32_t n, const void* d_in,
                                   void* d_out, milzma_result* results, void* hip_stream) {
  begin_call(ctx);
  try {
    return milzma_decode_units_impl(ctx, units, n, d_in, d_out, results, hip_stream);
  } catch (const std::exception& e) {  // (std::bad_alloc from a staging vector: never let it cross the C ABI)
    return unit_call_threw(ctx, e);
  }
}

extern "C" int milzma_decode_units_ex(milzma_ctx* ctx, const milzma_unit* units, uint32_t n, const void* d_in, void* d_out,
                                      milzma_result* results, void* hip_stream, uint32_t flags) {
  begin_call(ctx);
  try {
    if (ctx && (flags & ~(MILZMA_DECODE_GROW | MILZMA_DECODE_RESUME))) {
      ctx->err = "unknown flags";
      return MILZMA_INFRA_ERROR;
    }
    return milzma_decode_units_impl(ctx, units, n, d_in, d_out, results, hip_stream, flags);
  } catch (const std::exception& e) {
    return unit_call_threw(ctx, e);
  }
}

extern "C" int milzma_move_units(milzma_ctx* ctx, uint32_t n, const void* d_src, const uint64_t* src_off, void* d_dst,
                                 const uint64_t* dst_off, const uint64_t* len, void* hip_stream) {
  begin_call(ctx);
  try {
    return move_units_impl(ctx, n, d_src, src_off, d_dst, dst_off, len, static_cast<hipStream_t>(hip_stream));
  } catch (const std::exception& e) {
    if (ctx) ctx->err = std::string("host exception: ") + e.what();
    return MILZMA_INFRA_ERROR;
  }
}

extern "C" int milzma_decode_units_host(milzma_ctx* ctx, const milzma_unit* units, uint32_t n, const void* h_in,
                                        size_t in_bytes, void* h_out, size_t out_bytes, milzma_result* results) {
  begin_call(ctx);
  try {
    return milzma_decode_units_host_impl(ctx, units, n, h_in, in_bytes, h_out, out_bytes, results);
  } catch (const std::exception& e) {  // (std::bad_alloc from a staging vector: never let it cross the C ABI)
    if (ctx) ctx->err = std::string("host exception: ") + e.what();
    return MILZMA_INFRA_ERROR;
  }
}

extern "C" int milzma_lzma_decompress(milzma_ctx* ctx, const uint8_t* in, size_t in_len, const milzma_options* opt,
                                      milzma_output* out) {
  begin_call(ctx);
  try {
    return milzma_lzma_decompress_impl(ctx, in, in_len, opt, out);
  } catch (const std::exception& e) {  // (std::bad_alloc from a staging vector: never let it cross the C ABI)
    if (ctx) ctx->err = std::string("host exception: ") + e.what();
    if (out) out_fail(out, MILZMA_INFRA_ERROR, "%s", e.what());
    return MILZMA_INFRA_ERROR;
  }
}

extern "C" int milzma_lzma2_decompress(milzma_ctx* ctx, const uint8_t* in, size_t in_len, milzma_output* out) {
  begin_call(ctx);
  try {
    return milzma_lzma2_decompress_impl(ctx, in, in_len, out);
  } catch (const std::exception& e) {  // (std::bad_alloc from a staging vector: never let it cross the C ABI)
    if (ctx) ctx->err = std::string("host exception: ") + e.what();
    if (out) out_fail(out, MILZMA_INFRA_ERROR, "%s", e.what());
    return MILZMA_INFRA_ERROR;
  }
}

// Large whole-file calls are cut into groups of whole files that run on "lanes" (the context itself + further contexts on the same
// device), each group on its own host thread, its copies and kernel on the lane's own streams; uploads take turns in group order.
//  * default, 2 lanes, groups of >= 4096 decode units (a chip-full each): calls with >= 8192 units; the upload of group k + 1 and the
//    download + hand-over of group k - 1 run under group k's kernel.  One group alone cannot overlap its own three phases (every
//    stream takes the whole kernel), and this form does not need two kernels to run at once.
//  * MILZMA_LANES=3|4: groups of 512..2048 units, one per lane, kernels of different lanes running CONCURRENTLY (a stream's wave
//    is bound by its own instruction chain -- 104 cycles per decision with 4 waves on its SIMD, 80 alone: DESIGN.md 4.1 -- so a
//    group's kernel takes no longer next to the others than the single launch would, and starts after ITS share of the upload).
//    Measured (profiles/r03_batch_api.txt): 4096 files in one call 13.0 instead of 12.05 GB/s, 8192 files 14.7 instead of 13.6 --
//    but only if every lane's two streams get hardware queues of their own: the HIP runtime's default is 4 queues per device
//    (GPU_MAX_HW_QUEUES), streams beyond that share one and their kernels AND copies serialise (same call: 9.5 GB/s).  Hence opt-in,
//    for deployments that export GPU_MAX_HW_QUEUES >= 2 x lanes + 1.
// units_of(i): decode units file i contributes (1 per stream, blocks per .xz file).
namespace {

constexpr uint32_t kChipUnits = 4096, kMinGroupUnits = 512, kMaxGroupUnits = 2048, kMaxLanes = 4;

uint32_t lanes_wanted() {
  const char* e = env_get("MILZMA_LANES");
  return e ? std::min<uint32_t>(kMaxLanes, std::max(1, atoi(e))) : 2u;
}

template <class Units, class Call>
int grouped_batch(milzma_ctx* ctx, uint32_t n, const uint8_t* const* ins, const size_t* in_lens, milzma_output* outs, Units units_of,
                  Call call) {
  std::vector<uint32_t> cut{0};
  const uint32_t want = lanes_wanted();
  if (ctx && n && ins && in_lens && !env_get("MILZMA_NO_GROUPS")) {
    uint64_t total = 0, acc = 0;
    std::vector<uint32_t> u(n);
    for (uint32_t i = 0; i < n; i++) total += (u[i] = units_of(i));
    const bool small = want > 2;
    const uint64_t least = small ? kMinGroupUnits : kChipUnits;
    if (want > 1 && total >= 2 * least) {
      const uint64_t per = small ? std::min<uint64_t>(kMaxGroupUnits, std::max<uint64_t>(kMinGroupUnits, (total + want - 1) / want))
                                 : (total + total / kChipUnits - 1) / (total / kChipUnits);  // equal groups, each a chip-full or more
      for (uint32_t i = 0; i < n; i++) {
        acc += u[i];
        if (acc >= per && i + 1 < n && total - acc >= least / 2) {
          cut.push_back(i + 1);
          total -= acc;
          acc = 0;
        }
      }
    }
  }
  cut.push_back(n);
  const size_t groups = cut.size() - 1;
  if (groups <= 1) return call(ctx, n, ins, in_lens, outs);
  const size_t nl = std::min<size_t>(want, groups);
  while (ctx->lanes.size() + 1 < nl) {
    milzma_ctx* lane = nullptr;
    if (milzma_create(ctx->device, &lane) != MILZMA_OK) return call(ctx, n, ins, in_lens, outs);
    ctx->lanes.push_back(lane);
  }
  UploadTurn turn;
  std::vector<std::thread> th;
  std::vector<int> rc(nl, MILZMA_OK);
  const auto lane_body = [&](size_t k) {
    milzma_ctx* lane = k ? ctx->lanes[k - 1] : ctx;
    lane->turn = &turn;
    lane->budget_share = uint32_t(nl);
    for (size_t g = k; g < groups; g += nl) {
      lane->turn_no = uint32_t(g);
      lane->turn_done = false;
      const uint32_t lo = cut[g], m = cut[g + 1] - cut[g];
      int r = MILZMA_INFRA_ERROR;
      try {
        r = call(lane, m, ins + lo, in_lens + lo, outs + lo);
      } catch (const std::exception& e) {
        lane->err = std::string("host exception: ") + e.what();
        for (uint32_t i = lo; i < lo + m; i++) out_fail(&outs[i], MILZMA_INFRA_ERROR, "%s", e.what());
      }
      turn_release(lane);  // (a group that never reached its upload must not hold up the ones behind it)
      if (r != MILZMA_OK) rc[k] = r;
    }
    lane->turn = nullptr;
    lane->budget_share = 1;
  };
  // Lanes 1.. on threads of their own, lane 0 on the calling thread.  A thread that cannot be started (std::system_error) must not
  // take the process down through the vector's destructor while its siblings run: the lanes that did start are joined, and the
  // groups of the ones that did not are run here, one after the other.
  std::vector<size_t> not_started;
  for (size_t k = 1; k < nl; k++) {
    try {
      th.emplace_back(lane_body, k);
    } catch (const std::exception&) {
      not_started.push_back(k);
    }
  }
  lane_body(0);
  for (auto& t : th) t.join();
  for (size_t k : not_started) lane_body(k);
  for (milzma_ctx* lane : ctx->lanes) ctx->last_paths |= lane->last_paths;   // (what any group did, + the cut itself)
  ctx->last_paths |= MILZMA_PATH_GROUPED;
  int worst = MILZMA_OK;
  for (size_t k = 0; k < nl; k++)
    if (rc[k] != MILZMA_OK) {
      worst = rc[k];
      if (k) ctx->err = "lane " + std::to_string(k) + ": " + ctx->lanes[k - 1]->err;  // (always the failing lane's text, never a stale one)
    }
  return worst;
}

uint32_t xz_units_of(const uint8_t* in, size_t n) {
  std::vector<PlannedBlock> blocks;
  return plan_from_index(in, n, &blocks) && !blocks.empty() ? uint32_t(blocks.size()) : 1u;
}

}  // namespace

extern "C" int milzma_lzma_decompress_batch(milzma_ctx* ctx, uint32_t n, const uint8_t* const* ins, const size_t* in_lens,
                                            const milzma_options* opt, milzma_output* outs) {
  begin_call(ctx);
  try {
    return grouped_batch(
        ctx, n, ins, in_lens, outs, [](uint32_t) { return 1u; },
        [opt](milzma_ctx* c, uint32_t k, const uint8_t* const* i, const size_t* l, milzma_output* o) {
          return milzma_lzma_decompress_batch_impl(c, k, i, l, opt, o);
        });
  } catch (const std::exception& e) {  // (std::bad_alloc from a staging vector: never let it cross the C ABI)
    if (ctx) ctx->err = std::string("host exception: ") + e.what();
    for (uint32_t i = 0; i < n; i++) out_fail(&outs[i], MILZMA_INFRA_ERROR, "%s", e.what());
    return MILZMA_INFRA_ERROR;
  }
}

extern "C" int milzma_lzma2_decompress_batch(milzma_ctx* ctx, uint32_t n, const uint8_t* const* ins, const size_t* in_lens,
                                             milzma_output* outs) {
  begin_call(ctx);
  try {
    return grouped_batch(
        ctx, n, ins, in_lens, outs, [](uint32_t) { return 1u; },
        [](milzma_ctx* c, uint32_t k, const uint8_t* const* i, const size_t* l, milzma_output* o) {
          return milzma_lzma2_decompress_batch_impl(c, k, i, l, o);
        });
  } catch (const std::exception& e) {  // (std::bad_alloc from a staging vector: never let it cross the C ABI)
    if (ctx) ctx->err = std::string("host exception: ") + e.what();
    for (uint32_t i = 0; i < n; i++) out_fail(&outs[i], MILZMA_INFRA_ERROR, "%s", e.what());
    return MILZMA_INFRA_ERROR;
  }
}

extern "C" int milzma_xz_decompress_batch(milzma_ctx* ctx, uint32_t n, const uint8_t* const* ins, const size_t* in_lens,
                                          milzma_output* outs) {
  begin_call(ctx);
  try {
    return grouped_batch(
        ctx, n, ins, in_lens, outs, [&](uint32_t i) { return xz_units_of(ins[i], in_lens[i]); },
        [](milzma_ctx* c, uint32_t k, const uint8_t* const* i, const size_t* l, milzma_output* o) {
          return milzma_xz_decompress_batch_impl(c, k, i, l, o);
        });
  } catch (const std::exception& e) {  // (std::bad_alloc from a staging vector: never let it cross the C ABI)
    if (ctx) ctx->err = std::string("host exception: ") + e.what();
    for (uint32_t i = 0; i < n; i++) out_fail(&outs[i], MILZMA_INFRA_ERROR, "%s", e.what());
    return MILZMA_INFRA_ERROR;
  }
}

// ---- the whole-file batch calls in two halves --------------------------------------------------------------------
namespace {

template <class Call>
int batch_async(milzma_ctx* ctx, uint32_t n, const uint8_t* const* ins, const size_t* in_lens, milzma_output* outs, Call call) {
  if (!ctx) return MILZMA_INFRA_ERROR;
  if (ctx->batch_pending) {
    ctx->err = "a whole-file batch is already in flight on this context: call milzma_batch_wait first";
    return MILZMA_INFRA_ERROR;
  }
  if (n && (!ins || !in_lens || !outs)) {
    ctx->err = "null argument";
    return MILZMA_INFRA_ERROR;
  }
  try {
    std::vector<const uint8_t*> p(ins, ins + n);
    std::vector<size_t> l(in_lens, in_lens + n);
    ctx->batch_rc = MILZMA_OK;
    ctx->batch_thread = std::thread([ctx, n, outs, call, p = std::move(p), l = std::move(l)]() {
      ctx->batch_rc = call(ctx, n, p.data(), l.data(), outs);
    });
    ctx->batch_pending = true;
    return MILZMA_OK;
  } catch (const std::exception& e) {
    ctx->err = std::string("host exception: ") + e.what();
    return MILZMA_INFRA_ERROR;
  }
}

}  // namespace

extern "C" int milzma_lzma_decompress_batch_async(milzma_ctx* ctx, uint32_t n, const uint8_t* const* ins, const size_t* in_lens,
                                                  const milzma_options* opt, milzma_output* outs) {
  milzma_options o;
  milzma_default_options(&o);
  if (opt) o = *opt;
  return batch_async(ctx, n, ins, in_lens, outs, [o](milzma_ctx* c, uint32_t k, const uint8_t* const* i, const size_t* l, milzma_output* out) {
    return milzma_lzma_decompress_batch(c, k, i, l, &o, out);
  });
}

extern "C" int milzma_lzma2_decompress_batch_async(milzma_ctx* ctx, uint32_t n, const uint8_t* const* ins, const size_t* in_lens,
                                                   milzma_output* outs) {
  return batch_async(ctx, n, ins, in_lens, outs, [](milzma_ctx* c, uint32_t k, const uint8_t* const* i, const size_t* l, milzma_output* out) {
    return milzma_lzma2_decompress_batch(c, k, i, l, out);
  });
}

extern "C" int milzma_xz_decompress_batch_async(milzma_ctx* ctx, uint32_t n, const uint8_t* const* ins, const size_t* in_lens,
                                                milzma_output* outs) {
  return batch_async(ctx, n, ins, in_lens, outs, [](milzma_ctx* c, uint32_t k, const uint8_t* const* i, const size_t* l, milzma_output* out) {
    return milzma_xz_decompress_batch(c, k, i, l, out);
  });
}

extern "C" int milzma_batch_wait(milzma_ctx* ctx) {
  if (!ctx) return MILZMA_INFRA_ERROR;
  if (!ctx->batch_pending) {
    ctx->err = "no whole-file batch in flight on this context";
    return MILZMA_INFRA_ERROR;
  }
  if (ctx->batch_thread.joinable()) ctx->batch_thread.join();
  ctx->batch_pending = false;
  return ctx->batch_rc;
}

// Index of a well-formed .xz file -> one LZMA2 unit per block (offsets relative to the file's first byte, out_off / out_cap
// packed from 0 in file order): what milzma_xz_decompress_batch decodes ahead, for callers that keep files and
// output in device memory (bench.py --config xz).  The container checks (header / index / footer CRCs, block check
// values via milzma_crc_units) remain the caller's; the whole-file entry points do all of it.
extern "C" int milzma_xz_plan(const uint8_t* in, size_t in_len, milzma_unit* units, uint32_t cap, uint32_t* n_units,
                              uint32_t* check_id) {
  try {
    std::vector<PlannedBlock> blocks;
    if (!in || !n_units || !plan_from_index(in, in_len, &blocks)) return MILZMA_XZ_ERROR;
    *n_units = uint32_t(blocks.size());
    if (check_id) *check_id = in[in_len - 3];  // stream flags, second byte (footer copy)
    if (!units || cap < blocks.size()) return blocks.size() > cap ? MILZMA_INFRA_ERROR : MILZMA_OK;
    uint64_t out = 0;
    for (size_t k = 0; k < blocks.size(); k++) {
      milzma_unit& u = units[k];
      memset(&u, 0, sizeof u);
      u.kind = MILZMA_KIND_LZMA2;
      u.in_off = blocks[k].data_off;
      u.in_len = blocks[k].data_len;
      u.out_off = out;
      u.out_cap = blocks[k].unpacked;
      u.unpacked_size = blocks[k].unpacked;
      out += round_up(size_t(blocks[k].unpacked), 256);
    }
    return MILZMA_OK;
  } catch (const std::exception&) {
    return MILZMA_INFRA_ERROR;
  }
}

extern "C" int milzma_decode_units_async(milzma_ctx* ctx, const milzma_unit* units, uint32_t n, const void* d_in, void* d_out,
                                         void* hip_stream) {
  begin_call(ctx);
  try {
    return milzma_decode_units_async_impl(ctx, units, n, d_in, d_out, hip_stream);
  } catch (const std::exception& e) {
    return unit_call_threw(ctx, e);
  }
}

extern "C" int milzma_decode_units_wait(milzma_ctx* ctx, milzma_result* results) {
  try {
    return milzma_decode_units_wait_impl(ctx, results);
  } catch (const std::exception& e) {   // (the promotion rounds' staging; the batch is over either way)
    return unit_call_threw(ctx, e);
  }
}

// ------------------------------------------------------------------------------------------
// several GPUs of one node: one context + one host worker per device, work partitioned by
// compressed bytes (every public entry point of the reference builds a fresh decoder,
// src/lib.rs:44-105: streams / LZMA2 groups / XZ blocks never exchange anything)
// ------------------------------------------------------------------------------------------

struct milzma_multi {
  std::vector<milzma_ctx*> ctx;
  std::string err;
  // milzma_multi_decode_units_rooted: staging on the root device for what travels to / from the other devices, and how long it took
  DevBuf stage_in, stage_out;
  int stage_device = -1;
  float scatter_ms = 0.f, decode_ms = 0.f, gather_ms = 0.f;
};

namespace {

thread_local std::string g_multi_create_error;

// Longest-processing-time-first over (grouped) items; see milzma_partition in the header.
int partition_impl(const uint64_t* weights, const uint32_t* group, uint32_t n, uint32_t parts, uint32_t* part_of) {
  if (!part_of || parts == 0 || (n && !weights)) return MILZMA_INFRA_ERROR;
  struct Item {
    uint64_t w;
    uint32_t first;  // lowest member index (tie-break and determinism)
    std::vector<uint32_t> members;
  };
  std::vector<Item> items;
  std::unordered_map<uint32_t, size_t> of_group;
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t g = group ? group[i] : 0;
    if (g) {
      auto it = of_group.find(g);
      if (it != of_group.end()) {
        items[it->second].w += weights[i];
        items[it->second].members.push_back(i);
        continue;
      }
      of_group[g] = items.size();
    }
    items.push_back(Item{weights[i], i, {i}});
  }
  std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.w != b.w ? a.w > b.w : a.first < b.first; });
  std::vector<uint64_t> load(parts, 0);
  for (const Item& it : items) {
    uint32_t best = 0;
    for (uint32_t p = 1; p < parts; p++)
      if (load[p] < load[best]) best = p;
    load[best] += it.w;
    for (uint32_t i : it.members) part_of[i] = best;
  }
  return MILZMA_OK;
}

// runs fn(k) for every device index k on its own thread (the calling thread takes the last one).  Nothing thrown on a worker leaves
// it (an exception that escapes a std::thread is std::terminate, through the C ABI): failed(k, what) records it instead; a worker
// that cannot be started runs on the calling thread after the others.
template <class F, class G>
void per_device(size_t nd, F fn, G failed) {
  const auto guarded = [&](size_t k) {
    try {
      fn(k);
    } catch (const std::exception& e) {
      failed(k, e.what());
    }
  };
  std::vector<std::thread> th;
  std::vector<size_t> not_started;
  for (size_t k = 0; k + 1 < nd; k++) {
    try {
      th.emplace_back(guarded, k);
    } catch (const std::exception&) {
      not_started.push_back(k);
    }
  }
  if (nd) guarded(nd - 1);
  for (auto& t : th) t.join();
  for (size_t k : not_started) guarded(k);
}

int multi_fail(milzma_multi* m, const std::string& why) {
  if (m) m->err = why;
  return MILZMA_INFRA_ERROR;
}

// Whole-file batch over the devices: files partitioned by size, each device runs the single-device entry point on its share.
// every file that holds no result gets the infrastructure error (never left as the caller's zeroed "empty success")
void multi_outs_fail(uint32_t n, milzma_output* outs, const uint8_t* has_result, const char* why) {
  if (!outs) return;
  for (uint32_t i = 0; i < n; i++) {
    if (has_result && has_result[i]) continue;
    out_reset(&outs[i]);
    out_fail(&outs[i], MILZMA_INFRA_ERROR, "%s", why);
  }
}

template <class Call>
int multi_file_batch(milzma_multi* m, uint32_t n, const uint8_t* const* ins, const size_t* in_lens, milzma_output* outs, Call call) {
  if (!m || m->ctx.empty()) {
    multi_outs_fail(n, outs, nullptr, "no multi-device handle");
    return MILZMA_INFRA_ERROR;
  }
  m->err.clear();
  if (n == 0) return MILZMA_OK;
  if (!ins || !in_lens || !outs) {
    multi_outs_fail(n, outs, nullptr, "null argument");
    return multi_fail(m, "null argument");
  }
  std::vector<uint8_t> has_result;
  try {
    has_result.assign(n, 0);
    const uint32_t nd = uint32_t(m->ctx.size());
    std::vector<uint64_t> w(n);
    for (uint32_t i = 0; i < n; i++) w[i] = in_lens[i];
    std::vector<uint32_t> part(n);
    partition_impl(w.data(), nullptr, n, nd, part.data());
    std::vector<std::vector<uint32_t>> share(nd);
    for (uint32_t i = 0; i < n; i++) share[part[i]].push_back(i);
    std::vector<int> rc(nd, MILZMA_OK);
    per_device(
        nd,
        [&](size_t k) {
          const std::vector<uint32_t>& idx = share[k];
          if (idx.empty()) return;
          std::vector<const uint8_t*> sub_in(idx.size());
          std::vector<size_t> sub_len(idx.size());
          std::vector<milzma_output> sub_out(idx.size());
          for (size_t j = 0; j < idx.size(); j++) {
            sub_in[j] = ins[idx[j]];
            sub_len[j] = in_lens[idx[j]];
          }
          rc[k] = call(m->ctx[k], uint32_t(idx.size()), sub_in.data(), sub_len.data(), sub_out.data());
          for (size_t j = 0; j < idx.size(); j++) {  // (the single-device calls fill every slot, also when they fail)
            outs[idx[j]] = sub_out[j];
            has_result[idx[j]] = 1;
          }
        },
        [&](size_t k, const char* what) {
          rc[k] = MILZMA_INFRA_ERROR;
          m->ctx[k]->err = std::string("host exception: ") + what;
        });
    for (uint32_t k = 0; k < nd; k++)
      if (rc[k] != MILZMA_OK) {
        const std::string why = "device " + std::to_string(m->ctx[k]->device) + ": " + m->ctx[k]->err;
        multi_outs_fail(n, outs, has_result.data(), why.c_str());
        return multi_fail(m, why);
      }
    return MILZMA_OK;
  } catch (const std::exception& e) {
    const std::string why = std::string("host exception: ") + e.what();
    multi_outs_fail(n, outs, has_result.empty() ? nullptr : has_result.data(), why.c_str());
    return multi_fail(m, why);
  }
}

}  // namespace

extern "C" int milzma_partition(const uint64_t* weights, const uint32_t* group, uint32_t n, uint32_t parts, uint32_t* part_of) {
  try {
    return partition_impl(weights, group, n, parts, part_of);
  } catch (const std::exception&) {
    return MILZMA_INFRA_ERROR;
  }
}

extern "C" int milzma_multi_create(uint64_t device_mask, milzma_multi** out) {
  if (!out) return MILZMA_INFRA_ERROR;
  *out = nullptr;
  try {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
      (void)hipGetLastError();
      g_multi_create_error = "no usable HIP device (this library has no CPU decode path)";
      return MILZMA_INFRA_ERROR;
    }
    if (device_mask == 0) device_mask = count >= 64 ? ~uint64_t(0) : ((uint64_t(1) << count) - 1);
    // MILZMA_MULTI_REPLICAS=k (testing aid): k contexts per selected device, each treated as a device of its own -- the partition,
    // the per-device workers and the merge of their results run with several shares on a node that has one GPU.
    int replicas = 1;
    if (const char* e = env_get("MILZMA_MULTI_REPLICAS")) replicas = std::min(8, std::max(1, atoi(e)));
    auto* m = new milzma_multi();
    for (int d = 0; d < 64; d++) {
      if (!((device_mask >> d) & 1)) continue;
      for (int k = 0; k < replicas; k++) {
        milzma_ctx* c = nullptr;
        if (d >= count || milzma_create(d, &c) != MILZMA_OK) {
          g_multi_create_error = "device " + std::to_string(d) + ": " + (d >= count ? std::string("not present") : g_create_error);
          milzma_multi_destroy(m);
          return MILZMA_INFRA_ERROR;
        }
        m->ctx.push_back(c);
      }
    }
    *out = m;
    return MILZMA_OK;
  } catch (const std::exception& e) {
    g_multi_create_error = std::string("host exception: ") + e.what();
    return MILZMA_INFRA_ERROR;
  }
}

extern "C" void milzma_multi_destroy(milzma_multi* m) {
  if (!m) return;
  if (m->stage_device >= 0 && hipSetDevice(m->stage_device) == hipSuccess) {
    dev_release(m->stage_in);
    dev_release(m->stage_out);
  }
  for (milzma_ctx* c : m->ctx) milzma_destroy(c);
  delete m;
}

extern "C" uint32_t milzma_multi_devices(const milzma_multi* m, int* ordinals, uint32_t cap) {
  if (!m) return 0;
  for (uint32_t k = 0; ordinals && k < cap && k < m->ctx.size(); k++) ordinals[k] = m->ctx[k]->device;
  return uint32_t(m->ctx.size());
}

extern "C" const char* milzma_multi_last_error(const milzma_multi* m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }

extern "C" float milzma_multi_last_kernel_ms(const milzma_multi* m, uint32_t k, uint32_t* launches) {
  if (launches) *launches = 0;
  if (!m) return 0.f;
  if (k != UINT32_MAX) return k < m->ctx.size() ? milzma_last_kernel_ms(m->ctx[k], launches) : 0.f;
  float best = 0.f;
  for (milzma_ctx* c : m->ctx) {
    uint32_t l = 0;
    const float ms = milzma_last_kernel_ms(c, &l);
    if (ms >= best) {
      best = ms;
      if (launches) *launches = l;
    }
  }
  return best;
}

extern "C" int milzma_multi_decode_units(milzma_multi* m, const milzma_unit* units, uint32_t n, const uint32_t* device_of,
                                         const void* const* d_in, void* const* d_out, milzma_result* results) {
  if (!m || m->ctx.empty()) return MILZMA_INFRA_ERROR;
  m->err.clear();   // (what milzma_multi_last_error returns afterwards belongs to THIS call)
  try {
    if (n == 0) return MILZMA_OK;
    if (!units || !device_of || !d_in || !d_out || !results) return multi_fail(m, "null argument");
    const uint32_t nd = uint32_t(m->ctx.size());
    std::vector<std::vector<uint32_t>> share(nd);
    for (uint32_t i = 0; i < n; i++) {
      if (device_of[i] >= nd) return multi_fail(m, "unit " + std::to_string(i) + ": device index out of range");
      share[device_of[i]].push_back(i);
    }
    std::vector<int> rc(nd, MILZMA_OK);
    per_device(nd, [&](size_t k) {
      const std::vector<uint32_t>& idx = share[k];
      if (idx.empty()) {
        m->ctx[k]->last_ms = 0.f;
        m->ctx[k]->last_launches = 0;
        return;
      }
      std::vector<milzma_unit> sub(idx.size());
      std::vector<milzma_result> res(idx.size());
      for (size_t j = 0; j < idx.size(); j++) sub[j] = units[idx[j]];
      rc[k] = milzma_decode_units(m->ctx[k], sub.data(), uint32_t(sub.size()), d_in[k], d_out[k], res.data(), nullptr);
      if (rc[k] == MILZMA_OK)
        for (size_t j = 0; j < idx.size(); j++) results[idx[j]] = res[j];
    }, [&](size_t k, const char* what) {
      rc[k] = MILZMA_INFRA_ERROR;
      m->ctx[k]->err = std::string("host exception: ") + what;
    });
    for (uint32_t k = 0; k < nd; k++)
      if (rc[k] != MILZMA_OK) return multi_fail(m, "device " + std::to_string(m->ctx[k]->device) + ": " + m->ctx[k]->err);
    return MILZMA_OK;
  } catch (const std::exception& e) {
    return multi_fail(m, std::string("host exception: ") + e.what());
  }
}

// One ingest point (north_star: "input scatter and output gather over xGMI"): the whole batch lives in the memory of ONE device of
// the handle -- `root` -- and comes back there.  The units are partitioned by compressed bytes like everywhere else; the root's own
// share is decoded in place; every other device's share is packed on the root (one move kernel), crosses to that device with ONE
// device-to-device copy (hipMemcpyPeer: the direct xGMI link between the two GPUs where peer access exists), is decoded there, and
// its output crosses back the same way and is put in place by one more move kernel.  All devices work concurrently, each on its own
// host thread; nothing passes through host memory and there is no collective (each device talks to the root only).
extern "C" int milzma_multi_decode_units_rooted(milzma_multi* m, uint32_t root, const milzma_unit* units, uint32_t n, const void* d_in,
                                                void* d_out, milzma_result* results) {
  if (!m || m->ctx.empty()) return MILZMA_INFRA_ERROR;
  m->err.clear();
  try {
    using clk = std::chrono::steady_clock;
    const auto ms_since = [](clk::time_point t0) { return std::chrono::duration<float, std::milli>(clk::now() - t0).count(); };
    m->scatter_ms = m->decode_ms = m->gather_ms = 0.f;
    if (n == 0) return MILZMA_OK;
    const uint32_t nd = uint32_t(m->ctx.size());
    if (!units || !d_in || !d_out || !results) return multi_fail(m, "null argument");
    if (root >= nd) return multi_fail(m, "root: device index out of range");
    milzma_ctx* rc = m->ctx[root];
    std::vector<uint64_t> w(n);
    for (uint32_t i = 0; i < n; i++) w[i] = units[i].in_len + 1;
    std::vector<uint32_t> part(n);
    partition_impl(w.data(), nullptr, n, nd, part.data());
    // (the planner numbers parts 0..nd-1 by load: which part the root keeps does not matter, every part is about the same size)
    std::vector<std::vector<uint32_t>> share(nd);
    for (uint32_t i = 0; i < n; i++) share[part[i]].push_back(i);
    // packed layouts of the shares that travel
    std::vector<std::vector<milzma_unit>> sub(nd);
    std::vector<size_t> in_base(nd, 0), out_base(nd, 0), in_bytes(nd, 0), out_bytes(nd, 0);
    size_t in_total = 0, out_total = 0;
    std::vector<uint64_t> so, dof, ln;
    for (uint32_t k = 0; k < nd; k++) {
      if (k == root) continue;
      in_base[k] = in_total;
      out_base[k] = out_total;
      sub[k].resize(share[k].size());
      size_t io = 0, oo = 0;
      for (size_t j = 0; j < share[k].size(); j++) {
        const milzma_unit& u = units[share[k][j]];
        sub[k][j] = u;
        sub[k][j].in_off = io;
        sub[k][j].out_off = oo;
        so.push_back(u.in_off);
        dof.push_back(in_total + io);
        ln.push_back(u.in_len);
        io += round_up(size_t(u.in_len), 256);
        oo += round_up(size_t(u.out_cap), 256);
      }
      in_bytes[k] = io;
      out_bytes[k] = oo;
      in_total += io;
      out_total += oo;
    }
    if (!hip_ok(rc, hipSetDevice(rc->device), "hipSetDevice")) return multi_fail(m, rc->err);
    if (m->stage_device != rc->device) {  // (the staging follows the root)
      if (m->stage_device >= 0 && hipSetDevice(m->stage_device) == hipSuccess) {
        dev_release(m->stage_in);
        dev_release(m->stage_out);
      }
      (void)hipSetDevice(rc->device);
      m->stage_device = rc->device;
    }
    if (!dev_reserve(rc, m->stage_in, in_total + 512) || !dev_reserve(rc, m->stage_out, out_total + 512)) return multi_fail(m, rc->err);
    // 1. scatter, root side: pack what leaves
    const auto t_scatter = clk::now();
    if (!so.empty() && move_units_impl(rc, uint32_t(so.size()), d_in, so.data(), m->stage_in.p, dof.data(), ln.data(), work_stream(rc)) != MILZMA_OK)
      return multi_fail(m, rc->err);
    const float pack_ms = so.empty() ? 0.f : ms_since(t_scatter);
    // 2. every device: its share in, decode, its output back.  The way back is the waves' own where it can be: a device whose share
    //    is all in the fast kernel's class and that can reach the root's memory (peer access) runs its share as ONE streamed launch
    //     whose per-unit destinations are the caller's slices on the root -- every 64 KiB span crosses xGMI while the
    //    unit is still being decoded, nothing is left to gather when the kernel ends (equal streams end together: a copy behind the
    //    kernel could overlap nothing).  Otherwise (other classes, no peer access, a promoted LZMA2 unit, MILZMA_ROOTED_STREAM=0): one
    //    peer copy into the root's staging behind the decode, placed by the move kernel below.
    const char* const rooted_env = env_get("MILZMA_ROOTED_STREAM");
    const bool stream_back = !(rooted_env && !strcmp(rooted_env, "0"));
    std::vector<int> rcode(nd, MILZMA_OK);
    std::vector<uint8_t> wrote_home(nd, 0);
    std::vector<float> t_in(nd, 0.f), t_dec(nd, 0.f), t_out(nd, 0.f);
    std::vector<std::vector<milzma_result>> res(nd);
    per_device(
        nd,
        [&](size_t k) {
          milzma_ctx* c = m->ctx[k];
          c->last_ms = 0.f;
          c->last_launches = 0;
          if (share[k].empty()) return;
          res[k].resize(share[k].size());
          const auto bad = [&]() { rcode[k] = MILZMA_INFRA_ERROR; };
          if (k == root) {
            std::vector<milzma_unit> own(share[k].size());
            for (size_t j = 0; j < own.size(); j++) own[j] = units[share[k][j]];
            const auto t0 = clk::now();
            if (milzma_decode_units(c, own.data(), uint32_t(own.size()), d_in, d_out, res[k].data(), work_stream(c)) != MILZMA_OK) return bad();
            t_dec[k] = ms_since(t0);
            return;
          }
          if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice") || !dev_reserve(c, c->in, in_bytes[k] + 512) ||
              !dev_reserve(c, c->out, out_bytes[k] + 512))
            return bad();
          auto t0 = clk::now();
          if (!hip_ok(c, hipMemcpyPeer(c->in.p, c->device, static_cast<const uint8_t*>(m->stage_in.p) + in_base[k], rc->device, in_bytes[k]),
                      "device-to-device scatter"))
            return bad();
          t_in[k] = ms_since(t0);
          bool direct = stream_back && c->use_fast;
          uint64_t max_cap = 0;
          for (const milzma_unit& u : sub[k]) {
            direct = direct && classify(c, u) == kFast;
            max_cap = std::max<uint64_t>(max_cap, u.out_cap);
          }
          if (const char* peer_env = env_get("MILZMA_ROOTED_PEER"); peer_env && !strcmp(peer_env, "0")) direct = false;
          if (direct && c->device != rc->device) {
            int can = 0;
            direct = hipDeviceCanAccessPeer(&can, c->device, rc->device) == hipSuccess && can != 0;
            if (direct) {
              const hipError_t pe = hipDeviceEnablePeerAccess(rc->device, 0);
              direct = pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled;
            }
            (void)hipGetLastError();
          }
          if (direct && ensure_progress(c)) {
            size_t span = size_t(64) << 10;
            while ((size_t(max_cap) + span) / span + 1 > milzma_ctx::kMaxSpans) span *= 2;
            std::vector<uint64_t> ptrs(share[k].size() * 2);
            for (size_t j = 0; j < share[k].size(); j++) {
              const milzma_unit& u = units[share[k][j]];
              ptrs[2 * j] = uint64_t(reinterpret_cast<uintptr_t>(static_cast<uint8_t*>(d_out) + u.out_off));
              ptrs[2 * j + 1] = u.out_cap;
            }
            direct = span <= 0x80000000u && upload_host_ptrs(c, ptrs, work_stream(c));
            if (direct) {
              c->stream_span = uint32_t(span);
              c->stream_spans = uint32_t((size_t(max_cap) + 2 * span - 1) / span);
              c->stream_host = nullptr;
              c->stream_ptrs = static_cast<const uint64_t*>(c->hostptrs.p);
              c->stream_in_host = false;
            }
          } else {
            direct = false;
          }
          t0 = clk::now();
          const int dr = milzma_decode_units(c, sub[k].data(), uint32_t(sub[k].size()), c->in.p, c->out.p, res[k].data(), work_stream(c));
          c->stream_span = c->stream_spans = 0;
          c->stream_ptrs = nullptr;
          if (dr != MILZMA_OK) return bad();
          t_dec[k] = ms_since(t0);
          // (one launch, and it was the streamed one: every unit's bytes are at home.  A promoted unit ran again in a launch of its
          //  own, without destinations: then the whole share takes the copy.)
          if (direct && c->stream_active && c->last_launches == 1) {
            wrote_home[k] = 1;
            return;
          }
          t0 = clk::now();
          if (!hip_ok(c, hipMemcpyPeer(static_cast<uint8_t*>(m->stage_out.p) + out_base[k], rc->device, c->out.p, c->device, out_bytes[k]),
                      "device-to-device gather"))
            return bad();
          t_out[k] = ms_since(t0);
        },
        [&](size_t k, const char* what) {
          rcode[k] = MILZMA_INFRA_ERROR;
          m->ctx[k]->err = std::string("host exception: ") + what;
        });
    for (uint32_t k = 0; k < nd; k++)
      if (rcode[k] != MILZMA_OK) return multi_fail(m, "device " + std::to_string(m->ctx[k]->device) + ": " + m->ctx[k]->err);
    // 3. gather, root side: every travelled output into its place
    const auto t_gather = clk::now();
    so.clear();
    dof.clear();
    ln.clear();
    for (uint32_t k = 0; k < nd; k++)
      for (size_t j = 0; j < share[k].size(); j++) {
        const uint32_t i = share[k][j];
        results[i] = res[k][j];
        if (k == root || wrote_home[k]) continue;
        so.push_back(out_base[k] + sub[k][j].out_off);
        dof.push_back(units[i].out_off);
        ln.push_back(std::min<uint64_t>(res[k][j].out_len, units[i].out_cap));
      }
    if (!hip_ok(rc, hipSetDevice(rc->device), "hipSetDevice") ||
        (!so.empty() && move_units_impl(rc, uint32_t(so.size()), m->stage_out.p, so.data(), d_out, dof.data(), ln.data(), work_stream(rc)) != MILZMA_OK))
      return multi_fail(m, rc->err);
    const float place_ms = ms_since(t_gather);
    float in_max = 0.f, out_max = 0.f, dec_max = 0.f;
    for (uint32_t k = 0; k < nd; k++) {
      in_max = std::max(in_max, t_in[k]);
      out_max = std::max(out_max, t_out[k]);
      dec_max = std::max(dec_max, t_dec[k]);
    }
    m->scatter_ms = pack_ms + in_max;   // the packing on the root + the slowest device's copy in
    m->decode_ms = dec_max;
    m->gather_ms = out_max + place_ms;
    return MILZMA_OK;
  } catch (const std::exception& e) {
    return multi_fail(m, std::string("host exception: ") + e.what());
  }
}

extern "C" void milzma_multi_last_transfer_ms(const milzma_multi* m, float* scatter_ms, float* decode_ms, float* gather_ms) {
  if (scatter_ms) *scatter_ms = m ? m->scatter_ms : 0.f;
  if (decode_ms) *decode_ms = m ? m->decode_ms : 0.f;
  if (gather_ms) *gather_ms = m ? m->gather_ms : 0.f;
}

extern "C" int milzma_multi_decode_units_host(milzma_multi* m, const milzma_unit* units, uint32_t n, const void* h_in, size_t in_bytes,
                                              void* h_out, size_t out_bytes, milzma_result* results) {
  if (!m || m->ctx.empty()) return MILZMA_INFRA_ERROR;
  m->err.clear();
  try {
    if (n == 0) return MILZMA_OK;
    if (!units || !results || (in_bytes && !h_in) || (out_bytes && !h_out)) return multi_fail(m, "null argument");
    // the same descriptor checks as milzma_decode_units_host: nothing leaves the caller's buffers, no two outputs overlap
    {
      std::vector<std::pair<uint64_t, uint64_t>> spans;
      spans.reserve(n);
      for (uint32_t i = 0; i < n; i++) {
        const milzma_unit& u = units[i];
        if (u.in_off > in_bytes || u.in_len > in_bytes - u.in_off || u.out_off > out_bytes || u.out_cap > out_bytes - u.out_off)
          return multi_fail(m, "unit " + std::to_string(i) + ": input or output slice outside the buffers");
        if (u.out_cap) spans.emplace_back(u.out_off, u.out_off + u.out_cap);
      }
      std::sort(spans.begin(), spans.end());
      for (size_t k = 1; k < spans.size(); k++)
        if (spans[k].first < spans[k - 1].second) return multi_fail(m, "overlapping output slices");
    }
    const uint32_t nd = uint32_t(m->ctx.size());
    std::vector<uint64_t> w(n);
    for (uint32_t i = 0; i < n; i++) w[i] = units[i].in_len + 1;
    std::vector<uint32_t> part(n);
    partition_impl(w.data(), nullptr, n, nd, part.data());
    std::vector<std::vector<uint32_t>> share(nd);
    for (uint32_t i = 0; i < n; i++) share[part[i]].push_back(i);
    const uint8_t* hin = static_cast<const uint8_t*>(h_in);
    uint8_t* hout = static_cast<uint8_t*>(h_out);
    std::vector<int> rc(nd, MILZMA_OK);
    per_device(nd, [&](size_t k) {
      milzma_ctx* ctx = m->ctx[k];
      const std::vector<uint32_t>& idx = share[k];
      ctx->last_ms = 0.f;
      ctx->last_launches = 0;
      if (idx.empty()) return;
      // this device's share, packed: inputs and output slices at 256-byte aligned offsets of its own staging buffers
      std::vector<milzma_unit> sub(idx.size());
      size_t in_total = 0, out_total = 0;
      for (size_t j = 0; j < idx.size(); j++) {
        sub[j] = units[idx[j]];
        sub[j].in_off = in_total;
        sub[j].out_off = out_total;
        in_total += round_up(size_t(sub[j].in_len), 256);
        out_total += round_up(size_t(sub[j].out_cap), 256);
      }
      const auto bad = [&](const char* what) {
        if (what) ctx->err = what;
        rc[k] = MILZMA_INFRA_ERROR;
      };
      if (!hip_ok(ctx, hipSetDevice(ctx->device), "hipSetDevice") || !pin_reserve(ctx, ctx->pin_in, in_total) ||
          !pin_reserve(ctx, ctx->pin_out, out_total) || !dev_reserve(ctx, ctx->in, in_total + 512) ||
          !dev_reserve(ctx, ctx->out, out_total + 512))
        return bad(nullptr);
      uint8_t* pin = static_cast<uint8_t*>(ctx->pin_in.p);
      {  // gather || H2D in eight groups, as in the whole-file batch path
        const size_t groups = std::min<size_t>(8, sub.size());
        std::vector<size_t> first(groups + 1), bounds(groups + 1);
        for (size_t g = 0; g <= groups; g++) {
          first[g] = sub.size() * g / groups;
          bounds[g] = g == groups ? in_total : size_t(sub[first[g]].in_off);
        }
        if (!staged_h2d(ctx, ctx->in.p, pin, bounds, [&](size_t g) {
              parallel_for(first[g + 1] - first[g], [&](size_t j0) {
                const size_t j = first[g] + j0;
                memcpy(pin + sub[j].in_off, hin + units[idx[j]].in_off, size_t(sub[j].in_len));
              });
            }))
          return bad(nullptr);
      }
      std::vector<milzma_result> res(sub.size());
      if (milzma_decode_units(ctx, sub.data(), uint32_t(sub.size()), ctx->in.p, ctx->out.p, res.data(), work_stream(ctx)) != MILZMA_OK)
        return bad(nullptr);
      ChunkedCopy d2h;
      if (!d2h.start_d2h(ctx, ctx->pin_out.p, ctx->out.p, out_total)) return bad(nullptr);
      const uint8_t* pout = static_cast<const uint8_t*>(ctx->pin_out.p);
      std::vector<uint8_t> failed(sub.size(), 0);
      parallel_for(sub.size(), [&](size_t j) {
        const size_t got = size_t(std::min<uint64_t>(res[j].out_len, sub[j].out_cap));
        if (!d2h.wait_until(size_t(sub[j].out_off) + got)) {
          failed[j] = 1;
          return;
        }
        if (got) memcpy(hout + units[idx[j]].out_off, pout + sub[j].out_off, got);
        results[idx[j]] = res[j];
      });
      for (uint8_t f : failed)
        if (f) return bad("D2H output failed");
    }, [&](size_t k, const char* what) {
      rc[k] = MILZMA_INFRA_ERROR;
      m->ctx[k]->err = std::string("host exception: ") + what;
    });
    for (uint32_t k = 0; k < nd; k++)
      if (rc[k] != MILZMA_OK) return multi_fail(m, "device " + std::to_string(m->ctx[k]->device) + ": " + m->ctx[k]->err);
    return MILZMA_OK;
  } catch (const std::exception& e) {
    return multi_fail(m, std::string("host exception: ") + e.what());
  }
}

extern "C" int milzma_multi_lzma_decompress_batch(milzma_multi* m, uint32_t n, const uint8_t* const* ins, const size_t* in_lens,
                                                  const milzma_options* opt, milzma_output* outs) {
  try {
    return multi_file_batch(m, n, ins, in_lens, outs, [opt](milzma_ctx* c, uint32_t k, const uint8_t* const* i, const size_t* l, milzma_output* o) {
      return milzma_lzma_decompress_batch(c, k, i, l, opt, o);
    });
  } catch (const std::exception& e) {
    return multi_fail(m, std::string("host exception: ") + e.what());
  }
}

extern "C" int milzma_multi_lzma2_decompress_batch(milzma_multi* m, uint32_t n, const uint8_t* const* ins, const size_t* in_lens,
                                                   milzma_output* outs) {
  try {
    return multi_file_batch(m, n, ins, in_lens, outs, [](milzma_ctx* c, uint32_t k, const uint8_t* const* i, const size_t* l, milzma_output* o) {
      return milzma_lzma2_decompress_batch(c, k, i, l, o);
    });
  } catch (const std::exception& e) {
    return multi_fail(m, std::string("host exception: ") + e.what());
  }
}

extern "C" int milzma_multi_xz_decompress_batch(milzma_multi* m, uint32_t n, const uint8_t* const* ins, const size_t* in_lens,
                                                milzma_output* outs) {
  try {
    return multi_file_batch(m, n, ins, in_lens, outs, [](milzma_ctx* c, uint32_t k, const uint8_t* const* i, const size_t* l, milzma_output* o) {
      return milzma_xz_decompress_batch(c, k, i, l, o);
    });
  } catch (const std::exception& e) {
    return multi_fail(m, std::string("host exception: ") + e.what());
  }
}
