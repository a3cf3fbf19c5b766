// host.cpp -- host side of the C ABI (include/milzma.h).
//
// What runs here is what the reference runs *around* its hot loop: header and container
// parsing (LzmaParams::read_header, src/decode/lzma.rs:96-161; xz::decode_stream,
// src/decode/xz.rs), option handling (src/decode/options.rs), error rendering (src/error.rs)
// and -- new -- turning many streams / blocks into one batch of wavefront-sized decode units.
// All decoding happens in the HIP kernels; there is no CPU decode path in this library.

#include "host_internal.h"

using namespace milzma;
using namespace milzma::host;

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------

MILZMA_HOST_NS_BEGIN


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

MILZMA_HOST_NS_END



MILZMA_HOST_NS_BEGIN

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






MILZMA_HOST_NS_END

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
MILZMA_HOST_NS_BEGIN

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

size_t out_live_buffers() {   // (how many result buffers callers hold right now: the sanitizer harness asks after every sequence)
  OutPool& p = out_pool();
  std::lock_guard<std::mutex> lock(p.mu);
  return p.live.size();
}

size_t out_pooled_buffers() {   // (buffers the pool holds: 0 after milzma_pool_trim(0))
  OutPool& p = out_pool();
  std::lock_guard<std::mutex> lock(p.mu);
  size_t n = 0;
  for (auto* m : {&p.free_pinned, &p.free_by_cap})
    for (auto& kv : *m) n += kv.second.size();
  return n;
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
uint8_t* out_alloc(size_t n, bool pinned) {
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

// count buffers at once (a batch call wants thousands): everything the pool holds is taken under ONE lock -- sixteen host threads taking
// 4096 buffers one by one spent 10 ms of every whole-file call on this mutex (two acquisitions per buffer) --, what it does not hold is
// allocated outside it on the host threads, the registry grows under one more.  out[k] = nullptr where memory could not be had; returns
// false if any is.
bool out_alloc_many(const size_t* n, size_t count, bool pinned, uint8_t** out) {
  OutPool& p = out_pool();
  std::vector<OutHdr*> h(count, nullptr);
  try {
    {
      std::lock_guard<std::mutex> lock(p.mu);
      auto& m = pinned ? p.free_pinned : p.free_by_cap;
      for (size_t k = 0; k < count; k++) {
        const size_t cap = out_class(n[k]);
        auto it = m.lower_bound(cap);
        while (it != m.end() && it->second.empty()) it = m.erase(it);
        if (it != m.end() && it->first <= std::max(cap * 2, cap + (size_t(1) << 16))) {
          h[k] = it->second.back();
          it->second.pop_back();
          p.held -= size_t(h[k]->cap);
        }
      }
    }
    std::atomic<int> failed{0};
    parallel_for(count, [&](size_t k) {
      if (h[k]) return;
      const size_t cap = out_class(n[k]);
      if (pinned) {
        void* q = nullptr;
        if (hipHostMalloc(&q, sizeof(OutHdr) + cap, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
          (void)hipGetLastError();
          failed.store(1);
          return;
        }
        h[k] = static_cast<OutHdr*>(q);
      } else {
        h[k] = static_cast<OutHdr*>(malloc(sizeof(OutHdr) + cap));
        if (!h[k]) {
          failed.store(1);
          return;
        }
      }
      h[k]->cap = cap;
      h[k]->pinned = pinned ? 1 : 0;
    });
    std::lock_guard<std::mutex> lock(p.mu);
    p.live.reserve(p.live.size() + count);
    for (size_t k = 0; k < count; k++) {
      out[k] = h[k] ? reinterpret_cast<uint8_t*>(h[k] + 1) : nullptr;
      if (!h[k]) continue;
      p.live.insert(h[k] + 1);
      p.live_bytes += size_t(h[k]->cap);
      h[k] = nullptr;   // (handed out)
    }
    p.peak_live = std::max(p.peak_live, p.live_bytes);
    return failed.load() == 0;
  } catch (const std::bad_alloc&) {  // (the registry could not grow: what is not registered yet is not handed out)
    for (size_t k = 0; k < count; k++)
      if (h[k]) {
        hdr_free(h[k]);
        out[k] = nullptr;
      }
    return false;
  }
}

MILZMA_HOST_NS_END

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

MILZMA_HOST_NS_BEGIN

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
  ctx->err = "hipHostMalloc: no page-locked memory for the span counters";   // (a caller that gives up on this says why: the sanitizer harness's
  return false;                                                                //  push-mode fault injection found a write failing without a text)
}

// Launches `order` (unit indices) in class `cls`; kernel time is accumulated into ctx.
bool launch_class(milzma_ctx* ctx, LitClass cls, const std::vector<uint32_t>& order, uint32_t order_base,
                  const uint8_t* d_in, uint8_t* d_out, hipStream_t stream, bool grow = false, bool resume = false, bool feed = false) {
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
    uint32_t need = std::max<uint32_t>(4, ctx->slab_min_lclp);   // (at least what an LZMA2 unit can switch to; a batch that takes new members
                                                                 //  later -- MILZMA_KIND_START -- asks for room for theirs up front)
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
        if (feed) {   // (the generic kernel cannot park a unit)
          ctx->err = "MILZMA_DECODE_FEED: no memory for the literal-row slab of the fed units";
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
      // (the launch is the whole batch: the counters reach n -- or nobody counts: stream_feed)
      const bool want_stream = ctx->stream_span != 0 && (ctx->stream_host != nullptr || ctx->stream_ptrs != nullptr) &&
                               ((m == ctx->pend_n && !resume) || ctx->stream_feed);
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
            // (stream_feed: nobody reads the counters, and an earlier launch of this very call may still be adding to them -- found by
            //  ThreadSanitizer on the stand-in kernels)
            if (!ctx->stream_feed) memset(ctx->progress, 0, milzma_ctx::kMaxSpans * sizeof(uint32_t));
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
                                                   ctx->slice_quantum, ctx->slice_mode > 1, ctx->slice_ctx.p, grow,
                                                   feed ? (ctx->stream_feed ? 3u : 1u) : 0u,   // (bit 1: units parked for input deliver their last turn with their next launch)
                                                   ctx->stream_span, ctx->stream_spans,
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

MILZMA_HOST_NS_END

// Enqueue: descriptor upload, one launch per class, result download into a page-locked buffer -- all on `stream`,
// nothing waits for the GPU.
int milzma_decode_units_async_impl(milzma_ctx* ctx, const milzma_unit* units, uint32_t n, const void* d_in,
                                          void* d_out, void* hip_stream, uint32_t flags, const milzma_result* prev) {
  if (!ctx) return MILZMA_INFRA_ERROR;
  const bool resume = (flags & MILZMA_DECODE_RESUME) != 0;
  const bool feed = (flags & MILZMA_DECODE_FEED) != 0;
  const bool grow = resume || feed || (flags & MILZMA_DECODE_GROW) != 0;
  if (feed && !(ctx->use_fast && ctx->fast_spill)) {
    ctx->err = "MILZMA_DECODE_FEED needs the asm kernel's launch classes (MILZMA_KERNEL / MILZMA_SPILL = generic set)";
    return MILZMA_INFRA_ERROR;
  }
  // (a RESUME | FEED call may also START units -- MILZMA_KIND_START: continuous batching -- and so may be the first call of its batch)
  bool starts = false;
  if (resume && feed && units)
    for (uint32_t i = 0; i < n && !starts; i++) starts = (units[i].kind & MILZMA_KIND_START) != 0;
  if (resume && starts && prev && (!ctx->parked_valid || ctx->parked_n != n) && !ctx->pending) {
    ctx->parked_valid = true;   // a batch of n places, nothing parked in it yet
    ctx->parked_n = n;
    ctx->park_rec.assign(n, milzma_ctx::ParkRec());
    ctx->slab_live = false;
  }
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
  ctx->pend_flags = (grow ? MILZMA_DECODE_GROW : 0u) | (resume ? MILZMA_DECODE_RESUME : 0u) | (feed ? MILZMA_DECODE_FEED : 0u);
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
    if ((u.kind & (feed ? 0x0Fu : 0xFFu)) == MILZMA_KIND_RAW_LZMA && u.dict_size < 0x1000u) u.dict_size = 0x1000u;
  // (fed input: MILZMA_KIND_LAST_VIEW and MILZMA_KIND_PARTIAL are for the kernel; everything on the host side looks at the plain kind)
  ctx->feed_units.clear();
  std::vector<uint8_t> marks;   // MILZMA_KIND_START / _HOLD of each unit (host side only: the device sees the kind and LAST_VIEW)
  if (feed) {
    marks.resize(n);
    for (uint32_t i = 0; i < n; i++) {
      marks[i] = ctx->pend_units[i].kind & (MILZMA_KIND_START | MILZMA_KIND_HOLD);
      ctx->pend_units[i].kind &= uint8_t(~(MILZMA_KIND_START | MILZMA_KIND_HOLD));
    }
    ctx->feed_units = ctx->pend_units;
    for (milzma_unit& u : ctx->pend_units) u.kind &= uint8_t(~(MILZMA_KIND_LAST_VIEW | MILZMA_KIND_PARTIAL));
  }
  const milzma_unit* const units_up = feed ? ctx->feed_units.data() : ctx->pend_units.data();
  units = ctx->pend_units.data();

  // Partition by launch class; inside a class longest input first, so that the hardware's
  // in-order block dispatch behaves like longest-processing-time-first scheduling.
  std::vector<uint32_t> order[kNumLitClasses];
  std::vector<uint32_t> fresh[kNumLitClasses];   // RESUME | FEED: units that START in this call (launched like a first call's, beside the resumed ones)
  if (resume) {  // only what the previous call parked, each unit in the class the CONTEXT knows it was parked in
    for (uint32_t i = 0; i < n; i++) {
      const uint8_t mark = feed ? marks[i] : 0;
      if (mark & MILZMA_KIND_START) {   // a new member of the batch: its place must be free
        const milzma_ctx::ParkRec* rec = i < ctx->park_rec.size() ? &ctx->park_rec[i] : nullptr;
        LitClass c = classify(ctx, units[i]);
        if (c == kFast && units[i].kind == MILZMA_KIND_LZMA2) c = kFastSpill;
        const char* why = (rec && rec->parked && is_parked_result(prev[i])) ? "is parked: it cannot start again"
                          : (c != kFast && c != kFastSpill)                 ? "is outside the asm kernel's launch classes"
                                                                            : nullptr;
        if (why) {
          ctx->err = "MILZMA_KIND_START: unit " + std::to_string(i) + " " + why;
          ctx->pending = false;
          return MILZMA_INFRA_ERROR;
        }
        fresh[c].push_back(i);
        continue;
      }
      if (mark & MILZMA_KIND_HOLD) continue;   // parked, and to stay so in this call (its result and its record are kept)
      if (is_parked_result(prev[i])) {
        // (a unit parked by a FEED call comes back with another view by design -- the caller vouches that it starts at the unit's first
        //  unused byte, there is nothing here to hold that against)
        const milzma_ctx::ParkRec* rec = i < ctx->park_rec.size() ? &ctx->park_rec[i] : nullptr;
        const bool same_view = rec && (rec->fed || (units[i].in_off == rec->in_off && units[i].in_len == rec->in_len));
        const char* why = !rec || !rec->parked                                                        ? "was not parked by the previous call"
                          : (prev[i].status == MILZMA_ST_NEED_INPUT) != (rec->input != 0)             ? "was parked for another reason than its result says"
                          : !same_view || units[i].kind != rec->kind                                  ? "names another input than the one it was parked with"
                          : units[i].out_cap < rec->out_len                                           ? "has a slice smaller than the output it has produced"
                                                                                                      : nullptr;
        if (why) {  // (nothing is launched: a descriptor that does not fit the parked state would write outside its slice)
          ctx->err = "MILZMA_DECODE_RESUME: unit " + std::to_string(i) + " " + why;
          ctx->pending = false;
          return MILZMA_INFRA_ERROR;
        }
        order[rec->spill ? kFastSpill : kFast].push_back(i);
      }
    }
  } else {
    // (fed LZMA2 units: with a slab from the start -- a property switch beyond lc + lp = 3 cannot send them back to a start that has gone)
    for (uint32_t i = 0; i < n; i++) {
      LitClass c = classify(ctx, units[i]);
      if (feed && c == kFast && units[i].kind == MILZMA_KIND_LZMA2) c = kFastSpill;
      order[c].push_back(i);
    }
    if (feed)
      for (int c = 0; c < kNumLitClasses; c++)
        if (c != kFast && c != kFastSpill && !order[c].empty()) {
          ctx->err = "MILZMA_DECODE_FEED: unit " + std::to_string(order[c][0]) + " is outside the asm kernel's launch classes";
          ctx->pending = false;
          return MILZMA_INFRA_ERROR;
        }
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
  const size_t n_resumed = flat.size();   // (what follows in `flat` starts fresh)
  uint32_t fbase[kNumLitClasses];
  for (int c = 0; c < kNumLitClasses; c++) {
    std::stable_sort(fresh[c].begin(), fresh[c].end(), [&](uint32_t a, uint32_t b) { return units[a].in_len > units[b].in_len; });
    fbase[c] = uint32_t(flat.size());
    flat.insert(flat.end(), fresh[c].begin(), fresh[c].end());
  }

  if (!dev_reserve(ctx, ctx->units, size_t(n) * sizeof(milzma_unit)) ||
      !dev_reserve(ctx, ctx->order, size_t(n) * 2 * sizeof(uint32_t)) ||
      !dev_reserve(ctx, ctx->results, size_t(n) * sizeof(milzma_result)) || !dev_reserve(ctx, ctx->flags, 64 * sizeof(uint32_t)) ||
      !pin_reserve(ctx, ctx->pin_results, size_t(n) * (sizeof(milzma_result) + sizeof(uint32_t))))
    return fail();
  // (the order array is staged in page-locked memory behind the results so that its upload is asynchronous too)
  uint32_t* h_order = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(ctx->pin_results.p) + size_t(n) * sizeof(milzma_result));
  for (size_t k = 0; k < flat.size(); k++) h_order[k] = flat[k] | (resume && k < n_resumed ? 0x80000000u : 0u);  // (bit 31: resume from the parked state)
  if (!hip_ok(ctx, hipMemcpyAsync(ctx->units.p, units_up, size_t(n) * sizeof(milzma_unit), hipMemcpyHostToDevice, stream),
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
    if (!launch_class(ctx, LitClass(c), order[c], base[c], ctx->pend_in, ctx->pend_out, stream, grow, resume, feed) ||
        !launch_class(ctx, LitClass(c), fresh[c], fbase[c], ctx->pend_in, ctx->pend_out, stream, grow, false, feed))
      return fail();

  // (The results are fetched by the wait half, after the kernels: a copy queued behind a running kernel parks a DMA queue on
  //  that kernel's completion, and an unrelated upload of another context that lands on the same queue then waits for the whole
  //  kernel -- measured: 200 ms per grouped call, profiles/r03_batch_api.txt.)
  return MILZMA_OK;
}

// Wait: drain the stream, time the launches, rerun promoted LZMA2 units, hand the results over.
int milzma_decode_units_wait_impl(milzma_ctx* ctx, milzma_result* results) {
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
    if (!launch_class(ctx, next, again, n, ctx->pend_in, ctx->pend_out, stream, (ctx->pend_flags & MILZMA_DECODE_GROW) != 0, false,
                      (ctx->pend_flags & MILZMA_DECODE_FEED) != 0))
      return bail();
    if (!hip_ok(ctx,
                hipMemcpyAsync(results, ctx->results.p, size_t(n) * sizeof(milzma_result), hipMemcpyDeviceToHost, stream),
                "D2H results") ||
        !hip_ok(ctx, hipStreamSynchronize(stream), "hipStreamSynchronize") || !collect_kernel_ms(ctx))
      return bail();
  }
  if (ctx->pend_flags & MILZMA_DECODE_GROW) {
    bool any = false;
    for (uint32_t i = 0; i < n && !any; i++) any = is_parked_result(results[i]);
    ctx->parked_valid = any;
    ctx->parked_n = n;
    if (!any) ctx->slab_live = false;
    ctx->park_rec.assign(any ? n : 0, milzma_ctx::ParkRec());
    for (uint32_t i = 0; any && i < n; i++)
      if (is_parked_result(results[i])) {
        milzma_ctx::ParkRec& r = ctx->park_rec[i];
        r.parked = 1;
        r.input = results[i].status == MILZMA_ST_NEED_INPUT;
        r.fed = (ctx->pend_flags & MILZMA_DECODE_FEED) != 0;
        r.spill = (results[i].err_b & 0x100) ? 1 : 0;
        r.kind = uint8_t(ctx->pend_units[i].kind);
        r.in_off = ctx->pend_units[i].in_off;
        r.in_len = ctx->pend_units[i].in_len;
        r.out_len = results[i].out_len;
      }
  }
  return MILZMA_OK;
}

int milzma_decode_units_impl(milzma_ctx* ctx, const milzma_unit* units, uint32_t n, const void* d_in,
                                   void* d_out, milzma_result* results, void* hip_stream, uint32_t flags) {
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

int milzma_decode_units_host_impl(milzma_ctx* ctx, const milzma_unit* units, uint32_t n, const void* h_in,
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

MILZMA_HOST_NS_BEGIN

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

MILZMA_HOST_NS_END

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
    case MILZMA_ST_NEED_INPUT: return render(msg, cap, MILZMA_INFRA_ERROR, "unit parked at the end of its input view");
    default: return render(msg, cap, MILZMA_INFRA_ERROR, "unknown status %u", r->status);
  }
}

// ------------------------------------------------------------------------------------------
// CRC-32 (ISO-HDLC) / CRC-64 (XZ), slicing-by-8 (src/xz/crc.rs:1-4 names the polynomials)
// ------------------------------------------------------------------------------------------

MILZMA_HOST_NS_BEGIN

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

MILZMA_HOST_NS_END

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
MILZMA_HOST_NS_BEGIN

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

MILZMA_HOST_NS_END

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
int move_units_impl(milzma_ctx* ctx, uint32_t n, const void* d_src, const uint64_t* src_off, void* d_dst, const uint64_t* dst_off,
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
  PinLease lease;   // (the move list lives in pin_moves until the copy below has drained)
  if (!lease.take(ctx->pin_moves, "milzma_move_units")) {
    ctx->err = "the context's move-list buffer is in use by another call";
    return MILZMA_INFRA_ERROR;
  }
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

