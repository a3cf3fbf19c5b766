// tests/san/fake_hip.cpp -- TEST INFRASTRUCTURE.  A HIP runtime made of host memory and host threads, for the sanitizer builds of the
// library's host side (tests/san/Makefile: pipeline_asan / pipeline_tsan).  With it -- and tests/san/fake_kernels.cpp, where the CPU oracle
// stands in for the decode kernels -- the WHOLE of lzma_rs_amd/csrc/host.cpp runs without a GPU: staging, pools, streamed launches with their
// consumer threads, park / regrow / resume rounds, lanes, asynchronous calls, the multi-device entry points.  What is checked there is the
// host logic under AddressSanitizer / UndefinedBehaviorSanitizer / ThreadSanitizer; parity of the real kernels is the GPU suite's business.
//
// Model: "device memory" is malloc'd host memory (so ASan sees every out-of-bounds copy), page-locked memory likewise, a device pointer
// to mapped host memory is the host pointer.  A stream is an in-order chain of tasks, each on a thread of its own (std::async), so that
// work on different streams really overlaps and ThreadSanitizer sees the library's synchronisation, not the fake's.  Events are
// tasks that take the time.  FAKE_HIP_DEVICES (default 1) devices, all alike, peer access everywhere.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace {

struct FakeStream {
  std::mutex mu;
  std::shared_future<void> tail;
};
struct FakeEvent {
  std::mutex mu;
  std::shared_future<void> done;
  std::chrono::steady_clock::time_point at;
};

FakeStream g_null_stream;
std::mutex g_streams_mu;
std::map<FakeStream*, std::shared_ptr<FakeStream>> g_streams;   // (shared: a device-wide drain may still hold a stream another thread destroys)
thread_local int t_device = 0;

FakeStream* S(hipStream_t s) { return s ? reinterpret_cast<FakeStream*>(s) : &g_null_stream; }

void enqueue(FakeStream* s, std::function<void()> fn) {
  std::lock_guard<std::mutex> lock(s->mu);
  std::shared_future<void> prev = s->tail;
  s->tail = std::async(std::launch::async, [prev, fn = std::move(fn)] {
              if (prev.valid()) prev.wait();
              fn();
            }).share();
}

void drain(FakeStream* s) {
  std::shared_future<void> t;
  {
    std::lock_guard<std::mutex> lock(s->mu);
    t = s->tail;
  }
  if (t.valid()) t.wait();
}

void drain_all() {
  std::vector<std::shared_ptr<FakeStream>> all;
  {
    std::lock_guard<std::mutex> lock(g_streams_mu);
    for (auto& kv : g_streams) all.push_back(kv.second);
  }
  for (auto& s : all) drain(s.get());
  drain(&g_null_stream);
}

// Fault injection (pipeline_fuzz.cpp, PIPELINE_FAULTS): the n-th fallible runtime call from now on fails.
std::atomic<long> g_calls{0}, g_fail_at{-1};
bool fails_now() {
  const long k = g_calls.fetch_add(1) + 1;
  return k == g_fail_at.load();
}

int device_count() {
  const char* e = getenv("FAKE_HIP_DEVICES");
  const int n = e ? atoi(e) : 1;
  return n < 1 ? 1 : n > 8 ? 8 : n;
}

}  // namespace

// (tests/san/fake_kernels.cpp queues the stand-in kernels through this)
void fake_hip_enqueue(hipStream_t stream, std::function<void()> fn) { enqueue(S(stream), std::move(fn)); }
void fake_hip_fail_at(long n) {
  g_calls.store(0);
  g_fail_at.store(n);
}
long fake_hip_calls() { return g_calls.load(); }
bool fake_hip_launch_fails() { return fails_now(); }

extern "C" {

hipError_t hipGetDeviceCount(int* count) {
  *count = device_count();
  return hipSuccess;
}
hipError_t hipSetDevice(int device) {
  if (device < 0 || device >= device_count()) return hipErrorInvalidDevice;
  t_device = device;
  return hipSuccess;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t* prop, int device) {
  if (device < 0 || device >= device_count()) return hipErrorInvalidDevice;
  memset(prop, 0, sizeof *prop);
  strcpy(prop->name, "fake MI355X");
  strcpy(prop->gcnArchName, "gfx950:sramecc+:xnack-");
  prop->multiProcessorCount = 256;
  prop->totalGlobalMem = size_t(16) << 30;
  return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "fake HIP error"; }

hipError_t hipMalloc(void** p, size_t n) {
  *p = nullptr;
  if (fails_now()) return hipErrorOutOfMemory;
  *p = malloc(n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) {
  drain_all();  // (hipFree synchronises the device)
  free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) {
  *p = nullptr;
  if (fails_now()) return hipErrorOutOfMemory;
  *p = malloc(n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void* p) {
  drain_all();
  free(p);
  return hipSuccess;
}
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned int) {
  *dev = host;
  return hipSuccess;
}
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) {
  *free_b = size_t(12) << 30;
  *total_b = size_t(16) << 30;
  return hipSuccess;
}

hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind) {
  if (fails_now()) return hipErrorUnknown;
  drain(&g_null_stream);
  if (n) memmove(dst, src, n);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t stream) {
  if (fails_now()) return hipErrorUnknown;
  enqueue(S(stream), [=] {
    if (n) memmove(dst, src, n);
  });
  return hipSuccess;
}
hipError_t hipMemcpyPeer(void* dst, int, const void* src, int, size_t n) {
  if (fails_now()) return hipErrorUnknown;
  drain(&g_null_stream);
  if (n) memmove(dst, src, n);
  return hipSuccess;
}
hipError_t hipMemsetD16Async(hipDeviceptr_t dst, unsigned short v, size_t count, hipStream_t stream) {
  enqueue(S(stream), [=] {
    unsigned short* p = static_cast<unsigned short*>(dst);
    for (size_t i = 0; i < count; i++) p[i] = v;
  });
  return hipSuccess;
}

hipError_t hipStreamCreateWithFlags(hipStream_t* out, unsigned int) {
  if (fails_now()) return hipErrorOutOfMemory;
  auto sp = std::make_shared<FakeStream>();
  FakeStream* s = sp.get();
  {
    std::lock_guard<std::mutex> lock(g_streams_mu);
    g_streams[s] = std::move(sp);
  }
  *out = reinterpret_cast<hipStream_t>(s);
  return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t* out) { return hipStreamCreateWithFlags(out, 0); }
hipError_t hipStreamDestroy(hipStream_t stream) {
  if (!stream) return hipSuccess;
  FakeStream* s = S(stream);
  drain(s);
  std::shared_ptr<FakeStream> last;
  {
    std::lock_guard<std::mutex> lock(g_streams_mu);
    auto it = g_streams.find(s);
    if (it != g_streams.end()) {
      last = std::move(it->second);
      g_streams.erase(it);
    }
  }
  return hipSuccess;   // (`last` goes here, or with the last device-wide drain that holds it)
}
hipError_t hipStreamSynchronize(hipStream_t stream) {
  drain(S(stream));
  return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) {
  drain_all();
  return hipSuccess;
}

hipError_t hipEventCreateWithFlags(hipEvent_t* out, unsigned) {
  if (fails_now()) return hipErrorOutOfMemory;
  *out = reinterpret_cast<hipEvent_t>(new FakeEvent());
  return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* out) { return hipEventCreateWithFlags(out, 0); }
hipError_t hipEventDestroy(hipEvent_t e) {
  FakeEvent* ev = reinterpret_cast<FakeEvent*>(e);
  std::shared_future<void> d;
  {
    std::lock_guard<std::mutex> lock(ev->mu);
    d = ev->done;
  }
  if (d.valid()) d.wait();
  delete ev;
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t stream) {
  FakeEvent* ev = reinterpret_cast<FakeEvent*>(e);
  FakeStream* s = S(stream);
  std::lock_guard<std::mutex> lock(s->mu);
  std::shared_future<void> prev = s->tail;
  s->tail = std::async(std::launch::async, [prev, ev] {
              if (prev.valid()) prev.wait();
              std::lock_guard<std::mutex> l2(ev->mu);
              ev->at = std::chrono::steady_clock::now();
            }).share();
  std::lock_guard<std::mutex> l3(ev->mu);
  ev->done = s->tail;
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
  FakeEvent* ev = reinterpret_cast<FakeEvent*>(e);
  std::shared_future<void> d;
  {
    std::lock_guard<std::mutex> lock(ev->mu);
    d = ev->done;
  }
  if (d.valid()) d.wait();
  return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  hipEventSynchronize(a);
  hipEventSynchronize(b);
  FakeEvent *ea = reinterpret_cast<FakeEvent*>(a), *eb = reinterpret_cast<FakeEvent*>(b);
  std::chrono::steady_clock::time_point ta, tb;
  {
    std::lock_guard<std::mutex> lock(ea->mu);
    ta = ea->at;
  }
  {
    std::lock_guard<std::mutex> lock(eb->mu);
    tb = eb->at;
  }
  *ms = std::chrono::duration<float, std::milli>(tb - ta).count();
  return hipSuccess;
}

hipError_t hipDeviceCanAccessPeer(int* can, int, int) {
  *can = 1;
  return hipSuccess;
}
hipError_t hipDeviceEnablePeerAccess(int, unsigned int) { return hipSuccess; }

}  // extern "C"
