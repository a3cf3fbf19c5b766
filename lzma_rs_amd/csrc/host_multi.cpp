// host_multi.cpp -- the GPUs of one node behind one handle (see host_internal.h)
#include "host_internal.h"

using namespace milzma;
using namespace milzma::host;

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

MILZMA_HOST_NS_BEGIN

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

MILZMA_HOST_NS_END

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

