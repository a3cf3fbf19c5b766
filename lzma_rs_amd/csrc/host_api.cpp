// host_api.cpp -- the exported entry points' wrappers, grouped calls, the asynchronous halves (see host_internal.h)
#include "host_internal.h"

using namespace milzma;
using namespace milzma::host;

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
    if (ctx && (flags & ~(MILZMA_DECODE_GROW | MILZMA_DECODE_RESUME | MILZMA_DECODE_FEED))) {
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
MILZMA_HOST_NS_BEGIN

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

MILZMA_HOST_NS_END

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
MILZMA_HOST_NS_BEGIN

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

MILZMA_HOST_NS_END

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



// ---- push-mode streams (host_stream.cpp) -----------------------------------------------------------------------------------------
MILZMA_HIDDEN int milzma_streams_open_impl(milzma_ctx* ctx, uint32_t kind, uint32_t n, const milzma_options* options, milzma_streams** out);
MILZMA_HIDDEN int milzma_streams_write_impl(milzma_streams* S, uint32_t k, const uint32_t* idx, const void* const* data, const size_t* len, int32_t* status);
MILZMA_HIDDEN int milzma_streams_finish_impl(milzma_streams* S, milzma_output* outs);
MILZMA_HIDDEN void milzma_streams_close_impl(milzma_streams* S);
MILZMA_HIDDEN const char* milzma_streams_write_error_impl(const milzma_streams* S, uint32_t stream);
MILZMA_HIDDEN const char* milzma_streams_last_error_impl(const milzma_streams* S);
MILZMA_HIDDEN uint64_t milzma_streams_write_taken_impl(const milzma_streams* S, uint32_t stream);
MILZMA_HIDDEN int milzma_streams_output_impl(milzma_streams* S, uint32_t stream, uint64_t offset, void* dst, size_t cap, uint64_t* sink_len,
                                            int32_t* has_sink);

extern "C" int milzma_streams_open(milzma_ctx* ctx, uint32_t kind, uint32_t n, const milzma_options* options, milzma_streams** out) {
  begin_call(ctx);
  try {
    return milzma_streams_open_impl(ctx, kind, n, options, out);
  } catch (const std::exception& e) {
    if (ctx) ctx->err = std::string("host exception: ") + e.what();
    return MILZMA_INFRA_ERROR;
  }
}

extern "C" int milzma_streams_write(milzma_streams* s, uint32_t k, const uint32_t* idx, const void* const* data, const size_t* len, int32_t* status) {
  try {
    return milzma_streams_write_impl(s, k, idx, data, len, status);
  } catch (const std::exception&) {  // (std::bad_alloc while buffering: nothing was launched with half-built descriptors -- the round builds them first)
    if (status)
      for (uint32_t j = 0; j < k; j++) status[j] = MILZMA_INFRA_ERROR;
    return MILZMA_INFRA_ERROR;
  }
}

extern "C" int milzma_streams_finish(milzma_streams* s, milzma_output* outs) {
  try {
    return milzma_streams_finish_impl(s, outs);
  } catch (const std::exception&) {
    return MILZMA_INFRA_ERROR;
  }
}

extern "C" void milzma_streams_close(milzma_streams* s) { milzma_streams_close_impl(s); }
extern "C" const char* milzma_streams_write_error(const milzma_streams* s, uint32_t stream) { return milzma_streams_write_error_impl(s, stream); }
extern "C" const char* milzma_streams_last_error(const milzma_streams* s) { return milzma_streams_last_error_impl(s); }

extern "C" int milzma_streams_output(milzma_streams* s, uint32_t stream, uint64_t offset, void* dst, size_t cap, uint64_t* sink_len, int32_t* has_sink) {
  try {
    return milzma_streams_output_impl(s, stream, offset, dst, cap, sink_len, has_sink);
  } catch (const std::exception&) {
    return MILZMA_INFRA_ERROR;
  }
}
extern "C" uint64_t milzma_streams_write_taken(const milzma_streams* s, uint32_t stream) { return milzma_streams_write_taken_impl(s, stream); }
