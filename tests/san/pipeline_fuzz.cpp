// tests/san/pipeline_fuzz.cpp -- TEST INFRASTRUCTURE (built by tests/san/Makefile with fake_hip.cpp + fake_kernels.cpp, run by
// tests/test_host_pipeline_sanitized.py).  Drives the library's GPU entry points -- whole-file batches (.lzma, LZMA2, .xz), single files, the
// asynchronous halves on two contexts, large calls cut into groups, the multi-device calls incl. the one-ingest-point entry -- through the
// fake HIP runtime, under AddressSanitizer + UndefinedBehaviorSanitizer or ThreadSanitizer, and compares every result with the oracle where
// the stand-in kernels report a faithful status (streams that decode, and truncated ones).
//
//   pipeline_fuzz <dir with *.lzma / *.lzma2 / *.xz> <rounds> <rng seed>
//
// Which paths a run takes is a matter of the environment (MILZMA_STREAM_MIN, MILZMA_PINNED_OUT, MILZMA_TWO_PART, MILZMA_STREAM,
// FAKE_HIP_DEVICES, ...): the test runs the matrix.
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#include <sanitizer/lsan_interface.h>
#endif
#endif
#include <dirent.h>

#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "lzma_oracle.h"
#include "milzma.h"

namespace milzma { namespace host { size_t out_live_buffers(); size_t out_pooled_buffers(); } }   // (lzma_rs_amd/csrc/host.cpp, linked in: result buffers callers hold right now)

void fake_hip_fail_at(long n);   // tests/san/fake_hip.cpp: the n-th fallible runtime call from now on fails (-1: none)
long fake_hip_calls();

namespace {

struct Rng {
  uint64_t s;
  uint32_t next() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return uint32_t(s >> 16);
  }
  uint32_t below(uint32_t n) { return n ? next() % n : 0; }
};

enum Kind { LZMA, LZMA2, XZ };
struct Case {
  std::string name;
  Kind kind;
  std::vector<uint8_t> data;
};

uint64_t g_cases = 0, g_compared = 0, g_skipped = 0;

const uint8_t* ptr_of(const std::vector<uint8_t>& d) { return d.empty() ? reinterpret_cast<const uint8_t*>("") : d.data(); }

// the oracle's verdict on one file; faithful: the stand-in kernels report this kind of outcome as the real ones would
struct Want {
  orc_result r;
  bool faithful;
};
Want oracle_of(const Case& c) {
  Want w;
  memset(&w.r, 0, sizeof w.r);
  const uint8_t* p = ptr_of(c.data);
  if (c.kind == LZMA)
    orc_lzma_decompress(p, c.data.size(), nullptr, &w.r);
  else if (c.kind == LZMA2)
    orc_lzma2_decompress(p, c.data.size(), &w.r);
  else
    orc_xz_decompress(p, c.data.size(), &w.r);
  w.faithful = true;   // (the stand-in kernels report every error site with the real kernels' status: fake_kernels.cpp)
  return w;
}

bool check(const Case& c, const milzma_output& o, const char* via) {
  g_cases++;
  Want w = oracle_of(c);
  bool ok = true;
  if (w.faithful) {
    g_compared++;
    ok = o.kind == w.r.kind && strcmp(o.msg, w.r.msg) == 0 && o.len == w.r.out_len && (o.len == 0 || memcmp(o.data, w.r.out, o.len) == 0) &&
         o.in_consumed == w.r.in_consumed;
    if (!ok)
      printf("MISMATCH %s via %s (%zu bytes): got kind %d '%s' len %zu consumed %zu | oracle kind %d '%s' len %zu consumed %zu\n", c.name.c_str(), via,
             c.data.size(), o.kind, o.msg, o.len, o.in_consumed, w.r.kind, w.r.msg, w.r.out_len, w.r.in_consumed);
  } else {
    g_skipped++;
    ok = o.kind != MILZMA_OK && o.kind != MILZMA_INFRA_ERROR;   // (some reference error; which one is the real kernels' business)
    if (!ok) printf("MISMATCH %s via %s: oracle fails with '%s', the library says kind %d '%s'\n", c.name.c_str(), via, w.r.msg, o.kind, o.msg);
  }
  orc_free(w.r.out);
  return ok;
}

typedef int (*BatchFn)(milzma_ctx*, uint32_t, const uint8_t* const*, const size_t*, milzma_output*);
int lzma_batch(milzma_ctx* c, uint32_t n, const uint8_t* const* i, const size_t* l, milzma_output* o) { return milzma_lzma_decompress_batch(c, n, i, l, nullptr, o); }
int lzma_batch_async(milzma_ctx* c, uint32_t n, const uint8_t* const* i, const size_t* l, milzma_output* o) {
  return milzma_lzma_decompress_batch_async(c, n, i, l, nullptr, o);
}
typedef int (*MultiFn)(milzma_multi*, uint32_t, const uint8_t* const*, const size_t*, milzma_output*);
int multi_lzma(milzma_multi* m, uint32_t n, const uint8_t* const* i, const size_t* l, milzma_output* o) {
  return milzma_multi_lzma_decompress_batch(m, n, i, l, nullptr, o);
}

struct Batch {
  std::vector<const Case*> cases;
  std::vector<const uint8_t*> ins;
  std::vector<size_t> lens;
  std::vector<milzma_output> outs;
  void add(const Case* c) {
    cases.push_back(c);
    ins.push_back(ptr_of(c->data));
    lens.push_back(c->data.size());
  }
  void prepare() {
    outs.assign(cases.size(), milzma_output());
    for (auto& o : outs) memset(&o, 0, sizeof o);
  }
  bool verify(const char* via) {
    bool ok = true;
    for (size_t i = 0; i < cases.size(); i++) {
      ok = check(*cases[i], outs[i], via) && ok;
      milzma_free(outs[i].data);
    }
    return ok;
  }
};

bool run_batch(milzma_ctx* ctx, Kind kind, Batch& b, const char* via) {
  b.prepare();
  const uint32_t n = uint32_t(b.cases.size());
  int rc;
  if (kind == LZMA)
    rc = lzma_batch(ctx, n, b.ins.data(), b.lens.data(), b.outs.data());
  else if (kind == LZMA2)
    rc = milzma_lzma2_decompress_batch(ctx, n, b.ins.data(), b.lens.data(), b.outs.data());
  else
    rc = milzma_xz_decompress_batch(ctx, n, b.ins.data(), b.lens.data(), b.outs.data());
  if (rc == MILZMA_INFRA_ERROR) {
    printf("INFRA %s: %s\n", via, milzma_last_error(ctx));
    return false;
  }
  return b.verify(via);
}


// 3c (round 5).  RAW .lzma units of MIXED literal-row classes (lc + lp <= 3: no slab; 4 .. 12: rows in the slab, one stride per batch) through a
// caller's own GROW / move / RESUME loop from tiny slices, the known-size ones in slices that fit (they finish in the first call while the
// others park): the resumed subset then no longer holds the unit with the largest lc + lp, and the stride, the allocation and the other
// units' rows must stay what the first launch made them (the stride used to be re-derived from the resumed subset; a promotion
// launch used to wipe the whole slab).  One LZMA2 unit whose first chunk asks for lc + lp = 4 rides along: it is promoted into the slab's
// class by a launch of its own in the middle of the batch.  Also: a RESUME whose descriptors do not fit the parked states is refused.
bool raw_grow_loop(milzma_ctx* ctx, const std::vector<Case>& lzma_pool, const std::vector<Case>& lzma2_pool) {
  struct Item {
    const Case* c;
    milzma_unit u;
    size_t hdr;
    orc_result want;
  };
  std::vector<Item> items;
  bool have_big = false, have_small_unknown = false;
  for (const Case& c : lzma_pool) {
    if (items.size() >= 48) break;
    Item it;
    it.c = &c;
    memset(&it.u, 0, sizeof it.u);
    memset(&it.want, 0, sizeof it.want);
    size_t hl = 0;
    milzma_output ho;
    memset(&ho, 0, sizeof ho);
    if (milzma_lzma_read_header(ptr_of(c.data), c.data.size(), nullptr, &it.u, &hl, &ho) != MILZMA_OK) continue;
    orc_lzma_decompress(ptr_of(c.data), c.data.size(), nullptr, &it.want);
    if (it.want.kind != ORC_OK || it.want.out_len < 2000) {
      orc_free(it.want.out);
      continue;
    }
    it.hdr = hl;
    const uint32_t lclp = uint32_t(it.u.lc) + it.u.lp;
    have_big = have_big || lclp > 4;
    have_small_unknown = have_small_unknown || (lclp == 4 && it.u.unpacked_size == MILZMA_SIZE_UNKNOWN);
    items.push_back(it);
  }
  for (const Case& c : lzma2_pool) {   // one LZMA2 unit that will be promoted (lc + lp = 4 in its first chunk) and is long enough to park
    Item it;
    it.c = &c;
    memset(&it.u, 0, sizeof it.u);
    memset(&it.want, 0, sizeof it.want);
    if (c.name.find("22.lzma2") == std::string::npos || c.name.find('(') != std::string::npos) continue;
    orc_lzma2_decompress(ptr_of(c.data), c.data.size(), &it.want);
    if (it.want.kind != ORC_OK || it.want.out_len < 2000) {
      orc_free(it.want.out);
      continue;
    }
    it.u.kind = MILZMA_KIND_LZMA2;
    it.hdr = 0;
    items.push_back(it);
    break;
  }
  bool ok = true;
  if (!items.empty()) {
    const uint32_t n = uint32_t(items.size());
    std::vector<milzma_unit> units(n);
    size_t io = 0, oo = 0;
    for (uint32_t i = 0; i < n; i++) {
      milzma_unit& u = items[i].u;
      u.in_off = io;
      u.in_len = items[i].c->data.size() - items[i].hdr;
      io += (size_t(u.in_len) + 255) & ~size_t(255);
      // the units with the most literal rows get room at once (they finish in the first call), everything else starts far too small
      const bool roomy = u.kind == MILZMA_KIND_RAW_LZMA && uint32_t(u.lc) + u.lp > 4;
      u.out_off = oo;
      u.out_cap = roomy ? items[i].want.out_len + 300 : 700;
      oo += (size_t(u.out_cap) + 255) & ~size_t(255);
      units[i] = u;
    }
    std::vector<uint8_t> in(io + 512, 0), out(oo + 512, 0);
    for (uint32_t i = 0; i < n; i++) memcpy(in.data() + units[i].in_off, ptr_of(items[i].c->data) + items[i].hdr, size_t(units[i].in_len));
    std::vector<milzma_result> res(n);
    uint32_t flags = MILZMA_DECODE_GROW;
    int rounds_done = 0;
    bool refused_checked = false;
    for (;; rounds_done++) {
      if (milzma_decode_units_ex(ctx, units.data(), n, in.data(), out.data(), res.data(), nullptr, flags) != MILZMA_OK) {
        printf("INFRA raw grow loop, round %d: %s\n", rounds_done, milzma_last_error(ctx));
        ok = false;
        break;
      }
      std::vector<uint64_t> so, dof, ln;
      std::vector<milzma_unit> next = units;
      size_t total = 0;
      bool any = false;
      for (uint32_t i = 0; i < n; i++) {
        const bool parked = res[i].status == MILZMA_ST_OUT_FULL && res[i].err_a == MILZMA_PARKED;
        any = any || parked;
        next[i].out_off = total;
        next[i].out_cap = parked ? units[i].out_cap * 4 : units[i].out_cap;
        so.push_back(units[i].out_off);
        dof.push_back(total);
        ln.push_back(res[i].out_len < units[i].out_cap ? res[i].out_len : units[i].out_cap);
        total += (size_t(next[i].out_cap) + 255) & ~size_t(255);
      }
      if (!any) break;
      if (!refused_checked) {   // a RESUME that names a slice smaller than what a parked unit has produced is refused, nothing launched
        refused_checked = true;
        std::vector<milzma_unit> bad = units;
        std::vector<milzma_result> r2 = res;
        for (uint32_t i = 0; i < n; i++)
          if (res[i].status == MILZMA_ST_OUT_FULL && res[i].err_a == MILZMA_PARKED && res[i].out_len > 16) {
            bad[i].out_cap = res[i].out_len - 16;
            break;
          }
        if (milzma_decode_units_ex(ctx, bad.data(), n, in.data(), out.data(), r2.data(), nullptr, MILZMA_DECODE_RESUME) != MILZMA_INFRA_ERROR) {
          printf("MISMATCH a RESUME with a slice below the parked unit's output was not refused\n");
          ok = false;
          break;
        }
      }
      std::vector<uint8_t> bigger(total + 512, 0);
      if (milzma_move_units(ctx, n, out.data(), so.data(), bigger.data(), dof.data(), ln.data(), nullptr) != MILZMA_OK) {
        printf("INFRA move_units (raw grow loop): %s\n", milzma_last_error(ctx));
        ok = false;
        break;
      }
      out.swap(bigger);
      units = next;
      flags = MILZMA_DECODE_RESUME;
    }
    for (uint32_t i = 0; i < n && ok; i++) {
      g_cases++;
      g_compared++;
      const orc_result& w = items[i].want;
      if (res[i].status != MILZMA_ST_OK || res[i].out_len != w.out_len || memcmp(out.data() + units[i].out_off, w.out, w.out_len) != 0) {
        printf("MISMATCH raw unit %u (%s, lc %u lp %u) after %d grow rounds: status %u len %" PRIu64 " (want %zu)\n", i, items[i].c->name.c_str(),
               items[i].u.lc, items[i].u.lp, rounds_done, res[i].status, res[i].out_len, w.out_len);
        ok = false;
      }
    }
    if (ok && have_big && have_small_unknown && rounds_done < 2) {
      printf("MISMATCH the raw grow loop never resumed anything (%d rounds)\n", rounds_done);
      ok = false;
    }
  }
  for (Item& it : items) orc_free(it.want.out);
  return ok;
}

// 3d. fed input (MILZMA_DECODE_FEED): good .lzma payloads and LZMA2 streams arrive in three pieces, every view re-located into a fresh input
// buffer from the unit's first unused byte; some units start in slices that are too small as well (both parking reasons in one batch).
// The stand-in kernels keep what a unit has consumed as bytes (fake_kernels.cpp); what is under test is the host side: flags, descriptors as
// uploaded, previous results before a resuming launch, the parked-unit record -- and its refusals.
bool raw_feed_loop(milzma_ctx* ctx, const std::vector<Case>& lzma_pool, const std::vector<Case>& lzma2_pool) {
  struct Item {
    const Case* c;
    milzma_unit u;
    size_t hdr;
    orc_result want;
    uint64_t used = 0;
  };
  std::vector<Item> items;
  for (const Case& c : lzma_pool) {
    if (items.size() >= 24) break;
    Item it;
    it.c = &c;
    memset(&it.u, 0, sizeof it.u);
    memset(&it.want, 0, sizeof it.want);
    size_t hl = 0;
    milzma_output ho;
    memset(&ho, 0, sizeof ho);
    if (milzma_lzma_read_header(ptr_of(c.data), c.data.size(), nullptr, &it.u, &hl, &ho) != MILZMA_OK) continue;
    orc_lzma_decompress(ptr_of(c.data), c.data.size(), nullptr, &it.want);
    if (it.want.kind != ORC_OK || c.data.size() - hl < 200) {
      orc_free(it.want.out);
      continue;
    }
    it.hdr = hl;
    items.push_back(it);
  }
  for (const Case& c : lzma2_pool) {
    if (items.size() >= 30) break;
    Item it;
    it.c = &c;
    memset(&it.u, 0, sizeof it.u);
    memset(&it.want, 0, sizeof it.want);
    if (c.name.find('(') != std::string::npos) continue;
    orc_lzma2_decompress(ptr_of(c.data), c.data.size(), &it.want);
    if (it.want.kind != ORC_OK || c.data.size() < 200) {
      orc_free(it.want.out);
      continue;
    }
    it.u.kind = MILZMA_KIND_LZMA2;
    it.hdr = 0;
    items.push_back(it);
  }
  bool ok = true;
  const uint32_t n = uint32_t(items.size());
  if (n) {
    std::vector<milzma_unit> units(n);
    std::vector<milzma_result> res(n);
    size_t oo = 0;
    for (uint32_t i = 0; i < n; i++) {
      units[i] = items[i].u;
      units[i].out_off = oo;
      units[i].out_cap = i % 4 == 3 ? 500 : items[i].want.out_len + 300;   // (every fourth also parks for room once the oracle sees its last view)
      oo += (size_t(units[i].out_cap) + 255) & ~size_t(255);
    }
    std::vector<uint8_t> out(oo + 512, 0), in;
    uint32_t input_parks = 0, room_parks = 0;
    bool refused_checked = false;
    for (int round = 0; ok && round < 12; round++) {
      // the views of this round: a third / two thirds / all of every stream (later rounds: all), from the first unused byte, at a new place
      size_t io = size_t(round) * 7 + 3;
      in.assign(in.size(), 0);
      std::vector<uint8_t> fresh;
      bool any = false;
      for (uint32_t i = 0; i < n; i++) {
        const bool parked = round == 0 || ((res[i].status == MILZMA_ST_NEED_INPUT || res[i].status == MILZMA_ST_OUT_FULL) && res[i].err_a == MILZMA_PARKED);
        if (!parked) {
          units[i].kind = items[i].u.kind;
          continue;
        }
        any = true;
        const size_t len = items[i].c->data.size() - items[i].hdr;
        const size_t upto = round == 0 ? len / 3 : round == 1 ? len * 2 / 3 : len;
        const size_t from = size_t(items[i].used);
        units[i].in_off = io;
        units[i].in_len = upto > from ? upto - from : 0;
        units[i].kind = uint8_t(items[i].u.kind | (upto == len ? MILZMA_KIND_LAST_VIEW : 0));
        fresh.resize(io + size_t(units[i].in_len) + 64, 0);
        if (units[i].in_len) memcpy(fresh.data() + io, ptr_of(items[i].c->data) + items[i].hdr + from, size_t(units[i].in_len));
        io += size_t(units[i].in_len) + 1 + (i % 5);
      }
      if (!any) break;
      fresh.resize(io + 512, 0);
      in.swap(fresh);
      if (round > 0 && !refused_checked) {   // a RESUME whose results claim another parking reason than the context recorded is refused
        refused_checked = true;
        std::vector<milzma_result> lie = res;
        for (uint32_t i = 0; i < n; i++)
          if (lie[i].status == MILZMA_ST_NEED_INPUT && lie[i].err_a == MILZMA_PARKED) {
            lie[i].status = MILZMA_ST_OUT_FULL;
            break;
          }
        if (milzma_decode_units_ex(ctx, units.data(), n, in.data(), out.data(), lie.data(), nullptr, MILZMA_DECODE_RESUME | MILZMA_DECODE_FEED) !=
            MILZMA_INFRA_ERROR) {
          printf("MISMATCH a RESUME that lies about a unit's parking reason was not refused\n");
          ok = false;
          break;
        }
      }
      if (milzma_decode_units_ex(ctx, units.data(), n, in.data(), out.data(), res.data(), nullptr,
                                 MILZMA_DECODE_FEED | (round ? MILZMA_DECODE_RESUME : 0u)) != MILZMA_OK) {
        printf("INFRA raw feed loop, round %d: %s\n", round, milzma_last_error(ctx));
        ok = false;
        break;
      }
      std::vector<uint64_t> so, dof, ln;
      std::vector<milzma_unit> next = units;
      size_t total = 0;
      bool room = false;
      for (uint32_t i = 0; i < n; i++) {
        const bool pin = res[i].status == MILZMA_ST_NEED_INPUT && res[i].err_a == MILZMA_PARKED;
        const bool prm = res[i].status == MILZMA_ST_OUT_FULL && res[i].err_a == MILZMA_PARKED;
        if ((pin || prm) && (units[i].kind & MILZMA_KIND_LAST_VIEW ? pin : false)) {
          printf("MISMATCH unit %u parked for input on its last view\n", i);
          ok = false;
        }
        if (pin || prm) items[i].used += res[i].in_consumed;
        input_parks += pin;
        room_parks += prm;
        room = room || prm;
        next[i].out_off = total;
        next[i].out_cap = prm ? items[i].want.out_len + 300 : units[i].out_cap;
        so.push_back(units[i].out_off);
        dof.push_back(total);
        ln.push_back(res[i].out_len < units[i].out_cap ? res[i].out_len : units[i].out_cap);
        total += (size_t(next[i].out_cap) + 255) & ~size_t(255);
      }
      if (room) {
        std::vector<uint8_t> bigger(total + 512, 0);
        if (milzma_move_units(ctx, n, out.data(), so.data(), bigger.data(), dof.data(), ln.data(), nullptr) != MILZMA_OK) {
          printf("INFRA move_units (raw feed loop): %s\n", milzma_last_error(ctx));
          ok = false;
          break;
        }
        out.swap(bigger);
        units = next;
      }
    }
    for (uint32_t i = 0; i < n && ok; i++) {
      g_cases++;
      g_compared++;
      const orc_result& w = items[i].want;
      const uint64_t reader = items[i].used + res[i].in_consumed + items[i].hdr;
      if (res[i].status != MILZMA_ST_OK || res[i].out_len != w.out_len || memcmp(out.data() + units[i].out_off, w.out, w.out_len) != 0 ||
          reader != w.in_consumed) {
        printf("MISMATCH fed unit %u (%s): status %u len %" PRIu64 " (want %zu) reader %" PRIu64 " (want %zu)\n", i, items[i].c->name.c_str(), res[i].status,
               res[i].out_len, w.out_len, reader, w.in_consumed);
        ok = false;
      }
    }
    if (ok && (input_parks < n || room_parks == 0)) {
      printf("MISMATCH the raw feed loop parked %u times for input, %u for room (%u units)\n", input_parks, room_parks, n);
      ok = false;
    }
  }
  for (Item& it : items) orc_free(it.want.out);
  return ok;
}

// 3e. the push-mode API (milzma_streams_*: lzma_rs::decompress::Stream for a batch): good .lzma files written in four pieces each, on
// schedules of their own -- some streams join calls later (MILZMA_KIND_START inside a resuming call), all of them sit calls out
// (MILZMA_KIND_HOLD) --, one with a bad header, one never completed; finish against the one-shot oracle.
// copies > 1: the same files several times over -- a batch of >= 64 streams, whose result buffers are filled by the (stand-in) kernels
// themselves while they "decode" (milzma_streams::deliver, host_stream.cpp).
bool streams_api(milzma_ctx* ctx, const std::vector<Case>& lzma_pool, size_t copies = 1) {
  std::vector<const Case*> files;
  std::vector<orc_result> want;
  for (size_t rep = 0; rep < copies; rep++)
  for (const Case& c : lzma_pool) {
    if (files.size() >= 14 * copies) break;
    orc_result w;
    memset(&w, 0, sizeof w);
    orc_lzma_decompress(ptr_of(c.data), c.data.size(), nullptr, &w);
    if (w.kind != ORC_OK || c.data.size() < 120 || w.in_consumed != c.data.size()) {
      orc_free(w.out);
      continue;
    }
    files.push_back(&c);
    want.push_back(w);
  }
  bool ok = true;
  const uint32_t n = uint32_t(files.size()) + 2;   // + a stream with a bad header, + one that gets only half a header
  milzma_streams* S = nullptr;
  if (files.size() < 3) {
    for (orc_result& w : want) orc_free(w.out);
    return true;
  }
  if (milzma_streams_open(ctx, MILZMA_KIND_RAW_LZMA, n, nullptr, &S) != MILZMA_OK) {
    printf("INFRA streams_open: %s\n", milzma_last_error(ctx));
    ok = false;
  }
  const uint8_t bad_header[20] = {255, 1, 2, 3};
  for (int call = 0; ok && call < 11; call++) {
    std::vector<uint32_t> idx;
    std::vector<const void*> data;
    std::vector<size_t> len;
    for (uint32_t i = 0; i < uint32_t(files.size()); i++) {
      const int start = int(i % 4), piece = (call - start) / 1;
      if (call < start || (call - start) % 2 || piece / 2 >= 4) continue;   // every second call from its start on: four pieces
      const size_t total = files[i]->data.size(), q = piece / 2;
      const size_t a = total * q / 4, b = q == 3 ? total : total * (q + 1) / 4;
      idx.push_back(i);
      data.push_back(ptr_of(files[i]->data) + a);
      len.push_back(b - a);
    }
    if (call == 1) {
      idx.push_back(n - 2);
      data.push_back(bad_header);
      len.push_back(sizeof bad_header);
      idx.push_back(n - 1);
      data.push_back(ptr_of(files[0]->data));
      len.push_back(7);
    }
    if (call == 4) {   // the stream whose header write failed is dead: it refuses bytes (ErrorKind::WriteZero), it does not swallow them
      idx.push_back(n - 2);
      data.push_back(bad_header);
      len.push_back(5);
    }
    std::vector<int32_t> st(idx.size(), -1);
    if (milzma_streams_write(S, uint32_t(idx.size()), idx.data(), data.data(), len.data(), st.data()) != MILZMA_OK) {
      printf("INFRA streams_write, call %d: %s\n", call, milzma_streams_last_error(S));
      ok = false;
      break;
    }
    for (size_t j = 0; j < idx.size(); j++) {
      const bool should_fail = idx[j] == n - 2;
      const char* text = call == 4 ? "failed to write whole buffer" : "must be < 225";
      if ((st[j] != MILZMA_OK) != should_fail || (should_fail && !strstr(milzma_streams_write_error(S, idx[j]), text)) ||
          (!should_fail && milzma_streams_write_taken(S, idx[j]) != len[j]) || (should_fail && call == 4 && milzma_streams_write_taken(S, idx[j]) != 0)) {
        printf("MISMATCH streams_write: stream %u status %d (%s)\n", idx[j], st[j], milzma_streams_write_error(S, idx[j]));
        ok = false;
      }
    }
    if (ok && call == 5) {   // Stream::get_output between writes: whole rings only, none for the dead stream
      for (uint32_t i = 0; i < n && ok; i++) {
        uint64_t sink = 0;
        int32_t has = -1;
        std::vector<uint8_t> buf(1 << 16);
        if (milzma_streams_output(S, i, 0, buf.data(), buf.size(), &sink, &has) != MILZMA_OK || has != (i == n - 2 ? 0 : 1) ||
            (i < files.size() && sink > want[i].out_len)) {
          printf("MISMATCH streams_output: stream %u has %d sink %llu\n", i, has, (unsigned long long)sink);
          ok = false;
        }
      }
    }
  }
  if (ok) {
    std::vector<milzma_output> outs(n);
    if (milzma_streams_finish(S, outs.data()) == MILZMA_INFRA_ERROR) {
      printf("INFRA streams_finish: %s\n", milzma_streams_last_error(S));
      ok = false;
    }
    for (uint32_t i = 0; i < n && ok; i++) {
      g_cases++;
      g_compared++;
      const milzma_output& o = outs[i];
      if (i < files.size()) {
        if (o.kind != MILZMA_OK || o.len != want[i].out_len || (o.len && memcmp(o.data, want[i].out, o.len) != 0) || o.in_consumed != files[i]->data.size()) {
          printf("MISMATCH stream %u (%s): kind %d '%s' len %zu (want %zu)\n", i, files[i]->name.c_str(), o.kind, o.msg, o.len, want[i].out_len);
          ok = false;
        }
      } else if (o.kind != MILZMA_LZMA_ERROR || !strstr(o.msg, i == n - 2 ? "previous write error" : "failed to read header")) {
        printf("MISMATCH stream %u: kind %d '%s'\n", i, o.kind, o.msg);
        ok = false;
      }
    }
    for (milzma_output& o : outs) milzma_free(o.data);
  }
  milzma_streams_close(S);
  for (orc_result& w : want) orc_free(w.out);
  return ok;
}

// 3f. the push-mode calls under fault injection: open, four writes (every stream a quarter of its file per call), finish, close -- the n-th
// fallible runtime call of the sequence fails.  Whatever fails: a call that reports MILZMA_OK for a stream delivered what the oracle has, an
// infrastructure error carries a text, close and destroy release everything (the result buffers the waves deliver into, the slab, the
// staging: LeakSanitizer), nothing crashes or hangs.  copies: as in streams_api (>= 64 streams: the delivering path).
bool streams_faults(const std::vector<Case>& lzma_pool, long upto, long stride, size_t copies, long* infra_seen, long* good_seen, long* worst) {
  std::vector<const Case*> files;
  std::vector<orc_result> want;
  for (size_t rep = 0; rep < copies; rep++)
    for (const Case& c : lzma_pool) {
      if (files.size() >= 12 * copies) break;
      orc_result w;
      memset(&w, 0, sizeof w);
      orc_lzma_decompress(ptr_of(c.data), c.data.size(), nullptr, &w);
      if (w.kind != ORC_OK || c.data.size() < 120 || w.in_consumed != c.data.size()) {
        orc_free(w.out);
        continue;
      }
      files.push_back(&c);
      want.push_back(w);
    }
  bool ok = files.size() >= 3;
  const uint32_t n = uint32_t(files.size());
  long last = upto;
  for (long f = 0; ok && f <= last; f += (f == 0 ? 1 : stride)) {
    milzma_ctx* c = nullptr;
    fake_hip_fail_at(f == 0 ? -1 : f);
    if (milzma_create(0, &c) != MILZMA_OK) continue;
    milzma_streams* S = nullptr;
    bool infra = milzma_streams_open(c, MILZMA_KIND_RAW_LZMA, n, nullptr, &S) != MILZMA_OK;
    if (infra && milzma_last_error(c)[0] == 0) {
      printf("MISMATCH streams fault %ld: open failed without a text\n", f);
      ok = false;
    }
    for (int q = 0; ok && !infra && q < 4; q++) {
      std::vector<uint32_t> idx(n);
      std::vector<const void*> data(n);
      std::vector<size_t> len(n);
      for (uint32_t i = 0; i < n; i++) {
        const size_t total = files[i]->data.size(), a = total * size_t(q) / 4, b = q == 3 ? total : total * size_t(q + 1) / 4;
        idx[i] = i;
        data[i] = ptr_of(files[i]->data) + a;
        len[i] = b - a;
      }
      std::vector<int32_t> st(n, -1);
      if (milzma_streams_write(S, n, idx.data(), data.data(), len.data(), st.data()) != MILZMA_OK) {
        infra = true;
        if (milzma_streams_last_error(S)[0] == 0) {
          printf("MISMATCH streams fault %ld: write %d failed without a text\n", f, q);
          ok = false;
        }
      }
    }
    if (ok && S) {   // finish whatever became of the writes: a stream it calls decoded IS decoded
      std::vector<milzma_output> outs(n);
      for (milzma_output& o : outs) memset(&o, 0, sizeof o);
      const int rc = milzma_streams_finish(S, outs.data());
      for (uint32_t i = 0; i < n && ok; i++) {
        const milzma_output& o = outs[i];
        if (o.kind == MILZMA_OK && !infra && rc != MILZMA_INFRA_ERROR) {
          (*good_seen)++;
          g_compared++;
          if (o.len != want[i].out_len || (o.len && memcmp(o.data, want[i].out, o.len) != 0) || o.in_consumed != files[i]->data.size()) {
            printf("MISMATCH streams fault %ld: stream %u (%s) decoded to %zu bytes (want %zu)\n", f, i, files[i]->name.c_str(), o.len, want[i].out_len);
            ok = false;
          }
        } else if (o.kind == MILZMA_OK) {   // (decoded although a call before it failed: then it must still be right -- or empty-handed)
          if (o.len == want[i].out_len && (!o.len || memcmp(o.data, want[i].out, o.len) == 0)) (*good_seen)++;
          else (*infra_seen)++;
        } else {
          (*infra_seen)++;
        }
      }
      for (milzma_output& o : outs) milzma_free(o.data);
    } else if (infra) {
      (*infra_seen) += n;
    }
    *worst = std::max(*worst, fake_hip_calls());
    if (f == 0) {
      last = std::min(upto, fake_hip_calls() + 2);
      if (infra) {
        printf("MISMATCH streams fault run without a fault failed: %s\n", S ? milzma_streams_last_error(S) : milzma_last_error(c));
        ok = false;
      }
    }
    fake_hip_fail_at(-1);
    if (S) milzma_streams_close(S);
    milzma_destroy(c);
    if (milzma::host::out_live_buffers() != 0) {   // every result buffer of the sequence has been freed or was never handed out
      printf("MISMATCH streams fault %ld (%u streams): %zu result buffers are still held after finish, free and close\n", f, n, milzma::host::out_live_buffers());
      ok = false;
    }
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
    if (getenv("PIPELINE_LEAK_EACH")) {   // (debugging aid: which fault leaves something behind)
      if (!strcmp(getenv("PIPELINE_LEAK_EACH"), "trim")) milzma_pool_trim(0);
      if (__lsan_do_recoverable_leak_check()) {
        printf("LEAK after the push-mode sequence with fault %ld (%u streams, infra %d)\n", f, n, int(infra));
        ok = false;
      }
    }
#endif
#endif
  }
  for (orc_result& w : want) orc_free(w.out);
  return ok;
}

}  // namespace

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);   // (LeakSanitizer leaves through _exit: what was printed must be out by then)
  if (argc < 4) {
    fprintf(stderr, "usage: pipeline_fuzz <dir> <rounds> <rng seed>\n");
    return 2;
  }
  const std::string dir = argv[1];
  const int rounds = atoi(argv[2]);
  Rng rng{0x9E3779B97F4A7C15ull ^ strtoull(argv[3], nullptr, 0)};
  std::vector<Case> pool[3];
  if (DIR* dp = opendir(dir.c_str())) {
    while (dirent* e = readdir(dp)) {
      const std::string name = e->d_name;
      Kind k;
      if (name.size() > 5 && name.compare(name.size() - 5, 5, ".lzma") == 0)
        k = LZMA;
      else if (name.size() > 6 && name.compare(name.size() - 6, 6, ".lzma2") == 0)
        k = LZMA2;
      else if (name.size() > 3 && name.compare(name.size() - 3, 3, ".xz") == 0)
        k = XZ;
      else
        continue;
      if (FILE* f = fopen((dir + "/" + name).c_str(), "rb")) {
        Case c{name, k, {}};
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) c.data.insert(c.data.end(), buf, buf + n);
        fclose(f);
        pool[k].push_back(std::move(c));
      }
    }
    closedir(dp);
  }
  if (pool[LZMA].empty() || pool[LZMA2].empty() || pool[XZ].empty()) {
    fprintf(stderr, "need .lzma, .lzma2 and .xz files in %s\n", dir.c_str());
    return 2;
  }
  // truncated variants (the one failure the stand-in kernels report faithfully) + a few garbage tails
  for (int k = 0; k < 3; k++) {
    const size_t n0 = pool[k].size();
    for (size_t i = 0; i < n0; i++) {
      for (int v = 0; v < 2; v++) {
        Case c = pool[k][i];
        if (c.data.size() < 4) continue;
        c.data.resize(v == 0 ? c.data.size() / 2 : c.data.size() - 1 - rng.below(uint32_t(std::min<size_t>(c.data.size() - 1, 40))));
        c.name += v == 0 ? " (half)" : " (tail cut)";
        pool[k].push_back(std::move(c));
      }
    }
  }
  // mutated variants (PIPELINE_MUTATIONS per file): a flipped bit, a changed byte, an inserted / removed byte, changes near the end (where an
  // .xz file keeps its Index and footer).  Container damage is the host code's own business (compared with the oracle); a damaged payload
  // only has to come back as SOME reference error.
  if (const char* e = getenv("PIPELINE_MUTATIONS")) {
    const int per = atoi(e);
    for (int k = 0; k < 3; k++) {
      const size_t n0 = pool[k].size();
      for (size_t i = 0; i < n0; i++)
        for (int v = 0; v < per; v++) {
          Case c = pool[k][i];
          if (c.data.size() < 8) continue;
          const uint32_t ops = 1 + rng.below(2);
          for (uint32_t o = 0; o < ops; o++) {
            const bool tail = k == XZ && rng.below(2) == 0;   // (Index / footer region)
            const size_t at = tail ? c.data.size() - 1 - rng.below(uint32_t(std::min<size_t>(c.data.size() - 1, 48))) : rng.below(uint32_t(c.data.size()));
            switch (rng.below(4)) {
              case 0: c.data[at] ^= uint8_t(1u << rng.below(8)); break;
              case 1: c.data[at] = uint8_t(rng.next()); break;
              case 2: c.data.insert(c.data.begin() + at, uint8_t(rng.below(3) ? 0 : rng.next())); break;
              default: c.data.erase(c.data.begin() + at); break;
            }
          }
          c.name += " (mutation " + std::to_string(v) + ")";
          pool[k].push_back(std::move(c));
        }
    }
  }
  // PIPELINE_FAULTS=K: fault injection instead of the stages below.  One batch per entry point, K times over, the n-th fallible runtime
  // call (allocation, copy, stream / event creation, kernel launch) failing in run n: whatever fails, every file comes back either as the
  // oracle has it or with an infrastructure error and a text -- never as an empty success, never with a crash, a hang or a leak.
  if (const char* e = getenv("PIPELINE_FAULTS")) {
    const long upto = atol(e);
    const long stride = getenv("PIPELINE_FAULT_STRIDE") ? std::max(1L, atol(getenv("PIPELINE_FAULT_STRIDE"))) : 1;   // (every k-th call only)
    long worst = 0, infra_files = 0, good_files = 0;
    for (int k = 0; k < 3; k++) {
      Batch b;
      for (uint32_t i = 0; i < 8; i++) b.add(&pool[k][rng.below(uint32_t(pool[k].size()))]);
      long last = upto;   // (after the fault-free run: no further than the calls a batch makes)
      for (long n = 0; n <= last; n += (n == 0 ? 1 : stride)) {
        milzma_ctx* c = nullptr;
        fake_hip_fail_at(n == 0 ? -1 : n);
        if (milzma_create(0, &c) != MILZMA_OK) continue;   // (creation itself failed: reported, nothing to run)
        b.prepare();
        const uint32_t cnt = uint32_t(b.cases.size());
        const int rc = k == LZMA    ? lzma_batch(c, cnt, b.ins.data(), b.lens.data(), b.outs.data())
                       : k == LZMA2 ? milzma_lzma2_decompress_batch(c, cnt, b.ins.data(), b.lens.data(), b.outs.data())
                                    : milzma_xz_decompress_batch(c, cnt, b.ins.data(), b.lens.data(), b.outs.data());
        worst = std::max(worst, fake_hip_calls());
        if (n == 0) last = std::min(upto, fake_hip_calls() + 2);
        fake_hip_fail_at(-1);
        for (size_t i = 0; i < b.cases.size(); i++) {
          const milzma_output& o = b.outs[i];
          if (o.kind == MILZMA_INFRA_ERROR) {
            infra_files++;
            if (o.msg[0] == 0 || (o.data == nullptr && o.len != 0)) {
              printf("MISMATCH fault %ld, %s: an infrastructure error without a text: msg '%s' data %p len %zu rc %d, context says '%s'\n", n,
                     b.cases[i]->name.c_str(), o.msg, (void*)o.data, o.len, rc, milzma_last_error(c));
              return 1;
            }
          } else {
            good_files++;
            if (!check(*b.cases[i], o, "batch under fault injection")) {
              printf("  (fault at call %ld of entry point %d)\n", n, k);
              return 1;
            }
          }
          milzma_free(o.data);
        }
        milzma_destroy(c);
      }
    }
    // ... and the multi-device calls (FAKE_HIP_DEVICES / MILZMA_MULTI_REPLICAS): a whole-file batch, and the one-ingest-point unit call
    if (getenv("PIPELINE_FAULTS_MULTI")) {
      for (int form = 0; form < 2; form++) {
        Batch b;
        for (uint32_t i = 0; i < 9; i++) b.add(&pool[form == 0 ? XZ : LZMA2][rng.below(uint32_t(pool[form == 0 ? XZ : LZMA2].size()))]);
        // (form 1: LZMA2 units that decode, packed into one "device" buffer)
        std::vector<const Case*> cs;
        std::vector<orc_result> want;
        std::vector<milzma_unit> units;
        std::vector<uint8_t> in;
        size_t io = 0, oo = 0;
        if (form == 1) {
          for (const Case* c : b.cases) {
            Want w = oracle_of(*c);
            if (w.r.kind == ORC_OK) {
              milzma_unit u;
              memset(&u, 0, sizeof u);
              u.kind = MILZMA_KIND_LZMA2;
              u.in_off = io;
              u.in_len = c->data.size();
              u.out_off = oo;
              u.out_cap = w.r.out_len + 32;
              io += (c->data.size() + 255) & ~size_t(255);
              oo += size_t(u.out_cap);
              units.push_back(u);
              cs.push_back(c);
              want.push_back(w.r);
            } else {
              orc_free(w.r.out);
            }
          }
          in.assign(io + 512, 0);
          for (size_t i = 0; i < cs.size(); i++) memcpy(in.data() + units[i].in_off, ptr_of(cs[i]->data), cs[i]->data.size());
        }
        long last = upto;
        for (long n = 0; n <= last; n += (n == 0 ? 1 : stride)) {
          milzma_multi* m = nullptr;
          fake_hip_fail_at(n == 0 ? -1 : n);
          if (milzma_multi_create(0, &m) != MILZMA_OK) continue;
          if (form == 0) {
            b.prepare();
            const int rc = milzma_multi_xz_decompress_batch(m, uint32_t(b.cases.size()), b.ins.data(), b.lens.data(), b.outs.data());
            if (n == 0) last = std::min(upto, fake_hip_calls() + 2);
            fake_hip_fail_at(-1);
            for (size_t i = 0; i < b.cases.size(); i++) {
              const milzma_output& o = b.outs[i];
              if (o.kind == MILZMA_INFRA_ERROR) {
                infra_files++;
                if (o.msg[0] == 0) {
                  printf("MISMATCH multi fault %ld, %s: an infrastructure error without a text (rc %d, '%s')\n", n, b.cases[i]->name.c_str(), rc, milzma_multi_last_error(m));
                  return 1;
                }
              } else {
                good_files++;
                if (!check(*b.cases[i], o, "multi batch under fault injection")) return 1;
              }
              milzma_free(o.data);
            }
          } else if (!units.empty()) {
            std::vector<uint8_t> out(oo + 512, 0xAA);
            std::vector<milzma_result> res(units.size());
            const int rc = milzma_multi_decode_units_rooted(m, 0, units.data(), uint32_t(units.size()), in.data(), out.data(), res.data());
            if (n == 0) last = std::min(upto, fake_hip_calls() + 2);
            fake_hip_fail_at(-1);
            if (rc == MILZMA_OK) {
              for (size_t i = 0; i < units.size(); i++) {
                good_files++;
                if (res[i].status != MILZMA_ST_OK || res[i].out_len != want[i].out_len ||
                    (want[i].out_len && memcmp(out.data() + units[i].out_off, want[i].out, want[i].out_len) != 0)) {
                  printf("MISMATCH rooted call under fault %ld: unit %zu (%s) status %u\n", n, i, cs[i]->name.c_str(), res[i].status);
                  return 1;
                }
              }
            } else {
              infra_files += long(units.size());
              if (milzma_multi_last_error(m)[0] == 0) {
                printf("MISMATCH rooted call under fault %ld: failed without a text\n", n);
                return 1;
              }
            }
          }
          milzma_multi_destroy(m);
        }
        for (auto& w : want) orc_free(w.out);
      }
    }
    // ... and the push-mode calls: a dozen streams, and enough of them for the waves to deliver into the result buffers themselves
    long stream_infra = 0, stream_good = 0;
    if (!getenv("PIPELINE_FAULTS_MULTI") && !getenv("PIPELINE_NO_STREAM_FAULTS")) {
      if (!streams_faults(pool[LZMA], upto, stride, 1, &stream_infra, &stream_good, &worst)) return 1;
      if (!streams_faults(pool[LZMA], upto, stride * 7, 6, &stream_infra, &stream_good, &worst)) return 1;   // (every seventh call: 72 streams a run)
      infra_files += stream_infra;
      good_files += stream_good;
    }
    const size_t pooled = milzma_pool_trim(0);
    if (pooled != 0 || milzma::host::out_pooled_buffers() != 0 || milzma::host::out_live_buffers() != 0) {
      printf("MISMATCH: %zu bytes / %zu buffers still pooled after milzma_pool_trim(0), %zu still held by callers\n", pooled, milzma::host::out_pooled_buffers(),
             milzma::host::out_live_buffers());
      return 1;
    }
    printf("ok faults=%ld calls_per_batch<=%ld files_with_infra_error=%ld files_decoded=%ld compared=%" PRIu64 "\n", upto, worst, infra_files, good_files, g_compared);
    return 0;
  }
  milzma_ctx *ctx = nullptr, *ctx2 = nullptr;
  if (milzma_create(0, &ctx) != MILZMA_OK || milzma_create(0, &ctx2) != MILZMA_OK) {
    fprintf(stderr, "milzma_create: %s\n", milzma_last_error(nullptr));
    return 1;
  }
  bool ok = true;
  const char* kname[3] = {"lzma batch", "lzma2 batch", "xz batch"};
  for (int round = 0; round < rounds && ok; round++) {
    for (int k = 0; k < 3 && ok; k++) {
      // 1. one batch of a random selection (with repeats: two units of a batch may share their input bytes)
      Batch b;
      const uint32_t n = 1 + rng.below(uint32_t(pool[k].size()) * 2);
      for (uint32_t i = 0; i < n; i++) b.add(&pool[k][rng.below(uint32_t(pool[k].size()))]);
      ok = run_batch(ctx, Kind(k), b, kname[k]) && ok;
      // 2. single-file calls
      for (int j = 0; j < 3 && ok; j++) {
        const Case& c = pool[k][rng.below(uint32_t(pool[k].size()))];
        milzma_output o;
        memset(&o, 0, sizeof o);
        if (k == LZMA)
          milzma_lzma_decompress(ctx, ptr_of(c.data), c.data.size(), nullptr, &o);
        else if (k == LZMA2)
          milzma_lzma2_decompress(ctx, ptr_of(c.data), c.data.size(), &o);
        else
          milzma_xz_decompress(ctx, ptr_of(c.data), c.data.size(), &o);
        ok = check(c, o, "single") && ok;
        milzma_free(o.data);
      }
    }
    // 3. two calls in flight on two contexts (the asynchronous halves)
    {
      Batch a, b;
      for (uint32_t i = 0; i < 12; i++) {
        a.add(&pool[LZMA][rng.below(uint32_t(pool[LZMA].size()))]);
        b.add(&pool[XZ][rng.below(uint32_t(pool[XZ].size()))]);
      }
      a.prepare();
      b.prepare();
      const int ra = lzma_batch_async(ctx, uint32_t(a.cases.size()), a.ins.data(), a.lens.data(), a.outs.data());
      const int rb = milzma_xz_decompress_batch_async(ctx2, uint32_t(b.cases.size()), b.ins.data(), b.lens.data(), b.outs.data());
      if (ra != MILZMA_OK || rb != MILZMA_OK) {
        printf("INFRA async begin: %s / %s\n", milzma_last_error(ctx), milzma_last_error(ctx2));
        ok = false;
      }
      const int wa = milzma_batch_wait(ctx), wb = milzma_batch_wait(ctx2);
      if (wa == MILZMA_INFRA_ERROR || wb == MILZMA_INFRA_ERROR) {
        printf("INFRA async wait: %s / %s\n", milzma_last_error(ctx), milzma_last_error(ctx2));
        ok = false;
      }
      ok = a.verify("async lzma") && ok;
      ok = b.verify("async xz") && ok;
    }
  }
  // 3b. the unit-level calls a device-resident pipeline makes ("device" memory is host memory here): a caller's own GROW / move / RESUME
  //     loop from slices that start far too small, the asynchronous halves, and the .xz pipeline milzma_xz_plan -> decode -> milzma_crc_units
  //     (units of the generic kernel cannot be parked: they come back with a plain OUT_FULL and start over -- another loop, not this one)
  if (ok && !(getenv("MILZMA_KERNEL") && !strcmp(getenv("MILZMA_KERNEL"), "generic"))) {
    std::vector<const Case*> cs;
    for (const Case& c : pool[LZMA2]) {
      Want w = oracle_of(c);
      if (w.r.kind == ORC_OK && w.r.out_len > 0) cs.push_back(&c);
      orc_free(w.r.out);
    }
    const uint32_t n = uint32_t(cs.size());
    std::vector<milzma_unit> units(n);
    std::vector<orc_result> want(n);
    size_t io = 0;
    for (uint32_t i = 0; i < n; i++) {
      memset(&want[i], 0, sizeof want[i]);
      orc_lzma2_decompress(ptr_of(cs[i]->data), cs[i]->data.size(), &want[i]);
      memset(&units[i], 0, sizeof units[i]);
      units[i].kind = MILZMA_KIND_LZMA2;
      units[i].in_off = io;
      units[i].in_len = cs[i]->data.size();
      io += (cs[i]->data.size() + 255) & ~size_t(255);
    }
    std::vector<uint8_t> in(io + 512, 0);
    for (uint32_t i = 0; i < n; i++) memcpy(in.data() + units[i].in_off, ptr_of(cs[i]->data), cs[i]->data.size());
    // slices of 512 bytes to begin with; grown by 3x per round
    std::vector<uint8_t> out;
    size_t oo = 0;
    for (uint32_t i = 0; i < n; i++) {
      units[i].out_off = oo;
      units[i].out_cap = 512;
      oo += 512;
    }
    out.assign(oo + 512, 0);
    std::vector<milzma_result> res(n);
    uint32_t flags = MILZMA_DECODE_GROW;
    int rounds_done = 0;
    for (;; rounds_done++) {
      if (milzma_decode_units_ex(ctx, units.data(), n, in.data(), out.data(), res.data(), nullptr, flags) != MILZMA_OK) {
        printf("INFRA decode_units_ex: %s\n", milzma_last_error(ctx));
        ok = false;
        break;
      }
      std::vector<uint64_t> so, dof, ln;
      std::vector<uint8_t> bigger;
      size_t total = 0;
      std::vector<milzma_unit> next = units;
      bool any = false;
      for (uint32_t i = 0; i < n; i++) {
        const bool parked = res[i].status == MILZMA_ST_OUT_FULL && res[i].err_a == MILZMA_PARKED;
        any = any || parked;
        next[i].out_off = total;
        next[i].out_cap = parked ? units[i].out_cap * 3 : units[i].out_cap;
        so.push_back(units[i].out_off);
        dof.push_back(total);
        ln.push_back(res[i].out_len < units[i].out_cap ? res[i].out_len : units[i].out_cap);
        total += size_t(next[i].out_cap);
      }
      if (!any) break;
      bigger.assign(total + 512, 0);
      if (milzma_move_units(ctx, n, out.data(), so.data(), bigger.data(), dof.data(), ln.data(), nullptr) != MILZMA_OK) {
        printf("INFRA move_units: %s\n", milzma_last_error(ctx));
        ok = false;
        break;
      }
      out.swap(bigger);
      units = next;
      flags = MILZMA_DECODE_RESUME;
    }
    for (uint32_t i = 0; i < n && ok; i++) {
      g_cases++;
      g_compared++;
      if (res[i].status != MILZMA_ST_OK || res[i].out_len != want[i].out_len || memcmp(out.data() + units[i].out_off, want[i].out, want[i].out_len) != 0) {
        printf("MISMATCH unit %u (%s) after %d grow rounds: status %u len %" PRIu64 " (want %zu)\n", i, cs[i]->name.c_str(), rounds_done, res[i].status,
               res[i].out_len, want[i].out_len);
        ok = false;
      }
    }
    // the asynchronous halves + the CRCs of what was decoded
    if (ok && n) {
      std::vector<milzma_result> r2(n);
      std::vector<uint32_t> c32(n);
      std::vector<uint64_t> c64(n);
      if (milzma_decode_units_async(ctx, units.data(), n, in.data(), out.data(), nullptr) != MILZMA_OK ||
          milzma_decode_units_async(ctx, units.data(), n, in.data(), out.data(), nullptr) != MILZMA_INFRA_ERROR ||   // (a second one is refused)
          milzma_decode_units_wait(ctx, r2.data()) != MILZMA_OK ||
          milzma_crc_units(ctx, units.data(), n, out.data(), r2.data(), c32.data(), c64.data(), nullptr) != MILZMA_OK) {
        printf("INFRA async / crc: %s\n", milzma_last_error(ctx));
        ok = false;
      }
      for (uint32_t i = 0; i < n && ok; i++)
        if (r2[i].status != MILZMA_ST_OK || c32[i] != orc_crc32(want[i].out, want[i].out_len) || c64[i] != orc_crc64(want[i].out, want[i].out_len)) {
          printf("MISMATCH unit %u: CRCs of the device-side digest\n", i);
          ok = false;
        }
    }
    for (auto& w : want) orc_free(w.out);
    if (ok) ok = raw_grow_loop(ctx, pool[LZMA], pool[LZMA2]);
    if (ok) ok = raw_feed_loop(ctx, pool[LZMA], pool[LZMA2]);
    if (ok) ok = streams_api(ctx, pool[LZMA]);
    if (ok) ok = streams_api(ctx, pool[LZMA], 6);   // (>= 64 streams: the result buffers are filled by the kernels)
    // milzma_xz_plan: the Index of a good file -> one unit per block; decoded, every block's bytes where the plan put them
    for (const Case& c : pool[XZ]) {
      if (!ok) break;
      Want w = oracle_of(c);
      uint32_t nu = 0, check = 0;
      std::vector<milzma_unit> pu(256);
      if (w.r.kind == ORC_OK && milzma_xz_plan(ptr_of(c.data), c.data.size(), pu.data(), 256, &nu, &check) == MILZMA_OK && nu && nu <= 256) {
        size_t cap = 0;
        for (uint32_t k = 0; k < nu; k++) {
          pu[k].out_off = cap;
          cap += size_t(pu[k].out_cap);
        }
        std::vector<uint8_t> fin(c.data.size() + 512, 0), fout(cap + 512, 0);
        memcpy(fin.data(), ptr_of(c.data), c.data.size());
        std::vector<milzma_result> pr(nu);
        if (milzma_decode_units(ctx, pu.data(), nu, fin.data(), fout.data(), pr.data(), nullptr) != MILZMA_OK) {
          printf("INFRA planned decode: %s\n", milzma_last_error(ctx));
          ok = false;
        }
        size_t at = 0;
        for (uint32_t k = 0; k < nu && ok; k++) {
          g_cases++;
          g_compared++;
          if (pr[k].status != MILZMA_ST_OK || at + pr[k].out_len > w.r.out_len || memcmp(fout.data() + pu[k].out_off, w.r.out + at, size_t(pr[k].out_len)) != 0) {
            printf("MISMATCH planned block %u of %s\n", k, c.name.c_str());
            ok = false;
          }
          at += size_t(pr[k].out_len);
        }
        if (ok && at != w.r.out_len) {
          printf("MISMATCH planned blocks of %s: %zu bytes of %zu\n", c.name.c_str(), at, w.r.out_len);
          ok = false;
        }
      }
      orc_free(w.r.out);
    }
  }
  // 3c. application threads of their own, a context each, all at once (the result-buffer pool, the one-streamed-launch-per-device rule)
  if (ok) {
    std::vector<std::thread> th;
    std::vector<int> good(3, 1);
    for (int t = 0; t < 3; t++)
      th.emplace_back([&, t] {
        milzma_ctx* c3 = nullptr;
        if (milzma_create(0, &c3) != MILZMA_OK) {
          good[t] = 0;
          return;
        }
        Rng r2{0xABCDEF12345ull + uint64_t(t)};
        for (int it = 0; it < 2; it++) {
          const Kind k = Kind((t + it) % 3);
          Batch b;
          for (uint32_t i = 0; i < 25; i++) b.add(&pool[k][r2.below(uint32_t(pool[k].size()))]);
          b.prepare();
          const uint32_t n = uint32_t(b.cases.size());
          const int rc = k == LZMA    ? lzma_batch(c3, n, b.ins.data(), b.lens.data(), b.outs.data())
                         : k == LZMA2 ? milzma_lzma2_decompress_batch(c3, n, b.ins.data(), b.lens.data(), b.outs.data())
                                      : milzma_xz_decompress_batch(c3, n, b.ins.data(), b.lens.data(), b.outs.data());
          if (rc == MILZMA_INFRA_ERROR) good[t] = 0;
          for (size_t i = 0; i < b.cases.size(); i++) {   // (check() counts in globals: compared here, by hand)
            Want w = oracle_of(*b.cases[i]);
            const milzma_output& o = b.outs[i];
            if (w.faithful && !(o.kind == w.r.kind && strcmp(o.msg, w.r.msg) == 0 && o.len == w.r.out_len &&
                                (o.len == 0 || memcmp(o.data, w.r.out, o.len) == 0) && o.in_consumed == w.r.in_consumed))
              good[t] = 0;
            orc_free(w.r.out);
            milzma_free(o.data);
          }
        }
        milzma_destroy(c3);
      });
    for (auto& x : th) x.join();
    for (int t = 0; t < 3; t++)
      if (!good[t]) {
        printf("MISMATCH in application thread %d\n", t);
        ok = false;
      }
    g_cases += 150;
  }
  // 4. a call large enough to be cut into groups over lanes (>= 8192 units): tiny files, many times
  if (ok && getenv("PIPELINE_BIG")) {
    Batch b;
    const Case* tiny = &pool[LZMA][0];
    for (const Case& c : pool[LZMA])
      if (c.data.size() < tiny->data.size() && c.data.size() > 20) tiny = &c;
    for (uint32_t i = 0; i < 8192 + 300; i++) b.add(i % 7 == 0 ? &pool[LZMA][rng.below(uint32_t(pool[LZMA].size()))] : tiny);
    ok = run_batch(ctx, LZMA, b, "lzma batch in groups") && ok;
  }
  // 4b. decompress::Options on .lzma files (src/decode/options.rs): the three UnpackedSize modes with and without a provided size, a memory
  //     limit that does not bite -- through the batch call (one option set per batch) and the single-file call
  for (int v = 0; v < 6 && ok; v++) {
    milzma_options mo;
    orc_options oo;
    milzma_default_options(&mo);
    orc_default_options(&oo);
    mo.unpacked_size_mode = oo.unpacked_size_mode = v % 3;
    mo.provided_is_some = oo.provided_is_some = v >= 3;
    mo.provided = oo.provided = 20000;                   // (right for some files, wrong for most)
    mo.memlimit_is_some = oo.memlimit_is_some = v & 1;
    mo.memlimit = oo.memlimit = uint64_t(1) << 30;
    Batch b;
    for (uint32_t i = 0; i < 16; i++) b.add(&pool[LZMA][rng.below(uint32_t(pool[LZMA].size()))]);
    b.prepare();
    if (milzma_lzma_decompress_batch(ctx, uint32_t(b.cases.size()), b.ins.data(), b.lens.data(), &mo, b.outs.data()) == MILZMA_INFRA_ERROR) {
      printf("INFRA lzma batch with options: %s\n", milzma_last_error(ctx));
      ok = false;
      break;
    }
    for (size_t i = 0; i < b.cases.size() + 2 && ok; i++) {
      const Case& c = *b.cases[i % b.cases.size()];
      milzma_output single;
      memset(&single, 0, sizeof single);
      const bool batch = i < b.cases.size();
      if (!batch) milzma_lzma_decompress(ctx, ptr_of(c.data), c.data.size(), &mo, &single);
      const milzma_output& o = batch ? b.outs[i] : single;
      orc_result w;
      memset(&w, 0, sizeof w);
      orc_lzma_decompress(ptr_of(c.data), c.data.size(), &oo, &w);
      g_cases++;
      if (true) {
        g_compared++;
        if (!(o.kind == w.kind && strcmp(o.msg, w.msg) == 0 && o.len == w.out_len && (o.len == 0 || memcmp(o.data, w.out, o.len) == 0) &&
              o.in_consumed == w.in_consumed)) {
          printf("MISMATCH %s with options %d (%s): got kind %d '%s' len %zu consumed %zu | oracle kind %d '%s' len %zu consumed %zu\n", c.name.c_str(), v,
                 batch ? "batch" : "single", o.kind, o.msg, o.len, o.in_consumed, w.kind, w.msg, w.out_len, w.in_consumed);
          ok = false;
        }
      } else {
        g_skipped++;
        if (o.kind == MILZMA_OK || o.kind == MILZMA_INFRA_ERROR) {
          printf("MISMATCH %s with options %d: oracle fails with '%s', the library says kind %d '%s'\n", c.name.c_str(), v, w.msg, o.kind, o.msg);
          ok = false;
        }
      }
      orc_free(w.out);
      milzma_free(o.data);
    }
  }
  // 5. several devices behind one handle (FAKE_HIP_DEVICES > 1, or MILZMA_MULTI_REPLICAS)
  if (ok) {
    milzma_multi* m = nullptr;
    if (milzma_multi_create(0, &m) != MILZMA_OK) {
      printf("INFRA multi create: %s\n", milzma_multi_last_error(nullptr));
      ok = false;
    } else {
      for (int k = 0; k < 3 && ok; k++) {
        Batch b;
        for (uint32_t i = 0; i < 20; i++) b.add(&pool[k][rng.below(uint32_t(pool[k].size()))]);
        b.prepare();
        const uint32_t n = uint32_t(b.cases.size());
        const int rc = k == LZMA    ? multi_lzma(m, n, b.ins.data(), b.lens.data(), b.outs.data())
                       : k == LZMA2 ? milzma_multi_lzma2_decompress_batch(m, n, b.ins.data(), b.lens.data(), b.outs.data())
                                    : milzma_multi_xz_decompress_batch(m, n, b.ins.data(), b.lens.data(), b.outs.data());
        if (rc == MILZMA_INFRA_ERROR) {
          printf("INFRA multi batch: %s\n", milzma_multi_last_error(m));
          ok = false;
        }
        ok = b.verify("multi batch") && ok;
      }
      // the unit-level calls: host buffers, and the one-ingest-point entry ("device" memory is host memory under the fake runtime)
      if (ok) {
        std::vector<const Case*> cs;
        for (const Case& c : pool[LZMA2]) {
          Want w = oracle_of(c);
          if (w.r.kind == ORC_OK) cs.push_back(&c);
          orc_free(w.r.out);
        }
        std::vector<milzma_unit> units(cs.size());
        std::vector<uint8_t> in, out;
        size_t io = 0, oo = 0;
        std::vector<orc_result> want(cs.size());
        for (size_t i = 0; i < cs.size(); i++) {
          memset(&want[i], 0, sizeof want[i]);
          orc_lzma2_decompress(ptr_of(cs[i]->data), cs[i]->data.size(), &want[i]);
          memset(&units[i], 0, sizeof units[i]);
          units[i].kind = MILZMA_KIND_LZMA2;
          units[i].in_off = io;
          units[i].in_len = cs[i]->data.size();
          units[i].out_off = oo;
          units[i].out_cap = want[i].out_len + 16 + (i % 3) * 5;   // (unaligned slices on purpose)
          io += (cs[i]->data.size() + 255) & ~size_t(255);
          oo += size_t(units[i].out_cap);
        }
        in.assign(io + 512, 0);
        for (size_t i = 0; i < cs.size(); i++) memcpy(in.data() + units[i].in_off, ptr_of(cs[i]->data), cs[i]->data.size());
        for (int form = 0; form < 2 && ok; form++) {
          out.assign(oo + 512, 0xAA);
          std::vector<milzma_result> res(cs.size());
          const int rc = form == 0 ? milzma_multi_decode_units_host(m, units.data(), uint32_t(units.size()), in.data(), io, out.data(), oo, res.data())
                                   : milzma_multi_decode_units_rooted(m, 0, units.data(), uint32_t(units.size()), in.data(), out.data(), res.data());
          if (rc != MILZMA_OK) {
            printf("INFRA multi units (form %d): %s\n", form, milzma_multi_last_error(m));
            ok = false;
            break;
          }
          for (size_t i = 0; i < cs.size(); i++) {
            g_cases++;
            g_compared++;
            if (res[i].status != MILZMA_ST_OK || res[i].out_len != want[i].out_len ||
                (want[i].out_len && memcmp(out.data() + units[i].out_off, want[i].out, want[i].out_len) != 0) ||
                (form == 1 && out[units[i].out_off + want[i].out_len] != 0xAA)) {   // (the rooted form writes a unit's bytes and nothing else)
              printf("MISMATCH unit %zu (%s) of the multi-device unit call, form %d: status %u len %" PRIu64 "\n", i, cs[i]->name.c_str(), form,
                     res[i].status, res[i].out_len);
              ok = false;
            }
          }
        }
        for (auto& w : want) orc_free(w.out);
      }
      milzma_multi_destroy(m);
    }
  }
  milzma_destroy(ctx2);
  milzma_destroy(ctx);
  const size_t pooled = milzma_pool_trim(0);
  if (pooled != 0) {
    printf("MISMATCH: %zu bytes still pooled after milzma_pool_trim(0)\n", pooled);
    ok = false;
  }
  if (!ok) return 1;
  printf("ok cases=%" PRIu64 " compared=%" PRIu64 " unfaithful_status_skipped=%" PRIu64 "\n", g_cases, g_compared, g_skipped);
  return 0;
}
