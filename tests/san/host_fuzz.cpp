// tests/san/host_fuzz.cpp -- TEST INFRASTRUCTURE (built by tests/san/Makefile, run by tests/test_host_sanitized.py).
//
// Drives the GPU-free entry points of the library's host side under ASan + UBSan with hostile input:
//   milzma_lzma_read_header, milzma_xz_plan, milzma_partition, milzma_result_message, milzma_crc32 / milzma_crc64,
//   milzma_free / milzma_pool_trim (foreign and double-freed pointers), the whole-file calls without a context,
//   and -- through the MILZMA_TEST_HOOKS entry milzma_test_xz_walk -- the complete XZ container walk (xz::decode_stream,
//   src/decode/xz.rs:18-94, as restated in host.cpp) with the LZMA2 payloads decoded by the CPU oracle.
// Where no payload failed, the walk's verdict (kind, message, bytes, reader position) must equal the oracle's own XZ decoder:
// a differential check of the container logic on every seed and every mutation.
//
//   host_fuzz <seed dir> <mutations per seed> <rng seed>
//
// Exit status 0 and a final "ok ..." line, or the first mismatch / a sanitizer report.
#include <dirent.h>

#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lzma_oracle.h"
#include "milzma.h"

extern "C" {
typedef int (*milzma_test_lzma2_fn)(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len, size_t* consumed, void* user);
int milzma_test_xz_walk(const uint8_t* in, size_t in_len, milzma_test_lzma2_fn fn, void* user, milzma_output* out);
}

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

struct Stats {
  uint64_t cases = 0, compared = 0, payload_errors = 0, planned = 0, headers_ok = 0;
} g;

int lzma2_by_oracle(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len, size_t* consumed, void* user) {
  orc_result r;
  memset(&r, 0, sizeof r);
  const int kind = orc_lzma2_decompress(in, in_len, &r);
  if (kind != ORC_OK) {
    orc_free(r.out);
    *static_cast<bool*>(user) = true;
    return MILZMA_ST_INPUT_EOF;  // (which error: not this test's business -- only what the walk does around it)
  }
  *out = static_cast<uint8_t*>(malloc(r.out_len ? r.out_len : 1));
  if (r.out_len) memcpy(*out, r.out, r.out_len);
  *out_len = r.out_len;
  *consumed = r.in_consumed;
  orc_free(r.out);
  return MILZMA_ST_OK;
}

bool run_one(const std::vector<uint8_t>& d, Rng& rng, const char* what) {
  g.cases++;
  const uint8_t* p = d.empty() ? reinterpret_cast<const uint8_t*>("") : d.data();
  // --- .lzma header with a few option sets (src/decode/lzma.rs:96-161, options.rs)
  for (int v = 0; v < 4; v++) {
    milzma_options o;
    milzma_default_options(&o);
    o.unpacked_size_mode = v % 3;
    o.provided_is_some = v & 1;
    o.provided = rng.next();
    o.memlimit_is_some = v >> 1;
    o.memlimit = rng.next();
    milzma_unit u;
    size_t hl = 0;
    milzma_output out;
    const int k = milzma_lzma_read_header(p, d.size(), v == 3 ? nullptr : &o, &u, &hl, v == 2 ? nullptr : &out);
    if (k == MILZMA_OK) {
      g.headers_ok++;
      if (hl > d.size() || u.lc > 8 || u.lp > 4 || u.pb > 4 || u.dict_size < 4096) {
        printf("MISMATCH %s: read_header accepted hl=%zu lc=%u lp=%u pb=%u dict=%u\n", what, hl, u.lc, u.lp, u.pb, u.dict_size);
        return false;
      }
    }
  }
  // --- Index -> units (milzma_xz_plan): every unit must lie inside the file
  {
    uint32_t nu = 0, check = 0;
    std::vector<milzma_unit> units(64);
    const int k = milzma_xz_plan(p, d.size(), units.data(), uint32_t(units.size()), &nu, &check);
    if (k == MILZMA_OK) {
      g.planned++;
      for (uint32_t i = 0; i < nu && i < units.size(); i++)
        if (units[i].in_off > d.size() || units[i].in_len > d.size() - units[i].in_off || units[i].out_cap > MILZMA_MAX_UNIT_BYTES) {
          printf("MISMATCH %s: planned unit %u outside the file\n", what, i);
          return false;
        }
    }
    (void)milzma_xz_plan(p, d.size(), nullptr, 0, &nu, nullptr);
  }
  // --- the container walk, payloads by the oracle; against the oracle's own XZ decoder where no payload failed
  {
    bool payload_error = false;
    milzma_output out;
    memset(&out, 0, sizeof out);
    const int k = milzma_test_xz_walk(p, d.size(), lzma2_by_oracle, &payload_error, &out);
    orc_result r;
    memset(&r, 0, sizeof r);
    const int ok = orc_xz_decompress(p, d.size(), &r);
    if (payload_error) {
      g.payload_errors++;
    } else {
      g.compared++;
      const bool same = k == out.kind && out.kind == ok && strcmp(out.msg, r.msg) == 0 && out.len == r.out_len &&
                        (out.len == 0 || memcmp(out.data, r.out, out.len) == 0) && (ok != ORC_OK || out.in_consumed == r.in_consumed);
      if (!same) {
        printf("MISMATCH %s (%zu bytes): walk kind %d '%s' len %zu consumed %zu | oracle kind %d '%s' len %zu consumed %zu\n", what, d.size(),
               out.kind, out.msg, out.len, out.in_consumed, ok, r.msg, r.out_len, r.in_consumed);
        return false;
      }
    }
    uint8_t* data = out.data;
    milzma_free(data);
    milzma_free(data);  // twice: recognised through the registry, not through the (now pooled) buffer's bytes
    orc_free(r.out);
  }
  // --- the whole-file calls without a context: an infrastructure error in every slot, never an empty success
  if ((g.cases & 63) == 0) {
    const uint8_t* ins[2] = {p, p};
    const size_t lens[2] = {d.size(), d.size() / 2};
    milzma_output outs[2];
    memset(outs, 0, sizeof outs);
    if (milzma_multi_xz_decompress_batch(nullptr, 2, ins, lens, outs) != MILZMA_INFRA_ERROR || outs[0].kind != MILZMA_INFRA_ERROR ||
        outs[1].kind != MILZMA_INFRA_ERROR || outs[0].msg[0] == 0) {
      printf("MISMATCH %s: multi batch without a handle left an empty success\n", what);
      return false;
    }
  }
  return true;
}

void mutate(std::vector<uint8_t>& d, Rng& rng) {
  const uint32_t ops = 1 + rng.below(3);
  for (uint32_t k = 0; k < ops; k++) {
    switch (rng.below(6)) {
      case 0:
        if (!d.empty()) d[rng.below(uint32_t(d.size()))] ^= uint8_t(1u << rng.below(8));
        break;
      case 1:
        if (!d.empty()) d[rng.below(uint32_t(d.size()))] = uint8_t(rng.next());
        break;
      case 2:
        d.resize(rng.below(uint32_t(d.size()) + 1));
        break;
      case 3:
        if (!d.empty()) d.insert(d.begin() + rng.below(uint32_t(d.size())), uint8_t(rng.below(3) ? 0 : rng.next()));
        break;
      case 4:
        if (d.size() > 12) {  // the footer / index region is where the planner reads
          const size_t at = d.size() - 1 - rng.below(uint32_t(std::min<size_t>(d.size() - 1, 40)));
          d[at] = uint8_t(rng.next());
        }
        break;
      default:
        for (uint32_t i = rng.below(4); i && d.size() < (1u << 20); i--) d.push_back(uint8_t(rng.next()));
        break;
    }
  }
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: host_fuzz <seed dir> <mutations per seed> <rng seed>\n");
    return 2;
  }
  const std::string dir = argv[1];
  const long per_seed = atol(argv[2]);
  Rng rng{0x9E3779B97F4A7C15ull ^ strtoull(argv[3], nullptr, 0)};
  std::vector<std::pair<std::string, std::vector<uint8_t>>> seeds;
  if (DIR* dp = opendir(dir.c_str())) {
    while (dirent* e = readdir(dp)) {
      if (e->d_name[0] == '.') continue;
      const std::string path = dir + "/" + e->d_name;
      if (FILE* f = fopen(path.c_str(), "rb")) {
        std::vector<uint8_t> d;
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
        fclose(f);
        if (d.size() <= (4u << 20)) seeds.emplace_back(e->d_name, std::move(d));
      }
    }
    closedir(dp);
  }
  if (seeds.empty()) {
    fprintf(stderr, "no seeds in %s\n", dir.c_str());
    return 2;
  }
  // (pointers this library never handed out: ignored, never dereferenced -- ASan would see a read in front of them)
  {
    int on_stack = 0;
    milzma_free(&on_stack);
    void* foreign = malloc(32);
    milzma_free(foreign);
    free(foreign);
    milzma_free(nullptr);
  }
  for (const auto& s : seeds)
    if (!run_one(s.second, rng, s.first.c_str())) return 1;
  for (const auto& s : seeds) {
    const long reps = s.second.size() > (256u << 10) ? std::min<long>(per_seed, 20) : per_seed;  // (the 3 MB fixture: a few times)
    for (long i = 0; i < reps; i++) {
      std::vector<uint8_t> d = s.second;
      mutate(d, rng);
      const std::string what = s.first + " mutation " + std::to_string(i);
      if (!run_one(d, rng, what.c_str())) return 1;
    }
  }
  // planner and message renderer on arbitrary arguments
  for (int i = 0; i < 2000; i++) {
    const uint32_t n = rng.below(40), parts = 1 + rng.below(9);
    std::vector<uint64_t> w(n);
    std::vector<uint32_t> grp(n), part(n + 1, 0xFFFFFFFFu);
    for (uint32_t k = 0; k < n; k++) {
      w[k] = rng.next();
      grp[k] = rng.below(4) ? 0 : rng.below(5);
    }
    if (milzma_partition(w.data(), (i & 1) ? grp.data() : nullptr, n, parts, part.data()) != MILZMA_OK) return 1;
    for (uint32_t k = 0; k < n; k++)
      if (part[k] >= parts) {
        printf("MISMATCH partition: item %u -> part %u of %u\n", k, part[k], parts);
        return 1;
      }
    milzma_result r;
    memset(&r, 0, sizeof r);
    r.status = rng.below(40);
    r.err_a = (uint64_t(rng.next()) << 32) | rng.next();
    r.err_b = rng.next();
    char msg[32 + 388];
    (void)milzma_result_message(&r, rng.below(2), msg, 1 + rng.below(sizeof msg - 1));
  }
  const size_t pooled = milzma_pool_trim(0);
  if (pooled != 0) {
    printf("MISMATCH: %zu bytes still pooled after milzma_pool_trim(0)\n", pooled);
    return 1;
  }
  printf("ok cases=%" PRIu64 " walks_compared=%" PRIu64 " payload_errors=%" PRIu64 " plans_accepted=%" PRIu64 " headers_accepted=%" PRIu64 "\n",
         g.cases, g.compared, g.payload_errors, g.planned, g.headers_ok);
  return 0;
}
