// Micro-benchmark: which inner-loop style decodes adaptive range-coder bits fastest on gfx950?
//
// Every kernel decodes NBITS binary decisions per wave with the real LZMA arithmetic
// (11-bit adaptive probability, one normalisation byte at most per bit) over a synthetic
// input window, walking 6-level bit trees.  One wave = one independent serial chain, like
// one LZMA stream.  What differs is where the arithmetic runs and where the model lives:
//
//   A  SALU core (hand-ordered s_* sequence), model in a VGPR spread over lanes
//      (v_readlane / v_writelane)
//   B  SALU core, model in LDS (ds_read_u16 + v_readfirstlane, ds_write_b16)
//   C  VALU core computed redundantly by all lanes, model in LDS
//   D  VALU core, two independent chains per wave (lanes 0-31 / 32-63), model in LDS
//   V  like C but with the chain's uniformity hidden from hipcc so it really runs on the VALU
//   E  compiler-scheduled scalar C++ (what hipcc makes of the plain source), model in LDS
//
// Output: one line per (style, waves, lds bytes) with aggregate Gbit/s and cycles/bit/chain.
// Also probes same-wave global store -> load ordering without a fence (needed by the LZ77
// match copy when the output buffer doubles as the dictionary).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ void wl(uint32_t& reg, uint32_t v, uint32_t lane) {
  asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(reg) : "s"(v), "s"(lane));
}

// SALU arithmetic core of one decision.  In: range, code, p, tree. Out: updated; tree = 2*tree+bit.
__device__ __forceinline__ void salu_core(uint32_t& range, uint32_t& code, uint32_t& p, uint32_t& tree) {
  uint32_t bound, r1, c1, q, p1, u;
  asm volatile(
      "s_lshr_b32 %[bound], %[range], 11\n\t"
      "s_mul_i32 %[bound], %[bound], %[p]\n\t"
      "s_sub_u32 %[r1], %[range], %[bound]\n\t"
      "s_sub_u32 %[c1], %[code], %[bound]\n\t"
      "s_lshr_b32 %[q], %[p], 5\n\t"
      "s_sub_u32 %[p1], %[p], %[q]\n\t"
      "s_sub_u32 %[u], 0x800, %[p]\n\t"
      "s_lshr_b32 %[u], %[u], 5\n\t"
      "s_add_u32 %[u], %[p], %[u]\n\t"
      "s_cmp_ge_u32 %[code], %[bound]\n\t"
      "s_cselect_b32 %[range], %[r1], %[bound]\n\t"
      "s_cselect_b32 %[code], %[c1], %[code]\n\t"
      "s_cselect_b32 %[p], %[p1], %[u]\n\t"
      "s_addc_u32 %[tree], %[tree], %[tree]"
      : [range] "+s"(range), [code] "+s"(code), [p] "+s"(p), [tree] "+s"(tree), [bound] "=&s"(bound),
        [r1] "=&s"(r1), [c1] "=&s"(c1), [q] "=&s"(q), [p1] "=&s"(p1), [u] "=&s"(u)
      :
      : "scc");
}

struct Feed {  // 256-byte input window held one dword per lane
  uint32_t win;
  uint32_t pos;
};
__device__ __forceinline__ uint32_t feed_byte_s(Feed& f) {
  uint32_t w = rl(f.win, (f.pos >> 2) & 63);
  uint32_t b = (w >> ((f.pos & 3) * 8)) & 0xff;
  f.pos++;
  return b;
}

// ---------------- style A: SALU core, model in lanes of VGPRs ----------------
__global__ __launch_bounds__(64) void k_style_a(const uint32_t* in, uint32_t* out, int nsym) {
  extern __shared__ uint16_t lds[];
  const uint32_t lane = threadIdx.x;
  Feed f{in[(blockIdx.x * 64 + lane) & 0xffff], 0};
  uint32_t t0 = 0x400, t1 = 0x400, t2 = 0x400, t3 = 0x400;
  uint32_t range = 0xffffffffu, code = __builtin_amdgcn_readfirstlane(in[blockIdx.x & 0xffff]) >> 1;
  uint32_t acc = 0;
  for (int s = 0; s < nsym; s++) {
#define TREE6(T)                                                                     \
  {                                                                                  \
    uint32_t tree = 1;                                                               \
    _Pragma("unroll") for (int d = 0; d < 6; d++) {                                  \
      uint32_t idx = tree;                                                           \
      uint32_t p = rl(T, idx);                                                       \
      salu_core(range, code, p, tree);                                               \
      wl(T, p, idx);                                                                 \
      if (__builtin_expect(range < (1u << 24), 0)) {                                 \
        range <<= 8;                                                                 \
        code = (code << 8) | feed_byte_s(f);                                         \
      }                                                                              \
    }                                                                                \
    acc += tree;                                                                     \
  }
    TREE6(t0) TREE6(t1) TREE6(t2) TREE6(t3)
  }
  if (lane == 0) { out[blockIdx.x * 4] = acc; out[blockIdx.x * 4 + 1] = range ^ code; }
  if (nsym < 0) { lds[lane] = (uint16_t)t0; out[lane] = lds[lane ^ 1] + t1 + t2 + t3; }
}

// ---------------- style B: SALU core, model in LDS ----------------
__global__ __launch_bounds__(64) void k_style_b(const uint32_t* in, uint32_t* out, int nsym) {
  extern __shared__ uint16_t lds[];
  const uint32_t lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) lds[i] = 0x400;
  __syncthreads();
  Feed f{in[(blockIdx.x * 64 + lane) & 0xffff], 0};
  uint32_t range = 0xffffffffu, code = __builtin_amdgcn_readfirstlane(in[blockIdx.x & 0xffff]) >> 1;
  uint32_t acc = 0;
  for (int s = 0; s < nsym; s++) {
#pragma unroll
    for (int T = 0; T < 4; T++) {
      uint32_t tree = 1;
#pragma unroll
      for (int d = 0; d < 6; d++) {
        uint32_t idx = T * 64 + tree;
        uint32_t p = __builtin_amdgcn_readfirstlane((uint32_t)lds[idx]);
        salu_core(range, code, p, tree);
        lds[idx] = (uint16_t)p;
        if (__builtin_expect(range < (1u << 24), 0)) {
          range <<= 8;
          code = (code << 8) | feed_byte_s(f);
        }
      }
      acc += tree;
    }
  }
  if (lane == 0) { out[blockIdx.x * 4] = acc; out[blockIdx.x * 4 + 1] = range ^ code; }
}

// ---------------- style C/D: VALU core, per-lane state (C: all lanes same chain; D: 2 chains) --------
template <int CHAINS, bool FORCE_VALU = false>
__global__ __launch_bounds__(64) void k_style_cd(const uint32_t* in, uint32_t* out, int nsym) {
  extern __shared__ uint16_t lds[];
  const uint32_t lane = threadIdx.x;
  const uint32_t chain = (CHAINS == 1) ? 0 : (lane >> 5);
  uint16_t* model = lds + chain * 256;
  uint8_t* win = (uint8_t*)(lds + 1024) + chain * 256;  // input window in LDS, byte addressed
  for (int i = lane; i < 256 * CHAINS; i += 64) lds[i] = 0x400;
  for (int i = lane; i < 64 * CHAINS; i += 64) ((uint32_t*)(lds + 1024))[i] = in[(blockIdx.x * 128 + i) & 0xffff];
  __syncthreads();
  uint32_t pos = 0;
  uint32_t range = 0xffffffffu, code = in[(blockIdx.x * 2 + chain) & 0xffff] >> 1;
  if (FORCE_VALU) asm volatile("" : "+v"(range), "+v"(code));  // hide uniformity: keep the chain on the VALU
  uint32_t acc = 0;
  for (int s = 0; s < nsym; s++) {
#pragma unroll
    for (int T = 0; T < 4; T++) {
      uint32_t tree = 1;
#pragma unroll
      for (int d = 0; d < 6; d++) {
        uint32_t idx = T * 64 + tree;
        uint32_t p = model[idx];
        uint32_t bound = (range >> 11) * p;
        bool is1 = code >= bound;
        range = is1 ? range - bound : bound;
        code = is1 ? code - bound : code;
        uint32_t p1 = p - (p >> 5), p0 = p + ((2048 - p) >> 5);
        p = is1 ? p1 : p0;
        model[idx] = (uint16_t)p;
        tree = (tree << 1) | (is1 ? 1u : 0u);
        if (range < (1u << 24)) {
          range <<= 8;
          code = (code << 8) | win[pos & 255];
          pos++;
        }
      }
      acc += tree;
    }
  }
  if ((lane & 31) == 0) { out[blockIdx.x * 4 + chain * 2] = acc; out[blockIdx.x * 4 + chain * 2 + 1] = range ^ code; }
}

// ---------------- style E: plain C++ with readfirstlane'd values (compiler-scheduled SALU) -------
__global__ __launch_bounds__(64) void k_style_e(const uint32_t* in, uint32_t* out, int nsym) {
  extern __shared__ uint16_t lds[];
  const uint32_t lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) lds[i] = 0x400;
  __syncthreads();
  Feed f{in[(blockIdx.x * 64 + lane) & 0xffff], 0};
  uint32_t range = 0xffffffffu, code = __builtin_amdgcn_readfirstlane(in[blockIdx.x & 0xffff]) >> 1;
  uint32_t acc = 0;
  for (int s = 0; s < nsym; s++) {
#pragma unroll
    for (int T = 0; T < 4; T++) {
      uint32_t tree = 1;
#pragma unroll
      for (int d = 0; d < 6; d++) {
        uint32_t idx = T * 64 + tree;
        uint32_t p = __builtin_amdgcn_readfirstlane((uint32_t)lds[idx]);
        uint32_t bound = (range >> 11) * p;
        if (code < bound) {
          range = bound;
          p += (2048 - p) >> 5;
          tree = tree << 1;
        } else {
          range -= bound;
          code -= bound;
          p -= p >> 5;
          tree = (tree << 1) | 1;
        }
        lds[idx] = (uint16_t)p;
        if (__builtin_expect(range < (1u << 24), 0)) {
          range <<= 8;
          code = (code << 8) | feed_byte_s(f);
        }
      }
      acc += tree;
    }
  }
  if (lane == 0) { out[blockIdx.x * 4] = acc; out[blockIdx.x * 4 + 1] = range ^ code; }
}

// ---------------- probe: same-wave global store -> load without any fence ----------------
// Lane i stores a value, then lane (i+1)&63 loads it straight away (plain ops, no s_waitcnt
// inserted by us, no fence).  Counts stale reads.  Mimics LZ77 copies whose source is the
// bytes just written by the same wave.
__global__ __launch_bounds__(64) void k_store_load_probe(uint8_t* buf, uint32_t* stale, int iters) {
  const uint32_t lane = threadIdx.x;
  uint8_t* mine = buf + (size_t)blockIdx.x * 65536;
  uint32_t bad = 0;
  for (int it = 0; it < iters; it++) {
    uint32_t off = (it * 64) & 65535 & ~63u;
    uint8_t v = (uint8_t)(it * 7 + lane * 3 + blockIdx.x);
    mine[off + lane] = v;
    uint32_t other = (lane + 1) & 63;
    uint8_t got = *(volatile uint8_t*)(mine + off + other);
    uint8_t want = (uint8_t)(it * 7 + other * 3 + blockIdx.x);
    bad += (got != want);
  }
  atomicAdd(stale, bad);
}

// same, but the reader is the byte-serial overlapped copy pattern: out[pos+i] = out[pos+i-dist]
// with small dist, done in chunks of dist lanes (each chunk depends on the previous chunk's stores)
__global__ __launch_bounds__(64) void k_overlap_probe(uint8_t* buf, uint32_t* stale, int iters) {
  const uint32_t lane = threadIdx.x;
  uint8_t* mine = buf + (size_t)blockIdx.x * 65536;
  if (lane < 8) mine[lane] = (uint8_t)(lane * 31 + 1 + blockIdx.x);
  uint32_t pos = 8;
  for (int it = 0; it < iters && pos + 64 < 65536; it++) {
    uint32_t dist = 1 + (it % 7);
    uint32_t len = 5 + (it % 50);
    for (uint32_t done = 0; done < len;) {  // dist lanes at a time
      uint32_t n = min(dist, len - done);
      if (lane < n) mine[pos + done + lane] = mine[pos + done + lane - dist];
      done += n;
    }
    pos += len;
  }
  __syncthreads();
  // verify against a serial recomputation by lane 0
  if (lane == 0) {
    uint32_t bad = 0;
    uint32_t p = 8;
    uint8_t ref[8];
    for (int i = 0; i < 8; i++) ref[i] = (uint8_t)(i * 31 + 1 + blockIdx.x);
    (void)ref;
    for (int it = 0; it < iters && p + 64 < 65536; it++) {
      uint32_t dist = 1 + (it % 7);
      uint32_t len = 5 + (it % 50);
      for (uint32_t i = 0; i < len; i++) bad += (mine[p + i] != mine[p + i - dist]);
      p += len;
    }
    atomicAdd(stale, bad);
  }
}

template <typename K>
static void run(const char* name, K kernel, int waves, int lds_bytes, int nsym, int chains, const uint32_t* d_in, uint32_t* d_out) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kernel, dim3(waves), dim3(64), lds_bytes, 0, d_in, d_out, nsym / 8);  // warm
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kernel, dim3(waves), dim3(64), lds_bytes, 0, d_in, d_out, nsym);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  double bits = (double)waves * chains * nsym * 24.0;
  double gbps = bits / (ms * 1e-3) / 1e9;
  // cycles per bit per chain assuming all `waves` resident at once and 2.4 GHz
  double cyc = (ms * 1e-3) * 2.4e9 / ((double)nsym * 24.0);
  printf("%-6s waves=%5d lds=%6d chains/wave=%d  %8.3f ms  %8.2f Gbit/s  %7.1f cyc/bit/chain(if all resident)\n", name, waves,
         lds_bytes, chains, ms, gbps, cyc);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  clock=%d kHz  LDS/block=%zu\n", prop.name, prop.multiProcessorCount, prop.clockRate,
         prop.sharedMemPerBlock);
  std::vector<uint32_t> h(65536);
  uint32_t x = 12345;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
  uint32_t *d_in, *d_out;
  CHECK(hipMalloc(&d_in, h.size() * 4));
  CHECK(hipMalloc(&d_out, 65536 * 4 * 4));
  CHECK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipFuncSetAttribute((const void*)k_style_b, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHECK(hipFuncSetAttribute((const void*)k_style_e, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHECK(hipFuncSetAttribute((const void*)k_style_a, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHECK(hipFuncSetAttribute((const void*)k_style_cd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHECK(hipFuncSetAttribute((const void*)k_style_cd<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHECK(hipFuncSetAttribute((const void*)k_style_cd<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));

  const int nsym = 4000;  // 96k decisions per chain
  // latency probe: one wave per CU
  run("A", k_style_a, 256, 0, nsym, 1, d_in, d_out);
  run("B", k_style_b, 256, 4096, nsym, 1, d_in, d_out);
  run("C", k_style_cd<1>, 256, 4096, nsym, 1, d_in, d_out);
  run("D", k_style_cd<2>, 256, 4096, nsym, 2, d_in, d_out);
  run("E", k_style_e, 256, 4096, nsym, 1, d_in, d_out);
  run("V", k_style_cd<1, true>, 256, 4096, nsym, 1, d_in, d_out);
  // occupancy sweeps; LDS bytes per block emulate the model footprint (16 KiB -> 10 waves/CU)
  int wave_counts[] = {1024, 2560, 4096, 8192};
  for (int w : wave_counts) {
    run("A", k_style_a, w, 0, nsym, 1, d_in, d_out);
    run("A", k_style_a, w, 12288, nsym, 1, d_in, d_out);
    run("B", k_style_b, w, 16384, nsym, 1, d_in, d_out);
    run("B", k_style_b, w, 4096, nsym, 1, d_in, d_out);
    run("C", k_style_cd<1>, w, 16384, nsym, 1, d_in, d_out);
    run("C", k_style_cd<1>, w, 4096, nsym, 1, d_in, d_out);
    run("D", k_style_cd<2>, w / 2, 32768, nsym, 2, d_in, d_out);
    run("D", k_style_cd<2>, w / 2, 4096, nsym, 2, d_in, d_out);
    run("E", k_style_e, w, 16384, nsym, 1, d_in, d_out);
    run("V", k_style_cd<1, true>, w, 16384, nsym, 1, d_in, d_out);
    run("V", k_style_cd<1, true>, w, 4096, nsym, 1, d_in, d_out);
  }

  // ordering probes
  uint8_t* d_buf;
  uint32_t* d_stale;
  CHECK(hipMalloc(&d_buf, (size_t)2048 * 65536));
  CHECK(hipMalloc(&d_stale, 4));
  for (int rep = 0; rep < 2; rep++) {
    CHECK(hipMemset(d_stale, 0, 4));
    hipLaunchKernelGGL(k_store_load_probe, dim3(2048), dim3(64), 0, 0, d_buf, d_stale, 20000);
    CHECK(hipDeviceSynchronize());
    uint32_t stale = 0;
    CHECK(hipMemcpy(&stale, d_stale, 4, hipMemcpyDeviceToHost));
    printf("store->load probe (no fence): stale reads = %u of %llu\n", stale, 2048ull * 64 * 20000);
    CHECK(hipMemset(d_stale, 0, 4));
    hipLaunchKernelGGL(k_overlap_probe, dim3(2048), dim3(64), 0, 0, d_buf, d_stale, 2000);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&stale, d_stale, 4, hipMemcpyDeviceToHost));
    printf("overlapped-copy probe (no fence): mismatching bytes = %u\n", stale);
  }
  return 0;
}
