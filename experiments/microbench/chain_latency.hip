// chain_latency.hip -- where do the cycles of one range-decoder decision go on gfx950?
//
// The symbol loop is a strictly dependent chain SGPR -> VALU -> v_readlane -> SALU -> SGPR per decision.  Each kernel
// below runs one variant of that chain (the real instructions, dummy data that keeps every dependency) ITER x 8 times
// per wave and reports cycles per chain step from s_memtime, for one wave per CU ("lone": pure latency) and for 16
// waves per CU ("full": what the bench runs at).  Differences between variants attribute the latency to the hops.
//
//   hipcc --offload-arch=gfx950 -O2 experiments/microbench/chain_latency.hip -o /tmp/chain_latency && /tmp/chain_latency
//
// Variants
//   B     form B tree decision: v_lshrrev, v_mul, v_sub, v_readlane x2, s_sub, s_cselect_b64, s_addc, s_cmp, s_cbranch
//   A     form A tree decision: v_lshrrev, v_mul, s_nop, v_readlane, s_sub x2, s_cselect x2, s_addc, s_cmp, s_cbranch
//   Bs    form B with range >> 11 on the scalar side (s_lshr + v_mul with a scalar operand)
//   S     the scalar tail alone (s_sub, s_cselect_b64, s_addc, s_cmp, s_cbranch), bound constant
//   V3    three dependent vector instructions (v_lshrrev, v_mul, v_sub)
//   VS    vector -> scalar hop: v_readlane then a scalar instruction that needs it, then back into the vector op
//   SV    scalar -> vector hop: s_add then a vector instruction that reads it, v_readlane back
//   B+6   B followed by six independent vector instructions (what a deferred probability update costs today);
//   B~6   the same 17 instructions with the independent ones placed in the chain's gaps
//   S1    one dependent scalar instruction (s_add), V1 one dependent vector instruction (v_add): the issue floor
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int ITER = 2000;

#define REP8(x) x x x x x x x x

template <int VAR>
__global__ __launch_bounds__(64, 8) void chain(uint64_t* out, uint32_t seed) {
  uint32_t range = 0xF0000000u | seed, code = 0x12345678u ^ seed, sym = 1, prob = 1024 + (threadIdx.x & 7);
  uint32_t vt = 0, vb = 0, vr = prob, x0 = prob, x1 = prob + 1, x2 = 3, x3 = 4;
  uint64_t t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int i = 0; i < ITER; i++) {
    if constexpr (VAR == 0) {  // B
      asm volatile(REP8(
          "v_lshrrev_b32 %[vt], 11, s66\n\t"
          "v_mul_u32_u24 %[vb], %[vt], %[p]\n\t"
          "v_sub_u32 %[vr], s66, %[vb]\n\t"
          "v_readlane_b32 s66, %[vb], %[sym]\n\t"
          "v_readlane_b32 s74, %[vr], %[sym]\n\t"
          "s_sub_u32 s75, s67, s66\n\t"
          "s_cselect_b64 s[66:67], s[66:67], s[74:75]\n\t"
          "s_addc_u32 %[sym], %[sym], %[sym]\n\t"
          "s_or_b32 s66, s66, 0x40000000\n\t"  // (keeps the dummy range large: not part of the real chain)
          "s_cmp_lt_u32 s66, 0x1000000\n\t"
          "s_cbranch_scc1 1f\n\t"
          "1:\n\t")
          : [vt] "=&v"(vt), [vb] "=&v"(vb), [vr] "=&v"(vr), [sym] "+s"(sym)
          : [p] "v"(prob)
          : "s66", "s67", "s74", "s75", "scc");
    } else if constexpr (VAR == 1) {  // A
      asm volatile(REP8(
          "v_lshrrev_b32 %[vt], 11, s66\n\t"
          "v_mul_u32_u24 %[vb], %[vt], %[p]\n\t"
          "s_nop 0\n\t"
          "v_readlane_b32 s73, %[vb], %[sym]\n\t"
          "s_sub_u32 s74, s66, s73\n\t"
          "s_sub_u32 s75, s67, s73\n\t"
          "s_cselect_b32 s66, s73, s74\n\t"
          "s_cselect_b32 s67, s67, s75\n\t"
          "s_addc_u32 %[sym], %[sym], %[sym]\n\t"
          "s_or_b32 s66, s66, 0x40000000\n\t"
          "s_cmp_lt_u32 s66, 0x1000000\n\t"
          "s_cbranch_scc1 1f\n\t"
          "1:\n\t")
          : [vt] "=&v"(vt), [vb] "=&v"(vb), [sym] "+s"(sym)
          : [p] "v"(prob)
          : "s66", "s67", "s73", "s74", "s75", "scc");
    } else if constexpr (VAR == 2) {  // Bs
      asm volatile(REP8(
          "s_lshr_b32 s72, s66, 11\n\t"
          "v_mul_u32_u24 %[vb], s72, %[p]\n\t"
          "v_sub_u32 %[vr], s66, %[vb]\n\t"
          "v_readlane_b32 s66, %[vb], %[sym]\n\t"
          "v_readlane_b32 s74, %[vr], %[sym]\n\t"
          "s_sub_u32 s75, s67, s66\n\t"
          "s_cselect_b64 s[66:67], s[66:67], s[74:75]\n\t"
          "s_addc_u32 %[sym], %[sym], %[sym]\n\t"
          "s_or_b32 s66, s66, 0x40000000\n\t"
          "s_cmp_lt_u32 s66, 0x1000000\n\t"
          "s_cbranch_scc1 1f\n\t"
          "1:\n\t")
          : [vb] "=&v"(vb), [vr] "=&v"(vr), [sym] "+s"(sym)
          : [p] "v"(prob)
          : "s66", "s67", "s72", "s74", "s75", "scc");
    } else if constexpr (VAR == 3) {  // S
      asm volatile(REP8(
          "s_sub_u32 s75, s67, s66\n\t"
          "s_cselect_b64 s[66:67], s[66:67], s[74:75]\n\t"
          "s_addc_u32 %[sym], %[sym], %[sym]\n\t"
          "s_or_b32 s66, s66, 0x40000000\n\t"
          "s_cmp_lt_u32 s66, 0x1000000\n\t"
          "s_cbranch_scc1 1f\n\t"
          "1:\n\t")
          : [sym] "+s"(sym)
          :
          : "s66", "s67", "s74", "s75", "scc");
    } else if constexpr (VAR == 4) {  // V3
      asm volatile(REP8(
          "v_lshrrev_b32 %[vt], 11, %[vr]\n\t"
          "v_mul_u32_u24 %[vb], %[vt], %[p]\n\t"
          "v_sub_u32 %[vr], %[p], %[vb]\n\t")
          : [vt] "=&v"(vt), [vb] "=&v"(vb), [vr] "+v"(vr)
          : [p] "v"(prob));
    } else if constexpr (VAR == 5) {  // VS: v op -> readlane -> s op -> v op reading it
      asm volatile(REP8(
          "v_add_u32 %[vb], s73, %[p]\n\t"
          "s_nop 0\n\t"
          "v_readlane_b32 s73, %[vb], 3\n\t"
          "s_add_u32 s73, s73, 1\n\t")
          : [vb] "=&v"(vb)
          : [p] "v"(prob)
          : "s73", "scc");
    } else if constexpr (VAR == 6) {  // SV: s op -> v op (no readlane: the value never returns) + independent s chain
      asm volatile(REP8(
          "s_add_u32 s73, s73, 1\n\t"
          "v_add_u32 %[vb], s73, %[p]\n\t")
          : [vb] "=&v"(vb)
          : [p] "v"(prob)
          : "s73", "scc");
    } else if constexpr (VAR == 9) {  // B + 6 independent vector instructions after the chain
      asm volatile(REP8(
          "v_lshrrev_b32 %[vt], 11, s66\n\t"
          "v_mul_u32_u24 %[vb], %[vt], %[p]\n\t"
          "v_sub_u32 %[vr], s66, %[vb]\n\t"
          "v_readlane_b32 s66, %[vb], %[sym]\n\t"
          "v_readlane_b32 s74, %[vr], %[sym]\n\t"
          "s_sub_u32 s75, s67, s66\n\t"
          "s_cselect_b64 s[66:67], s[66:67], s[74:75]\n\t"
          "s_addc_u32 %[sym], %[sym], %[sym]\n\t"
          "s_or_b32 s66, s66, 0x40000000\n\t"
          "s_cmp_lt_u32 s66, 0x1000000\n\t"
          "s_cbranch_scc1 1f\n\t"
          "1:\n\t"
          "v_mad_u32_u24 %[x0], %[x0], 31, %[p]\n\t"
          "v_lshrrev_b32 %[x0], 5, %[x0]\n\t"
          "v_mad_u32_u24 %[x1], %[x1], 31, %[p]\n\t"
          "v_lshrrev_b32 %[x1], 5, %[x1]\n\t"
          "v_add_u32 %[x2], %[x2], %[p]\n\t"
          "v_xor_b32 %[x3], %[x3], %[p]\n\t")
          : [vt] "=&v"(vt), [vb] "=&v"(vb), [vr] "=&v"(vr), [sym] "+s"(sym), [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3)
          : [p] "v"(prob)
          : "s66", "s67", "s74", "s75", "scc");
    } else if constexpr (VAR == 10) {  // the same 17 instructions, the independent ones in the chain's gaps
      asm volatile(REP8(
          "v_lshrrev_b32 %[vt], 11, s66\n\t"
          "v_mad_u32_u24 %[x0], %[x0], 31, %[p]\n\t"
          "v_mul_u32_u24 %[vb], %[vt], %[p]\n\t"
          "v_mad_u32_u24 %[x1], %[x1], 31, %[p]\n\t"
          "v_sub_u32 %[vr], s66, %[vb]\n\t"
          "v_readlane_b32 s66, %[vb], %[sym]\n\t"
          "v_readlane_b32 s74, %[vr], %[sym]\n\t"
          "v_lshrrev_b32 %[x0], 5, %[x0]\n\t"
          "v_lshrrev_b32 %[x1], 5, %[x1]\n\t"
          "s_sub_u32 s75, s67, s66\n\t"
          "v_add_u32 %[x2], %[x2], %[p]\n\t"
          "s_cselect_b64 s[66:67], s[66:67], s[74:75]\n\t"
          "v_xor_b32 %[x3], %[x3], %[p]\n\t"
          "s_addc_u32 %[sym], %[sym], %[sym]\n\t"
          "s_or_b32 s66, s66, 0x40000000\n\t"
          "s_cmp_lt_u32 s66, 0x1000000\n\t"
          "s_cbranch_scc1 1f\n\t"
          "1:\n\t")
          : [vt] "=&v"(vt), [vb] "=&v"(vb), [vr] "=&v"(vr), [sym] "+s"(sym), [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3)
          : [p] "v"(prob)
          : "s66", "s67", "s74", "s75", "scc");
    } else if constexpr (VAR == 7) {  // S1
      asm volatile(REP8("s_add_u32 s73, s73, 1\n\t") ::: "s73", "scc");
    } else {  // V1
      asm volatile(REP8("v_add_u32 %[vb], 1, %[vb]\n\t") : [vb] "+v"(prob));
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (uint64_t(range ^ code ^ sym ^ prob ^ vt ^ vb ^ vr ^ x0 ^ x1 ^ x2 ^ x3) & 0);
}

template <int VAR>
int run(const char* name, int steps_per_rep, uint64_t* d_out, std::vector<uint64_t>& h) {
  for (int waves : {256, 4096}) {
    chain<VAR><<<waves, 64>>>(d_out, 7);
    CHECK(hipDeviceSynchronize());
    chain<VAR><<<waves, 64>>>(d_out, 7);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), d_out, waves * 8, hipMemcpyDeviceToHost));
    double sum = 0, mx = 0;
    for (int i = 0; i < waves; i++) {
      sum += double(h[i]);
      if (double(h[i]) > mx) mx = double(h[i]);
    }
    const double per = sum / waves / (double(ITER) * 8);
    printf("%-3s %-5s  %7.1f cycles per chain step (%d instructions -> %.1f cycles each), slowest wave %.1f\n", name,
           waves == 256 ? "lone" : "full", per, steps_per_rep, per / steps_per_rep, mx / (double(ITER) * 8));
  }
  return 0;
}

template <int VAR>
int sweep(const char* name, int steps_per_rep, uint64_t* d_out, std::vector<uint64_t>& h) {
  // more chains per SIMD: does the CU issue more per cycle (latency-bound) or the same (issue-bound)?
  for (int per_simd : {1, 2, 3, 4, 5, 6, 8}) {
    const int waves = 256 * 4 * per_simd;
    chain<VAR><<<waves, 64>>>(d_out, 7);
    CHECK(hipDeviceSynchronize());
    chain<VAR><<<waves, 64>>>(d_out, 7);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), d_out, waves * 8, hipMemcpyDeviceToHost));
    double sum = 0;
    for (int i = 0; i < waves; i++) sum += double(h[i]);
    const double per = sum / waves / (double(ITER) * 8);
    printf("%-3s %d waves/SIMD: %7.1f cycles per chain step per wave, %.2f instructions per cycle per CU\n", name, per_simd, per,
           4.0 * per_simd * steps_per_rep / per);
  }
  return 0;
}

int main() {
  uint64_t* d_out;
  CHECK(hipMalloc(&d_out, 8192 * 8));
  std::vector<uint64_t> h(8192);
  if (run<0>("B", 11, d_out, h) || run<1>("A", 12, d_out, h) || run<2>("Bs", 11, d_out, h) || run<3>("S", 6, d_out, h) ||
      run<4>("V3", 3, d_out, h) || run<5>("VS", 4, d_out, h) || run<6>("SV", 2, d_out, h) || run<7>("S1", 1, d_out, h) ||
      run<8>("V1", 1, d_out, h))
    return 1;
  if (run<9>("B+6", 17, d_out, h) || run<10>("B~6", 17, d_out, h)) return 1;
  if (sweep<0>("B", 11, d_out, h) || sweep<3>("S", 6, d_out, h) || sweep<4>("V3", 3, d_out, h) || sweep<7>("S1", 1, d_out, h) ||
      sweep<8>("V1", 1, d_out, h))
    return 1;
  return 0;
}
