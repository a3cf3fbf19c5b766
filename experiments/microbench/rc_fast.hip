// Micro-benchmark 2: candidate per-bit sequences for the fast kernel (lane-resident model,
// range/code as wave-uniform values held in VGPRs, probability update under a one-lane EXEC).
//   F  branch on the decoded bit (two short paths), ~14.5 instructions per bit
//   G  branchless (v_cndmask / v_min), ~20 instructions per bit
//   P  F with a packed (2 x u16 per dword) model register, as used for the literal table
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

struct RC { uint32_t range, code, win, off, rem, eof; };

#define NORM_ASM                                  \
      "v_cmp_gt_u32 vcc, %[top], %[range]\n\t"    \
      "s_cbranch_vccz 4f\n\t"                     \
      "v_lshlrev_b32 %[range], 8, %[range]\n\t"   \
      "s_cmp_eq_u32 %[rem], 0\n\t"                \
      "s_cbranch_scc1 3f\n\t"                     \
      "s_lshr_b32 %[ss], %[off], 2\n\t"           \
      "v_readlane_b32 %[sw], %[win], %[ss]\n\t"   \
      "s_lshl_b32 %[ss], %[off], 3\n\t"           \
      "s_lshr_b32 %[sw], %[sw], %[ss]\n\t"        \
      "s_and_b32 %[sw], %[sw], 0xff\n\t"          \
      "v_lshl_or_b32 %[code], %[code], 8, %[sw]\n\t" \
      "s_add_u32 %[off], %[off], 1\n\t"           \
      "s_sub_u32 %[rem], %[rem], 1\n\t"           \
      "s_branch 4f\n"                             \
      "3:\n\t"                                    \
      "v_lshlrev_b32 %[code], 8, %[code]\n\t"     \
      "s_mov_b32 %[eof], 1\n"                     \
      "4:"

__device__ __forceinline__ void bitF(RC& rc, uint32_t& T, uint32_t idx, uint32_t& sym, uint32_t top) {
  uint32_t sp, sw, ss, vb, vt;
  asm volatile(
      "v_readlane_b32 %[sp], %[T], %[idx]\n\t"
      "v_lshrrev_b32 %[vb], 11, %[range]\n\t"
      "s_lshl_b64 exec, 1, %[idx]\n\t"
      "v_mul_u32_u24 %[vb], %[vb], %[sp]\n\t"
      "v_cmp_lt_u32 vcc, %[code], %[vb]\n\t"
      "s_cbranch_vccz 1f\n\t"
      "v_sub_u32 %[vt], 0x800, %[T]\n\t"
      "v_lshrrev_b32 %[vt], 5, %[vt]\n\t"
      "v_add_u32 %[T], %[T], %[vt]\n\t"
      "s_mov_b64 exec, -1\n\t"
      "v_mov_b32 %[range], %[vb]\n\t"
      "s_lshl_b32 %[sym], %[sym], 1\n\t"
      "s_branch 2f\n"
      "1:\n\t"
      "v_lshrrev_b32 %[vt], 5, %[T]\n\t"
      "v_sub_u32 %[T], %[T], %[vt]\n\t"
      "s_mov_b64 exec, -1\n\t"
      "v_sub_u32 %[range], %[range], %[vb]\n\t"
      "v_sub_u32 %[code], %[code], %[vb]\n\t"
      "s_lshl1_add_u32 %[sym], %[sym], 1\n"
      "2:\n\t" NORM_ASM
      : [T] "+v"(T), [range] "+v"(rc.range), [code] "+v"(rc.code), [sym] "+s"(sym), [off] "+s"(rc.off),
        [rem] "+s"(rc.rem), [eof] "+s"(rc.eof), [sp] "=&s"(sp), [sw] "=&s"(sw), [ss] "=&s"(ss), [vb] "=&v"(vb), [vt] "=&v"(vt)
      : [idx] "s"(idx), [win] "v"(rc.win), [top] "s"(top)
      : "vcc", "scc");
}

__device__ __forceinline__ void bitG(RC& rc, uint32_t& T, uint32_t idx, uint32_t& sym, uint32_t top) {
  uint32_t sp, sw, ss, vb, vt, vu;
  asm volatile(
      "v_readlane_b32 %[sp], %[T], %[idx]\n\t"
      "v_lshrrev_b32 %[vb], 11, %[range]\n\t"
      "v_mul_u32_u24 %[vb], %[vb], %[sp]\n\t"
      "v_cmp_ge_u32 vcc, %[code], %[vb]\n\t"
      "v_sub_u32 %[vt], %[range], %[vb]\n\t"
      "v_sub_u32 %[vu], %[code], %[vb]\n\t"
      "v_cndmask_b32 %[range], %[vb], %[vt], vcc\n\t"
      "v_min_u32 %[code], %[code], %[vu]\n\t"
      "s_cmp_lg_u64 vcc, 0\n\t"
      "s_addc_u32 %[sym], %[sym], %[sym]\n\t"
      "s_lshl_b64 exec, 1, %[idx]\n\t"
      "v_cndmask_b32 %[vt], 31, 0, vcc\n\t"
      "v_cndmask_b32 %[vu], 64, 0, vcc\n\t"
      "v_add_u32 %[vt], %[T], %[vt]\n\t"
      "v_lshrrev_b32 %[vt], 5, %[vt]\n\t"
      "v_sub_u32 %[vt], %[vu], %[vt]\n\t"
      "v_add_u32 %[T], %[T], %[vt]\n\t"
      "s_mov_b64 exec, -1\n\t" NORM_ASM
      : [T] "+v"(T), [range] "+v"(rc.range), [code] "+v"(rc.code), [sym] "+s"(sym), [off] "+s"(rc.off),
        [rem] "+s"(rc.rem), [eof] "+s"(rc.eof), [sp] "=&s"(sp), [sw] "=&s"(sw), [ss] "=&s"(ss), [vb] "=&v"(vb), [vt] "=&v"(vt), [vu] "=&v"(vu)
      : [idx] "s"(idx), [win] "v"(rc.win), [top] "s"(top)
      : "vcc", "scc");
}

// packed model register: probability = 16-bit half `sh` (0 or 16) of lane idx
__device__ __forceinline__ void bitP(RC& rc, uint32_t& T, uint32_t idx, uint32_t sh, uint32_t& sym, uint32_t top) {
  uint32_t sp, sw, ss, vb;
  asm volatile(
      "v_readlane_b32 %[sp], %[T], %[idx]\n\t"
      "v_lshrrev_b32 %[vb], 11, %[range]\n\t"
      "s_lshr_b32 %[sp], %[sp], %[sh]\n\t"
      "s_and_b32 %[sp], %[sp], 0xffff\n\t"
      "s_lshl_b64 exec, 1, %[idx]\n\t"
      "v_mul_u32_u24 %[vb], %[vb], %[sp]\n\t"
      "v_cmp_lt_u32 vcc, %[code], %[vb]\n\t"
      "s_cbranch_vccz 1f\n\t"
      "s_sub_u32 %[sp], 0x800, %[sp]\n\t"
      "s_lshr_b32 %[sp], %[sp], 5\n\t"
      "s_lshl_b32 %[sp], %[sp], %[sh]\n\t"
      "v_add_u32 %[T], %[T], %[sp]\n\t"
      "s_mov_b64 exec, -1\n\t"
      "v_mov_b32 %[range], %[vb]\n\t"
      "s_lshl_b32 %[sym], %[sym], 1\n\t"
      "s_branch 2f\n"
      "1:\n\t"
      "s_lshr_b32 %[sp], %[sp], 5\n\t"
      "s_lshl_b32 %[sp], %[sp], %[sh]\n\t"
      "v_sub_u32 %[T], %[T], %[sp]\n\t"
      "s_mov_b64 exec, -1\n\t"
      "v_sub_u32 %[range], %[range], %[vb]\n\t"
      "v_sub_u32 %[code], %[code], %[vb]\n\t"
      "s_lshl1_add_u32 %[sym], %[sym], 1\n"
      "2:\n\t" NORM_ASM
      : [T] "+v"(T), [range] "+v"(rc.range), [code] "+v"(rc.code), [sym] "+s"(sym), [off] "+s"(rc.off),
        [rem] "+s"(rc.rem), [eof] "+s"(rc.eof), [sp] "=&s"(sp), [sw] "=&s"(sw), [ss] "=&s"(ss), [vb] "=&v"(vb)
      : [idx] "s"(idx), [sh] "s"(sh), [win] "v"(rc.win), [top] "s"(top)
      : "vcc", "scc");
}

template <int STYLE>
__global__ __launch_bounds__(64) void k_fast(const uint32_t* in, uint32_t* out, int nsym) {
  const uint32_t lane = threadIdx.x;
  RC rc;
  rc.win = in[(blockIdx.x * 64 + lane) & 0xffff];
  rc.range = 0xffffffffu;
  rc.code = in[blockIdx.x & 0xffff] >> 1;
  rc.off = 0; rc.rem = 1u << 30; rc.eof = 0;
  uint32_t t0 = STYLE == 2 ? 0x04000400u : 0x400u, t1 = t0, t2 = t0, t3 = t0;
  uint32_t acc = 0;
  const uint32_t top = 1u << 24;
  for (int s = 0; s < nsym; s++) {
#define TREE(T)                                                              \
  {                                                                          \
    uint32_t sym = 1;                                                        \
    _Pragma("unroll") for (int d = 0; d < 6; d++) {                          \
      if (STYLE == 0) bitF(rc, T, sym, sym, top);                            \
      else if (STYLE == 1) bitG(rc, T, sym, sym, top);                       \
      else bitP(rc, T, sym, (d & 1) * 16, sym, top);                         \
    }                                                                        \
    acc += sym;                                                              \
  }
    TREE(t0) TREE(t1) TREE(t2) TREE(t3)
    if (rc.off >= 192) { rc.off -= 192; rc.win = in[(s * 64 + blockIdx.x + lane) & 0xffff]; }
  }
  if (lane == 0) { out[blockIdx.x * 4] = acc; out[blockIdx.x * 4 + 1] = rc.range ^ rc.code; out[blockIdx.x * 4 + 2] = rc.eof; }
  if (nsym < 0) out[lane] = t0 + t1 + t2 + t3;
}


// H: range wave-uniform in an SGPR (bound/select on the SALU), code in a VGPR, probability update on
// the VALU under a one-lane EXEC; no branch except the (cold, compiler-placed) normalisation.
struct RCH { uint32_t range /*SGPR*/, code /*VGPR*/, win, off, rem, eof; };

__device__ __forceinline__ void normH(RCH& rc) {
  rc.range <<= 8;
  if (rc.rem == 0) { rc.eof = 1; rc.code <<= 8; return; }
  uint32_t w = __builtin_amdgcn_readlane(rc.win, rc.off >> 2);
  uint32_t b = (w >> ((rc.off & 3) * 8)) & 0xff;
  asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(rc.code) : "s"(b));
  rc.off++; rc.rem--;
}

template <int UPD>
__device__ __forceinline__ void bitH(RCH& rc, uint32_t& T, uint32_t idx, uint32_t& sym) {
  uint32_t sp, sb, sr1, vt, vu;
  if (UPD == 0) {
    asm volatile(
      "v_readlane_b32 %[sp], %[T], %[idx]\n\t"
      "s_lshr_b32 %[sb], %[range], 11\n\t"
      "s_mul_i32 %[sb], %[sb], %[sp]\n\t"
      "v_cmp_ge_u32 vcc, %[code], %[sb]\n\t"
      "v_subrev_u32 %[vt], %[sb], %[code]\n\t"
      "v_min_u32 %[code], %[code], %[vt]\n\t"
      "s_lshl_b64 exec, 1, %[idx]\n\t"
      "v_cndmask_b32 %[vt], 31, 0, vcc\n\t"
      "v_cndmask_b32 %[vu], 64, 0, vcc\n\t"
      "v_add_u32 %[vt], %[T], %[vt]\n\t"
      "v_lshrrev_b32 %[vt], 5, %[vt]\n\t"
      "v_sub_u32 %[vt], %[vu], %[vt]\n\t"
      "v_add_u32 %[T], %[T], %[vt]\n\t"
      "s_mov_b64 exec, -1\n\t"
      "s_sub_u32 %[sr1], %[range], %[sb]\n\t"
      "s_cmp_lg_u64 vcc, 0\n\t"
      "s_cselect_b32 %[range], %[sr1], %[sb]\n\t"
      "s_addc_u32 %[sym], %[sym], %[sym]"
      : [T] "+v"(T), [range] "+s"(rc.range), [code] "+v"(rc.code), [sym] "+s"(sym), [sp] "=&s"(sp), [sb] "=&s"(sb),
        [sr1] "=&s"(sr1), [vt] "=&v"(vt), [vu] "=&v"(vu)
      : [idx] "s"(idx)
      : "vcc", "scc");
  } else {
    uint32_t s31, s64;
    asm volatile(
      "v_readlane_b32 %[sp], %[T], %[idx]\n\t"
      "s_lshr_b32 %[sb], %[range], 11\n\t"
      "s_mul_i32 %[sb], %[sb], %[sp]\n\t"
      "v_cmp_ge_u32 vcc, %[code], %[sb]\n\t"
      "v_subrev_u32 %[vt], %[sb], %[code]\n\t"
      "s_sub_u32 %[sr1], %[range], %[sb]\n\t"
      "v_min_u32 %[code], %[code], %[vt]\n\t"
      "s_cmp_lg_u64 vcc, 0\n\t"
      "s_cselect_b32 %[range], %[sr1], %[sb]\n\t"
      "s_cselect_b32 %[s31], 0, 31\n\t"
      "s_cselect_b32 %[s64], 0, 64\n\t"
      "s_lshl_b64 exec, 1, %[idx]\n\t"
      "v_add_u32 %[vt], %[T], %[s31]\n\t"
      "v_lshrrev_b32 %[vt], 5, %[vt]\n\t"
      "v_sub_u32 %[vt], %[s64], %[vt]\n\t"
      "v_add_u32 %[T], %[T], %[vt]\n\t"
      "s_mov_b64 exec, -1\n\t"
      "s_cmp_lg_u64 vcc, 0\n\t"
      "s_addc_u32 %[sym], %[sym], %[sym]"
      : [T] "+v"(T), [range] "+s"(rc.range), [code] "+v"(rc.code), [sym] "+s"(sym), [sp] "=&s"(sp), [sb] "=&s"(sb),
        [sr1] "=&s"(sr1), [vt] "=&v"(vt), [s31] "=&s"(s31), [s64] "=&s"(s64)
      : [idx] "s"(idx)
      : "vcc", "scc");
  }
  if (__builtin_expect(rc.range < (1u << 24), 0)) normH(rc);
}

template <int UPD>
__global__ __launch_bounds__(64) void k_h(const uint32_t* in, uint32_t* out, int nsym) {
  const uint32_t lane = threadIdx.x;
  RCH rc;
  rc.win = in[(blockIdx.x * 64 + lane) & 0xffff];
  rc.range = 0xffffffffu;
  rc.code = in[blockIdx.x & 0xffff] >> 1;
  rc.off = 0; rc.rem = 1u << 30; rc.eof = 0;
  uint32_t t0 = 0x400u, t1 = t0, t2 = t0, t3 = t0;
  uint32_t acc = 0;
  for (int s = 0; s < nsym; s++) {
#define TREEH0(T) { uint32_t sym = 1; _Pragma("unroll") for (int d = 0; d < 6; d++) bitH<UPD>(rc, T, sym, sym); acc += sym; }
    TREEH0(t0) TREEH0(t1) TREEH0(t2) TREEH0(t3)
    if (rc.off >= 192) { rc.off -= 192; rc.win = in[(s * 64 + blockIdx.x + lane) & 0xffff]; }
  }
  if (lane == 0) { out[blockIdx.x * 4] = acc; out[blockIdx.x * 4 + 1] = rc.range ^ __builtin_amdgcn_readfirstlane(rc.code); out[blockIdx.x * 4 + 2] = rc.eof; }
  if (nsym < 0) out[lane] = t0 + t1 + t2 + t3;
}

#define TREEH(T) { uint32_t sym = 1; _Pragma("unroll") for (int d = 0; d < 6; d++) bitH<0>(rc, T, sym, sym); acc += sym; }
template <int UNR>
__global__ __launch_bounds__(64) void k_h_big(const uint32_t* in, uint32_t* out, int nsym) {
  const uint32_t lane = threadIdx.x;
  RCH rc;
  rc.win = in[(blockIdx.x * 64 + lane) & 0xffff];
  rc.range = 0xffffffffu;
  rc.code = in[blockIdx.x & 0xffff] >> 1;
  rc.off = 0; rc.rem = 1u << 30; rc.eof = 0;
  uint32_t t0 = 0x400u, t1 = t0, t2 = t0, t3 = t0;
  uint32_t acc = 0;
  for (int s = 0; s < nsym; s += UNR) {
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      TREEH(t0) TREEH(t1) TREEH(t2) TREEH(t3)
      if (rc.off >= 192) { rc.off -= 192; rc.win = in[(s * 64 + blockIdx.x + lane) & 0xffff]; }
    }
  }
  if (lane == 0) { out[blockIdx.x * 4] = acc; out[blockIdx.x * 4 + 1] = rc.range ^ __builtin_amdgcn_readfirstlane(rc.code); out[blockIdx.x * 4 + 2] = rc.eof; }
  if (nsym < 0) out[lane] = t0 + t1 + t2 + t3;
}

template <typename K>
static void run(const char* name, K kernel, int waves, int nsym, const uint32_t* d_in, uint32_t* d_out) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kernel, dim3(waves), dim3(64), 0, 0, d_in, d_out, nsym / 8);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(waves), dim3(64), 0, 0, d_in, d_out, nsym);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double bits = (double)waves * nsym * 24.0;
  printf("%-3s waves=%5d  %8.3f ms  %8.2f Gbit/s  %7.1f cyc/bit/chain@2.4GHz(if all resident)\n", name, waves, best,
         bits / (best * 1e-3) / 1e9, (best * 1e-3) * 2.4e9 / ((double)nsym * 24.0));
  fflush(stdout);
}

int main() {
  std::vector<uint32_t> h(65536);
  uint32_t x = 12345;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
  uint32_t *d_in, *d_out;
  CHECK(hipMalloc(&d_in, h.size() * 4)); CHECK(hipMalloc(&d_out, 65536 * 4 * 4));
  CHECK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const int nsym = 4000;
  int wave_counts[] = {256, 4096};
  for (int w : wave_counts) {
    run("F", k_fast<0>, w, nsym, d_in, d_out);
    run("G", k_fast<1>, w, nsym, d_in, d_out);
    run("P", k_fast<2>, w, nsym, d_in, d_out);
    run("H0", k_h<0>, w, nsym, d_in, d_out);
    run("H1", k_h<1>, w, nsym, d_in, d_out);
    run("Hx4", k_h_big<4>, w, nsym, d_in, d_out);
    run("Hx16", k_h_big<16>, w, nsym, d_in, d_out);
    run("Hx32", k_h_big<32>, w, nsym, d_in, d_out);
  }
  return 0;
}
