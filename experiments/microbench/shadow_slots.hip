// shadow_slots.hip -- round 3 micro-benchmarks on the decision chain of the symbol loop (gfx950).
//
//   hipcc --offload-arch=gfx950 -O2 experiments/microbench/shadow_slots.hip -o /tmp/shadow_slots && /tmp/shadow_slots
//
// 1. "Shadow slots".  A form-B tree decision is v_lshrrev, v_mul, v_sub, v_readlane x2, s_sub, s_cselect_b64, s_addc,
//    s_cmp, s_cbranch.  The scalar instruction that consumes a v_readlane result issues ~14 cycles later than a dependent
//    instruction normally would (profiles/r02_chain_latency.txt).  A wave issues in order, one instruction per ~4.6 cycles:
//    independent instructions placed between the v_readlane pair and the s_sub should therefore be (nearly) free.  Variants
//    put k independent vector / scalar instructions (a probability update: v_mad, v_lshrrev, v_cndmask; scalar glue) either
//    INTO that shadow or AFTER the decision, lone wave and 3 / 4 waves per SIMD.
// 2. Lane-speculative tree walk (verdict item 1c).  Every lane walks the path to its own leaf with range / code / input
//    look-ahead per lane, all on the vector ALU (12 instructions per level incl. normalisation), one find-first + three
//    v_readlane per WALK instead of one vector -> scalar hop per LEVEL.  Compared with 6 form-B decisions.
// 3. Side questions: v_mul_u32_u24_sdwa (16-bit operand select) and VGPR index mode at the rate of the plain instruction?
//    does a half-empty EXEC make vector instructions cheaper?  what does the not-taken normalisation branch cost?
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int ITER = 1500;
#define REP8(x) x x x x x x x x

// the decision up to the hop, and its scalar tail (the s_or keeps the dummy range large: not part of the real chain)
#define D_HEAD                                   \
  "v_lshrrev_b32 %[vt], 11, s66\n\t"             \
  "v_mul_u32_u24 %[vb], %[vt], %[p]\n\t"         \
  "v_sub_u32 %[vr], s66, %[vb]\n\t"              \
  "v_readlane_b32 s66, %[vb], %[sym]\n\t"        \
  "v_readlane_b32 s74, %[vr], %[sym]\n\t"
#define D_TAIL                                            \
  "s_sub_u32 s75, s67, s66\n\t"                           \
  "s_cselect_b64 s[66:67], s[66:67], s[74:75]\n\t"        \
  "s_addc_u32 %[sym], %[sym], %[sym]\n\t"                 \
  "s_or_b32 s66, s66, 0x40000000\n\t"                     \
  "s_cmp_lt_u32 s66, 0x1000000\n\t"                       \
  "s_cbranch_scc1 1f\n\t"                                 \
  "1:\n\t"
#define D_TAIL_NOBR                                       \
  "s_sub_u32 s75, s67, s66\n\t"                           \
  "s_cselect_b64 s[66:67], s[66:67], s[74:75]\n\t"        \
  "s_addc_u32 %[sym], %[sym], %[sym]\n\t"                 \
  "s_or_b32 s66, s66, 0x40000000\n\t"
// independent work: a probability update on other registers (mask in an SGPR pair), scalar glue
#define V1 "v_mad_u32_u24 %[x0], %[x1], 31, %[p]\n\t"
#define V2 V1 "v_lshrrev_b32 %[x0], 5, %[x0]\n\t"
#define V3 V2 "v_cndmask_b32_e64 %[x1], %[x1], %[x0], s[90:91]\n\t"
#define V4 V3 "v_mad_u32_u24 %[x2], %[x3], 31, %[p]\n\t"
#define V5 V4 "v_lshrrev_b32 %[x2], 5, %[x2]\n\t"
#define V6 V5 "v_cndmask_b32_e64 %[x3], %[x3], %[x2], s[90:91]\n\t"
#define S1 "s_add_u32 s80, s80, 1\n\t"
#define S2 S1 "s_and_b32 s81, s80, 7\n\t"
#define S3 S2 "s_lshl_b32 s82, s81, 2\n\t"

#define OPS                                                                                                          \
  : [vt] "=&v"(vt), [vb] "=&v"(vb), [vr] "=&v"(vr), [sym] "+s"(sym), [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2),    \
    [x3] "+v"(x3)                                                                                                    \
  : [p] "v"(prob)                                                                                                    \
  : "s66", "s67", "s74", "s75", "s80", "s81", "s82", "s90", "s91", "scc", "vcc"

enum {
  kB, kV1s, kV2s, kV3s, kV4s, kV6s, kV2a, kV3a, kV6a, kS1s, kS2s, kS3s, kS2a, kS3a, kV2S1s, kV2S1a, kNoBr, kSpec6, kB6,
  kV3chain, kV3sdwa, kV3idx, kV3half, kCount
};

template <int VAR>
__global__ __launch_bounds__(64, 8) void chain(uint64_t* out, uint32_t seed) {
  uint32_t sym = 1, prob = 1024 + (threadIdx.x & 7);
  uint32_t vt = 0, vb = 0, vr = prob, x0 = prob, x1 = prob + 1, x2 = 3, x3 = 4;
  uint64_t t0, t1;
  asm volatile("s_mov_b32 s66, 0xF0000007\n\ts_mov_b32 s67, 0x12345678\n\ts_mov_b32 s80, 0\n\ts_mov_b64 s[90:91], 0x5\n\t"
               "s_mov_b32 s68, 0x01020304\n\ts_mov_b32 s69, 0x1000000" ::: "s66", "s67", "s68", "s69", "s80", "s90", "s91");
  if constexpr (VAR == kV3idx) asm volatile("s_mov_b32 s83, 0\n\ts_set_gpr_idx_on s83, gpr_idx(SRC1)" ::: "s83", "m0");
  if constexpr (VAR == kV3half) asm volatile("s_mov_b64 exec, 0xffffffff");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int i = 0; i < ITER; i++) {
    if constexpr (VAR == kB) asm volatile(REP8(D_HEAD D_TAIL) OPS);
    else if constexpr (VAR == kV1s) asm volatile(REP8(D_HEAD V1 D_TAIL) OPS);
    else if constexpr (VAR == kV2s) asm volatile(REP8(D_HEAD V2 D_TAIL) OPS);
    else if constexpr (VAR == kV3s) asm volatile(REP8(D_HEAD V3 D_TAIL) OPS);
    else if constexpr (VAR == kV4s) asm volatile(REP8(D_HEAD V4 D_TAIL) OPS);
    else if constexpr (VAR == kV6s) asm volatile(REP8(D_HEAD V6 D_TAIL) OPS);
    else if constexpr (VAR == kV2a) asm volatile(REP8(D_HEAD D_TAIL V2) OPS);
    else if constexpr (VAR == kV3a) asm volatile(REP8(D_HEAD D_TAIL V3) OPS);
    else if constexpr (VAR == kV6a) asm volatile(REP8(D_HEAD D_TAIL V6) OPS);
    else if constexpr (VAR == kS1s) asm volatile(REP8(D_HEAD S1 D_TAIL) OPS);
    else if constexpr (VAR == kS2s) asm volatile(REP8(D_HEAD S2 D_TAIL) OPS);
    else if constexpr (VAR == kS3s) asm volatile(REP8(D_HEAD S3 D_TAIL) OPS);
    else if constexpr (VAR == kS2a) asm volatile(REP8(D_HEAD D_TAIL S2) OPS);
    else if constexpr (VAR == kS3a) asm volatile(REP8(D_HEAD D_TAIL S3) OPS);
    else if constexpr (VAR == kV2S1s) asm volatile(REP8(D_HEAD V2 S1 D_TAIL) OPS);
    else if constexpr (VAR == kV2S1a) asm volatile(REP8(D_HEAD D_TAIL V2 S1) OPS);
    else if constexpr (VAR == kNoBr) asm volatile(REP8(D_HEAD D_TAIL_NOBR) OPS);
    else if constexpr (VAR == kB6) {  // six decisions = one 6-level walk, the reference for kSpec6 (REP8 -> 48 decisions per iteration)
      asm volatile(REP8(D_HEAD D_TAIL D_HEAD D_TAIL D_HEAD D_TAIL D_HEAD D_TAIL D_HEAD D_TAIL D_HEAD D_TAIL) OPS);
    } else if constexpr (VAR == kSpec6) {
      // v40 range, v[42:43] = (look-ahead, code) pair for v_lshlrev_b64, v44..: temporaries; s[84:85] alive mask, s[80:81] this
      // level's "my path takes the 1 branch" mask (constant per level in the real thing), s69 = 2^24
#define LVL                                              \
  "v_lshrrev_b32 v44, 11, v40\n\t"                     \
  "v_mul_u32_u24 v45, v44, %[p]\n\t"                   \
  "v_sub_u32 v46, v40, v45\n\t"                       \
  "v_cndmask_b32_e64 v47, 0, v45, s[90:91]\n\t"        \
  "v_cndmask_b32_e64 v40, v45, v46, s[90:91]\n\t"     \
  "v_sub_u32 v43, v43, v47\n\t"                       \
  "v_cmp_lt_u32_e64 s[84:85], v43, v40\n\t"            \
  "v_or_b32 v40, 0x40000000, v40\n\t"                  \
  "v_cmp_gt_u32 vcc, s69, v40\n\t"                      \
  "s_nop 1\n\t"                                          \
  "v_cndmask_b32_e64 v48, 0, 8, vcc\n\t"                \
  "v_lshlrev_b32 v40, v48, v40\n\t"                   \
  "v_lshlrev_b64 v[42:43], v48, v[42:43]\n\t"
      asm volatile(REP8(
          "v_mov_b32 v40, s66\n\t"
          "v_mov_b32 v43, s67\n\t"
          "v_mov_b32 v42, s68\n\t"
          LVL LVL LVL LVL LVL LVL
          "s_or_b64 s[84:85], s[84:85], 0x20\n\t"       // (dummy data: make sure one lane "survives")
          "s_ff1_i32_b64 s86, s[84:85]\n\t"
          "s_nop 3\n\t"
          "v_readlane_b32 s66, v40, s86\n\t"
          "v_readlane_b32 s67, v43, s86\n\t"
          "v_readlane_b32 s68, v42, s86\n\t"
          "s_or_b32 s66, s66, 0x40000000\n\t")
          :
          : [p] "v"(prob)
          : "s66", "s67", "s68", "s84", "s85", "s86", "scc", "vcc", "v40", "v42", "v43", "v44", "v45", "v46", "v47", "v48");
    } else if constexpr (VAR == kV3chain || VAR == kV3idx || VAR == kV3half) {
      asm volatile(REP8("v_lshrrev_b32 %[vt], 11, %[vr]\n\t"
                        "v_mul_u32_u24 %[vb], %[vt], %[p]\n\t"
                        "v_sub_u32 %[vr], %[p], %[vb]\n\t")
                   : [vt] "=&v"(vt), [vb] "=&v"(vb), [vr] "+v"(vr)
                   : [p] "v"(prob));
    } else if constexpr (VAR == kV3sdwa) {
      asm volatile(REP8("v_lshrrev_b32 %[vt], 11, %[vr]\n\t"
                        "v_mul_u32_u24_sdwa %[vb], %[vt], %[p] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
                        "v_sub_u32 %[vr], %[p], %[vb]\n\t")
                   : [vt] "=&v"(vt), [vb] "=&v"(vb), [vr] "+v"(vr)
                   : [p] "v"(prob));
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if constexpr (VAR == kV3idx) asm volatile("s_set_gpr_idx_off");
  if constexpr (VAR == kV3half) asm volatile("s_mov_b64 exec, -1");
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (uint64_t(sym ^ prob ^ vt ^ vb ^ vr ^ x0 ^ x1 ^ x2 ^ x3) & 0);
}

template <int VAR>
int run(const char* name, int instr, int decisions, uint64_t* d_out, std::vector<uint64_t>& h) {
  printf("%-7s (%2d instructions per step)", name, instr);
  for (int per_simd : {0, 2, 3, 4, 8}) {  // 0: one wave per CU
    const int waves = per_simd ? 256 * 4 * per_simd : 256;
    for (int rep = 0; rep < 2; rep++) {
      chain<VAR><<<waves, 64>>>(d_out, 7);
      CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(h.data(), d_out, waves * 8, hipMemcpyDeviceToHost));
    double sum = 0;
    for (int i = 0; i < waves; i++) sum += double(h[i]);
    const double per = sum / waves / (double(ITER) * 8);
    printf("  %s %7.1f", per_simd == 0 ? "lone" : per_simd == 2 ? "2/SIMD" : per_simd == 3 ? "3/SIMD" : per_simd == 4 ? "4/SIMD" : "8/SIMD", per);
  }
  printf("   cycles per step (%d decision%s)\n", decisions, decisions == 1 ? "" : "s");
  return 0;
}

int main() {
  uint64_t* d_out;
  CHECK(hipMalloc(&d_out, 8192 * 8));
  std::vector<uint64_t> h(8192);
  printf("# shadow slots: k independent instructions between the v_readlane pair and the s_sub (s) or after the decision (a)\n");
  if (run<kB>("B", 11, 1, d_out, h) || run<kV1s>("B+1Vs", 12, 1, d_out, h) || run<kV2s>("B+2Vs", 13, 1, d_out, h) || run<kV3s>("B+3Vs", 14, 1, d_out, h) ||
      run<kV4s>("B+4Vs", 15, 1, d_out, h) || run<kV6s>("B+6Vs", 17, 1, d_out, h) || run<kV2a>("B+2Va", 13, 1, d_out, h) ||
      run<kV3a>("B+3Va", 14, 1, d_out, h) || run<kV6a>("B+6Va", 17, 1, d_out, h) || run<kS1s>("B+1Ss", 12, 1, d_out, h) ||
      run<kS2s>("B+2Ss", 13, 1, d_out, h) || run<kS3s>("B+3Ss", 14, 1, d_out, h) || run<kS2a>("B+2Sa", 13, 1, d_out, h) ||
      run<kS3a>("B+3Sa", 14, 1, d_out, h) || run<kV2S1s>("B+2V1Ss", 14, 1, d_out, h) || run<kV2S1a>("B+2V1Sa", 14, 1, d_out, h) ||
      run<kNoBr>("B-norm", 9, 1, d_out, h))
    return 1;
  printf("# lane-speculative 6-level walk (all vector, 13 instructions per level + 10 per walk) against six form-B decisions\n");
  if (run<kB6>("6xB", 66, 6, d_out, h) || run<kSpec6>("spec6", 88, 6, d_out, h)) return 1;
  printf("# three dependent vector instructions: plain, with an SDWA operand select, in VGPR index mode, with half of EXEC off\n");
  if (run<kV3chain>("V3", 3, 0, d_out, h) || run<kV3sdwa>("V3sdwa", 3, 0, d_out, h) || run<kV3idx>("V3idx", 3, 0, d_out, h) ||
      run<kV3half>("V3half", 3, 0, d_out, h))
    return 1;
  return 0;
}
