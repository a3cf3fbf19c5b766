// order_slots.hip -- round 4 micro-benchmark: does the ORDER of a form-B tree decision's ten instructions matter?
//
//   hipcc --offload-arch=gfx950 -O2 experiments/microbench/order_slots.hip -o /tmp/order_slots && /tmp/order_slots
//
// A form-B decision is  v_lshrrev, v_mul, v_sub, v_readlane x2 | s_sub, s_cselect_b64, s_addc, s_cmp, s_cbranch  (the normalisation
// test).  The vector head of decision k+1 depends on the s_cselect_b64 of decision k only: the s_addc / s_cmp / s_cbranch behind it
// are in the way merely because a wave issues in order.  Variants hoist the next decision's vector instructions above them
// (speculatively: the rare normalisation stub would have to redo them) or interleave the two groups, and one variant takes v_sub off
// the chain (range - bound as vt * (2048 - p) + (range & 2047): v_mad in parallel with v_mul).  Lone wave and 2 / 3 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int ITER = 1500;
#define REP8(x) x x x x x x x x

#define LSHR "v_lshrrev_b32 %[vt], 11, s66\n\t"
#define MUL "v_mul_u32_u24 %[vb], %[vt], %[p]\n\t"
#define SUB "v_sub_u32 %[vr], s66, %[vb]\n\t"
#define RL0 "v_readlane_b32 s66, %[vb], %[sym]\n\t"
#define RL1 "v_readlane_b32 s74, %[vr], %[sym]\n\t"
#define SSUB "s_sub_u32 s75, s67, s66\n\t"
#define CSEL "s_cselect_b64 s[66:67], s[66:67], s[74:75]\n\t"
#define KEEP "s_or_b32 s66, s66, 0x40000000\n\t"   /* keeps the dummy range large: not part of the real chain */
#define ADDC "s_addc_u32 %[sym], %[sym], %[sym]\n\t"
#define CMP "s_cmp_lt_u32 s66, 0x1000000\n\t"
#define BR "s_cbranch_scc1 9f\n\t"
// (KEEP clobbers SCC: in the orders below the s_addc therefore reads a stale SCC -- it is a cost model, not a decoder)

// baseline, as the loop emits it today
#define STEP_B LSHR MUL SUB RL0 RL1 SSUB CSEL KEEP ADDC CMP BR
// the whole vector head hoisted above s_addc / s_cmp / s_cbranch
#define STEP_H3 RL0 RL1 SSUB CSEL KEEP LSHR MUL SUB ADDC CMP BR
// interleaved: every dependent vector instruction is followed by an independent scalar one
#define STEP_I RL0 RL1 SSUB CSEL KEEP LSHR ADDC MUL CMP SUB BR
// only v_lshrrev hoisted (the stub would redo one instruction)
#define STEP_H1 RL0 RL1 SSUB CSEL KEEP LSHR ADDC CMP BR MUL SUB
// v_lshrrev and v_mul hoisted
#define STEP_H2 RL0 RL1 SSUB CSEL KEEP LSHR ADDC MUL CMP BR SUB
// v_sub off the chain: lo = range & 2047 (scalar or vector), vr = vt * pc + lo with pc = 2048 - p (once per walk)
#define STEP_P "v_and_b32 %[x0], s66, %[m7ff]\n\t" LSHR MUL "v_mad_u32_u24 %[vr], %[vt], %[pc], %[x0]\n\t" RL0 RL1 SSUB CSEL KEEP ADDC CMP BR
#define STEP_PS "s_and_b32 s80, s66, 0x7ff\n\t" LSHR MUL "v_mad_u32_u24 %[vr], %[vt], %[pc], s80\n\t" RL0 RL1 SSUB CSEL KEEP ADDC CMP BR
// both: parallel v_mad + interleaving
#define STEP_PI RL0 RL1 SSUB CSEL KEEP LSHR "s_and_b32 s80, s66, 0x7ff\n\t" MUL ADDC "v_mad_u32_u24 %[vr], %[vt], %[pc], s80\n\t" CMP BR
// calibration: one vector instruction less on the chain (no v_sub), one scalar less (no s_addc), the range shift on the scalar ALU
#define STEP_M1V LSHR MUL RL0 "v_readlane_b32 s74, %[vb], %[sym]\n\t" SSUB CSEL KEEP ADDC CMP BR
#define STEP_M1S LSHR MUL SUB RL0 RL1 SSUB CSEL KEEP CMP BR
#define STEP_R11S "s_lshr_b32 s80, s66, 11\n\t" "v_mul_u32_u24 %[vb], s80, %[p]\n\t" SUB RL0 RL1 SSUB CSEL KEEP ADDC CMP BR
// the test on the vector ALU early in the head, its branch in the shadow of the hop (the stub would redo the head: a second hop)
#define STEP_VT LSHR "v_cmp_gt_u32 vcc, 0x2000, %[vt]\n\t" MUL SUB RL0 RL1 "s_cbranch_vccnz 9f\n\t" SSUB CSEL KEEP ADDC


// ---- second batch: other decision forms ----------------------------------------------------------------------------------------
#define RLB "v_readlane_b32 s72, %[vb], %[sym]\n\t"
#define A_TAIL "s_sub_u32 s74, s66, s72\n\t" "s_sub_u32 s75, s67, s72\n\t" "s_cselect_b32 s66, s72, s74\n\t" "s_cselect_b32 s67, s67, s75\n\t" KEEP ADDC CMP BR
// form A: every lane computes its bound, one v_readlane, range - bound and the two selects on the scalar ALU
#define STEP_A LSHR MUL "s_nop 0\n\t" RLB A_TAIL
// form A with the range shift on the scalar ALU too: two vector instructions in all
#define STEP_AR "s_lshr_b32 s80, s66, 11\n\t" "v_mul_u32_u24 %[vb], s80, %[p]\n\t" "s_nop 0\n\t" RLB A_TAIL
// all scalar, the node's probability fetched on the chain: one vector instruction (the v_readlane)
#define STEP_S1 "v_readlane_b32 s82, %[p], %[sym]\n\t" "s_lshr_b32 s80, s66, 11\n\t" "s_mul_i32 s72, s80, s82\n\t" A_TAIL
// all scalar, both children's probabilities read ahead (their hop hides behind this decision's scalar chain)
#define STEP_S2 "s_lshl_b32 s81, %[sym], 1\n\t" "v_readlane_b32 s84, %[p], s81\n\t" "s_or_b32 s85, s81, 1\n\t" "v_readlane_b32 s86, %[p], s85\n\t" \
  "s_lshr_b32 s80, s66, 11\n\t" "s_mul_i32 s72, s80, s82\n\t" "s_sub_u32 s74, s66, s72\n\t" "s_sub_u32 s75, s67, s72\n\t" \
  "s_cselect_b32 s66, s72, s74\n\t" "s_cselect_b32 s67, s67, s75\n\t" "s_cselect_b32 s82, s86, s84\n\t" KEEP ADDC CMP BR
// form B with the two v_readlane swapped, and without the second one (what does a v_readlane cost?)
#define STEP_BSW LSHR MUL SUB RL1 RL0 SSUB CSEL KEEP ADDC CMP BR
#define STEP_M1R LSHR MUL SUB RL0 SSUB CSEL KEEP ADDC CMP BR
// form B without the normalisation test, and with the test but without s_addc and without v_sub (floor of the form)
#define STEP_NOT LSHR MUL SUB RL0 RL1 SSUB CSEL KEEP ADDC

// ---- third batch: the normalisation stub (taken at EVERY decision here; in the loop: once per input byte, 0.12 per decision) -----------
//   N0: as the loop has it: v_readlane of the input byte first -- the s_lshl_b64 behind it waits out the vector -> scalar hop.
//   N1: the byte was read ahead into an SGPR (s82) by the previous stub; this stub's own v_readlane (for the NEXT normalisation) is its
//       last instruction but the branch back, so its hop can run under the next decision's vector head.
//   N2: N1 without the v_readlane at all (what hiding it completely would give).
#define TAKEN "s_cmp_gt_u32 s66, 0\n\t" "s_cbranch_scc1 1f\n\t" "s_branch 2f\n\t" "1:\n\t"
#define STEP_N0 LSHR MUL SUB RL0 RL1 SSUB CSEL KEEP ADDC TAKEN \
  "v_readlane_b32 s82, %[x0], s81\n\t" "s_lshl_b64 s[84:85], s[84:85], 8\n\t" "s_or_b32 s85, s85, s82\n\t" "s_add_u32 s81, s81, 1\n\t" "s_cbranch_scc0 2f\n\t" "2:\n\t"
#define STEP_N1 LSHR MUL SUB RL0 RL1 SSUB CSEL KEEP ADDC TAKEN \
  "s_lshl_b64 s[84:85], s[84:85], 8\n\t" "s_or_b32 s85, s85, s82\n\t" "s_add_u32 s81, s81, 1\n\t" "s_cbranch_scc1 2f\n\t" "v_readlane_b32 s82, %[x0], s81\n\t" "s_branch 2f\n\t" "2:\n\t"
#define STEP_N2 LSHR MUL SUB RL0 RL1 SSUB CSEL KEEP ADDC TAKEN \
  "s_lshl_b64 s[84:85], s[84:85], 8\n\t" "s_or_b32 s85, s85, s82\n\t" "s_add_u32 s81, s81, 1\n\t" "s_cbranch_scc1 2f\n\t" "s_branch 2f\n\t" "2:\n\t"

#define OPS                                                                                                          \
  : [vt] "=&v"(vt), [vb] "=&v"(vb), [vr] "=&v"(vr), [sym] "+s"(sym), [x0] "+v"(x0)                                   \
  : [p] "v"(prob), [pc] "v"(probc), [m7ff] "v"(m7ff)                                                                                   \
  : "s66", "s67", "s72", "s74", "s75", "s80", "s81", "s82", "s84", "s85", "s86", "scc", "vcc"

enum { kB, kH3, kI, kH1, kH2, kP, kPS, kPI, kM1V, kM1S, kR11S, kVT, kA, kAR, kS1, kS2, kBSW, kM1R, kNOT, kN0, kN1, kN2, kCount };

template <int VAR>
__global__ __launch_bounds__(64, 8) void chain(uint64_t* out, uint32_t seed) {
  uint32_t sym = 1, prob = 1024 + (threadIdx.x & 7), probc = 2048 - prob;
  uint32_t vt = 0, vb = 0, vr = prob, x0 = prob, m7ff = 0x7ff;
  uint64_t t0, t1;
  asm volatile("s_mov_b32 s66, 0xF0000007\n\ts_mov_b32 s67, 0x12345678\n\ts_mov_b32 s74, 0x12345\n\ts_mov_b32 s80, 0" ::: "s66", "s67", "s74", "s80");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int i = 0; i < ITER; i++) {
#define RUN(S) asm volatile(REP8(S) "s_branch 8f\n\t9:\n\ts_nop 0\n\t8:\n\t" OPS)
    if constexpr (VAR == kB) RUN(STEP_B);
    else if constexpr (VAR == kH3) RUN(STEP_H3);
    else if constexpr (VAR == kI) RUN(STEP_I);
    else if constexpr (VAR == kH1) RUN(STEP_H1);
    else if constexpr (VAR == kH2) RUN(STEP_H2);
    else if constexpr (VAR == kP) RUN(STEP_P);
    else if constexpr (VAR == kPS) RUN(STEP_PS);
    else if constexpr (VAR == kPI) RUN(STEP_PI);
    else if constexpr (VAR == kM1V) RUN(STEP_M1V);
    else if constexpr (VAR == kM1S) RUN(STEP_M1S);
    else if constexpr (VAR == kR11S) RUN(STEP_R11S);
    else if constexpr (VAR == kVT) RUN(STEP_VT);
    else if constexpr (VAR == kA) RUN(STEP_A);
    else if constexpr (VAR == kAR) RUN(STEP_AR);
    else if constexpr (VAR == kS1) RUN(STEP_S1);
    else if constexpr (VAR == kS2) RUN(STEP_S2);
    else if constexpr (VAR == kBSW) RUN(STEP_BSW);
    else if constexpr (VAR == kM1R) RUN(STEP_M1R);
    else if constexpr (VAR == kNOT) RUN(STEP_NOT);
    else if constexpr (VAR == kN0) RUN(STEP_N0);
    else if constexpr (VAR == kN1) RUN(STEP_N1);
    else if constexpr (VAR == kN2) RUN(STEP_N2);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (uint64_t(sym ^ prob ^ vt ^ vb ^ vr ^ x0) & 0);
}

template <int VAR>
int run(const char* name, int instr, uint64_t* d_out, std::vector<uint64_t>& h) {
  printf("%-6s (%2d instructions per decision)", name, instr);
  for (int per_simd : {0, 2, 3, 4, 5}) {  // 0: one wave per CU
    const int waves = per_simd ? 256 * 4 * per_simd : 256;
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
      chain<VAR><<<waves, 64>>>(d_out, 7);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(h.data(), d_out, waves * 8, hipMemcpyDeviceToHost));
      double sum = 0;
      for (int i = 0; i < waves; i++) sum += double(h[i]);
      best = std::min(best, sum / waves / (double(ITER) * 8));
    }
    printf("  %s %6.1f", per_simd == 0 ? "lone" : per_simd == 2 ? "2/SIMD" : per_simd == 3 ? "3/SIMD" : per_simd == 4 ? "4/SIMD" : "5/SIMD", best);
  }
  printf("   cycles per decision\n");
  return 0;
}

int main(int argc, char**) {
  uint64_t* d_out;
  CHECK(hipMalloc(&d_out, 8192 * 8));
  std::vector<uint64_t> h(8192);
  const bool second = argc > 1;
  const bool third = argc > 2;
  for (int pass = 0; pass < 2; pass++) {   // twice: run-to-run spread
    if (third) {
      if (run<kB>("B", 11, d_out, h) || run<kN0>("N0", 18, d_out, h) || run<kN1>("N1", 19, d_out, h) || run<kN2>("N2", 18, d_out, h)) return 1;
    } else if (!second) {
      if (run<kB>("B", 11, d_out, h) || run<kH1>("H1", 11, d_out, h) || run<kH2>("H2", 11, d_out, h) || run<kH3>("H3", 11, d_out, h) ||
          run<kI>("I", 11, d_out, h) || run<kP>("P", 12, d_out, h) || run<kPS>("PS", 12, d_out, h) || run<kPI>("PI", 12, d_out, h) ||
          run<kM1V>("B-1V", 10, d_out, h) || run<kM1S>("B-1S", 10, d_out, h) || run<kR11S>("R11S", 11, d_out, h) || run<kVT>("VT", 11, d_out, h))
        return 1;
    } else {
      if (run<kB>("B", 11, d_out, h) || run<kA>("A", 13, d_out, h) || run<kAR>("A-R11S", 13, d_out, h) || run<kS1>("S1", 12, d_out, h) ||
          run<kS2>("S2", 16, d_out, h) || run<kBSW>("B-swap", 11, d_out, h) || run<kM1R>("B-1RL", 10, d_out, h) || run<kNOT>("B-test", 9, d_out, h))
        return 1;
    }
    printf("\n");
  }
  return 0;
}
