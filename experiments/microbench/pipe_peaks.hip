// pipe_peaks.hip -- round 5: what the issue pipes of a gfx950 SIMD / CU deliver, in instructions per cycle, measured with the
// WALL clock (hipEvents around the launch) and cross-checked with s_memtime -- the hardware ceiling the symbol loop's
// instruction mix is priced against in DESIGN.md section 4 / bench.py's roofline_issue.
//
//   hipcc --offload-arch=gfx950 -O2 experiments/microbench/pipe_peaks.hip -o build/pipe_peaks && build/pipe_peaks
//
// Every body is INDEPENDENT work (eight rotating destination registers, nothing waits for a result sooner than eight
// instructions later), so what is measured is the pipe, not a dependency chain.  Each variant runs with 1, 2, 4, 5 and 8 waves
// per SIMD (1024 x W one-wave blocks on 256 CUs x 4 SIMDs).  Printed per variant and W:
//   ipc_simd  = instructions of the body's kind per shader cycle per SIMD (wall clock, assumed 2.4 GHz unless --mhz)
//   ipc_cu    = the same per CU
//   tick      = s_memtime ticks per wall-clock nanosecond (2.4 = the counter runs at the shader clock)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define REP4(x) x x x x

// ---- bodies: 32 instructions of one kind each (4 x 8 rotating destinations) -------------------------------------------------
#define V8(op, src) \
  op " %[a0], " src ", %[a0]\n\t" op " %[a1], " src ", %[a1]\n\t" op " %[a2], " src ", %[a2]\n\t" op " %[a3], " src ", %[a3]\n\t" \
  op " %[a4], " src ", %[a4]\n\t" op " %[a5], " src ", %[a5]\n\t" op " %[a6], " src ", %[a6]\n\t" op " %[a7], " src ", %[a7]\n\t"
#define BODY_VADD REP4(V8("v_add_u32", "%[k]"))
#define BODY_VMUL24 REP4(V8("v_mul_u32_u24", "%[k]"))
#define BODY_VLSHR REP4(V8("v_lshrrev_b32", "1"))
#define BODY_VSUBS REP4(V8("v_sub_u32", "s66"))                    /* a scalar operand, as the loop's v_sub vr, range, vb */
#define V8M(op) \
  op " %[a0], %[a0], 31, %[k]\n\t" op " %[a1], %[a1], 31, %[k]\n\t" op " %[a2], %[a2], 31, %[k]\n\t" op " %[a3], %[a3], 31, %[k]\n\t" \
  op " %[a4], %[a4], 31, %[k]\n\t" op " %[a5], %[a5], 31, %[k]\n\t" op " %[a6], %[a6], 31, %[k]\n\t" op " %[a7], %[a7], 31, %[k]\n\t"
#define BODY_VMAD24 REP4(V8M("v_mad_u32_u24"))
#define BODY_VBFE REP4(V8M("v_bfe_u32"))
#define BODY_VMULLO REP4(V8("v_mul_lo_u32", "%[k]"))
#define RL8 \
  "v_readlane_b32 s72, %[a0], s80\n\t" "v_readlane_b32 s73, %[a1], s80\n\t" "v_readlane_b32 s74, %[a2], s80\n\t" "v_readlane_b32 s75, %[a3], s80\n\t" \
  "v_readlane_b32 s76, %[a4], s80\n\t" "v_readlane_b32 s77, %[a5], s80\n\t" "v_readlane_b32 s78, %[a6], s80\n\t" "v_readlane_b32 s79, %[a7], s80\n\t"
#define BODY_VREADLANE REP4(RL8)
#define CMP8 \
  "v_cmp_eq_u32_e64 s[72:73], %[a0], %[k]\n\t" "v_cmp_eq_u32_e64 s[74:75], %[a1], %[k]\n\t" "v_cmp_eq_u32_e64 s[76:77], %[a2], %[k]\n\t" \
  "v_cmp_eq_u32_e64 s[78:79], %[a3], %[k]\n\t" "v_cmp_eq_u32_e64 s[72:73], %[a4], %[k]\n\t" "v_cmp_eq_u32_e64 s[74:75], %[a5], %[k]\n\t" \
  "v_cmp_eq_u32_e64 s[76:77], %[a6], %[k]\n\t" "v_cmp_eq_u32_e64 s[78:79], %[a7], %[k]\n\t"
#define BODY_VCMP REP4(CMP8)
#define CND8 \
  "v_cndmask_b32_e64 %[a0], %[a0], %[k], s[82:83]\n\t" "v_cndmask_b32_e64 %[a1], %[a1], %[k], s[82:83]\n\t" "v_cndmask_b32_e64 %[a2], %[a2], %[k], s[82:83]\n\t" \
  "v_cndmask_b32_e64 %[a3], %[a3], %[k], s[82:83]\n\t" "v_cndmask_b32_e64 %[a4], %[a4], %[k], s[82:83]\n\t" "v_cndmask_b32_e64 %[a5], %[a5], %[k], s[82:83]\n\t" \
  "v_cndmask_b32_e64 %[a6], %[a6], %[k], s[82:83]\n\t" "v_cndmask_b32_e64 %[a7], %[a7], %[k], s[82:83]\n\t"
#define BODY_VCNDMASK REP4(CND8)
#define S8(op) \
  op " s72, s72, s80\n\t" op " s73, s73, s80\n\t" op " s74, s74, s80\n\t" op " s75, s75, s80\n\t" \
  op " s76, s76, s80\n\t" op " s77, s77, s80\n\t" op " s78, s78, s80\n\t" op " s79, s79, s80\n\t"
#define BODY_SADD REP4(S8("s_add_u32"))
#define BODY_SMUL REP4(S8("s_mul_i32"))
#define CS8 \
  "s_cselect_b64 s[72:73], s[74:75], s[76:77]\n\t" "s_cselect_b64 s[84:85], s[74:75], s[76:77]\n\t" "s_cselect_b64 s[86:87], s[74:75], s[76:77]\n\t" \
  "s_cselect_b64 s[88:89], s[74:75], s[76:77]\n\t" "s_cselect_b64 s[72:73], s[74:75], s[76:77]\n\t" "s_cselect_b64 s[84:85], s[74:75], s[76:77]\n\t" \
  "s_cselect_b64 s[86:87], s[74:75], s[76:77]\n\t" "s_cselect_b64 s[88:89], s[74:75], s[76:77]\n\t"
#define BODY_SCSEL64 REP4(CS8)
// compare + never-taken conditional branch (the loop's range < 2^24 test): 16 pairs
#define CB2 "s_cmp_lt_u32 s66, 0x1000000\n\t" "s_cbranch_scc1 9f\n\t"
#define BODY_SCMPBR REP4(CB2 CB2 CB2 CB2)
// never-taken branches alone (branch unit): 32
#define BODY_BR REP4("s_cbranch_scc1 9f\n\t" "s_cbranch_scc1 9f\n\t" "s_cbranch_scc1 9f\n\t" "s_cbranch_scc1 9f\n\t" "s_cbranch_scc1 9f\n\t" "s_cbranch_scc1 9f\n\t" "s_cbranch_scc1 9f\n\t" "s_cbranch_scc1 9f\n\t")
// taken branches: 32 hops to the next instruction's label
#define TB(n) "s_branch 7" #n "f\n\t" "s_nop 0\n\t" "7" #n ":\n\t"
#define BODY_TAKEN REP4(TB(0) TB(1) TB(2) TB(3) TB(4) TB(5) TB(6) TB(7))
// mixes: 16 vector + 16 scalar, alternating (do the two pipes issue side by side from ONE wave? from several?)
#define VS(v, s) "v_add_u32 %[" #v "], %[k], %[" #v "]\n\t" "s_add_u32 " #s ", " #s ", s80\n\t"
#define BODY_MIX_VS REP4(VS(a0, s72) VS(a1, s73) VS(a2, s74) VS(a3, s75))
// the loop's own blend per adaptive tree decision (form B): 5 V (two of them v_readlane), 4 S, 1 never-taken branch -- independent copies
#define DEC(vt, vb, vr, s0, s1) \
  "v_lshrrev_b32 %[" #vt "], 11, s66\n\t" "v_mul_u32_u24 %[" #vb "], %[" #vt "], %[k]\n\t" "v_sub_u32 %[" #vr "], s66, %[" #vb "]\n\t" \
  "v_readlane_b32 " #s0 ", %[" #vb "], s80\n\t" "v_readlane_b32 " #s1 ", %[" #vr "], s80\n\t" \
  "s_sub_u32 s84, s67, " #s0 "\n\t" "s_cselect_b64 s[86:87], s[66:67], s[74:75]\n\t" "s_addc_u32 s85, s85, s85\n\t" \
  "s_cmp_lt_u32 s66, 0x1000000\n\t" "s_cbranch_scc1 9f\n\t"
#define BODY_DECMIX DEC(a0, a1, a2, s72, s73) DEC(a3, a4, a5, s76, s77) DEC(a6, a7, a0, s78, s79) DEC(a1, a2, a3, s72, s73)

// ---- round 5, second batch: the other forms the symbol loop executes (tools/emu/profile.py --pipes lists them) ----
#define F_VLSHR_S(r) "v_lshrrev_b32 %[" #r "], 11, s66\n\t"
#define BODY_VLSHR_S REP4(F_VLSHR_S(a0) F_VLSHR_S(a1) F_VLSHR_S(a2) F_VLSHR_S(a3) F_VLSHR_S(a4) F_VLSHR_S(a5) F_VLSHR_S(a6) F_VLSHR_S(a7))
#define F_VADD_S(r) "v_add_u32 %[" #r "], s66, %[" #r "]\n\t"
#define BODY_VADD_S REP4(F_VADD_S(a0) F_VADD_S(a1) F_VADD_S(a2) F_VADD_S(a3) F_VADD_S(a4) F_VADD_S(a5) F_VADD_S(a6) F_VADD_S(a7))
#define F_VSUB_V(r) "v_sub_u32 %[" #r "], %[k], %[" #r "]\n\t"
#define BODY_VSUB_V REP4(F_VSUB_V(a0) F_VSUB_V(a1) F_VSUB_V(a2) F_VSUB_V(a3) F_VSUB_V(a4) F_VSUB_V(a5) F_VSUB_V(a6) F_VSUB_V(a7))
#define F_VMUL24_S(r) "v_mul_u32_u24 %[" #r "], s80, %[" #r "]\n\t"
#define BODY_VMUL24_S REP4(F_VMUL24_S(a0) F_VMUL24_S(a1) F_VMUL24_S(a2) F_VMUL24_S(a3) F_VMUL24_S(a4) F_VMUL24_S(a5) F_VMUL24_S(a6) F_VMUL24_S(a7))
#define F_VMAD24_S(r) "v_mad_u32_u24 %[" #r "], %[" #r "], 31, s80\n\t"
#define BODY_VMAD24_S REP4(F_VMAD24_S(a0) F_VMAD24_S(a1) F_VMAD24_S(a2) F_VMAD24_S(a3) F_VMAD24_S(a4) F_VMAD24_S(a5) F_VMAD24_S(a6) F_VMAD24_S(a7))
#define F_VBFE_S(r) "v_bfe_u32 %[" #r "], s66, %[" #r "], 1\n\t"
#define BODY_VBFE_S REP4(F_VBFE_S(a0) F_VBFE_S(a1) F_VBFE_S(a2) F_VBFE_S(a3) F_VBFE_S(a4) F_VBFE_S(a5) F_VBFE_S(a6) F_VBFE_S(a7))
#define F_VCND_VCC(r) "v_cndmask_b32 %[" #r "], %[" #r "], %[k], vcc\n\t"
#define BODY_VCND_VCC REP4(F_VCND_VCC(a0) F_VCND_VCC(a1) F_VCND_VCC(a2) F_VCND_VCC(a3) F_VCND_VCC(a4) F_VCND_VCC(a5) F_VCND_VCC(a6) F_VCND_VCC(a7))
#define F_VCMP_VCC_S(r) "v_cmp_eq_u32 vcc, s80, %[" #r "]\n\t"
#define BODY_VCMP_VCC_S REP4(F_VCMP_VCC_S(a0) F_VCMP_VCC_S(a1) F_VCMP_VCC_S(a2) F_VCMP_VCC_S(a3) F_VCMP_VCC_S(a4) F_VCMP_VCC_S(a5) F_VCMP_VCC_S(a6) F_VCMP_VCC_S(a7))
#define F_VCMP_VCC_V(r) "v_cmp_eq_u32 vcc, %[k], %[" #r "]\n\t"
#define BODY_VCMP_VCC_V REP4(F_VCMP_VCC_V(a0) F_VCMP_VCC_V(a1) F_VCMP_VCC_V(a2) F_VCMP_VCC_V(a3) F_VCMP_VCC_V(a4) F_VCMP_VCC_V(a5) F_VCMP_VCC_V(a6) F_VCMP_VCC_V(a7))
#define F_VMOV_S(r) "v_mov_b32 %[" #r "], s66\n\t"
#define BODY_VMOV_S REP4(F_VMOV_S(a0) F_VMOV_S(a1) F_VMOV_S(a2) F_VMOV_S(a3) F_VMOV_S(a4) F_VMOV_S(a5) F_VMOV_S(a6) F_VMOV_S(a7))
#define F_VMOV_C(r) "v_mov_b32 %[" #r "], 0\n\t"
#define BODY_VMOV_C REP4(F_VMOV_C(a0) F_VMOV_C(a1) F_VMOV_C(a2) F_VMOV_C(a3) F_VMOV_C(a4) F_VMOV_C(a5) F_VMOV_C(a6) F_VMOV_C(a7))
#define F_VAND_L(r) "v_and_b32 %[" #r "], 0xffff, %[" #r "]\n\t"
#define BODY_VAND_L REP4(F_VAND_L(a0) F_VAND_L(a1) F_VAND_L(a2) F_VAND_L(a3) F_VAND_L(a4) F_VAND_L(a5) F_VAND_L(a6) F_VAND_L(a7))
#define F_VXOR(r) "v_xor_b32 %[" #r "], %[k], %[" #r "]\n\t"
#define BODY_VXOR REP4(F_VXOR(a0) F_VXOR(a1) F_VXOR(a2) F_VXOR(a3) F_VXOR(a4) F_VXOR(a5) F_VXOR(a6) F_VXOR(a7))
#define F_VLSHLOR(r) "v_lshl_or_b32 %[" #r "], %[" #r "], 16, %[k]\n\t"
#define BODY_VLSHLOR REP4(F_VLSHLOR(a0) F_VLSHLOR(a1) F_VLSHLOR(a2) F_VLSHLOR(a3) F_VLSHLOR(a4) F_VLSHLOR(a5) F_VLSHLOR(a6) F_VLSHLOR(a7))
#define F_VLSHLADD_S(r) "v_lshl_add_u32 %[" #r "], %[" #r "], 3, s80\n\t"
#define BODY_VLSHLADD_S REP4(F_VLSHLADD_S(a0) F_VLSHLADD_S(a1) F_VLSHLADD_S(a2) F_VLSHLADD_S(a3) F_VLSHLADD_S(a4) F_VLSHLADD_S(a5) F_VLSHLADD_S(a6) F_VLSHLADD_S(a7))
#define F_VASHR(r) "v_ashrrev_i32 %[" #r "], 5, %[" #r "]\n\t"
#define BODY_VASHR REP4(F_VASHR(a0) F_VASHR(a1) F_VASHR(a2) F_VASHR(a3) F_VASHR(a4) F_VASHR(a5) F_VASHR(a6) F_VASHR(a7))
#define RLC8 "v_readlane_b32 s72, %[a0], 5\n\t" "v_readlane_b32 s73, %[a1], 6\n\t" "v_readlane_b32 s74, %[a2], 7\n\t" "v_readlane_b32 s75, %[a3], 8\n\t" "v_readlane_b32 s76, %[a4], 9\n\t" "v_readlane_b32 s77, %[a5], 10\n\t" "v_readlane_b32 s78, %[a6], 11\n\t" "v_readlane_b32 s79, %[a7], 12\n\t"
#define BODY_VRL_CONST REP4(RLC8)
#define RFL8 "v_readfirstlane_b32 s72, %[a0]\n\t" "v_readfirstlane_b32 s73, %[a1]\n\t" "v_readfirstlane_b32 s74, %[a2]\n\t" "v_readfirstlane_b32 s75, %[a3]\n\t" "v_readfirstlane_b32 s76, %[a4]\n\t" "v_readfirstlane_b32 s77, %[a5]\n\t" "v_readfirstlane_b32 s78, %[a6]\n\t" "v_readfirstlane_b32 s79, %[a7]\n\t"
#define BODY_VRFL REP4(RFL8)
#define BODY_SADDC REP4(S8("s_addc_u32"))
#define LS8 "s_lshl_b64 s[72:73], s[72:73], 8\n\t" "s_lshl_b64 s[84:85], s[84:85], 8\n\t" "s_lshl_b64 s[86:87], s[86:87], 8\n\t" "s_lshl_b64 s[88:89], s[88:89], 8\n\t" "s_lshl_b64 s[72:73], s[72:73], 8\n\t" "s_lshl_b64 s[84:85], s[84:85], 8\n\t" "s_lshl_b64 s[86:87], s[86:87], 8\n\t" "s_lshl_b64 s[88:89], s[88:89], 8\n\t"
#define BODY_SLSHL64 REP4(LS8)

// ---- third batch: is the 24-cycle v_cndmask_b32 (vcc, VOP2) of the second batch real where the loop uses it -- behind a v_cmp that wrote vcc? ----
#define PV(a, b) "v_cmp_eq_u32 vcc, s80, %[" #a "]\n\t" "v_add_u32 %[" #a "], %[k], %[" #a "]\n\t" "v_xor_b32 %[" #b "], %[k], %[" #b "]\n\t" "v_cndmask_b32 %[" #b "], %[" #b "], %[k], vcc\n\t"
#define BODY_PAIR_VCC REP4(PV(a0, a1) PV(a2, a3)) REP4(PV(a4, a5) PV(a6, a7))
#define PS(a, b) "v_cmp_eq_u32_e64 s[82:83], s80, %[" #a "]\n\t" "v_add_u32 %[" #a "], %[k], %[" #a "]\n\t" "v_xor_b32 %[" #b "], %[k], %[" #b "]\n\t" "v_cndmask_b32_e64 %[" #b "], %[" #b "], %[k], s[82:83]\n\t"
#define BODY_PAIR_SGPR REP4(PS(a0, a1) PS(a2, a3)) REP4(PS(a4, a5) PS(a6, a7))
#define F_VCND_VCC64(r) "v_cndmask_b32_e64 %[" #r "], %[" #r "], %[k], vcc\n\t"
#define BODY_VCND_VCC64 REP4(F_VCND_VCC64(a0) F_VCND_VCC64(a1) F_VCND_VCC64(a2) F_VCND_VCC64(a3) F_VCND_VCC64(a4) F_VCND_VCC64(a5) F_VCND_VCC64(a6) F_VCND_VCC64(a7))
#define BODY_VCND_AFTER "v_cmp_eq_u32 vcc, s80, %[a0]\n\t" "s_nop 4\n\t" BODY_VCND_VCC
#define RLM8 \
  "v_readlane_b32 s72, %[a0], m0\n\t" "v_readlane_b32 s73, %[a1], m0\n\t" "v_readlane_b32 s74, %[a2], m0\n\t" "v_readlane_b32 s75, %[a3], m0\n\t" \
  "v_readlane_b32 s76, %[a4], m0\n\t" "v_readlane_b32 s77, %[a5], m0\n\t" "v_readlane_b32 s78, %[a6], m0\n\t" "v_readlane_b32 s79, %[a7], m0\n\t"
#define BODY_VRL_M0 "s_mov_b32 m0, 5\n\t" REP4(RLM8)

// ---- round 6: the VDIRECT direct-bit block (tools/gen_fast_loop.py) and the scalar block it stands for, as the DEPENDENT chains they are in the loop ----
#define F_VSUBCO(r) "v_sub_co_u32 %[" #r "], vcc, %[k], %[" #r "]\n\t"
#define BODY_VSUBCO REP4(F_VSUBCO(a0) F_VSUBCO(a1) F_VSUBCO(a2) F_VSUBCO(a3) F_VSUBCO(a4) F_VSUBCO(a5) F_VSUBCO(a6) F_VSUBCO(a7))
#define F_VMIN(r) "v_min_u32 %[" #r "], %[k], %[" #r "]\n\t"
#define BODY_VMIN REP4(F_VMIN(a0) F_VMIN(a1) F_VMIN(a2) F_VMIN(a3) F_VMIN(a4) F_VMIN(a5) F_VMIN(a6) F_VMIN(a7))
#define VDB "v_lshrrev_b32 %[a0], 1, %[a0]\n\t" "v_addc_co_u32 %[a1], vcc, %[a1], %[a1], vcc\n\t" "v_sub_co_u32 %[a2], vcc, %[a3], %[a0]\n\t" "v_min_u32 %[a3], %[a3], %[a2]\n\t"
#define BODY_VDBLOCK REP4(VDB VDB) REP4(VDB VDB)
#define SDB "s_lshr_b32 s66, s66, 1\n\t" "s_sub_u32 s75, s67, s66\n\t" "s_cselect_b32 s67, s67, s75\n\t" "s_addc_u32 s87, s87, s87\n\t"
#define BODY_SDBLOCK REP4(SDB SDB) REP4(SDB SDB)

#define OPS \
  : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6), [a7] "+v"(a7) \
  : [k] "v"(k) \
  : "s66", "s67", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "m0", "scc", "vcc"

enum { kVADD, kVMUL24, kVLSHR, kVSUBS, kVMAD24, kVBFE, kVMULLO, kVREADLANE, kVCMP, kVCNDMASK, kSADD, kSMUL, kSCSEL64, kSCMPBR, kBR, kTAKEN,
       kMIX_VS, kDECMIX, kVRL_CONST, kVRFL, kVLSHR_S, kVADD_S, kVSUB_V, kVMUL24_S, kVMAD24_S, kVBFE_S, kVCND_VCC, kVCMP_VCC_S, kVCMP_VCC_V, kVMOV_S, kVMOV_C, kVAND_L, kVXOR, kVLSHLOR, kVLSHLADD_S, kVASHR, kSADDC, kSLSHL64, kPAIR_VCC, kPAIR_SGPR, kVCND_VCC64, kVCND_AFTER, kVRL_M0, kVSUBCO, kVMIN, kVDBLOCK, kSDBLOCK, kCount };
static const char* kNames[kCount] = {"v_add_u32", "v_mul_u32_u24", "v_lshrrev_b32", "v_sub_u32 (sgpr src)", "v_mad_u32_u24", "v_bfe_u32", "v_mul_lo_u32",
                                     "v_readlane_b32", "v_cmp_eq_u32 -> sgpr pair", "v_cndmask_b32 (sgpr mask)", "s_add_u32", "s_mul_i32", "s_cselect_b64",
                                     "s_cmp + s_cbranch (not taken)", "s_cbranch (not taken)", "s_branch (taken)", "mix 1 V : 1 S", "form-B decision blend",
    "v_readlane_b32 (const lane)", "v_readfirstlane_b32", "v_lshrrev_b32 (sgpr src)", "v_add_u32 (sgpr src)", "v_sub_u32 (vgpr srcs)", "v_mul_u32_u24 (sgpr src)", "v_mad_u32_u24 (sgpr src)", "v_bfe_u32 (sgpr src)", "v_cndmask_b32 (vcc, e32)", "v_cmp_eq_u32 vcc (sgpr src)", "v_cmp_eq_u32 vcc (vgpr srcs)", "v_mov_b32 (from sgpr)", "v_mov_b32 (inline const)", "v_and_b32 (32-bit literal)", "v_xor_b32", "v_lshl_or_b32", "v_lshl_add_u32 (sgpr src)", "v_ashrrev_i32", "s_addc_u32", "s_lshl_b64",
    "v_cmp vcc + 2 fillers + v_cndmask vcc", "the same through an SGPR pair (e64)", "v_cndmask_b32_e64 (vcc)", "v_cndmask_b32 vcc after a v_cmp", "v_readlane_b32 (m0 lane)",
    "v_sub_co_u32 (vcc out)", "v_min_u32", "VDIRECT bit block (dependent)", "scalar bit block (dependent)"};
// instructions of the body per iteration, split by pipe: {valu, salu, branch}
static const int kCounts[kCount][3] = {{32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0},
                                       {0, 32, 0}, {0, 32, 0}, {0, 32, 0}, {0, 16, 16}, {0, 0, 32}, {0, 0, 32}, {16, 16, 0}, {20, 16, 4},
    {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {0, 32, 0}, {0, 32, 0},
    {64, 0, 0}, {64, 0, 0}, {32, 0, 0}, {33, 0, 0}, {32, 1, 0},
    {32, 0, 0}, {32, 0, 0}, {64, 0, 0}, {0, 64, 0}};

template <int VAR>
__global__ __launch_bounds__(64) void body(uint64_t* out, int iters) {
  uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, k = 3;
  uint64_t t0, t1;
  asm volatile("s_mov_b32 s66, 0xF0000007\n\ts_mov_b32 s67, 0x12345678\n\ts_mov_b32 s80, 5\n\ts_mov_b64 s[82:83], 0x55\n\t"
               "s_mov_b32 s74, 1\n\ts_mov_b32 s75, 2\n\ts_mov_b32 s76, 3\n\ts_mov_b32 s77, 4\n\ts_cmp_eq_u32 s80, 0" ::
               : "s66", "s67", "s74", "s75", "s76", "s77", "s80", "s82", "s83", "scc");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int i = 0; i < iters; i++) {
#define RUN(B) asm volatile(B B B B "s_branch 8f\n\t9:\n\ts_nop 0\n\t8:\n\t" OPS)
    if constexpr (VAR == kVADD) RUN(BODY_VADD);
    else if constexpr (VAR == kVMUL24) RUN(BODY_VMUL24);
    else if constexpr (VAR == kVLSHR) RUN(BODY_VLSHR);
    else if constexpr (VAR == kVSUBS) RUN(BODY_VSUBS);
    else if constexpr (VAR == kVMAD24) RUN(BODY_VMAD24);
    else if constexpr (VAR == kVBFE) RUN(BODY_VBFE);
    else if constexpr (VAR == kVMULLO) RUN(BODY_VMULLO);
    else if constexpr (VAR == kVREADLANE) RUN(BODY_VREADLANE);
    else if constexpr (VAR == kVCMP) RUN(BODY_VCMP);
    else if constexpr (VAR == kVCNDMASK) RUN(BODY_VCNDMASK);
    else if constexpr (VAR == kSADD) RUN(BODY_SADD);
    else if constexpr (VAR == kSMUL) RUN(BODY_SMUL);
    else if constexpr (VAR == kSCSEL64) RUN(BODY_SCSEL64);
    else if constexpr (VAR == kSCMPBR) RUN(BODY_SCMPBR);
    else if constexpr (VAR == kBR) RUN(BODY_BR);
    else if constexpr (VAR == kTAKEN) RUN(BODY_TAKEN);
    else if constexpr (VAR == kMIX_VS) RUN(BODY_MIX_VS);
    else if constexpr (VAR == kDECMIX) RUN(BODY_DECMIX);
    else if constexpr (VAR == kVRL_CONST) RUN(BODY_VRL_CONST);
    else if constexpr (VAR == kVRFL) RUN(BODY_VRFL);
    else if constexpr (VAR == kVLSHR_S) RUN(BODY_VLSHR_S);
    else if constexpr (VAR == kVADD_S) RUN(BODY_VADD_S);
    else if constexpr (VAR == kVSUB_V) RUN(BODY_VSUB_V);
    else if constexpr (VAR == kVMUL24_S) RUN(BODY_VMUL24_S);
    else if constexpr (VAR == kVMAD24_S) RUN(BODY_VMAD24_S);
    else if constexpr (VAR == kVBFE_S) RUN(BODY_VBFE_S);
    else if constexpr (VAR == kVCND_VCC) RUN(BODY_VCND_VCC);
    else if constexpr (VAR == kVCMP_VCC_S) RUN(BODY_VCMP_VCC_S);
    else if constexpr (VAR == kVCMP_VCC_V) RUN(BODY_VCMP_VCC_V);
    else if constexpr (VAR == kVMOV_S) RUN(BODY_VMOV_S);
    else if constexpr (VAR == kVMOV_C) RUN(BODY_VMOV_C);
    else if constexpr (VAR == kVAND_L) RUN(BODY_VAND_L);
    else if constexpr (VAR == kVXOR) RUN(BODY_VXOR);
    else if constexpr (VAR == kVLSHLOR) RUN(BODY_VLSHLOR);
    else if constexpr (VAR == kVLSHLADD_S) RUN(BODY_VLSHLADD_S);
    else if constexpr (VAR == kVASHR) RUN(BODY_VASHR);
    else if constexpr (VAR == kSADDC) RUN(BODY_SADDC);
    else if constexpr (VAR == kSLSHL64) RUN(BODY_SLSHL64);
    else if constexpr (VAR == kPAIR_VCC) RUN(BODY_PAIR_VCC);
    else if constexpr (VAR == kPAIR_SGPR) RUN(BODY_PAIR_SGPR);
    else if constexpr (VAR == kVCND_VCC64) RUN(BODY_VCND_VCC64);
    else if constexpr (VAR == kVCND_AFTER) RUN(BODY_VCND_AFTER);
    else if constexpr (VAR == kVRL_M0) RUN(BODY_VRL_M0);
    else if constexpr (VAR == kVSUBCO) RUN(BODY_VSUBCO);
    else if constexpr (VAR == kVMIN) RUN(BODY_VMIN);
    else if constexpr (VAR == kVDBLOCK) RUN(BODY_VDBLOCK);
    else RUN(BODY_SDBLOCK);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (threadIdx.x == 0) out[blockIdx.x] = (t1 - t0) + (uint64_t)((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345);
}

static std::vector<int> g_waves = {1, 2, 4, 5, 8};
static const char* g_only = nullptr;   // --only <substring of a variant's name>
static int g_reps = 3;

template <int VAR>
static int run(uint64_t* d_out, double mhz, int iters) {
  if (g_only && !strstr(kNames[VAR], g_only)) return 0;
  printf("%-30s", kNames[VAR]);
  for (int w : g_waves) {
    int blocks = 1024 * w;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(body<VAR>, dim3(blocks), dim3(64), 0, 0, d_out, 50);   // warm-up
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < g_reps; rep++) {
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(body<VAR>, dim3(blocks), dim3(64), 0, 0, d_out, iters);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    std::vector<uint64_t> t(blocks);
    CHECK(hipMemcpy(t.data(), d_out, blocks * sizeof(uint64_t), hipMemcpyDeviceToHost));
    double mean_ticks = 0;
    for (auto x : t) mean_ticks += (double)x;
    mean_ticks /= blocks;
    double cycles = best * 1e-3 * mhz * 1e6;                       // shader cycles of the whole launch (wall clock)
    double per_simd = 4.0 * w * iters;                             // bodies one SIMD executed (four per loop iteration)
    const int* c = kCounts[VAR];
    printf(" | W=%d %7.3f ms", w, best);
    if (c[0]) printf(" V %.3f", per_simd * c[0] / cycles);
    if (c[1]) printf(" S %.3f", per_simd * c[1] / cycles);
    if (c[2]) printf(" B %.3f", per_simd * c[2] / cycles);
    printf(" tick/ns %.2f", mean_ticks / (best * 1e6));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
  }
  printf("\n");
  fflush(stdout);
  return 0;
}

template <int VAR>
static int run_all(uint64_t* d_out, double mhz, int iters) {
  if (run<VAR>(d_out, mhz, iters)) return 1;
  if constexpr (VAR + 1 < kCount) return run_all<VAR + 1>(d_out, mhz, iters);
  return 0;
}

int main(int argc, char** argv) {
  double mhz = 2400;
  int iters = 5000;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--mhz") && i + 1 < argc) mhz = atof(argv[++i]);
    if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
    if (!strcmp(argv[i], "--only") && i + 1 < argc) g_only = argv[++i];
    if (!strcmp(argv[i], "--reps") && i + 1 < argc) g_reps = atoi(argv[++i]);
    if (!strcmp(argv[i], "--waves") && i + 1 < argc) g_waves = {atoi(argv[++i])};
  }
  uint64_t* d_out;
  CHECK(hipMalloc(&d_out, 8192 * sizeof(uint64_t)));
  printf("# pipe_peaks: instructions per shader cycle per SIMD (wall clock at %.0f MHz; 1024 x W one-wave blocks, %d iterations of 4 x 32 instructions)\n", mhz, iters);
  printf("# V = vector ALU, S = scalar ALU, B = branch; tick/ns = s_memtime ticks per wave / wall ns of the launch (<= the counter's rate in GHz)\n");
  return run_all<0>(d_out, mhz, iters);
}
