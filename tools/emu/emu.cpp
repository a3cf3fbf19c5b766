// emu.cpp -- a functional emulator for the gfx950 instruction subset the generated symbol loop uses
// (tools/gen_fast_loop.py).  Test / tuning infrastructure, never part of the product:
//   * runs the asm loop on the CPU, one wavefront, so that generator changes can be checked bit-exactly
//     against the oracle without a GPU (tests/test_asm_emulator.py);
//   * counts how often every instruction executes (and every branch is taken), which gives the exact
//     scalar / vector / branch instruction mix per output byte of a workload (tools/emu/profile.py).
// The program is pre-parsed by tools/emu/asmprog.py into "opcode nargs kind value ..." lines.
// Semantics follow the "AMD Instinct MI300 / CDNA3 ISA" descriptions of each opcode; EXEC is all ones.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

enum Kind : int { K_S = 0, K_V = 1, K_I = 2, K_L = 3, K_VCC = 4 };

struct Arg {
  int kind;
  uint32_t val;
};

struct Ins {
  int op;
  int n;
  Arg a[6];
};

#define OPS(X)                                                                                                        \
  X(s_mov_b32) X(s_movk_i32) X(s_not_b32) X(s_brev_b32) X(s_add_u32) X(s_addc_u32) X(s_sub_u32) X(s_subb_u32)          \
  X(s_and_b32) X(s_or_b32) X(s_xor_b32) X(s_andn2_b32) X(s_lshl_b32) X(s_lshr_b32) X(s_ashr_i32) X(s_min_u32)          \
  X(s_max_u32) X(s_max_i32) X(s_min_i32) X(s_mul_i32) X(s_cselect_b32) X(s_cselect_b64) X(s_lshl2_add_u32)             \
  X(s_lshl1_add_u32) X(s_lshl3_add_u32) X(s_lshl4_add_u32) X(s_and_b64) X(s_or_b64) X(s_mov_b64) X(s_lshl_b64) X(s_lshr_b64) X(s_bfe_u32)               \
  X(s_bfm_b32) X(s_flbit_i32_b32) X(s_ff1_i32_b32) X(s_ff1_i32_b64) X(s_bcnt1_i32_b32) X(s_cmp_eq_u32)                 \
  X(s_cmp_lg_u32) X(s_cmp_gt_u32) X(s_cmp_ge_u32) X(s_cmp_lt_u32) X(s_cmp_le_u32) X(s_cmp_lt_i32) X(s_cmp_gt_i32)      \
  X(s_cmpk_eq_u32) X(s_cmpk_lg_u32) X(s_cmpk_gt_u32) X(s_cmpk_ge_u32) X(s_cmpk_lt_u32) X(s_cmpk_le_u32)                \
  X(s_bitset1_b32) X(s_bitcmp1_b32) X(s_bitcmp0_b32) X(s_bitcmp1_b64) X(s_branch) X(s_cbranch_scc0) X(s_cbranch_scc1)                   \
  X(s_cbranch_vccnz) X(s_cbranch_vccz) X(s_cbranch_execz) X(s_nop) X(s_waitcnt) X(s_getpc_b64) X(s_setpc_b64)          \
  X(s_call_b64) X(s_set_gpr_idx_on) X(s_set_gpr_idx_off) X(s_addk_i32) X(s_sleep) X(s_abs_i32) X(s_setprio)  \
  X(s_getreg_b32) X(s_memtime) X(s_memrealtime) X(s_load_dword)                         \
  X(v_mov_b32) X(v_readlane_b32) X(v_readfirstlane_b32) X(v_writelane_b32) X(v_lshrrev_b32) X(v_lshlrev_b32)           \
  X(v_ashrrev_i32) X(v_add_u32) X(v_sub_u32) X(v_subrev_u32) X(v_and_b32) X(v_or_b32) X(v_xor_b32)                     \
  X(v_mul_u32_u24) X(v_mad_u32_u24) X(v_mad_i32_i24) X(v_mul_lo_u32) X(v_cndmask_b32) X(v_cmp_lt_u32) X(v_cmp_eq_u32) X(v_cmp_gt_u32)   \
  X(v_cmp_ge_u32) X(v_cmp_le_u32) X(v_cmp_ne_u32) X(v_lshl_or_b32) X(v_lshl_add_u32) X(v_add_lshl_u32)                 \
  X(v_and_or_b32) X(v_add3_u32) X(v_bfe_u32) X(v_ffbh_u32) X(v_cvt_f32_u32) X(v_cvt_u32_f32) X(v_rcp_f32)              \
  X(v_add_f32) X(v_mul_f32) X(v_min_u32) X(v_max_u32) X(v_max_i32) X(v_movrels_b32) X(v_movreld_b32) X(v_bfi_b32)                   \
  X(v_alignbit_b32) X(v_or3_b32) X(v_xad_u32) X(v_sub_co_u32) X(v_addc_co_u32) X(v_mbcnt_lo_u32_b32) X(v_mbcnt_hi_u32_b32)              \
  X(ds_read_b128) X(ds_write_b128) X(ds_read_b32) X(ds_write_b32) X(ds_read_u8) X(ds_write_b8) X(ds_read_b64)          \
  X(ds_write_b64) X(buffer_load_ubyte) X(buffer_store_byte) X(buffer_load_dword) X(buffer_store_dword)                 \
  X(buffer_load_dwordx2) X(buffer_load_dwordx4) X(buffer_store_dwordx2) X(buffer_store_dwordx4)                        \
  X(ds_bpermute_b32)

enum Op : int {
#define X(n) OP_##n,
  OPS(X)
#undef X
      OP_COUNT
};

const char* kOpNames[] = {
#define X(n) #n,
    OPS(X)
#undef X
};

struct Emu {
  std::vector<Ins> prog;
  std::vector<uint64_t> counts, taken;
  uint32_t s[128];
  uint64_t vcc;
  uint32_t scc;
  uint32_t m0;
  uint32_t idx_mode;  // bit0 SRC0, bit1 SRC1, bit2 SRC2, bit3 DST
  bool idx_on;
  uint32_t v[256][64];
  std::vector<uint8_t> lds;
  uint8_t* mem;
  uint64_t mem_size;
  uint64_t executed;
  uint32_t hwreg;
  std::string err;
  Emu() : vcc(0), scc(0), m0(0), idx_mode(0), idx_on(false), lds(65536 * 3, 0), mem(nullptr), mem_size(0), executed(0), hwreg(0) {
    memset(s, 0, sizeof(s));
    memset(v, 0, sizeof(v));
  }
};

inline float as_f(uint32_t x) {
  float f;
  memcpy(&f, &x, 4);
  return f;
}
inline uint32_t as_u(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  return x;
}

// scalar-side read of an operand (SGPR, immediate, vcc low); lane-side read adds VGPRs
inline uint32_t rs(Emu& e, const Arg& a) {
  switch (a.kind) {
    case K_S: return a.val == 124 ? e.m0 : e.s[a.val];  // (124 = M0)
    case K_I: return a.val;
    case K_VCC: return uint32_t(e.vcc);
    default: e.err = "scalar read of a non-scalar operand"; return 0;
  }
}
inline uint64_t rs64(Emu& e, const Arg& a) {
  switch (a.kind) {
    case K_S: return uint64_t(e.s[a.val]) | (uint64_t(e.s[a.val + 1]) << 32);
    case K_I: return uint64_t(int64_t(int32_t(a.val)));
    case K_VCC: return e.vcc;
    default: e.err = "64-bit scalar read of a non-scalar operand"; return 0;
  }
}
inline void ws(Emu& e, const Arg& a, uint32_t x) {
  if (a.kind == K_S) {
    if (a.val == 124) e.m0 = x;
    else e.s[a.val] = x;
  } else if (a.kind == K_VCC) e.vcc = (e.vcc & 0xFFFFFFFF00000000ull) | x;
  else e.err = "scalar write to a non-scalar operand";
}
inline void ws64(Emu& e, const Arg& a, uint64_t x) {
  if (a.kind == K_S) {
    e.s[a.val] = uint32_t(x);
    e.s[a.val + 1] = uint32_t(x >> 32);
  } else if (a.kind == K_VCC) e.vcc = x;
  else e.err = "64-bit scalar write to a non-scalar operand";
}
inline uint32_t vreg(Emu& e, const Arg& a, int srcpos) {  // VGPR index after s_set_gpr_idx relocation
  uint32_t r = a.val;
  if (e.idx_on && (e.idx_mode >> srcpos) & 1u) r += e.m0 & 0xFFu;
  return r & 255u;
}
inline uint32_t rl(Emu& e, const Arg& a, int lane, int srcpos) {
  if (a.kind == K_V) return e.v[vreg(e, a, srcpos)][lane];
  return rs(e, a);
}

template <typename F>
inline void vop2(Emu& e, const Ins& I, F f) {  // d = f(src0, src1)
  uint32_t d = vreg(e, I.a[0], 3);
  uint32_t tmp[64];
  for (int l = 0; l < 64; l++) tmp[l] = f(rl(e, I.a[1], l, 0), rl(e, I.a[2], l, 1));
  memcpy(e.v[d], tmp, sizeof(tmp));
}
template <typename F>
inline void vop3(Emu& e, const Ins& I, F f) {
  uint32_t d = vreg(e, I.a[0], 3);
  uint32_t tmp[64];
  for (int l = 0; l < 64; l++) tmp[l] = f(rl(e, I.a[1], l, 0), rl(e, I.a[2], l, 1), rl(e, I.a[3], l, 2));
  memcpy(e.v[d], tmp, sizeof(tmp));
}
template <typename F>
inline void vcmp(Emu& e, const Ins& I, F f) {
  uint64_t m = 0;
  for (int l = 0; l < 64; l++)
    if (f(rl(e, I.a[1], l, 0), rl(e, I.a[2], l, 1))) m |= 1ull << l;
  ws64(e, I.a[0], m);
}

struct Rsrc {
  uint64_t base;
  uint32_t records;
};
inline Rsrc rsrc(Emu& e, const Arg& a) {
  Rsrc r;
  r.base = uint64_t(e.s[a.val]) | (uint64_t(e.s[a.val + 1] & 0xFFFFu) << 32);
  r.records = e.s[a.val + 2];
  return r;
}

long run(Emu& e, int start, long max_steps) {
  int pc = start;
  const int n = int(e.prog.size());
  long steps = 0;
  while (pc < n) {
    if (++steps > max_steps) {
      e.err = "step limit reached";
      return -1;
    }
    const Ins& I = e.prog[pc];
    e.counts[pc]++;
    int next = pc + 1;
    switch (I.op) {
      case OP_s_mov_b32: ws(e, I.a[0], rs(e, I.a[1])); break;
      case OP_s_mov_b64: ws64(e, I.a[0], rs64(e, I.a[1])); break;
      case OP_s_movk_i32: ws(e, I.a[0], uint32_t(int32_t(int16_t(I.a[1].val)))); break;
      case OP_s_addk_i32: {
        int64_t r = int64_t(int32_t(rs(e, I.a[0]))) + int16_t(I.a[1].val);
        ws(e, I.a[0], uint32_t(r));
        e.scc = (r > INT32_MAX || r < INT32_MIN);
      } break;
      case OP_s_not_b32: { uint32_t r = ~rs(e, I.a[1]); ws(e, I.a[0], r); e.scc = r != 0; } break;
      case OP_s_abs_i32: { int32_t x = int32_t(rs(e, I.a[1])); uint32_t r = x < 0 ? uint32_t(-int64_t(x)) : uint32_t(x); ws(e, I.a[0], r); e.scc = r != 0; } break;
      case OP_s_brev_b32: {
        uint32_t x = rs(e, I.a[1]), r = 0;
        for (int i = 0; i < 32; i++) r |= ((x >> i) & 1u) << (31 - i);
        ws(e, I.a[0], r);
      } break;
      case OP_s_add_u32: { uint64_t r = uint64_t(rs(e, I.a[1])) + rs(e, I.a[2]); ws(e, I.a[0], uint32_t(r)); e.scc = uint32_t(r >> 32); } break;
      case OP_s_addc_u32: { uint64_t r = uint64_t(rs(e, I.a[1])) + rs(e, I.a[2]) + e.scc; ws(e, I.a[0], uint32_t(r)); e.scc = uint32_t(r >> 32); } break;
      case OP_s_sub_u32: { uint32_t a = rs(e, I.a[1]), b = rs(e, I.a[2]); ws(e, I.a[0], a - b); e.scc = b > a; } break;
      case OP_s_subb_u32: { uint32_t a = rs(e, I.a[1]), b = rs(e, I.a[2]); uint64_t bb = uint64_t(b) + e.scc; ws(e, I.a[0], uint32_t(a - bb)); e.scc = bb > a; } break;
      case OP_s_and_b32: { uint32_t r = rs(e, I.a[1]) & rs(e, I.a[2]); ws(e, I.a[0], r); e.scc = r != 0; } break;
      case OP_s_andn2_b32: { uint32_t r = rs(e, I.a[1]) & ~rs(e, I.a[2]); ws(e, I.a[0], r); e.scc = r != 0; } break;
      case OP_s_or_b32: { uint32_t r = rs(e, I.a[1]) | rs(e, I.a[2]); ws(e, I.a[0], r); e.scc = r != 0; } break;
      case OP_s_xor_b32: { uint32_t r = rs(e, I.a[1]) ^ rs(e, I.a[2]); ws(e, I.a[0], r); e.scc = r != 0; } break;
      case OP_s_and_b64: { uint64_t r = rs64(e, I.a[1]) & rs64(e, I.a[2]); ws64(e, I.a[0], r); e.scc = r != 0; } break;
      case OP_s_or_b64: { uint64_t r = rs64(e, I.a[1]) | rs64(e, I.a[2]); ws64(e, I.a[0], r); e.scc = r != 0; } break;
      case OP_s_lshl_b32: { uint32_t r = rs(e, I.a[1]) << (rs(e, I.a[2]) & 31u); ws(e, I.a[0], r); e.scc = r != 0; } break;
      case OP_s_lshl_b64: { uint64_t r = rs64(e, I.a[1]) << (rs(e, I.a[2]) & 63u); ws64(e, I.a[0], r); e.scc = r != 0; } break;
      case OP_s_lshr_b64: { uint64_t r = rs64(e, I.a[1]) >> (rs(e, I.a[2]) & 63u); ws64(e, I.a[0], r); e.scc = r != 0; } break;
      case OP_s_lshr_b32: { uint32_t r = rs(e, I.a[1]) >> (rs(e, I.a[2]) & 31u); ws(e, I.a[0], r); e.scc = r != 0; } break;
      case OP_s_ashr_i32: { uint32_t r = uint32_t(int32_t(rs(e, I.a[1])) >> (rs(e, I.a[2]) & 31u)); ws(e, I.a[0], r); e.scc = r != 0; } break;
      case OP_s_min_u32: { uint32_t a = rs(e, I.a[1]), b = rs(e, I.a[2]); ws(e, I.a[0], a < b ? a : b); e.scc = a < b; } break;
      case OP_s_max_u32: { uint32_t a = rs(e, I.a[1]), b = rs(e, I.a[2]); ws(e, I.a[0], a > b ? a : b); e.scc = a > b; } break;
      case OP_s_max_i32: { int32_t a = int32_t(rs(e, I.a[1])), b = int32_t(rs(e, I.a[2])); ws(e, I.a[0], uint32_t(a > b ? a : b)); e.scc = a > b; } break;
      case OP_s_min_i32: { int32_t a = int32_t(rs(e, I.a[1])), b = int32_t(rs(e, I.a[2])); ws(e, I.a[0], uint32_t(a < b ? a : b)); e.scc = a < b; } break;
      case OP_s_mul_i32: ws(e, I.a[0], rs(e, I.a[1]) * rs(e, I.a[2])); break;
      case OP_s_cselect_b32: ws(e, I.a[0], e.scc ? rs(e, I.a[1]) : rs(e, I.a[2])); break;
      case OP_s_cselect_b64: ws64(e, I.a[0], e.scc ? rs64(e, I.a[1]) : rs64(e, I.a[2])); break;
      case OP_s_lshl1_add_u32:
      case OP_s_lshl2_add_u32:
      case OP_s_lshl3_add_u32:
      case OP_s_lshl4_add_u32: {
        int sh = I.op == OP_s_lshl1_add_u32 ? 1 : I.op == OP_s_lshl2_add_u32 ? 2 : I.op == OP_s_lshl3_add_u32 ? 3 : 4;
        uint64_t r = (uint64_t(rs(e, I.a[1])) << sh) + rs(e, I.a[2]);
        ws(e, I.a[0], uint32_t(r));
        e.scc = (r >> 32) != 0;
      } break;
      case OP_s_bfe_u32: {
        uint32_t x = rs(e, I.a[1]), c = rs(e, I.a[2]);
        uint32_t off = c & 31u, w = (c >> 16) & 0x7Fu;
        uint32_t r = w == 0 ? 0 : (w >= 32 ? (x >> off) : ((x >> off) & ((1u << w) - 1u)));
        ws(e, I.a[0], r);
        e.scc = r != 0;
      } break;
      case OP_s_bfm_b32: ws(e, I.a[0], ((1u << (rs(e, I.a[1]) & 31u)) - 1u) << (rs(e, I.a[2]) & 31u)); break;
      case OP_s_flbit_i32_b32: { uint32_t x = rs(e, I.a[1]); ws(e, I.a[0], x ? uint32_t(__builtin_clz(x)) : 0xFFFFFFFFu); } break;
      case OP_s_ff1_i32_b32: { uint32_t x = rs(e, I.a[1]); ws(e, I.a[0], x ? uint32_t(__builtin_ctz(x)) : 0xFFFFFFFFu); } break;
      case OP_s_ff1_i32_b64: { uint64_t x = rs64(e, I.a[1]); ws(e, I.a[0], x ? uint32_t(__builtin_ctzll(x)) : 0xFFFFFFFFu); } break;
      case OP_s_bcnt1_i32_b32: { uint32_t r = uint32_t(__builtin_popcount(rs(e, I.a[1]))); ws(e, I.a[0], r); e.scc = r != 0; } break;
      case OP_s_cmp_eq_u32: e.scc = rs(e, I.a[0]) == rs(e, I.a[1]); break;
      case OP_s_cmp_lg_u32: e.scc = rs(e, I.a[0]) != rs(e, I.a[1]); break;
      case OP_s_cmp_gt_u32: e.scc = rs(e, I.a[0]) > rs(e, I.a[1]); break;
      case OP_s_cmp_ge_u32: e.scc = rs(e, I.a[0]) >= rs(e, I.a[1]); break;
      case OP_s_cmp_lt_u32: e.scc = rs(e, I.a[0]) < rs(e, I.a[1]); break;
      case OP_s_cmp_le_u32: e.scc = rs(e, I.a[0]) <= rs(e, I.a[1]); break;
      case OP_s_cmp_lt_i32: e.scc = int32_t(rs(e, I.a[0])) < int32_t(rs(e, I.a[1])); break;
      case OP_s_cmp_gt_i32: e.scc = int32_t(rs(e, I.a[0])) > int32_t(rs(e, I.a[1])); break;
      case OP_s_cmpk_eq_u32: e.scc = rs(e, I.a[0]) == (I.a[1].val & 0xFFFFu); break;
      case OP_s_cmpk_lg_u32: e.scc = rs(e, I.a[0]) != (I.a[1].val & 0xFFFFu); break;
      case OP_s_cmpk_gt_u32: e.scc = rs(e, I.a[0]) > (I.a[1].val & 0xFFFFu); break;
      case OP_s_cmpk_ge_u32: e.scc = rs(e, I.a[0]) >= (I.a[1].val & 0xFFFFu); break;
      case OP_s_cmpk_lt_u32: e.scc = rs(e, I.a[0]) < (I.a[1].val & 0xFFFFu); break;
      case OP_s_cmpk_le_u32: e.scc = rs(e, I.a[0]) <= (I.a[1].val & 0xFFFFu); break;
      case OP_s_bitset1_b32: ws(e, I.a[0], rs(e, I.a[0]) | (1u << (rs(e, I.a[1]) & 31u))); break;
      case OP_s_bitcmp1_b32: e.scc = (rs(e, I.a[0]) >> (rs(e, I.a[1]) & 31u)) & 1u; break;
      case OP_s_bitcmp0_b32: e.scc = !((rs(e, I.a[0]) >> (rs(e, I.a[1]) & 31u)) & 1u); break;
      case OP_s_bitcmp1_b64: e.scc = uint32_t((rs64(e, I.a[0]) >> (rs(e, I.a[1]) & 63u)) & 1u); break;
      case OP_s_branch: next = int(I.a[0].val); e.taken[pc]++; break;
      case OP_s_cbranch_scc0: if (!e.scc) { next = int(I.a[0].val); e.taken[pc]++; } break;
      case OP_s_cbranch_scc1: if (e.scc) { next = int(I.a[0].val); e.taken[pc]++; } break;
      case OP_s_cbranch_vccnz: if (e.vcc != 0) { next = int(I.a[0].val); e.taken[pc]++; } break;
      case OP_s_cbranch_vccz: if (e.vcc == 0) { next = int(I.a[0].val); e.taken[pc]++; } break;
      case OP_s_cbranch_execz: break;  // EXEC is never zero here
      case OP_s_getreg_b32: ws(e, I.a[0], e.hwreg); break;  // (whatever field is asked for: the value the test set)
      case OP_s_load_dword: {  // sdst, sbase (pair: address in the emulated memory), offset
        const uint64_t a = rs64(e, I.a[1]) + I.a[2].val;
        uint32_t x = 0;
        if (a + 4 > e.mem_size) { e.err = "s_load_dword outside the emulated memory"; break; }
        memcpy(&x, e.mem + a, 4);
        ws(e, I.a[0], x);
      } break;
      case OP_s_memtime:
      case OP_s_memrealtime: ws64(e, I.a[0], e.executed + uint64_t(steps)); break;
      case OP_s_setprio:
      case OP_s_nop:
      case OP_s_sleep:
      case OP_s_waitcnt: break;
      case OP_s_getpc_b64: ws64(e, I.a[0], uint64_t(pc + 1) * 4u); break;
      case OP_s_setpc_b64: { uint64_t t = rs64(e, I.a[0]); if (t & 3u) e.err = "s_setpc_b64 to a misaligned address"; next = int(t >> 2); e.taken[pc]++; } break;
      case OP_s_call_b64: ws64(e, I.a[0], uint64_t(pc + 1) * 4u); next = int(I.a[1].val); e.taken[pc]++; break;
      case OP_s_set_gpr_idx_on: e.m0 = (e.m0 & ~0xFFu) | (rs(e, I.a[0]) & 0xFFu); e.idx_mode = I.a[1].val; e.idx_on = true; break;
      case OP_s_set_gpr_idx_off: e.idx_on = false; break;

      case OP_v_mov_b32: {
        uint32_t d = vreg(e, I.a[0], 3);
        uint32_t tmp[64];
        for (int l = 0; l < 64; l++) tmp[l] = rl(e, I.a[1], l, 0);
        memcpy(e.v[d], tmp, sizeof(tmp));
      } break;
      case OP_v_movrels_b32: {  // D = VGPR[src + M0]
        uint32_t sidx = (I.a[1].val + e.m0) & 255u;
        memcpy(e.v[I.a[0].val], e.v[sidx], sizeof(e.v[0]));
      } break;
      case OP_v_movreld_b32: {  // VGPR[dst + M0] = S0
        uint32_t didx = (I.a[0].val + e.m0) & 255u;
        uint32_t tmp[64];
        for (int l = 0; l < 64; l++) tmp[l] = rl(e, I.a[1], l, 0);
        memcpy(e.v[didx], tmp, sizeof(tmp));
      } break;
      case OP_v_readlane_b32: ws(e, I.a[0], e.v[vreg(e, I.a[1], 0)][rs(e, I.a[2]) & 63u]); break;
      case OP_v_readfirstlane_b32: ws(e, I.a[0], e.v[vreg(e, I.a[1], 0)][0]); break;
      case OP_v_writelane_b32: e.v[I.a[0].val][rs(e, I.a[2]) & 63u] = rs(e, I.a[1]); break;
      case OP_v_lshrrev_b32: vop2(e, I, [](uint32_t a, uint32_t b) { return b >> (a & 31u); }); break;
      case OP_v_lshlrev_b32: vop2(e, I, [](uint32_t a, uint32_t b) { return b << (a & 31u); }); break;
      case OP_v_ashrrev_i32: vop2(e, I, [](uint32_t a, uint32_t b) { return uint32_t(int32_t(b) >> (a & 31u)); }); break;
      case OP_v_add_u32: vop2(e, I, [](uint32_t a, uint32_t b) { return a + b; }); break;
      case OP_v_sub_u32: vop2(e, I, [](uint32_t a, uint32_t b) { return a - b; }); break;
      case OP_v_subrev_u32: vop2(e, I, [](uint32_t a, uint32_t b) { return b - a; }); break;
      case OP_v_and_b32: vop2(e, I, [](uint32_t a, uint32_t b) { return a & b; }); break;
      case OP_v_or_b32: vop2(e, I, [](uint32_t a, uint32_t b) { return a | b; }); break;
      case OP_v_xor_b32: vop2(e, I, [](uint32_t a, uint32_t b) { return a ^ b; }); break;
      case OP_v_min_u32: vop2(e, I, [](uint32_t a, uint32_t b) { return a < b ? a : b; }); break;
      case OP_v_max_u32: vop2(e, I, [](uint32_t a, uint32_t b) { return a > b ? a : b; }); break;
      case OP_v_max_i32: vop2(e, I, [](uint32_t a, uint32_t b) { return uint32_t(int32_t(a) > int32_t(b) ? int32_t(a) : int32_t(b)); }); break;
      case OP_v_mul_u32_u24: vop2(e, I, [](uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }); break;
      case OP_v_mul_lo_u32: vop2(e, I, [](uint32_t a, uint32_t b) { return a * b; }); break;
      case OP_v_mad_u32_u24: vop3(e, I, [](uint32_t a, uint32_t b, uint32_t c) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu) + c; }); break;
      case OP_v_mad_i32_i24:   // (operands: the low 24 bits, sign-extended)
        vop3(e, I, [](uint32_t a, uint32_t b, uint32_t c) {
          return uint32_t((int32_t(a << 8) >> 8) * (int32_t(b << 8) >> 8)) + c;
        });
        break;
      case OP_v_lshl_or_b32: vop3(e, I, [](uint32_t a, uint32_t b, uint32_t c) { return (a << (b & 31u)) | c; }); break;
      case OP_v_lshl_add_u32: vop3(e, I, [](uint32_t a, uint32_t b, uint32_t c) { return (a << (b & 31u)) + c; }); break;
      case OP_v_add_lshl_u32: vop3(e, I, [](uint32_t a, uint32_t b, uint32_t c) { return (a + b) << (c & 31u); }); break;
      case OP_v_and_or_b32: vop3(e, I, [](uint32_t a, uint32_t b, uint32_t c) { return (a & b) | c; }); break;
      case OP_v_or3_b32: vop3(e, I, [](uint32_t a, uint32_t b, uint32_t c) { return a | b | c; }); break;
      case OP_v_add3_u32: vop3(e, I, [](uint32_t a, uint32_t b, uint32_t c) { return a + b + c; }); break;
      case OP_v_xad_u32: vop3(e, I, [](uint32_t a, uint32_t b, uint32_t c) { return (a ^ b) + c; }); break;
      case OP_v_bfi_b32: vop3(e, I, [](uint32_t a, uint32_t b, uint32_t c) { return (a & b) | (~a & c); }); break;
      case OP_v_alignbit_b32: vop3(e, I, [](uint32_t a, uint32_t b, uint32_t c) { return uint32_t(((uint64_t(a) << 32) | b) >> (c & 31u)); }); break;
      case OP_v_bfe_u32: vop3(e, I, [](uint32_t a, uint32_t b, uint32_t c) {
          uint32_t off = b & 31u, w = c & 31u;
          return w == 0 ? 0u : ((a >> off) & ((1u << w) - 1u));
        }); break;
      case OP_v_cndmask_b32: {
        uint32_t d = vreg(e, I.a[0], 3);
        uint64_t m = rs64(e, I.a[3]);
        uint32_t tmp[64];
        for (int l = 0; l < 64; l++) tmp[l] = (m >> l) & 1u ? rl(e, I.a[2], l, 1) : rl(e, I.a[1], l, 0);
        memcpy(e.v[d], tmp, sizeof(tmp));
      } break;
      case OP_v_cmp_lt_u32: vcmp(e, I, [](uint32_t a, uint32_t b) { return a < b; }); break;
      case OP_v_cmp_eq_u32: vcmp(e, I, [](uint32_t a, uint32_t b) { return a == b; }); break;
      case OP_v_cmp_ne_u32: vcmp(e, I, [](uint32_t a, uint32_t b) { return a != b; }); break;
      case OP_v_cmp_gt_u32: vcmp(e, I, [](uint32_t a, uint32_t b) { return a > b; }); break;
      case OP_v_cmp_ge_u32: vcmp(e, I, [](uint32_t a, uint32_t b) { return a >= b; }); break;
      case OP_v_cmp_le_u32: vcmp(e, I, [](uint32_t a, uint32_t b) { return a <= b; }); break;
      case OP_v_sub_co_u32: {  // vdst, sdst(carry), a, b
        uint32_t d = vreg(e, I.a[0], 3);
        uint64_t m = 0;
        uint32_t tmp[64];
        for (int l = 0; l < 64; l++) {
          uint32_t a = rl(e, I.a[2], l, 0), b = rl(e, I.a[3], l, 1);
          tmp[l] = a - b;
          if (b > a) m |= 1ull << l;
        }
        memcpy(e.v[d], tmp, sizeof(tmp));
        ws64(e, I.a[1], m);
      } break;
      case OP_v_addc_co_u32: {  // vdst, sdst(carry out), a, b, ssrc(carry in)
        uint32_t d = vreg(e, I.a[0], 3);
        const uint64_t cin = rs64(e, I.a[4]);
        uint64_t m = 0;
        uint32_t tmp[64];
        for (int l = 0; l < 64; l++) {
          const uint64_t sum = uint64_t(rl(e, I.a[2], l, 0)) + uint64_t(rl(e, I.a[3], l, 1)) + ((cin >> l) & 1u);
          tmp[l] = uint32_t(sum);
          if (sum >> 32) m |= 1ull << l;
        }
        memcpy(e.v[d], tmp, sizeof(tmp));
        ws64(e, I.a[1], m);
      } break;
      case OP_v_mbcnt_lo_u32_b32:
      case OP_v_mbcnt_hi_u32_b32: {
        uint32_t d = vreg(e, I.a[0], 3);
        uint32_t tmp[64];
        for (int l = 0; l < 64; l++) {
          uint32_t mask = rl(e, I.a[1], l, 0), acc = rl(e, I.a[2], l, 1);
          uint32_t lm;
          if (I.op == OP_v_mbcnt_lo_u32_b32) lm = l >= 32 ? 0xFFFFFFFFu : ((1u << l) - 1u);
          else lm = l <= 32 ? 0u : ((1u << (l - 32)) - 1u);
          tmp[l] = acc + uint32_t(__builtin_popcount(mask & lm));
        }
        memcpy(e.v[d], tmp, sizeof(tmp));
      } break;
      case OP_v_ffbh_u32: {
        uint32_t d = vreg(e, I.a[0], 3);
        uint32_t tmp[64];
        for (int l = 0; l < 64; l++) { uint32_t x = rl(e, I.a[1], l, 0); tmp[l] = x ? uint32_t(__builtin_clz(x)) : 0xFFFFFFFFu; }
        memcpy(e.v[d], tmp, sizeof(tmp));
      } break;
      case OP_v_cvt_f32_u32: {
        uint32_t d = vreg(e, I.a[0], 3);
        uint32_t tmp[64];
        for (int l = 0; l < 64; l++) tmp[l] = as_u(float(rl(e, I.a[1], l, 0)));
        memcpy(e.v[d], tmp, sizeof(tmp));
      } break;
      case OP_v_cvt_u32_f32: {
        uint32_t d = vreg(e, I.a[0], 3);
        uint32_t tmp[64];
        for (int l = 0; l < 64; l++) {
          float f = as_f(rl(e, I.a[1], l, 0));
          tmp[l] = !(f > 0.0f) ? 0u : (f >= 4294967296.0f ? 0xFFFFFFFFu : uint32_t(f));
        }
        memcpy(e.v[d], tmp, sizeof(tmp));
      } break;
      case OP_v_rcp_f32: {
        uint32_t d = vreg(e, I.a[0], 3);
        uint32_t tmp[64];
        for (int l = 0; l < 64; l++) tmp[l] = as_u(1.0f / as_f(rl(e, I.a[1], l, 0)));
        memcpy(e.v[d], tmp, sizeof(tmp));
      } break;
      case OP_v_add_f32: vop2(e, I, [](uint32_t a, uint32_t b) { return as_u(as_f(a) + as_f(b)); }); break;
      case OP_v_mul_f32: vop2(e, I, [](uint32_t a, uint32_t b) { return as_u(as_f(a) * as_f(b)); }); break;

      case OP_ds_read_b128:
      case OP_ds_read_b64:
      case OP_ds_read_b32:
      case OP_ds_read_u8: {
        int nb = I.op == OP_ds_read_b128 ? 16 : I.op == OP_ds_read_b64 ? 8 : I.op == OP_ds_read_b32 ? 4 : 1;
        uint32_t off = I.n > 2 ? I.a[2].val : 0;
        for (int l = 0; l < 64; l++) {
          uint32_t a = e.v[I.a[1].val][l] + off;
          if (a + nb > e.lds.size()) { e.err = "LDS read out of range"; break; }
          if (nb == 1) e.v[I.a[0].val][l] = e.lds[a];
          else for (int k = 0; k < nb / 4; k++) memcpy(&e.v[I.a[0].val + k][l], &e.lds[a + 4 * k], 4);
        }
      } break;
      case OP_ds_write_b128:
      case OP_ds_write_b64:
      case OP_ds_write_b32:
      case OP_ds_write_b8: {
        int nb = I.op == OP_ds_write_b128 ? 16 : I.op == OP_ds_write_b64 ? 8 : I.op == OP_ds_write_b32 ? 4 : 1;
        uint32_t off = I.n > 2 ? I.a[2].val : 0;
        for (int l = 0; l < 64; l++) {
          uint32_t a = e.v[I.a[0].val][l] + off;
          if (a + nb > e.lds.size()) { e.err = "LDS write out of range"; break; }
          if (nb == 1) e.lds[a] = uint8_t(e.v[I.a[1].val][l]);
          else for (int k = 0; k < nb / 4; k++) memcpy(&e.lds[a + 4 * k], &e.v[I.a[1].val + k][l], 4);
        }
      } break;
      case OP_ds_bpermute_b32: {
        uint32_t tmp[64];
        for (int l = 0; l < 64; l++) tmp[l] = e.v[I.a[2].val][(e.v[I.a[1].val][l] >> 2) & 63u];
        memcpy(e.v[I.a[0].val], tmp, sizeof(tmp));
      } break;
      case OP_buffer_load_ubyte:
      case OP_buffer_load_dword: {  // vdst, vaddr, srsrc, soffset   (offen)
        Rsrc r = rsrc(e, I.a[2]);
        uint32_t so = rs(e, I.a[3]);
        int nb = I.op == OP_buffer_load_ubyte ? 1 : 4;
        uint32_t tmp[64];
        for (int l = 0; l < 64; l++) {
          uint64_t off = uint64_t(e.v[I.a[1].val][l]) + so;
          uint32_t val = 0;
          if (off + nb <= r.records) {
            uint64_t a = r.base + off;
            if (a + nb > e.mem_size) { e.err = "buffer load outside the emulated memory"; break; }
            memcpy(&val, e.mem + a, nb);
          }
          tmp[l] = val;
        }
        memcpy(e.v[I.a[0].val], tmp, sizeof(tmp));
      } break;
      case OP_buffer_load_dwordx2:
      case OP_buffer_load_dwordx4: {  // v[d : d + n - 1], vaddr, srsrc, soffset   (offen): dword k of a lane into register d + k
        Rsrc r = rsrc(e, I.a[2]);
        uint32_t so = rs(e, I.a[3]);
        const int nd = I.op == OP_buffer_load_dwordx2 ? 2 : 4;
        uint32_t addr[64];
        memcpy(addr, e.v[I.a[1].val], sizeof(addr));
        for (int l = 0; l < 64; l++) {
          uint64_t off = uint64_t(addr[l]) + so;
          for (int k = 0; k < nd; k++) {
            uint32_t val = 0;
            if (off + 4 * k + 4 <= r.records) {
              uint64_t a = r.base + off + 4 * k;
              if (a + 4 > e.mem_size) { e.err = "buffer load outside the emulated memory"; break; }
              memcpy(&val, e.mem + a, 4);
            }
            e.v[I.a[0].val + k][l] = val;
          }
        }
      } break;
      case OP_buffer_store_dwordx2:
      case OP_buffer_store_dwordx4: {
        Rsrc r = rsrc(e, I.a[2]);
        uint32_t so = rs(e, I.a[3]);
        const int nd = I.op == OP_buffer_store_dwordx2 ? 2 : 4;
        for (int l = 0; l < 64; l++) {
          uint64_t off = uint64_t(e.v[I.a[1].val][l]) + so;
          for (int k = 0; k < nd; k++)
            if (off + 4 * k + 4 <= r.records) {
              uint64_t a = r.base + off + 4 * k;
              if (a + 4 > e.mem_size) { e.err = "buffer store outside the emulated memory"; break; }
              memcpy(e.mem + a, &e.v[I.a[0].val + k][l], 4);
            }
        }
      } break;
      case OP_buffer_store_byte:
      case OP_buffer_store_dword: {  // vdata, vaddr, srsrc, soffset   (offen); lanes in ascending order
        Rsrc r = rsrc(e, I.a[2]);
        uint32_t so = rs(e, I.a[3]);
        int nb = I.op == OP_buffer_store_byte ? 1 : 4;
        for (int l = 0; l < 64; l++) {
          uint64_t off = uint64_t(e.v[I.a[1].val][l]) + so;
          if (off + nb <= r.records) {
            uint64_t a = r.base + off;
            if (a + nb > e.mem_size) { e.err = "buffer store outside the emulated memory"; break; }
            memcpy(e.mem + a, &e.v[I.a[0].val][l], nb);
          }
        }
      } break;
      default: e.err = std::string("unimplemented opcode ") + kOpNames[I.op]; break;
    }
    if (!e.err.empty()) {
      char buf[64];
      snprintf(buf, sizeof(buf), " (instruction %d)", pc);
      e.err += buf;
      return -1;
    }
    pc = next;
  }
  e.executed += uint64_t(steps);
  return steps;
}

}  // namespace

extern "C" {

void* emu_create(const char* program) {
  static std::map<std::string, int> names;
  if (names.empty())
    for (int i = 0; i < OP_COUNT; i++) names[kOpNames[i]] = i;
  Emu* e = new Emu();
  const char* p = program;
  while (*p) {
    const char* eol = strchr(p, '\n');
    std::string line(p, eol ? size_t(eol - p) : strlen(p));
    p = eol ? eol + 1 : p + line.size();
    if (line.empty()) continue;
    char op[64];
    int n = 0, used = 0;
    if (sscanf(line.c_str(), "%63s %d%n", op, &n, &used) < 2) { e->err = "bad program line: " + line; return e; }
    auto it = names.find(op);
    if (it == names.end()) { e->err = std::string("unknown opcode: ") + op; return e; }
    Ins I;
    memset(&I, 0, sizeof(I));
    I.op = it->second;
    I.n = n;
    const char* q = line.c_str() + used;
    for (int i = 0; i < n && i < 6; i++) {
      int k, adv = 0;
      unsigned long long v;
      if (sscanf(q, " %d %llu%n", &k, &v, &adv) < 2) { e->err = "bad operand in: " + line; return e; }
      I.a[i].kind = k;
      I.a[i].val = uint32_t(v);
      q += adv;
    }
    e->prog.push_back(I);
  }
  e->counts.assign(e->prog.size(), 0);
  e->taken.assign(e->prog.size(), 0);
  return e;
}
void emu_destroy(void* h) { delete static_cast<Emu*>(h); }
const char* emu_error(void* h) { return static_cast<Emu*>(h)->err.c_str(); }
void emu_set_mem(void* h, uint8_t* mem, uint64_t size) {
  Emu* e = static_cast<Emu*>(h);
  e->mem = mem;
  e->mem_size = size;
}
void emu_set_s(void* h, int i, uint32_t x) { static_cast<Emu*>(h)->s[i] = x; }
uint32_t emu_get_s(void* h, int i) { return static_cast<Emu*>(h)->s[i]; }
void emu_set_v(void* h, int i, const uint32_t* lanes) { memcpy(static_cast<Emu*>(h)->v[i], lanes, 256); }
void emu_get_v(void* h, int i, uint32_t* lanes) { memcpy(lanes, static_cast<Emu*>(h)->v[i], 256); }
void emu_set_hwreg(void* h, uint32_t x) { static_cast<Emu*>(h)->hwreg = x; }
void emu_set_vcc(void* h, uint64_t x) { static_cast<Emu*>(h)->vcc = x; }
void emu_lds_write(void* h, uint32_t off, const uint8_t* src, uint32_t n) { memcpy(&static_cast<Emu*>(h)->lds[off], src, n); }
void emu_lds_read(void* h, uint32_t off, uint8_t* dst, uint32_t n) { memcpy(dst, &static_cast<Emu*>(h)->lds[off], n); }
long emu_run(void* h, int start, long max_steps) { return run(*static_cast<Emu*>(h), start, max_steps); }
void emu_counts(void* h, uint64_t* counts, uint64_t* taken) {
  Emu* e = static_cast<Emu*>(h);
  memcpy(counts, e->counts.data(), e->counts.size() * 8);
  memcpy(taken, e->taken.data(), e->taken.size() * 8);
}
void emu_reset_counts(void* h) {
  Emu* e = static_cast<Emu*>(h);
  std::fill(e->counts.begin(), e->counts.end(), 0);
  std::fill(e->taken.begin(), e->taken.end(), 0);
}
int emu_size(void* h) { return int(static_cast<Emu*>(h)->prog.size()); }
}
