// decode_fast.hip.h -- the fast LZMA / LZMA2 decode kernel for pb <= 2 and lc + lp <= 3
// (which includes the ubiquitous lc3/lp0/pb2).
//
// Same contract and same reference behaviour as decode_generic.hip.h; what differs is where the
// probability model lives and how one binary decision is issued.  Measured on MI355X
// (experiments/microbench): a lone wave issues one instruction per ~5-7 cycles, a taken branch
// costs ~40, the scalar ALU is one instruction per cycle per CU, and an LDS round trip adds ~80
// cycles to every decision.  So:
//
//   * the whole model sits in VGPRs, one probability per lane ("lane-resident"): a probability
//     is fetched with v_readlane and updated in place under a one-lane EXEC mask -- no LDS at
//     all, so 16 waves (= 16 streams) fit per CU and a 4096-stream batch is resident at once;
//   * `range` is an SGPR and `code` a (wave-uniform) VGPR, which splits the ~20 instructions of
//     a decision evenly between the scalar ALU and the four SIMDs;
//   * a decision contains no branch; the only branch is the (cold, out-of-line) normalisation;
//   * the literal table (8 rows x 0x300 probabilities) is packed two-per-dword and split: the
//     plain sub-tables (256 probabilities per row) live in 16 VGPRs, the two matched sub-tables
//     (512 per row, 8 KiB in all) in LDS, which still leaves room for 16 waves per CU.  The row
//     of the current literal is staged into unpacked registers (dynamic row -> hipcc's
//     s_set_gpr_idx indexing / one ds_read_b128 prefetched a symbol ahead), decoded, packed back.
//
// Reference (behaviour mirrored): src/decode/rangecoder.rs, src/decode/lzma.rs:164-593,
// src/decode/lzbuffer.rs, src/decode/lzma2.rs:52-229.
#pragma once
#include "device_common.h"

namespace milzma {

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t kNoByte = 0xFFFFFFFFu;

// Cycle attribution for tuning (make prof): per-unit s_memtime deltas, reported through
// milzma_result.err_a / err_b of successful units.  Not part of the product build.
#ifdef MILZMA_PROFILE
#define MILZMA_PROF_NOW() uint32_t(__builtin_amdgcn_s_memtime())
#define MILZMA_PROF_ADD(acc, t0) (acc) += MILZMA_PROF_NOW() - (t0)
#else
#define MILZMA_PROF_NOW() 0u
#define MILZMA_PROF_ADD(acc, t0) (void)(t0)
#endif

struct FastDecoder {
  // ---- input: 256-byte window (dword per lane) that always has >= 64 bytes ahead of `off` at the
  //      start of a symbol, so normalisation never has to refill mid-symbol ---------------------
  rsrc_t in_rsrc;          // input bytes from a 64-byte aligned base up to the 64-byte block of the last valid byte
  uint32_t wbase;          // virtual position of the window's first byte (multiple of 64)
  uint32_t off;            // next byte = window byte `off`
  uint32_t rem;            // bytes until the decoder's EOF (reader end or take() limit)
  uint32_t end;            // virtual position one past the unit's last byte
  uint32_t win, win_next;  // VGPR: window [wbase, wbase+256) and [wbase+192, wbase+448)
  // ---- range coder ---------------------------------------------------------------------------
  uint32_t range;  // SGPR
  uint32_t code;   // VGPR, same value in every lane
  uint32_t eof;    // sticky: normalize() wanted a byte past the limit
  uint32_t k2048, k31;  // VGPR constants for the probability update (v_cndmask cannot take two scalars)
  // ---- model: one probability per lane -----------------------------------------------------------
  uint32_t m_ismatch;   // [state*4 + pos_state] (48); 48 len.choice 49 len.choice2 50 replen.choice 51 replen.choice2
  uint32_t m_rep;       // [k*12 + state], k: 0 is_rep, 1 is_rep_g0, 2 is_rep_g1, 3 is_rep_g2
  uint32_t m_rep0long;  // [state*4 + pos_state] (48); align tree node n at lane 48 + n
  u32x4* m_posslot;     // -> kernel-local vector: [len_state] -> tree of 64, node n at lane n
  uint32_t m_posdec_a;  // pos_decoders[0..63] at lane idx (reverse trees of slots 4..11)
  uint32_t m_posdec_b;  // slot 12 node n at lane n, slot 13 node n at lane 32 + n
  uint32_t m_len_lm, m_rlen_lm;  // low[ps][8] at lane ps*8+n, mid[ps][8] at lane 32+ps*8+n
  uint32_t m_len_h0, m_len_h1, m_rlen_h0, m_rlen_h1;  // high tree nodes 1..63 / 64..127
  uint32_t m_len_h2, m_len_h3, m_rlen_h2, m_rlen_h3;  // high tree nodes 128..191 / 192..255
  // Literal rows, packed 2 x u16 per dword (probability n of a sub-table: lane n & 63, half (n >> 6) & 1
  // of dword n >> 7).  Plain sub-table of row r: (*lit_plain)[2r], (*lit_plain)[2r + 1] (a kernel-local
  // vector: dynamically indexed vectors must be their own allocas for hipcc to keep them -- and the
  // rest of this struct -- in registers).  Matched sub-tables of row r: lit_matched[r*256 + lane*4 + k],
  // k = 0,1 for match bit 0 and k = 2,3 for match bit 1.
  u32x16* lit_plain;
  uint32_t* lit_matched;  // LDS, 8 rows x 64 lanes x 4 dwords
  uint32_t mrow[4];       // VGPR: this lane's 4 dwords of the prefetched matched row
  uint32_t mrow_row;      // which row mrow holds (0xFFFFFFFF = none)
  // ---- output / LZ state (wave-uniform) -----------------------------------------------------------
  rsrc_t out_rsrc;  // the unit's output slice (out_cap bytes)
  uint32_t out_lim;
  uint32_t lim_is_mem;
  uint32_t dict_base, len, dict_size;
  uint32_t lc, lp, pb;
  uint32_t state, rep0, rep1, rep2, rep3;
  uint32_t prev, mb;  // last output byte / byte after the last match source; kNoByte = not known
  // A short match is split in two: its load is issued when the match is decoded, its store (and the
  // extraction of prev / mb from the loaded lanes) happens when the next symbol needs them or the
  // next match is about to be copied.  The ~1.5 us HBM/MALL latency of the load then overlaps with
  // decoding the next symbol instead of stalling the wave (rocprof: s_waitcnt was 28 % of wave time).
  uint32_t pend_n;    // bytes of the pending match (0 = none); its lanes 0..pend_n hold src[0..pend_n]
  uint32_t pend_pos;  // where they go
  uint32_t pend_val;  // VGPR
  uint32_t status;
  uint32_t prof_lit, prof_copy, prof_match, prof_nlit;  // MILZMA_PROFILE only
  milzma_result* res;  // error arguments go straight to the result record (keeps them out of SGPRs)

  __device__ __forceinline__ void fail(uint32_t st, uint64_t a = 0, uint64_t b = 0) {
    status = st;
    res->err_a = a;  // every lane stores the same two values (no lane-dependent branch next to a loop exit:
    res->err_b = b;  // hipcc would treat everything live across that loop as divergent)
  }

  // hipcc's register allocator may split the live range of a scalar around a region that does not
  // use it and park the value in a VGPR; getting it back for an inline-asm "s" operand is then an
  // illegal copy.  An empty asm that names the scalars inside such regions keeps them in SGPRs.
  __device__ __forceinline__ void pin_scalars() const { asm volatile("" ::"s"(range), "s"(off), "s"(rem)); }

  // ---- reader ----------------------------------------------------------------------------------
  __device__ __forceinline__ uint32_t load_window(uint32_t wpos) const {
    // lane's dword of the window starting at virtual position wpos (0 beyond the resource's end).
    // The value is passed through a v_mov so that the load is waited for HERE (once per 192 input
    // bytes): a window register that is still "in flight" across the symbol loop makes hipcc put an
    // s_waitcnt vmcnt(0) at the top of every iteration, which also waits for every outstanding
    // output store (rocprof: 28 % of wave time in s_waitcnt).
    uint32_t w = buf_load_u32(in_rsrc, wpos + threadIdx.x * 4u);
    asm volatile("v_mov_b32 %0, %0" : "+v"(w));
    return w;
  }
  __device__ __forceinline__ uint32_t vpos() const { return wbase + off; }
  __device__ __forceinline__ void seek(uint32_t v) {
    wbase = v & ~63u;
    off = v - wbase;
    win = load_window(wbase);
    win_next = load_window(wbase + 192u);
  }
  // call between symbols: keeps >= 64 bytes of window ahead of `off`
  __device__ __forceinline__ void slide() {
    if (__builtin_expect(off >= 192u, 0)) {
      win = win_next;
      wbase += 192u;
      off -= 192u;
      win_next = load_window(wbase + 192u);
    }
  }
  __device__ __forceinline__ uint32_t take_byte() {  // precondition rem > 0, off < 256
    const uint32_t w = readlane(win, off >> 2);
    const uint32_t b = (w >> ((off & 3u) * 8u)) & 0xffu;
    off++;
    rem--;
    return b;
  }
  // header bytes between symbols (may cross the window end)
  __device__ __forceinline__ uint32_t header_byte() {
    slide();
    return take_byte();
  }

  // ---- range decoder -------------------------------------------------------------------------------
  __device__ __forceinline__ void set_code(uint32_t c) { asm volatile("v_mov_b32 %0, %1" : "=v"(code) : "s"(c)); }

  __device__ __forceinline__ bool rc_init() {  // RangeDecoder::new (rangecoder.rs:20-30)
    if (rem < 5) {
      off += rem;
      rem = 0;
      return false;
    }
    slide();
    (void)take_byte();
    uint32_t c = take_byte();
    c = (c << 8) | take_byte();
    c = (c << 8) | take_byte();
    c = (c << 8) | take_byte();
    set_code(c);
    range = 0xFFFFFFFFu;
    return true;
  }

  // RangeDecoder::normalize (rangecoder.rs:59-69); cold path of every decision
  __device__ __forceinline__ void normalize() {
    range <<= 8;
    if (__builtin_expect(rem == 0, 0)) {
      eof = 1;
      asm volatile("v_lshlrev_b32 %0, 8, %0" : "+v"(code));
    } else {
      const uint32_t b = take_byte();
      asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(code) : "s"(b));
    }
  }

  // RangeDecoder::decode_bit (rangecoder.rs:92-120) on the probability held by one lane of T.
  // Returns 2*sym + bit.  10 scalar + 8 vector instructions, no branch.  The lane is
  //   LANE_IS_A:   a            LANE_A_PLUS_B: a + b            LANE_A_AND_63: a & 63
  // computed inside the asm: left to hipcc, some of these scalar adds get moved to the VALU and the
  // result handed to v_readlane as a VGPR, which does not assemble.
  enum { LANE_IS_A = 0, LANE_A_PLUS_B = 1, LANE_A_AND_63 = 2 };
#define MILZMA_BIT_BODY                                                                                      \
  "v_readlane_b32 %[sp], %[T], %[ln]\n\t"                                                                    \
  "s_lshr_b32 %[sb], %[range], 11\n\t"                                                                       \
  "s_mul_i32 %[sb], %[sb], %[sp]\n\t"            /* bound */                                                 \
  "v_cmp_ge_u32 vcc, %[code], %[sb]\n\t"          /* bit (all lanes agree) */                                \
  "v_subrev_u32 %[vt], %[sb], %[code]\n\t"                                                                   \
  "v_min_u32 %[code], %[code], %[vt]\n\t"         /* code -= bound if bit */                                 \
  "s_lshl_b64 exec, 1, %[ln]\n\t"                 /* only the owning lane updates its probability */         \
  "v_cndmask_b32 %[vt], %[k2048], %[k31], vcc\n\t"  /* K = bit ? 31 : 2048 */                          \
  "v_sub_u32 %[vt], %[vt], %[T]\n\t"                                                                         \
  "v_ashrrev_i32 %[vt], 5, %[vt]\n\t"             /* (K - p) >> 5 arithmetic: -(p >> 5) or (2048 - p) >> 5 */ \
  "v_add_u32 %[T], %[T], %[vt]\n\t"               /* bit ? p - (p >> 5) : p + ((2048 - p) >> 5) */           \
  "s_mov_b64 exec, -1\n\t"                                                                                   \
  "s_sub_u32 %[sr1], %[range], %[sb]\n\t"                                                                    \
  "s_cmp_lg_u64 vcc, 0\n\t"                                                                                  \
  "s_cselect_b32 %[range], %[sr1], %[sb]\n\t"                                                                \
  "s_addc_u32 %[sym], %[sym], %[sym]"
  template <int MODE>
  __device__ __forceinline__ uint32_t bitm(uint32_t& T, uint32_t a, uint32_t b, uint32_t sym) {
    uint32_t sp, sb, sr1, vt, ln;
    if (MODE == LANE_IS_A) {
      asm volatile(MILZMA_BIT_BODY
                   : [T] "+v"(T), [range] "+s"(range), [code] "+v"(code), [sym] "+s"(sym), [sp] "=&s"(sp),
                     [sb] "=&s"(sb), [sr1] "=&s"(sr1), [vt] "=&v"(vt)
                   : [ln] "s"(a), [k2048] "v"(k2048), [k31] "v"(k31)
                   : "vcc", "scc");
    } else if (MODE == LANE_A_PLUS_B) {
      asm volatile("s_add_u32 %[ln], %[a], %[b]\n\t" MILZMA_BIT_BODY
                   : [T] "+v"(T), [range] "+s"(range), [code] "+v"(code), [sym] "+s"(sym), [sp] "=&s"(sp),
                     [sb] "=&s"(sb), [sr1] "=&s"(sr1), [vt] "=&v"(vt), [ln] "=&s"(ln)
                   : [a] "s"(a), [b] "s"(b), [k2048] "v"(k2048), [k31] "v"(k31)
                   : "vcc", "scc");
    } else {
      asm volatile("s_and_b32 %[ln], %[a], 63\n\t" MILZMA_BIT_BODY
                   : [T] "+v"(T), [range] "+s"(range), [code] "+v"(code), [sym] "+s"(sym), [sp] "=&s"(sp),
                     [sb] "=&s"(sb), [sr1] "=&s"(sr1), [vt] "=&v"(vt), [ln] "=&s"(ln)
                   : [a] "s"(a), [k2048] "v"(k2048), [k31] "v"(k31)
                   : "vcc", "scc");
    }
    if (__builtin_expect(range < kTop, 0)) normalize();
    return sym;
  }
  __device__ __forceinline__ uint32_t bit(uint32_t& T, uint32_t lane, uint32_t sym) {
    return bitm<LANE_IS_A>(T, lane, 0, sym);
  }
  // probability at lane base + node, node = the tree walk's running symbol
  __device__ __forceinline__ uint32_t bit_at(uint32_t& T, uint32_t base, uint32_t node, uint32_t sym) {
    return bitm<LANE_A_PLUS_B>(T, base, node, sym);
  }
  // probability at lane node & 63 (levels 6 and 7 of the 8-level trees)
  __device__ __forceinline__ uint32_t bit_lo6(uint32_t& T, uint32_t node, uint32_t sym) {
    return bitm<LANE_A_AND_63>(T, node, 0, sym);
  }

  // RangeDecoder::get_bit (rangecoder.rs:71-82): one direct bit; returns 2*sym + bit
  __device__ __forceinline__ uint32_t direct_bit(uint32_t sym) {
    uint32_t vt;
    asm volatile(
        "s_lshr_b32 %[range], %[range], 1\n\t"
        "v_cmp_ge_u32 vcc, %[code], %[range]\n\t"
        "v_subrev_u32 %[vt], %[range], %[code]\n\t"
        "v_min_u32 %[code], %[code], %[vt]\n\t"
        "s_cmp_lg_u64 vcc, 0\n\t"
        "s_addc_u32 %[sym], %[sym], %[sym]"
        : [range] "+s"(range), [code] "+v"(code), [sym] "+s"(sym), [vt] "=&v"(vt)
        :
        : "vcc", "scc");
    if (__builtin_expect(range < kTop, 0)) normalize();
    return sym;
  }

  // a decision whose probability register is one of two, chosen by a wave-uniform flag
  // (selects instead of a branch: a taken branch costs more than the three extra v_cndmask)
  template <bool MASK63 = true>
  __device__ __forceinline__ uint32_t bit_sel(uint32_t& a, uint32_t& b, uint32_t pick_b, uint32_t lane, uint32_t sym) {
    const bool pb_ = pick_b != 0;
    uint32_t x = pb_ ? b : a;
    sym = MASK63 ? bit_lo6(x, lane, sym) : bit(x, lane, sym);
    a = pb_ ? a : x;
    b = pb_ ? x : b;
    return sym;
  }

  // bit-tree of NBITS levels rooted at lane `base` + 1 of T (node n at lane base + n), n < 64 - base
  template <uint32_t NBITS>
  __device__ __forceinline__ uint32_t tree(uint32_t& T, uint32_t base) {
    uint32_t sym = 1;
#pragma unroll
    for (uint32_t i = 0; i < NBITS; i++) sym = bit_at(T, base, sym, sym);
    return sym - (1u << NBITS);
  }

  // reverse bit-tree (rangecoder.rs:136-151): same walk, result is the bit-reversed path
  __device__ __forceinline__ uint32_t reverse_tree(uint32_t& T, uint32_t base, uint32_t nbits) {
    uint32_t sym = 1;  // nbits is 1..5; unrolled with one forward exit (backward branches are dear)
#pragma unroll
    for (uint32_t i = 0; i < 5; i++) {
      if (i >= nbits) break;
      sym = bit_at(T, base, sym, sym);
    }
    return __builtin_bitreverse32(sym << (32u - nbits));  // drops the leading 1, reverses the nbits below it
  }

  // LenDecoder::decode (rangecoder.rs:256-269); which = 0 len_decoder, 1 rep_len_decoder
  __device__ __forceinline__ uint32_t len_decode(uint32_t which, uint32_t pos_state, uint32_t& lm, uint32_t& h0,
                                                 uint32_t& h1, uint32_t& h2, uint32_t& h3) {
    if (!bit(m_ismatch, 48u + which * 2u, 0)) return tree<3>(lm, pos_state * 8u);
    if (!bit(m_ismatch, 49u + which * 2u, 0)) return tree<3>(lm, 32u + pos_state * 8u) + 8u;
    uint32_t sym = 1;
#pragma unroll
    for (uint32_t i = 0; i < 6; i++) sym = bit(h0, sym, sym);
    sym = bit_lo6(h1, sym, sym);
    sym = bit_sel(h2, h3, (sym >> 6) & 1u, sym, sym);  // nodes 128..191 in h2, 192..255 in h3
    return sym - 256u + 16u;
  }

  // ---- model reset (DecoderState::new / reset_state) ---------------------------------------------------
  __device__ __forceinline__ void reset_model() {
    const uint32_t p = 0x400u, pp = 0x04000400u;
    m_ismatch = m_rep = m_rep0long = p;
    *m_posslot = u32x4{p, p, p, p};
    m_posdec_a = m_posdec_b = p;
    m_len_lm = m_rlen_lm = m_len_h0 = m_len_h1 = m_rlen_h0 = m_rlen_h1 = p;
    m_len_h2 = m_len_h3 = m_rlen_h2 = m_rlen_h3 = p;
#pragma unroll
    for (int i = 0; i < 16; i++) (*lit_plain)[i] = pp;
#pragma unroll
    for (int r = 0; r < 8; r++) {
      uint4* p4 = reinterpret_cast<uint4*>(lit_matched + r * 256 + threadIdx.x * 4);
      *p4 = uint4{pp, pp, pp, pp};
    }
    mrow_row = 0xFFFFFFFFu;
    state = 0;
    rep0 = rep1 = rep2 = rep3 = 0;
  }

  // ---- output window (same scheme as the generic kernel) --------------------------------------------------
  __device__ __forceinline__ uint32_t opos() const { return dict_base + len; }
  __device__ __forceinline__ void finish_pending() {
    if (pend_n != 0) {
      buf_store_u8(out_rsrc, threadIdx.x < pend_n ? pend_pos + threadIdx.x : kOob, pend_val);
      prev = readlane(pend_val, pend_n - 1u);
      mb = readlane(pend_val, pend_n);
      pend_n = 0;
    }
  }
  __device__ __forceinline__ uint32_t fetch_out(uint32_t pos) {
    finish_pending();
    return readfirst(buf_load_u8(out_rsrc, pos));
  }
  __device__ __forceinline__ void limit_error() {
    if (lim_is_mem)
      fail(MILZMA_ST_MEMLIMIT, out_lim);
    else
      fail(MILZMA_ST_OUT_FULL);
  }
  __device__ __forceinline__ bool append_literal(uint32_t byte) {
    const uint32_t pos = opos();
    if (__builtin_expect(pos >= out_lim, 0)) {
      limit_error();
      return false;
    }
    buf_store_u8(out_rsrc, threadIdx.x == 0 ? pos : kOob, byte);
    len++;
    prev = byte;
    mb = kNoByte;
    return true;
  }
  __device__ __forceinline__ bool append_lz(uint32_t mlen, uint32_t dist, uint32_t size_known, uint32_t target) {
    if (__builtin_expect(dist > dict_size, 0)) {
      fail(MILZMA_ST_LZ_DIST_DICT, dist, dict_size);
      return false;
    }
    if (__builtin_expect(dist > len, 0)) {
      fail(MILZMA_ST_LZ_DIST_OUT, dist, len);
      return false;
    }
    const uint32_t pos = opos();
    uint32_t n = mlen;
    bool clipped = false;
    if (__builtin_expect(pos + mlen > out_lim || pos + mlen < pos, 0)) {
      n = out_lim > pos ? out_lim - pos : 0;
      clipped = true;
    }
    const uint32_t src = pos - dist;
    const bool periodic = dist <= n;
    const float rcp = periodic ? __builtin_amdgcn_rcpf(float(dist)) : 0.0f;
    const uint32_t t_copy = MILZMA_PROF_NOW();
    finish_pending();  // its store must precede this match's loads (the source may overlap it)
    if (__builtin_expect(n < kWave && !clipped, 1)) {
      // one chunk: issue the load now, store later (finish_pending)
      const uint32_t i = threadIdx.x;
      const uint32_t j = periodic ? small_mod(i, dist, rcp) : i;
      pend_val = buf_load_u8(out_rsrc, i <= n ? src + j : kOob);
      pend_pos = pos;
      pend_n = n;
      len += mlen;
      MILZMA_PROF_ADD(prof_copy, t_copy);
      return true;
    }
    for (uint32_t i0 = 0; i0 <= n; i0 += kWave) {
      pin_scalars();
      const uint32_t i = i0 + threadIdx.x;
      const uint32_t j = periodic ? small_mod(i, dist, rcp) : i;
      const uint32_t val = buf_load_u8(out_rsrc, i <= n ? src + j : kOob);
      buf_store_u8(out_rsrc, i < n ? pos + i : kOob, val);
      if (n > 0 && n - 1 >= i0 && n - 1 < i0 + kWave) prev = readlane(val, (n - 1) & 63u);
      if (n >= i0 && n < i0 + kWave) mb = readlane(val, n & 63u);
    }
    if (__builtin_expect(clipped, 0)) {
      if (!lim_is_mem && size_known && out_lim - dict_base >= target) {
        len += mlen;
        prev = mb = kNoByte;
        return true;
      }
      len += n;
      limit_error();
      return false;
    }
    len += mlen;
    return true;
  }

  // ---- literal (lzma.rs:526-561) -----------------------------------------------------------------------
  __device__ __forceinline__ uint32_t literal_row() const {
    return ((len & ((1u << lp) - 1u)) << lc) + (prev >> (8u - lc));
  }
  // issues the LDS read of this lane's slice of a matched row; consumed (if at all) by decode_literal
  __device__ __forceinline__ void prefetch_matched_row(uint32_t row) {
    const uint4 w = *reinterpret_cast<const uint4*>(lit_matched + row * 256u + threadIdx.x * 4u);
    mrow[0] = w.x;
    mrow[1] = w.y;
    mrow[2] = w.z;
    mrow[3] = w.w;
    mrow_row = row;
  }

  __device__ __forceinline__ bool decode_literal(uint32_t* byte_out) {
    finish_pending();
    if (__builtin_expect(prev == kNoByte, 0)) prev = len == 0 ? 0 : fetch_out(opos() - 1);
    const uint32_t row = literal_row();
    const bool matched = state >= 7;
    uint32_t match_byte = 0;
    if (matched) {
      const uint32_t dist = rep0 + 1;
      if (__builtin_expect(dist > dict_size || dist == 0, 0)) {
        fail(MILZMA_ST_MATCH_DIST_DICT, uint64_t(rep0) + 1, dict_size);
        return false;
      }
      if (__builtin_expect(dist > len, 0)) {
        fail(MILZMA_ST_MATCH_DIST_OUT, dist, len);
        return false;
      }
      match_byte = mb != kNoByte ? mb : fetch_out(opos() - dist);
    }
    // stage the row: packed dwords -> registers holding one probability per lane
    const uint32_t p0 = (*lit_plain)[row * 2u], p1 = (*lit_plain)[row * 2u + 1u];
    uint32_t u[12];
    u[0] = p0 & 0xffffu;
    u[1] = p0 >> 16;
    u[2] = p1 & 0xffffu;
    u[3] = p1 >> 16;
    if (matched) {
      if (__builtin_expect(mrow_row != row, 0)) prefetch_matched_row(row);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        u[4 + 2 * k] = mrow[k] & 0xffffu;
        u[5 + 2 * k] = mrow[k] >> 16;
      }
    } else {
#pragma unroll
      for (int k = 4; k < 12; k++) u[k] = 0;
    }
    uint32_t sym = 1;
    if (matched) {
      // probs[((1 + match_bit) << 8) + sym]: slot 4 (1 + match_bit) + (sym >> 6), lane sym & 63.
      // While the decoded bits equal the match byte's bits stay here; the first mismatch jumps
      // into the plain chain at the next level (one taken branch per matched literal).
      uint32_t mbit;
#define MILZMA_MATCHED(A, B, LANE_MASK, NEXT)               \
  mbit = (match_byte >> 7) & 1u;                           \
  match_byte <<= 1;                                        \
  sym = bit_sel(A, B, mbit, sym, sym);                     \
  if ((sym & 1u) != mbit) goto NEXT;
      MILZMA_MATCHED(u[4], u[8], 63u, plain1)
      MILZMA_MATCHED(u[4], u[8], 63u, plain2)
      MILZMA_MATCHED(u[4], u[8], 63u, plain3)
      MILZMA_MATCHED(u[4], u[8], 63u, plain4)
      MILZMA_MATCHED(u[4], u[8], 63u, plain5)
      MILZMA_MATCHED(u[4], u[8], 63u, plain6)
      MILZMA_MATCHED(u[5], u[9], 63u, plain7)
#undef MILZMA_MATCHED
      mbit = (match_byte >> 7) & 1u;
      if (sym & 64u)  // nodes 192..255
        sym = bit_sel(u[7], u[11], mbit, sym, sym);
      else
        sym = bit_sel(u[6], u[10], mbit, sym, sym);
      goto literal_done;
    }
    sym = bit(u[0], sym, sym);
  plain1:
    sym = bit(u[0], sym, sym);
  plain2:
    sym = bit(u[0], sym, sym);
  plain3:
    sym = bit(u[0], sym, sym);
  plain4:
    sym = bit(u[0], sym, sym);
  plain5:
    sym = bit(u[0], sym, sym);
  plain6:
    sym = bit_lo6(u[1], sym, sym);
  plain7:
    sym = bit_sel(u[2], u[3], (sym >> 6) & 1u, sym, sym);
  literal_done:
    // pack the row back
    (*lit_plain)[row * 2u] = u[0] | (u[1] << 16);
    (*lit_plain)[row * 2u + 1u] = u[2] | (u[3] << 16);
    if (matched) {
      uint4 w;
      w.x = u[4] | (u[5] << 16);
      w.y = u[6] | (u[7] << 16);
      w.z = u[8] | (u[9] << 16);
      w.w = u[10] | (u[11] << 16);
      *reinterpret_cast<uint4*>(lit_matched + row * 256u + threadIdx.x * 4u) = w;
      mrow_row = 0xFFFFFFFFu;
    }
    *byte_out = sym & 0xffu;
    return true;
  }

  // ---- distance (lzma.rs:563-592) ----------------------------------------------------------------------
  __device__ __forceinline__ uint32_t decode_distance(uint32_t length) {
    const uint32_t len_state = length > 3 ? 3 : length;
    uint32_t ps = (*m_posslot)[len_state];
    const uint32_t pos_slot = tree<6>(ps, 0);
    (*m_posslot)[len_state] = ps;
    if (pos_slot < 4) return pos_slot;
    const uint32_t ndb = (pos_slot >> 1) - 1;
    uint32_t result = (2u | (pos_slot & 1u)) << ndb;
    if (pos_slot < 12) {
      result += reverse_tree(m_posdec_a, result - pos_slot, ndb);  // nodes at pos_decoders[result - pos_slot + n]
    } else if (pos_slot < 14) {
      result += reverse_tree(m_posdec_b, (pos_slot - 12u) * 32u, 5);
    } else {
      uint32_t d = 0, cnt = ndb - 4;  // 2..26 direct bits
      for (; cnt >= 4; cnt -= 4) {
        d = direct_bit(d);
        d = direct_bit(d);
        d = direct_bit(d);
        d = direct_bit(d);
      }
      if (cnt & 2) {
        d = direct_bit(d);
        d = direct_bit(d);
      }
      if (cnt & 1) d = direct_bit(d);
      result += d << 4;
      result += reverse_tree(m_rep0long, 48, 4);  // align decoder
    }
    return result;
  }

  // ---- process_mode(Finish) (lzma.rs:435-524) ---------------------------------------------------------------
  __device__ __forceinline__ bool process(uint32_t size_known, uint32_t target, uint32_t target_clamped) {
    const uint32_t pb_mask = (1u << pb) - 1u;
    for (;;) {
      if (size_known) {
        if (len >= target && !target_clamped) break;
      } else if (rem == 0 && readfirst(code) == 0) {  // is_finished_ok
        break;
      }
      slide();
      if (state >= 7 && pend_n == 0 && prev != kNoByte) prefetch_matched_row(literal_row());  // a literal here would be a matched one
      const uint32_t pos_state = len & pb_mask;
      if (!bit(m_ismatch, state * 4u + pos_state, 0)) {
        uint32_t byte;
        if (__builtin_expect(eof, 0)) return fail(MILZMA_ST_INPUT_EOF), false;
        const uint32_t t_lit = MILZMA_PROF_NOW();
        if (!decode_literal(&byte)) return false;
        MILZMA_PROF_ADD(prof_lit, t_lit);
        if (__builtin_expect(eof, 0)) return fail(MILZMA_ST_INPUT_EOF), false;
        if (!append_literal(byte)) return false;
        state = state < 4 ? 0 : (state < 10 ? state - 3 : state - 6);
        continue;
      }
      uint32_t mlen;
      if (bit(m_rep, state, 0)) {
        if (!bit(m_rep, 12u + state, 0)) {
          if (!bit(m_rep0long, state * 4u + pos_state, 0)) {
            if (__builtin_expect(eof, 0)) return fail(MILZMA_ST_INPUT_EOF), false;
            state = state < 7 ? 9 : 11;
            if (!append_lz(1, rep0 + 1, size_known, target)) return false;
            continue;
          }
        } else {
          uint32_t dist;
          if (!bit(m_rep, 24u + state, 0)) {
            dist = rep1;
          } else {
            if (!bit(m_rep, 36u + state, 0)) {
              dist = rep2;
            } else {
              dist = rep3;
              rep3 = rep2;
            }
            rep2 = rep1;
          }
          rep1 = rep0;
          rep0 = dist;
        }
        mlen = len_decode(1, pos_state, m_rlen_lm, m_rlen_h0, m_rlen_h1, m_rlen_h2, m_rlen_h3);
        state = state < 7 ? 8 : 11;
        if (__builtin_expect(eof, 0)) return fail(MILZMA_ST_INPUT_EOF), false;
      } else {
        rep3 = rep2;
        rep2 = rep1;
        rep1 = rep0;
        const uint32_t t_len = MILZMA_PROF_NOW();
        mlen = len_decode(0, pos_state, m_len_lm, m_len_h0, m_len_h1, m_len_h2, m_len_h3);
        MILZMA_PROF_ADD(prof_nlit, t_len);
        state = state < 7 ? 7 : 10;
        const uint32_t t_dist = MILZMA_PROF_NOW();
        rep0 = decode_distance(mlen);
        MILZMA_PROF_ADD(prof_match, t_dist);
        if (__builtin_expect(eof, 0)) return fail(MILZMA_ST_INPUT_EOF), false;
        if (__builtin_expect(rep0 == 0xFFFFFFFFu, 0)) {
          if (rem == 0 && readfirst(code) == 0) return true;
          return fail(MILZMA_ST_MARKER_TRAILING), false;
        }
      }
      if (!append_lz(mlen + 2, rep0 + 1, size_known, target)) return false;
    }
    if (size_known && (target_clamped || len != target)) {
      status = MILZMA_ST_SIZE_MISMATCH;  // the caller fills in the two sizes
      return false;
    }
    return true;
  }
};

__device__ __forceinline__ bool fast_props_ok(uint32_t lc, uint32_t lp, uint32_t pb) { return pb <= 2 && lc + lp <= 3; }

__global__ __launch_bounds__(64, 4) void decode_fast_kernel(const milzma_unit* __restrict__ units,
                                                            const uint32_t* __restrict__ order, uint32_t n_units,
                                                            const uint8_t* in_base, uint8_t* out_base,
                                                            milzma_result* results) {
  if (blockIdx.x >= n_units) return;
  const uint32_t uidx = order[blockIdx.x];
  const milzma_unit u = units[uidx];
  milzma_result* res = results + uidx;

  if (u.in_len > MILZMA_MAX_UNIT_BYTES || u.out_cap > MILZMA_MAX_UNIT_BYTES || u.lc > 8 || u.lp > 4 || u.pb > 4 ||
      (u.kind != MILZMA_KIND_RAW_LZMA && u.kind != MILZMA_KIND_LZMA2)) {
    store_result(res, MILZMA_ST_BAD_UNIT, 0, 0, 0, 0, 0, 0);
    return;
  }
  const bool raw = u.kind == MILZMA_KIND_RAW_LZMA;
  if (raw && !fast_props_ok(u.lc, u.lp, u.pb)) {
    store_result(res, MILZMA_ST_NEED_GENERIC, 0, 0, 0, 0, 0, 0);
    return;
  }

  __shared__ uint32_t lds_matched[8 * 64 * 4];
  u32x16 lit_plain;
  u32x4 posslot;
  const uint32_t t_kernel0 = MILZMA_PROF_NOW();
  (void)t_kernel0;
  FastDecoder d;
  d.lit_plain = &lit_plain;
  d.lit_matched = lds_matched;
  d.m_posslot = &posslot;
  d.out_rsrc = make_rsrc(out_base + u.out_off, uint32_t(u.out_cap));
  d.status = MILZMA_ST_OK;
  d.res = res;
  d.prof_lit = d.prof_copy = d.prof_match = d.prof_nlit = 0;
  d.eof = 0;
  d.dict_base = 0;
  d.len = 0;
  d.prev = 0;
  d.pend_n = 0;
  d.pend_pos = 0;
  asm volatile("v_mov_b32 %0, 0" : "=v"(d.pend_val));
  d.mb = kNoByte;
  d.range = 0;
  d.set_code(0);
  asm volatile("v_mov_b32 %0, 0x800\n\tv_mov_b32 %1, 31" : "=v"(d.k2048), "=v"(d.k31));
  {
    const uint8_t* p = in_base + u.in_off;
    const uint32_t a0 = uint32_t(reinterpret_cast<uintptr_t>(p) & 63u);
    d.end = a0 + uint32_t(u.in_len);
    d.in_rsrc = make_rsrc(p - a0, (d.end + 63u) & ~63u);
    d.rem = uint32_t(u.in_len);
    d.seek(a0);
  }
  const uint32_t a0 = d.vpos();
  uint32_t chunks = 0;
  bool ok = true;

  if (raw) {
    d.lc = u.lc;
    d.lp = u.lp;
    d.pb = u.pb;
    d.dict_size = u.dict_size;
    const uint64_t mem_eff = u.memlimit < uint64_t(u.dict_size) ? u.memlimit : UINT64_MAX;
    d.lim_is_mem = mem_eff <= u.out_cap ? 1u : 0u;
    d.out_lim = uint32_t(d.lim_is_mem ? mem_eff : u.out_cap);
  } else {
    d.lc = d.lp = d.pb = 0;
    d.dict_size = 0xFFFFFFFFu;
    d.lim_is_mem = 0;
    d.out_lim = uint32_t(u.out_cap);
  }
  d.reset_model();

  for (bool first = true;; first = false) {
    uint32_t known = 1, clamped = 0;
    uint64_t target64 = 0;
    uint32_t rem_after = 0;  // bytes that remain visible after the chunk's take() window
    if (raw) {
      if (!first) break;
      known = u.unpacked_size != MILZMA_SIZE_UNKNOWN ? 1u : 0u;
      target64 = u.unpacked_size;
      clamped = (known && target64 > 0xFFFFFFFFull) ? 1u : 0u;
    } else {
      if (d.rem == 0) {
        d.fail(MILZMA_ST_L2_STATUS_EOF);
        ok = false;
        break;
      }
      const uint32_t status = d.header_byte();
      if (status == 0) break;
      chunks++;
      if (status == 1 || status == 2) {
        if (d.rem < 2) {
          d.off += d.rem;
          d.rem = 0;
          d.fail(MILZMA_ST_L2_UNPACKED_EOF);
          ok = false;
          break;
        }
        uint32_t n = d.header_byte() << 8;
        n = (n | d.header_byte()) + 1;
        d.finish_pending();
        if (status == 1) {
          d.dict_base += d.len;
          d.len = 0;
        }
        if (d.rem < n) {
          d.off += d.rem;
          d.rem = 0;
          d.fail(MILZMA_ST_L2_STORED_EOF, n);
          ok = false;
          break;
        }
        const uint32_t pos = d.opos();
        if (uint64_t(pos) + n > d.out_lim) {
          d.fail(MILZMA_ST_OUT_FULL);
          ok = false;
          break;
        }
        const uint32_t v = d.vpos();
        for (uint32_t i0 = 0; i0 < n; i0 += kWave) {  // uniform trip count, lanes predicated by offset
          d.pin_scalars();
          const uint32_t i = i0 + threadIdx.x;
          buf_store_u8(d.out_rsrc, i < n ? pos + i : kOob, buf_load_u8(d.in_rsrc, i < n ? v + i : kOob));
        }
        d.len += n;
        d.prev = d.mb = kNoByte;
        d.rem -= n;
        d.seek(v + n);
        continue;
      }
      if ((status & 0x80u) == 0) {
        d.fail(MILZMA_ST_L2_INVALID_STATUS, status);
        ok = false;
        break;
      }
      const uint32_t reset = (status >> 5) & 3u;
      if (d.rem < 2) {
        d.off += d.rem;
        d.rem = 0;
        d.fail(MILZMA_ST_L2_UNPACKED_EOF);
        ok = false;
        break;
      }
      uint32_t unpacked = (status & 0x1Fu) << 16;
      unpacked |= d.header_byte() << 8;
      unpacked = (unpacked | d.header_byte()) + 1;
      if (d.rem < 2) {
        d.off += d.rem;
        d.rem = 0;
        d.fail(MILZMA_ST_L2_PACKED_EOF);
        ok = false;
        break;
      }
      uint32_t packed = d.header_byte() << 8;
      packed = (packed | d.header_byte()) + 1;
      if (reset == 3) {
        d.finish_pending();
        d.dict_base += d.len;
        d.len = 0;
        d.prev = d.mb = kNoByte;
      }
      if (reset >= 1) {
        if (reset >= 2) {
          if (d.rem == 0) {
            d.fail(MILZMA_ST_L2_PROPS_EOF);
            ok = false;
            break;
          }
          uint32_t pbv = d.header_byte();
          if (pbv >= 225) {
            d.fail(MILZMA_ST_L2_PROPS_INVALID, pbv);
            ok = false;
            break;
          }
          const uint32_t lc = pbv % 9;
          pbv /= 9;
          const uint32_t lp = pbv % 5;
          pbv /= 5;
          if (lc + lp > 4) {
            d.fail(MILZMA_ST_L2_LCLP, lc, lp);
            ok = false;
            break;
          }
          if (!fast_props_ok(lc, lp, pbv)) {  // valid props this kernel is not specialised for
            d.fail(MILZMA_ST_NEED_GENERIC);
            ok = false;
            break;
          }
          d.lc = lc;
          d.lp = lp;
          d.pb = pbv;
        }
        d.reset_model();
      }
      if (d.rem > packed) {
        rem_after = d.rem - packed;
        d.rem = packed;
      } else if (d.rem < packed) {
        chunks |= 0x80000000u;
      }
      target64 = uint64_t(unpacked) + d.len;
      clamped = target64 > 0xFFFFFFFFull ? 1u : 0u;
    }
    if (!d.rc_init()) {
      d.rem += rem_after;
      d.fail(MILZMA_ST_RC_INIT);
      ok = false;
      break;
    }
    ok = d.process(known, uint32_t(target64), clamped);
    if (!ok && d.status == MILZMA_ST_SIZE_MISMATCH) d.fail(MILZMA_ST_SIZE_MISMATCH, target64, uint64_t(d.len));
    d.rem += rem_after;
    if (!ok) break;
  }
  d.finish_pending();
  const uint64_t total = uint64_t(d.dict_base) + d.len;
  uint64_t flushed = total;
  if (!ok) flushed = raw ? (total / d.dict_size) * d.dict_size : d.dict_base;
  // err_a / err_b were written by fail(); zero them on success (all lanes store the same values)
  res->status = d.status;
  res->chunks = chunks;
  res->out_len = total;
  res->out_flushed = flushed;
  res->in_consumed = d.vpos() - a0;
  if (d.status == MILZMA_ST_OK) {
#ifdef MILZMA_PROFILE
    res->err_a = (uint64_t(d.prof_lit) << 32) | d.prof_copy;
    res->err_b = (uint64_t(d.prof_match) << 32) | ((MILZMA_PROF_NOW() - t_kernel0) >> 4);
    res->out_flushed = (uint64_t(d.prof_nlit) << 32) | uint32_t(res->out_flushed);
#else
    res->err_a = 0;
    res->err_b = 0;
#endif
  }
}

}  // namespace milzma
