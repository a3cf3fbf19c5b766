// decode_generic.hip.h -- the general LZMA / LZMA2 decode kernel (any lc/lp/pb).
//
// One wavefront per unit.  The probability model sits in LDS as u16 (literal table spilled to
// HBM when 0x300 << (lc+lp) does not fit the launch's LDS class); range/code/state/reps are
// wave-uniform (hipcc keeps them in SGPRs and issues the arithmetic on the scalar ALU); the
// lanes cooperate on the input window, the LZ77 copy and model resets.  The output slice in
// HBM doubles as the dictionary: a match reads out[pos - dist .. ] straight from the slice.
//
// Mirrors, in behaviour and error order, the reference's
//   RangeDecoder            src/decode/rangecoder.rs:7-151
//   BitTree / LenDecoder    src/decode/rangecoder.rs:153-270
//   DecoderState            src/decode/lzma.rs:164-593 (process_mode(Finish) path, update = true)
//   LzCircularBuffer        src/decode/lzbuffer.rs:167-321  (RAW_LZMA units)
//   LzAccumBuffer           src/decode/lzbuffer.rs:38-165   (LZMA2 units)
//   Lzma2Decoder            src/decode/lzma2.rs:52-229      (packet walk, done by the wave itself)
#pragma once
#include "device_common.h"

namespace milzma {

// model layout, u16 indices (src/decode/lzma.rs:165-186)
enum : uint32_t {
  M_IS_MATCH = 0,        // [12 << 4]
  M_IS_REP = 192,        // [12]
  M_IS_REP_G0 = 204,     // [12]
  M_IS_REP_G1 = 216,     // [12]
  M_IS_REP_G2 = 228,     // [12]
  M_IS_REP_0LONG = 240,  // [12 << 4]
  M_POS_SLOT = 432,      // [4][64]
  M_ALIGN = 688,         // [16]
  M_POS_DEC = 704,       // [115] (+1 pad)
  M_LEN = 820,           // LenDecoder: choice, choice2, low[16][8], mid[16][8], high[256]
  M_REP_LEN = 1334,      // LenDecoder
  M_SMALL_END = 1848,    // literal table follows when it lives in LDS
  L_CHOICE = 0,
  L_CHOICE2 = 1,
  L_LOW = 2,
  L_MID = 130,
  L_HIGH = 258,
};

template <bool LIT_IN_LDS>
struct GenericDecoder {
  // ---- wave-uniform state -------------------------------------------------------------
  Reader rd;
  uint32_t range, code;
  uint16_t* model;        // LDS
  uint16_t* lit;          // LDS (LIT_IN_LDS) or HBM scratch
  uint8_t* out;           // unit's output slice
  uint32_t out_lim;       // first position that may not be written
  bool lim_is_mem;        // out_lim comes from Options.memlimit (else from out_cap)
  uint32_t dict_base;     // bytes before the last dictionary reset (LZMA2); 0 for RAW
  uint32_t len;           // LzBuffer::len(): bytes since the last dictionary reset
  uint32_t dict_size;     // RAW: LzCircularBuffer.dict_size; LZMA2: unbounded (0xFFFFFFFF)
  uint32_t lc, lp, pb;
  uint32_t state;
  uint32_t rep0, rep1, rep2, rep3;
  uint32_t prev, mb;      // last output byte / byte following the last match source
  bool prev_valid, mb_valid;
  bool eof;               // sticky: normalize() wanted a byte past the reader's limit
  uint32_t status;
  uint64_t err_a, err_b;

  __device__ __forceinline__ void fail(uint32_t st, uint64_t a = 0, uint64_t b = 0) {
    status = st;
    err_a = a;
    err_b = b;
  }

  // ---- model ---------------------------------------------------------------------------
  __device__ __forceinline__ void reset_model() {  // DecoderState::new / reset_state: everything to 0x400
    const uint32_t lane = threadIdx.x;
    uint32_t* m32 = reinterpret_cast<uint32_t*>(model);
    for (uint32_t i = lane; i < M_SMALL_END / 2; i += kWave) m32[i] = 0x04000400u;
    const uint32_t nlit = (0x300u << (lc + lp)) / 2;
    uint32_t* l32 = reinterpret_cast<uint32_t*>(lit);
    for (uint32_t i = lane; i < nlit; i += kWave) l32[i] = 0x04000400u;
    state = 0;
    rep0 = rep1 = rep2 = rep3 = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }

  // ---- range decoder ---------------------------------------------------------------------
  // RangeDecoder::new (rangecoder.rs:20-30)
  __device__ __forceinline__ bool rc_init() {
    if (rd.lim - rd.v < 5) {
      rd.v = rd.lim;  // read_u8 / read_exact consumed what there was
      return false;
    }
    (void)reader_byte(rd);  // first byte is ignored
    uint32_t c = reader_byte(rd);
    c = (c << 8) | reader_byte(rd);
    c = (c << 8) | reader_byte(rd);
    c = (c << 8) | reader_byte(rd);
    code = c;
    range = 0xFFFFFFFFu;
    return true;
  }

  // RangeDecoder::normalize (rangecoder.rs:59-69): at most one byte
  __device__ __forceinline__ void normalize() {
    if (__builtin_expect(range < kTop, 0)) {
      range <<= 8;
      if (__builtin_expect(reader_eof(rd), 0)) {
        eof = true;
        code <<= 8;
      } else {
        code = (code << 8) | reader_byte(rd);
      }
    }
  }

  // RangeDecoder::decode_bit (rangecoder.rs:92-120)
  template <typename P>
  __device__ __forceinline__ uint32_t decode_bit(P* probs, uint32_t idx) {
    uint32_t p = probs[idx];
    const uint32_t bound = (range >> 11) * p;
    uint32_t bit;
    if (code < bound) {
      range = bound;
      p += (0x800u - p) >> 5;
      bit = 0;
    } else {
      range -= bound;
      code -= bound;
      p -= p >> 5;
      bit = 1;
    }
    probs[idx] = uint16_t(p);
    normalize();
    return bit;
  }

  // RangeDecoder::get (rangecoder.rs:71-90): direct bits, MSB first
  __device__ __forceinline__ uint32_t direct_bits(uint32_t count) {
    uint32_t result = 0;
    for (uint32_t i = 0; i < count; i++) {
      range >>= 1;
      const uint32_t bit = code >= range ? 1u : 0u;
      if (bit) code -= range;
      normalize();
      result = (result << 1) | bit;
    }
    return result;
  }

  // RangeDecoder::parse_bit_tree (rangecoder.rs:122-134)
  template <uint32_t NBITS>
  __device__ __forceinline__ uint32_t bit_tree(uint16_t* probs) {
    uint32_t tmp = 1;
#pragma unroll
    for (uint32_t i = 0; i < NBITS; i++) tmp = (tmp << 1) | decode_bit(probs, tmp);
    return tmp - (1u << NBITS);
  }

  // RangeDecoder::parse_reverse_bit_tree (rangecoder.rs:136-151)
  __device__ __forceinline__ uint32_t reverse_bit_tree(uint16_t* probs, uint32_t nbits) {
    uint32_t result = 0, tmp = 1;
    for (uint32_t i = 0; i < nbits; i++) {
      const uint32_t bit = decode_bit(probs, tmp);
      tmp = (tmp << 1) | bit;
      result |= bit << i;
    }
    return result;
  }

  // LenDecoder::decode (rangecoder.rs:256-269)
  __device__ __forceinline__ uint32_t len_decode(uint16_t* ld, uint32_t pos_state) {
    if (!decode_bit(ld, L_CHOICE)) return bit_tree<3>(ld + L_LOW + pos_state * 8);
    if (!decode_bit(ld, L_CHOICE2)) return bit_tree<3>(ld + L_MID + pos_state * 8) + 8;
    return bit_tree<8>(ld + L_HIGH) + 16;
  }

  // ---- output window -------------------------------------------------------------------
  __device__ __forceinline__ uint32_t opos() const { return dict_base + len; }

  // one byte back from the output slice (same-wave stores are visible to later loads)
  __device__ __forceinline__ uint32_t fetch_out(uint32_t pos) {
    uint32_t b = 0;
    if (threadIdx.x == 0) b = out[pos];
    return readfirst(b);
  }

  __device__ __forceinline__ void limit_error() {
    if (lim_is_mem)
      fail(MILZMA_ST_MEMLIMIT, out_lim);  // lzbuffer.rs:210-217
    else
      fail(MILZMA_ST_OUT_FULL);
  }

  // LzBuffer::append_literal
  __device__ __forceinline__ bool append_literal(uint32_t byte) {
    const uint32_t pos = opos();
    if (__builtin_expect(pos >= out_lim, 0)) {
      limit_error();
      return false;
    }
    if (threadIdx.x == 0) out[pos] = uint8_t(byte);
    len++;
    prev = byte;
    prev_valid = true;
    mb_valid = false;
    return true;
  }

  // LzBuffer::append_lz (lzbuffer.rs:123-141, 272-297): byte-serial semantics (overlapping
  // copies replicate the last `dist` bytes) done 64 lanes at a time with a periodic index, so
  // a copy never reads what it wrote itself.  Also picks up the byte that follows the source
  // (the next literal's match byte, lzma.rs:537-538) and the last byte copied.
  __device__ __forceinline__ bool append_lz(uint32_t mlen, uint32_t dist, bool size_known, uint32_t target) {
    if (dist > dict_size) {
      fail(MILZMA_ST_LZ_DIST_DICT, dist, dict_size);
      return false;
    }
    if (dist > len) {
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
    const uint8_t* src = out + (pos - dist);
    uint8_t* dst = out + pos;
    const bool periodic = dist <= n;  // source runs into the destination: index modulo dist
    const float rcp = periodic ? __builtin_amdgcn_rcpf(float(dist)) : 0.0f;
    for (uint32_t i0 = 0; i0 <= n; i0 += kWave) {
      const uint32_t i = i0 + threadIdx.x;
      uint32_t val = 0;
      if (i <= n) {
        const uint32_t j = periodic ? small_mod(i, dist, rcp) : i;
        val = src[j];
        if (i < n) dst[i] = uint8_t(val);
      }
      if (n > 0 && n - 1 >= i0 && n - 1 < i0 + kWave) prev = readlane(val, (n - 1) & 63u);
      if (n >= i0 && n < i0 + kWave) mb = readlane(val, n & 63u);
    }
    if (__builtin_expect(clipped, 0)) {
      // A final match of a known-size stream may overshoot the caller's slice: the reference
      // appends it all and then reports the size mismatch (lzma.rs:513-521).  Anything else
      // that does not fit is a limit condition.
      if (!lim_is_mem && size_known && out_lim - dict_base >= target) {
        len += mlen;
        prev_valid = mb_valid = false;
        return true;
      }
      len += n;
      limit_error();
      return false;
    }
    len += mlen;
    prev_valid = true;
    mb_valid = true;
    return true;
  }

  // ---- DecoderState --------------------------------------------------------------------
  // decode_literal (lzma.rs:526-561)
  __device__ __forceinline__ bool decode_literal(uint32_t* byte_out) {
    if (!prev_valid) {
      prev = len == 0 ? 0 : fetch_out(opos() - 1);
      prev_valid = true;
    }
    const uint32_t lit_state = ((len & ((1u << lp) - 1u)) << lc) + (prev >> (8u - lc));
    auto* probs = lit + lit_state * 0x300u;
    uint32_t result = 1;
    if (state >= 7) {
      const uint32_t dist = rep0 + 1;
      if (dist > dict_size || dist == 0) {  // last_n (lzbuffer.rs:240-255, 96-106)
        fail(MILZMA_ST_MATCH_DIST_DICT, uint64_t(rep0) + 1, dict_size);
        return false;
      }
      if (dist > len) {
        fail(MILZMA_ST_MATCH_DIST_OUT, dist, len);
        return false;
      }
      uint32_t match_byte = mb_valid ? mb : fetch_out(opos() - dist);
      while (result < 0x100) {
        const uint32_t match_bit = (match_byte >> 7) & 1u;
        match_byte <<= 1;
        const uint32_t bit = decode_bit(probs, ((1u + match_bit) << 8) + result);
        result = (result << 1) | bit;
        if (match_bit != bit) break;
      }
    }
    while (result < 0x100) result = (result << 1) | decode_bit(probs, result);
    *byte_out = result - 0x100;
    return true;
  }

  // decode_distance (lzma.rs:563-592)
  __device__ __forceinline__ uint32_t decode_distance(uint32_t length) {
    const uint32_t len_state = length > 3 ? 3 : length;
    const uint32_t pos_slot = bit_tree<6>(model + M_POS_SLOT + len_state * 64);
    if (pos_slot < 4) return pos_slot;
    const uint32_t num_direct_bits = (pos_slot >> 1) - 1;
    uint32_t result = (2u | (pos_slot & 1u)) << num_direct_bits;
    if (pos_slot < 14) {
      result += reverse_bit_tree(model + M_POS_DEC + (result - pos_slot), num_direct_bits);
    } else {
      result += direct_bits(num_direct_bits - 4) << 4;
      result += reverse_bit_tree(model + M_ALIGN, 4);
    }
    return result;
  }

  // process_mode(Finish) (lzma.rs:435-524) for one LZMA payload.
  //   size_known / target : DecoderState.unpacked_size, compared with len (since dict reset)
  //   target64            : the value the reference prints on mismatch
  __device__ __forceinline__ bool process(bool size_known, uint32_t target, bool target_clamped, uint64_t target64) {
    const uint32_t pb_mask = (1u << pb) - 1u;
    for (;;) {
      if (size_known) {
        if (len >= target && !target_clamped) break;
      } else if (code == 0 && reader_eof(rd)) {  // is_finished_ok (rangecoder.rs:49-52)
        break;
      }
      // process_next_inner (lzma.rs:278-393)
      const uint32_t pos_state = len & pb_mask;
      if (!decode_bit(model, M_IS_MATCH + (state << 4) + pos_state)) {
        uint32_t byte;
        if (eof) return fail(MILZMA_ST_INPUT_EOF), false;
        if (!decode_literal(&byte)) return false;
        if (eof) return fail(MILZMA_ST_INPUT_EOF), false;
        if (!append_literal(byte)) return false;
        state = state < 4 ? 0 : (state < 10 ? state - 3 : state - 6);
        continue;
      }
      uint32_t mlen;
      if (decode_bit(model, M_IS_REP + state)) {
        if (!decode_bit(model, M_IS_REP_G0 + state)) {
          if (!decode_bit(model, M_IS_REP_0LONG + (state << 4) + pos_state)) {
            if (eof) return fail(MILZMA_ST_INPUT_EOF), false;
            state = state < 7 ? 9 : 11;
            if (!append_lz(1, rep0 + 1, size_known, target)) return false;
            continue;
          }
        } else {
          uint32_t dist;
          if (!decode_bit(model, M_IS_REP_G1 + state)) {
            dist = rep1;
          } else {
            if (!decode_bit(model, M_IS_REP_G2 + state)) {
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
        mlen = len_decode(model + M_REP_LEN, pos_state);
        state = state < 7 ? 8 : 11;
        if (eof) return fail(MILZMA_ST_INPUT_EOF), false;
      } else {
        rep3 = rep2;
        rep2 = rep1;
        rep1 = rep0;
        mlen = len_decode(model + M_LEN, pos_state);
        state = state < 7 ? 7 : 10;
        rep0 = decode_distance(mlen);
        if (eof) return fail(MILZMA_ST_INPUT_EOF), false;
        if (rep0 == 0xFFFFFFFFu) {  // end-of-stream marker (lzma.rs:372-382)
          if (code == 0 && reader_eof(rd)) break;  // Finished: the known-size check below still applies (lzma.rs:513-521)
          return fail(MILZMA_ST_MARKER_TRAILING), false;
        }
      }
      if (!append_lz(mlen + 2, rep0 + 1, size_known, target)) return false;
    }
    if (size_known && (target_clamped || len != target))  // lzma.rs:513-521
      return fail(MILZMA_ST_SIZE_MISMATCH, target64, uint64_t(len)), false;
    return true;
  }
};

// lit_cap_lclp: largest lc+lp whose literal table fits this launch's LDS (LIT_IN_LDS); for the HBM-spill class the
//               largest lc+lp among the launch's units: it sizes a block's slice of the scratch slab.
// lit_scratch : HBM, (0x300 << lit_cap_lclp) u16 per block, used when !LIT_IN_LDS.
template <bool LIT_IN_LDS>
__global__ __launch_bounds__(64) void decode_generic_kernel(const milzma_unit* __restrict__ units,
                                                            const uint32_t* __restrict__ order, uint32_t n_units,
                                                            const uint8_t* in_base, uint8_t* out_base,
                                                            milzma_result* results, uint32_t lit_cap_lclp,
                                                            uint16_t* lit_scratch) {
  extern __shared__ uint16_t lds[];
  if (blockIdx.x >= n_units) return;
  const uint32_t uidx = order[blockIdx.x];
  const milzma_unit u = units[uidx];
  milzma_result* res = results + uidx;

  GenericDecoder<LIT_IN_LDS> d;
  d.model = lds;
  d.lit = LIT_IN_LDS ? lds + M_SMALL_END : lit_scratch + size_t(blockIdx.x) * (size_t(0x300u) << lit_cap_lclp);
  d.out = out_base + u.out_off;
  d.status = MILZMA_ST_OK;
  d.err_a = d.err_b = 0;
  d.eof = false;
  d.dict_base = 0;
  d.len = 0;
  d.prev = 0;
  d.mb = 0;
  d.prev_valid = true;
  d.mb_valid = false;
  d.range = 0;
  d.code = 0;

  if (u.in_len > MILZMA_MAX_UNIT_BYTES || u.out_cap > MILZMA_MAX_UNIT_BYTES || u.lc > 8 || u.lp > 4 || u.pb > 4 ||
      (u.kind != MILZMA_KIND_RAW_LZMA && u.kind != MILZMA_KIND_LZMA2)) {
    store_result(res, MILZMA_ST_BAD_UNIT, 0, 0, 0, 0, 0, 0);
    return;
  }
  reader_init(d.rd, in_base + u.in_off, uint32_t(u.in_len));
  const uint32_t a0 = d.rd.v;
  const bool raw = u.kind == MILZMA_KIND_RAW_LZMA;
  uint32_t chunks = 0;
  bool ok = true;

  if (raw) {
    // LzmaDecoder::new (lzma.rs:607-613)
    d.lc = u.lc;
    d.lp = u.lp;
    d.pb = u.pb;
    if (LIT_IN_LDS && d.lc + d.lp > lit_cap_lclp) {
      store_result(res, MILZMA_ST_NEED_LCLP, 0, 0, 0, 0, d.lc + d.lp, 0);
      return;
    }
    d.dict_size = u.dict_size;
    // LzCircularBuffer::set (lzbuffer.rs:206-221): the ring only grows up to dict_size, so the
    // limit bites iff memlimit < dict_size, at position == memlimit.
    const uint64_t mem_eff = u.memlimit < uint64_t(u.dict_size) ? u.memlimit : UINT64_MAX;
    d.lim_is_mem = mem_eff <= u.out_cap;
    d.out_lim = uint32_t(d.lim_is_mem ? mem_eff : u.out_cap);
  } else {
    d.lc = d.lp = d.pb = 0;        // Lzma2Decoder::new (lzma2.rs:23-34)
    d.dict_size = 0xFFFFFFFFu;     // LzAccumBuffer: bounded only by bytes since the last reset
    d.lim_is_mem = false;
    d.out_lim = uint32_t(u.out_cap);
  }
  d.reset_model();

  // RAW: one pass = LzmaDecoder::decompress (lzma.rs:635-648).
  // LZMA2: Lzma2Decoder::decompress (lzma2.rs:52-82), the wave walks the packets itself.
  for (bool first = true;; first = false) {
    bool known = true, clamped = false;
    uint64_t target64 = 0;
    uint32_t saved_lim = d.rd.lim;
    if (raw) {
      if (!first) break;
      known = u.unpacked_size != MILZMA_SIZE_UNKNOWN;
      target64 = u.unpacked_size;
      clamped = known && target64 > 0xFFFFFFFFull;
    } else {
      if (reader_eof(d.rd)) {
        d.fail(MILZMA_ST_L2_STATUS_EOF);
        ok = false;
        break;
      }
      const uint32_t status = reader_byte(d.rd);
      if (status == 0) break;
      chunks++;
      if (status == 1 || status == 2) {
        // parse_uncompressed (lzma2.rs:195-229)
        if (d.rd.lim - d.rd.v < 2) {
          d.rd.v = d.rd.lim;
          d.fail(MILZMA_ST_L2_UNPACKED_EOF);
          ok = false;
          break;
        }
        uint32_t n = reader_byte(d.rd) << 8;
        n = (n | reader_byte(d.rd)) + 1;
        if (status == 1) {  // LzAccumBuffer::reset (lzbuffer.rs:72-78)
          d.dict_base += d.len;
          d.len = 0;
        }
        if (d.rd.lim - d.rd.v < n) {
          d.rd.v = d.rd.lim;
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
        const uint8_t* src = d.rd.base + d.rd.v;
        for (uint32_t i = threadIdx.x; i < n; i += kWave) d.out[pos + i] = src[i];
        d.len += n;
        d.prev_valid = d.mb_valid = false;
        reader_seek(d.rd, d.rd.v + n);
        continue;
      }
      // parse_lzma (lzma2.rs:84-193)
      if ((status & 0x80u) == 0) {
        d.fail(MILZMA_ST_L2_INVALID_STATUS, status);
        ok = false;
        break;
      }
      const uint32_t reset = (status >> 5) & 3u;  // 0 none, 1 state, 2 +props, 3 +dict
      if (d.rd.lim - d.rd.v < 2) {
        d.rd.v = d.rd.lim;
        d.fail(MILZMA_ST_L2_UNPACKED_EOF);
        ok = false;
        break;
      }
      uint32_t unpacked = (status & 0x1Fu) << 16;
      unpacked |= reader_byte(d.rd) << 8;
      unpacked = (unpacked | reader_byte(d.rd)) + 1;
      if (d.rd.lim - d.rd.v < 2) {
        d.rd.v = d.rd.lim;
        d.fail(MILZMA_ST_L2_PACKED_EOF);
        ok = false;
        break;
      }
      uint32_t packed = reader_byte(d.rd) << 8;
      packed = (packed | reader_byte(d.rd)) + 1;
      if (reset == 3) {
        d.dict_base += d.len;
        d.len = 0;
        d.prev_valid = d.mb_valid = false;
      }
      if (reset >= 1) {
        if (reset >= 2) {
          if (reader_eof(d.rd)) {
            d.fail(MILZMA_ST_L2_PROPS_EOF);
            ok = false;
            break;
          }
          uint32_t pbv = reader_byte(d.rd);
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
          if (LIT_IN_LDS && lc + lp > lit_cap_lclp) {
            d.fail(MILZMA_ST_NEED_LCLP, lc + lp);
            ok = false;
            break;
          }
          d.lc = lc;
          d.lp = lp;
          d.pb = pbv;
        }
        d.reset_model();  // DecoderState::reset_state (lzma.rs:216-249)
      }
      // input.take(packed): the decoder's EOF for this chunk; leftovers are NOT skipped after
      saved_lim = d.rd.lim;
      if (d.rd.lim - d.rd.v > packed)
        d.rd.lim = d.rd.v + packed;
      else if (d.rd.lim - d.rd.v < packed)
        chunks |= 0x80000000u;  // the window was cut short by the reader's own end
      target64 = uint64_t(unpacked) + d.len;
      clamped = target64 > 0xFFFFFFFFull;
    }
    if (!d.rc_init()) {
      d.rd.lim = saved_lim;
      d.fail(MILZMA_ST_RC_INIT);
      ok = false;
      break;
    }
    ok = d.process(known, uint32_t(target64), clamped, target64);
    d.rd.lim = saved_lim;
    if (!ok) break;
  }
  const uint64_t total = uint64_t(d.dict_base) + d.len;
  uint64_t flushed = total;
  if (!ok) {
    // LzCircularBuffer: whole rings reach the sink as they fill, the tail only at finish().
    // LzAccumBuffer: the sink sees data at dictionary resets and at finish().
    flushed = raw ? (total / d.dict_size) * d.dict_size : d.dict_base;
  }
  store_result(res, d.status, chunks, total, flushed, d.rd.v - a0, d.err_a, d.err_b);
}

}  // namespace milzma
