// device_common.h -- pieces shared by the decode kernels (gfx950 only).
//
// One wavefront (64 lanes) owns one decode unit.  Everything that describes the stream's
// progress (range, code, positions, LZMA state, reps) is wave-uniform and lives in SGPRs; the
// lanes are used for what is naturally parallel: the 256-byte input window, LZ77 copies, model
// initialisation and (in the fast kernel) the probability model itself.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "milzma.h"

namespace milzma {

constexpr uint32_t kWave = 64;
constexpr uint32_t kTop = 1u << 24;

// Raw buffer resources: per-lane predication without control flow.  A lane whose offset is
// out of range (we use 0xFFFFFFFF) loads 0 / has its store dropped, so "only some lanes touch
// memory" needs neither a divergent branch nor an EXEC dance -- which also keeps hipcc's
// uniformity analysis from declaring the surrounding loops' scalars divergent.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr uint32_t kOob = 0xFFFFFFFFu;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ uint32_t buf_load_u8(rsrc_t r, uint32_t off) {
  return uint32_t(__builtin_amdgcn_raw_buffer_load_b8(r, off, 0, 0));
}
__device__ __forceinline__ uint32_t buf_load_u32(rsrc_t r, uint32_t off) {
  return uint32_t(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ void buf_store_u8(rsrc_t r, uint32_t off, uint32_t v) {
  __builtin_amdgcn_raw_buffer_store_b8(uint8_t(v), r, off, 0, 0);
}

// Wave-uniform read of one lane's register.
__device__ __forceinline__ uint32_t readlane(uint32_t v, uint32_t lane) {
  return __builtin_amdgcn_readlane(v, lane);
}
__device__ __forceinline__ uint32_t readfirst(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// Compressed-input reader: a 2 x 256-byte window of the unit's input held one dword per lane
// (coalesced 256-byte loads), consumed byte-wise through v_readlane.  Positions are "virtual":
// offset from the 256-byte-aligned address just below the unit's first byte, so window loads
// are always aligned and never leave the 256-byte block of a valid byte.
// Replaces the `R: io::BufRead` + read_u8() of src/decode/rangecoder.rs:26-27,64.
struct Reader {
  const uint8_t* base;  // 256-byte aligned
  uint32_t v;           // next byte to hand out
  uint32_t end;         // one past the unit's last byte (the underlying reader's EOF)
  uint32_t lim;         // current EOF as seen by the decoder (end, or the io::Read::take() limit)
  uint32_t w0, w1;      // per-lane dwords of window k = v >> 8 and of window k + 1
};

__device__ __forceinline__ uint32_t reader_load_window(const Reader& r, uint32_t k) {
  // window k holds bytes [256k, 256k + 256); only touch memory if it contains a valid byte
  if ((k << 8) < r.end) return reinterpret_cast<const uint32_t*>(r.base + (size_t(k) << 8))[threadIdx.x];
  return 0;
}

__device__ __forceinline__ void reader_seek(Reader& r, uint32_t v) {
  r.v = v;
  r.w0 = reader_load_window(r, v >> 8);
  r.w1 = reader_load_window(r, (v >> 8) + 1);
}

__device__ __forceinline__ void reader_init(Reader& r, const uint8_t* p, uint32_t len) {
  const uint32_t a0 = uint32_t(reinterpret_cast<uintptr_t>(p) & 255u);
  r.base = p - a0;
  r.end = a0 + len;
  r.lim = r.end;
  reader_seek(r, a0);
}

// Precondition: r.v < r.lim.
__device__ __forceinline__ uint32_t reader_byte(Reader& r) {
  const uint32_t w = readlane(r.w0, (r.v >> 2) & 63u);
  const uint32_t b = (w >> ((r.v & 3u) * 8u)) & 0xffu;
  r.v++;
  if (__builtin_expect((r.v & 255u) == 0, 0)) {
    r.w0 = r.w1;
    r.w1 = reader_load_window(r, (r.v >> 8) + 1);
  }
  return b;
}

__device__ __forceinline__ bool reader_eof(const Reader& r) { return r.v >= r.lim; }

// float-reciprocal modulo, exact for i < 1024 and 1 <= d < 1024 (LZ copies: i <= 273)
__device__ __forceinline__ uint32_t small_mod(uint32_t i, uint32_t d, float rcp_d) {
  const uint32_t q = uint32_t((float(i) + 0.5f) * rcp_d);
  return i - q * d;
}

// Shared tail: publishes the unit's result.  Every lane stores the same values to the same
// addresses: a lane-dependent branch anywhere in the kernel makes hipcc structurise the whole
// control flow graph (every uniform branch turns into a mask-and-Flow-block sequence).
__device__ __forceinline__ void store_result(milzma_result* res, uint32_t status, uint32_t chunks, uint64_t out_len,
                                             uint64_t out_flushed, uint64_t in_consumed, uint64_t a, uint64_t b) {
  res->status = status;
  res->chunks = chunks;
  res->out_len = out_len;
  res->out_flushed = out_flushed;
  res->in_consumed = in_consumed;
  res->err_a = a;
  res->err_b = b;
}

}  // namespace milzma
