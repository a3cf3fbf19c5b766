// crc_units.hip.h -- CRC-32 (ISO-HDLC) and CRC-64 (XZ) of the decoded output of every unit, on the GPU.
//
// validate_block_check (src/decode/xz.rs:292-333) digests each block's output with the CRCs named in
// src/xz/crc.rs:1-4.  On the host that is a second pass over every output byte on one core; here it is a
// pass over data that is still in HBM / L2: one wavefront per unit, lane l digests the l-th of 64 equal
// chunks of the unit's output (byte-wise table lookups from LDS, both CRCs at once) and the 64 partial CRCs
// go back to the host, which folds them with the usual GF(2) "shift by the length of what follows"
// operator (host.cpp: crc_fold).  Plain HIP C++: 1 MiB per unit is 16 KiB per lane, ~0.5 ms.
#pragma once
#include "device_common.h"

namespace milzma {

struct CrcParts {     // per unit
  uint32_t c32[64];   // CRC-32 of chunk l (init / xorout all ones); 0 for an empty chunk
  uint64_t c64[64];   // CRC-64/XZ of chunk l
  uint32_t chunk;     // bytes per chunk (a multiple of 16; the last non-empty chunk may be shorter)
  uint32_t valid;     // 0 if the unit's status was not OK
};

__global__ __launch_bounds__(64) void crc_units_kernel(const milzma_unit* __restrict__ units, uint32_t n_units,
                                                       const uint8_t* __restrict__ out_base,
                                                       const milzma_result* __restrict__ results,
                                                       CrcParts* __restrict__ parts) {
  const uint32_t u = blockIdx.x;
  if (u >= n_units) return;
  __shared__ uint32_t t32[256];
  __shared__ uint64_t t64[256];
  for (uint32_t i = threadIdx.x; i < 256; i += kWave) {
    uint32_t c = i;
    uint64_t d = i;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
      d = (d & 1) ? (d >> 1) ^ 0xC96C5795D7870F42ull : d >> 1;
    }
    t32[i] = c;
    t64[i] = d;
  }
  __syncthreads();
  CrcParts* p = parts + u;
  const milzma_result r = results[u];
  if (r.status != MILZMA_ST_OK) {
    if (threadIdx.x == 0) p->valid = 0;
    return;
  }
  const uint64_t len = r.out_len;
  const uint8_t* base = out_base + units[u].out_off;
  uint64_t chunk = ((len + 63) / 64 + 15) & ~uint64_t(15);
  if (chunk == 0) chunk = 16;
  const uint64_t begin = uint64_t(threadIdx.x) * chunk;
  const uint64_t end = begin + chunk < len ? begin + chunk : len;
  uint32_t c32 = 0xFFFFFFFFu;
  uint64_t c64 = ~uint64_t(0);
  uint64_t i = begin;
  if ((reinterpret_cast<uintptr_t>(base) & 15) == 0) {
    for (; i + 16 <= end; i += 16) {
      const uint4 w = *reinterpret_cast<const uint4*>(base + i);
      const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const uint32_t byte = (ws[k] >> (8 * b)) & 0xFFu;
          c32 = t32[(c32 ^ byte) & 0xFFu] ^ (c32 >> 8);
          c64 = t64[(uint32_t(c64) ^ byte) & 0xFFu] ^ (c64 >> 8);
        }
      }
    }
  }
  for (; i < end; i++) {
    const uint32_t byte = base[i];
    c32 = t32[(c32 ^ byte) & 0xFFu] ^ (c32 >> 8);
    c64 = t64[(uint32_t(c64) ^ byte) & 0xFFu] ^ (c64 >> 8);
  }
  p->c32[threadIdx.x] = ~c32;  // an empty chunk yields ~(~0) = 0, the neutral element of the fold
  p->c64[threadIdx.x] = ~c64;
  if (threadIdx.x == 0) {
    p->chunk = uint32_t(chunk);
    p->valid = 1;
  }
}

}  // namespace milzma
