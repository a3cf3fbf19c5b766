// kernels.hip -- HIP translation unit: decode kernels for gfx950 and their launchers.
#include "kernels.h"

#include <algorithm>
#include <mutex>

#include "decode_generic.hip.h"
#include "decode_fast_asm.hip.h"
#include "crc_units.hip.h"

namespace milzma {

hipError_t launch_generic(LitClass cls, const milzma_unit* d_units, const uint32_t* d_order, uint32_t n,
                          const uint8_t* d_in, uint8_t* d_out, milzma_result* d_results, uint16_t* d_scratch,
                          uint32_t spill_lclp, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const dim3 grid(n), block(kWave);
  switch (cls) {
    case kLitLds3: {
      const size_t lds = (M_SMALL_END + (0x300u << 3)) * sizeof(uint16_t);
      hipLaunchKernelGGL(decode_generic_kernel<true>, grid, block, lds, stream, d_units, d_order, n, d_in, d_out,
                         d_results, 3u, nullptr);
      break;
    }
    case kLitLds4: {
      const size_t lds = (M_SMALL_END + (0x300u << 4)) * sizeof(uint16_t);
      hipLaunchKernelGGL(decode_generic_kernel<true>, grid, block, lds, stream, d_units, d_order, n, d_in, d_out,
                         d_results, 4u, nullptr);
      break;
    }
    default: {
      const size_t lds = M_SMALL_END * sizeof(uint16_t);
      hipLaunchKernelGGL(decode_generic_kernel<false>, grid, block, lds, stream, d_units, d_order, n, d_in, d_out,
                         d_results, spill_lclp, d_scratch);
      break;
    }
  }
  return hipGetLastError();
}

// Blocks of a fast kernel the CURRENT device holds at once: occupancy API x CU count, asked once per device and kernel (the cache
// is keyed by the device ordinal and guarded: the lanes' and the multi-device workers' threads come through here concurrently, and
// devices of one node may differ in CU count / partition mode); the data-sheet figure -- 256 CUs x 16 -- if the API fails.  Decides
// when the priority rotation starts and how many persistent waves a time-sliced launch gets: a wrong value costs time, never
// correctness.
namespace {
enum { kKernFast = 0, kKernSliced = 1 };
uint32_t resident_blocks(int kern, uint32_t lds_pad) {
  static std::mutex mu;
  static uint32_t cached[64][2] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    dev = -1;
  }
  const bool cacheable = lds_pad == 0 && dev >= 0 && dev < 64;
  if (cacheable) {
    std::lock_guard<std::mutex> lock(mu);
    if (cached[dev][kern]) return cached[dev][kern];
  }
  uint32_t r = 256u * 16u;
  int per_cu = 0;
  hipDeviceProp_t prop;
  const hipError_t e = kern == kKernFast ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_fast_asm_kernel, int(kWave), lds_pad)
                                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_fast_asm_sliced_kernel, int(kWave), lds_pad);
  if (e == hipSuccess && per_cu > 0 && dev >= 0 && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    r = uint32_t(per_cu) * uint32_t(prop.multiProcessorCount);
  else
    (void)hipGetLastError();
  if (cacheable) {
    std::lock_guard<std::mutex> lock(mu);
    cached[dev][kern] = r;
  }
  return r;
}
}  // namespace

uint32_t fast_resident_blocks(uint32_t lds_pad) { return resident_blocks(kKernFast, lds_pad); }

hipError_t launch_fast(const milzma_unit* d_units, const uint32_t* d_order, uint32_t n, const uint8_t* d_in, uint8_t* d_out,
                       milzma_result* d_results, hipStream_t stream, uint32_t lds_pad, uint32_t* d_flag, const uint8_t* d_slab,
                       uint32_t slab_bytes) {
  if (n == 0) return hipSuccess;
  // d_flag: a device word per launch that the launch's last block raises: the waves rotate their priorities (finish
  // together) only once no block is waiting for a slot any more; until then staggered finishes refill slots early
  // (5120 streams: 13.5 vs 11.2 GB/s).  A launch that is a whole number of rounds rotates from the start (8192: 17.2 vs 16.2).
  const uint32_t resident = fast_resident_blocks(lds_pad);
  if (hipError_t e = hipMemsetAsync(d_flag, n % resident == 0 ? 1 : 0, sizeof(uint32_t), stream); e != hipSuccess) return e;
  // lds_pad: unused dynamic LDS (MILZMA_LDS_PAD, tuning only): what an LDS-resident window of that size would do to occupancy
  hipLaunchKernelGGL(decode_fast_asm_kernel, dim3(n), dim3(kWave), lds_pad, stream, d_units, d_order, n, d_in, d_out, d_results, d_flag, d_slab,
                     slab_bytes);
  return hipGetLastError();
}

// the parked states of a batch are indexed by unit
size_t slice_ctx_bytes() { return size_t(SliceCtx::kDwords) * sizeof(uint32_t); }
size_t slice_queue_bytes(uint32_t cap) { return sizeof(SliceQueue) + size_t(cap) * sizeof(uint32_t); }

hipError_t launch_fast_sliced(const milzma_unit* d_units, const uint32_t* d_order, uint32_t n, const uint8_t* d_in, uint8_t* d_out,
                              milzma_result* d_results, hipStream_t stream, uint32_t lds_pad, uint32_t* d_flag, void* d_queue,
                              uint32_t cap, uint32_t quantum, bool always_park, void* d_ctxmem, bool grow, uint32_t feed, uint32_t span_bytes,
                              uint32_t n_spans, uint32_t* progress, uint8_t* host_out, uint32_t* in_ready, const uint64_t* host_ptrs,
                              const uint8_t* d_slab, uint32_t slab_bytes) {
  if (n == 0) return hipSuccess;
  auto* q = static_cast<SliceQueue*>(d_queue);
  auto* ring = reinterpret_cast<uint32_t*>(q + 1);
  if (hipError_t e = hipMemsetAsync(d_flag, 1, sizeof(uint32_t), stream); e != hipSuccess) return e;  // all waves start together: rotate
  // persistent waves: what the chip holds of THIS kernel (it keeps more registers alive than the ordinary one; never more than that one's)
  const uint32_t resident = std::min(fast_resident_blocks(lds_pad), resident_blocks(kKernSliced, lds_pad));
  const uint32_t waves = std::min(n, resident);
  hipLaunchKernelGGL(slice_queue_init_kernel, dim3(64), dim3(256), 0, stream, q, ring, d_order, n, cap, quantum, always_park ? 1u : 0u, waves,
                     d_units, d_in, d_out, d_results, d_flag, static_cast<uint32_t*>(d_ctxmem), grow ? 1u : 0u, feed,
                     uint32_t(slice_ctx_bytes() / sizeof(uint32_t)), progress ? span_bytes : 0u, n_spans, progress, host_out, progress ? in_ready : nullptr, progress ? host_ptrs : nullptr,
                     d_slab, slab_bytes);
  hipLaunchKernelGGL(decode_fast_asm_sliced_kernel, dim3(waves), dim3(kWave), lds_pad, stream, q);
  return hipGetLastError();
}

// d_dst[dst_off[i], +len[i]) = d_src[src_off[i], +len[i]): one block per range, 16 bytes per lane and step where both ends are
// 16-byte aligned (the library's own slices are 256-byte aligned), bytes otherwise.  HBM-bound: what carries the output of
// parked units into their larger slices (milzma_move_units).
__global__ __launch_bounds__(256) void move_units_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const uint64_t* __restrict__ offs,
                                                         uint32_t n) {
  const uint32_t i = blockIdx.x;
  if (i >= n) return;
  const uint64_t so = offs[i], dof = offs[n + i], len = offs[2 * size_t(n) + i];
  const uint8_t* s = src + so;
  uint8_t* d = dst + dof;
  uint64_t done = 0;
  if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15u) == 0) {
    const uint64_t vecs = len / 16;
    const uint4* s4 = reinterpret_cast<const uint4*>(s);
    uint4* d4 = reinterpret_cast<uint4*>(d);
    for (uint64_t k = threadIdx.x; k < vecs; k += blockDim.x) d4[k] = s4[k];
    done = vecs * 16;
  }
  for (uint64_t k = done + threadIdx.x; k < len; k += blockDim.x) d[k] = s[k];
}

hipError_t launch_move_units(const uint8_t* d_src, uint8_t* d_dst, const uint64_t* d_offs, uint32_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(move_units_kernel, dim3(n), dim3(256), 0, stream, d_src, d_dst, d_offs, n);
  return hipGetLastError();
}

// One block per (unit, 16 KiB piece) of the units' literal-row slabs (1536 B << lc+lp each: a multiple of 16 KiB from lc + lp = 4 on... of 8 KiB
// at 4: the tail piece is short): 16-byte stores of 0x0400 0x0400 ...
__global__ __launch_bounds__(256) void slab_init_kernel(uint8_t* __restrict__ slab, uint32_t slab_bytes, const uint32_t* __restrict__ order, uint32_t n) {
  const uint32_t k = blockIdx.x;
  if (k >= n) return;
  const uint32_t unit = order[k] & 0x7FFFFFFFu;
  const size_t lo = size_t(blockIdx.y) * 16384u;
  if (lo >= slab_bytes) return;
  const size_t hi = lo + 16384u < slab_bytes ? lo + 16384u : size_t(slab_bytes);
  uint4* p = reinterpret_cast<uint4*>(slab + size_t(unit) * slab_bytes + lo);
  const uint4 v = make_uint4(0x04000400u, 0x04000400u, 0x04000400u, 0x04000400u);
  for (size_t i = threadIdx.x; i < (hi - lo) / 16; i += blockDim.x) p[i] = v;
}

hipError_t launch_slab_init(uint8_t* d_slab, uint32_t slab_bytes, const uint32_t* d_order, uint32_t n, hipStream_t stream) {
  if (n == 0 || slab_bytes == 0) return hipSuccess;
  const uint32_t pieces = uint32_t((size_t(slab_bytes) + 16383u) / 16384u);   // (<= 384 at lc + lp = 12)
  for (uint32_t i = 0; i < n; i += 65535u * 32u) {                            // (grid.x limit: 2^31 - 1; kept far below it)
    const uint32_t m = std::min<uint32_t>(n - i, 65535u * 32u);
    hipLaunchKernelGGL(slab_init_kernel, dim3(m, pieces), dim3(256), 0, stream, d_slab, slab_bytes, d_order + i, m);
  }
  return hipGetLastError();
}

uint32_t stream_lead_bytes(uint32_t in_len) { return stream_lead(in_len); }

hipError_t launch_crc_units(const milzma_unit* d_units, uint32_t n, const uint8_t* d_out, const milzma_result* d_results,
                            void* d_parts, hipStream_t stream) {
  static_assert(sizeof(CrcParts) == kCrcPartsBytes, "CrcParts layout");
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(crc_units_kernel, dim3(n), dim3(kWave), 0, stream, d_units, n, d_out, d_results,
                     static_cast<CrcParts*>(d_parts));
  return hipGetLastError();
}

}  // namespace milzma
