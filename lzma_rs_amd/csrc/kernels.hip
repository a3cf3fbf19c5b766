// kernels.hip -- HIP translation unit: decode kernels for gfx950 and their launchers.
#include "kernels.h"

#include <algorithm>

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

// Blocks of the fast kernel the current device holds at once (occupancy API x CU count, asked once per instantiation;
// the data-sheet figures -- 256 CUs x 16, or x 9 for the LC4 instantiation -- if the API fails).  Only decides whether the
// priority rotation starts with the launch: a wrong value costs time, never correctness.
uint32_t fast_resident_blocks(bool lc4, uint32_t lds_pad) {
  static uint32_t cached[2] = {0, 0};
  if (lds_pad == 0 && cached[lc4]) return cached[lc4];
  uint32_t r = lc4 ? 256u * 9u : 256u * 16u;
  int dev = 0, per_cu = 0;
  hipDeviceProp_t prop;
  const hipError_t e = lc4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_fast_asm_kernel<16>, int(kWave), lds_pad)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_fast_asm_kernel<8>, int(kWave), lds_pad);
  if (e == hipSuccess && per_cu > 0 && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
      prop.multiProcessorCount > 0)
    r = uint32_t(per_cu) * uint32_t(prop.multiProcessorCount);
  else
    (void)hipGetLastError();
  if (lds_pad == 0) cached[lc4] = r;
  return r;
}

hipError_t launch_fast(const milzma_unit* d_units, const uint32_t* d_order, uint32_t n, const uint8_t* d_in, uint8_t* d_out,
                       milzma_result* d_results, hipStream_t stream, uint32_t lds_pad, bool lc4, uint32_t* d_flag) {
  if (n == 0) return hipSuccess;
  // d_flag: a device word per launch that the launch's last block raises: the waves rotate their priorities (finish
  // together) only once no block is waiting for a slot any more; until then staggered finishes refill slots early
  // (5120 streams: 13.5 vs 11.2 GB/s).  A launch that is a whole number of rounds rotates from the start (8192: 17.2 vs 16.2).
  const uint32_t resident = fast_resident_blocks(lc4, lds_pad);
  if (hipError_t e = hipMemsetAsync(d_flag, n % resident == 0 ? 1 : 0, sizeof(uint32_t), stream); e != hipSuccess) return e;
  // lds_pad: unused dynamic LDS (MILZMA_LDS_PAD, tuning only): what an LDS-resident window of that size would do to occupancy
  if (lc4)
    hipLaunchKernelGGL(decode_fast_asm_kernel<16>, dim3(n), dim3(kWave), lds_pad, stream, d_units, d_order, n, d_in, d_out, d_results,
                       d_flag);
  else
    hipLaunchKernelGGL(decode_fast_asm_kernel<8>, dim3(n), dim3(kWave), lds_pad, stream, d_units, d_order, n, d_in, d_out, d_results,
                       d_flag);
  return hipGetLastError();
}

bool fast8_takes_pb4() {
#ifdef MILZMA_LOOP_NO_PB4
  return false;
#else
  return true;
#endif
}

size_t slice_ctx_bytes(bool lc4) { return size_t(lc4 ? SliceCtx<16>::kDwords : SliceCtx<8>::kDwords) * sizeof(uint32_t); }
size_t slice_queue_bytes(uint32_t cap) { return sizeof(SliceQueue) + size_t(cap) * sizeof(uint32_t); }

hipError_t launch_fast_sliced(const milzma_unit* d_units, const uint32_t* d_order, uint32_t n, const uint8_t* d_in, uint8_t* d_out,
                              milzma_result* d_results, hipStream_t stream, uint32_t lds_pad, bool lc4, uint32_t* d_flag, void* d_queue,
                              uint32_t cap, uint32_t quantum, bool always_park, void* d_ctxmem) {
  if (n == 0) return hipSuccess;
  auto* q = static_cast<SliceQueue*>(d_queue);
  auto* ring = reinterpret_cast<uint32_t*>(q + 1);
  if (hipError_t e = hipMemsetAsync(d_flag, 1, sizeof(uint32_t), stream); e != hipSuccess) return e;  // all waves start together: rotate
  // persistent waves: what the chip holds of THIS kernel (it keeps more registers alive than the ordinary one; never more than that one's)
  static uint32_t cached[2] = {0, 0};
  uint32_t resident = fast_resident_blocks(lc4, lds_pad);
  if (lds_pad == 0 && cached[lc4]) {
    resident = cached[lc4];
  } else {
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    const hipError_t e = lc4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_fast_asm_sliced_kernel<16>, int(kWave), lds_pad)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_fast_asm_sliced_kernel<8>, int(kWave), lds_pad);
    if (e == hipSuccess && per_cu > 0 && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
      resident = std::min(resident, uint32_t(per_cu) * uint32_t(prop.multiProcessorCount));
    else
      (void)hipGetLastError();
    if (lds_pad == 0) cached[lc4] = resident;
  }
  const uint32_t waves = std::min(n, resident);
  hipLaunchKernelGGL(slice_queue_init_kernel, dim3(64), dim3(256), 0, stream, q, ring, d_order, n, cap, quantum, always_park ? 1u : 0u, waves,
                     d_units, d_in, d_out, d_results, d_flag, static_cast<uint32_t*>(d_ctxmem));
  if (lc4)
    hipLaunchKernelGGL(decode_fast_asm_sliced_kernel<16>, dim3(waves), dim3(kWave), lds_pad, stream, q);
  else
    hipLaunchKernelGGL(decode_fast_asm_sliced_kernel<8>, dim3(waves), dim3(kWave), lds_pad, stream, q);
  return hipGetLastError();
}

hipError_t launch_crc_units(const milzma_unit* d_units, uint32_t n, const uint8_t* d_out, const milzma_result* d_results,
                            void* d_parts, hipStream_t stream) {
  static_assert(sizeof(CrcParts) == kCrcPartsBytes, "CrcParts layout");
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(crc_units_kernel, dim3(n), dim3(kWave), 0, stream, d_units, n, d_out, d_results,
                     static_cast<CrcParts*>(d_parts));
  return hipGetLastError();
}

}  // namespace milzma
