// tests/san/kernel_stubs.cpp -- TEST INFRASTRUCTURE.  The sanitized host build (tests/san/Makefile) links lzma_rs_amd/csrc/host.cpp
// without the HIP translation unit: these stand in for the kernel launchers (nothing is ever launched: there is no GPU in that
// test, milzma_create fails, and the entry points under test are the GPU-free ones).
#include "kernels.h"

namespace milzma {
hipError_t launch_generic(LitClass, const milzma_unit*, const uint32_t*, uint32_t, const uint8_t*, uint8_t*, milzma_result*, uint16_t*, uint32_t,
                          hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_fast(const milzma_unit*, const uint32_t*, uint32_t, const uint8_t*, uint8_t*, milzma_result*, hipStream_t, uint32_t, uint32_t*,
                       const uint8_t*, uint32_t) { return hipErrorNoDevice; }
uint32_t fast_resident_blocks(uint32_t) { return 4096; }
size_t slice_ctx_bytes() { return 1; }
size_t slice_queue_bytes(uint32_t cap) { return cap; }
hipError_t launch_fast_sliced(const milzma_unit*, const uint32_t*, uint32_t, const uint8_t*, uint8_t*, milzma_result*, hipStream_t, uint32_t,
                              uint32_t*, void*, uint32_t, uint32_t, bool, void*, bool, uint32_t, uint32_t, uint32_t, uint32_t*, uint8_t*, uint32_t*, const uint64_t*, const uint8_t*, uint32_t) { return hipErrorNoDevice; }
uint32_t stream_lead_bytes(uint32_t in_len) { return in_len; }
hipError_t launch_slab_init(uint8_t*, uint32_t, const uint32_t*, uint32_t, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_move_units(const uint8_t*, uint8_t*, const uint64_t*, uint32_t, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_crc_units(const milzma_unit*, uint32_t, const uint8_t*, const milzma_result*, void*, hipStream_t) { return hipErrorNoDevice; }
}  // namespace milzma
