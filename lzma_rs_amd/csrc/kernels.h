// kernels.h -- host-visible launchers of the HIP decode kernels (internal to the library).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "milzma.h"

namespace milzma {

// Launch classes of the generic kernel: how much of the model lives in LDS.
enum LitClass : int {
  kLitLds3 = 0,   // literal table for lc+lp <= 3 in LDS (15 984 B per wave: 10 waves per CU)
  kLitLds4 = 1,   // lc+lp <= 4 in LDS (28 272 B per wave: 5 waves per CU)
  kLitSpill = 2,  // literal table in HBM scratch (lc+lp up to 12), small tables in LDS
  kNumLitClasses = 3
};

// bytes of HBM scratch one block of the spill class needs
constexpr size_t kSpillBytesPerBlock = size_t(0x300u << 12) * sizeof(uint16_t);

// `order[0..n)` lists the unit indices this launch decodes (one 64-thread block each).
hipError_t launch_generic(LitClass cls, const milzma_unit* d_units, const uint32_t* d_order, uint32_t n,
                          const uint8_t* d_in, uint8_t* d_out, milzma_result* d_results, uint16_t* d_scratch,
                          hipStream_t stream);

}  // namespace milzma
