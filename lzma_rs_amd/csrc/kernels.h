// kernels.h -- host-visible launchers of the HIP decode kernels (internal to the library).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "milzma.h"

namespace milzma {

// Launch classes: which kernel, and for the generic one how much of the model lives in LDS.
enum LitClass : int {
  kFast = 0,      // decode_fast_asm_kernel: lc+lp <= 3 (any pb), model in VGPR lanes, 8 KiB LDS, 16 waves per CU
  kFastSpill = 1, // the same kernel, HBM variant of the loop: lc+lp >= 4 (any lc <= 8, lp <= 4, pb <= 4): the literal rows in a slab in HBM
                  // (1536 B << lc+lp per unit), eight register rows / eight LDS rows as direct-mapped caches over it
  kLitLds3 = 2,   // generic kernel, literal table for lc+lp <= 3 in LDS (15 984 B per wave: 10 waves per CU)
  kLitLds4 = 3,   // generic kernel, lc+lp <= 4 in LDS (28 272 B per wave: 5 waves per CU)
  kLitSpill = 4,  // generic kernel, literal table in an HBM scratch slab sized by the launch's largest lc+lp (up to 12), small tables in LDS
  kNumLitClasses = 5
};

// bytes of HBM scratch one block of the spill class needs when the largest lc + lp of its launch is `lclp` (<= 12)
constexpr size_t spill_bytes_per_block(uint32_t lclp) { return (size_t(0x300u) << lclp) * sizeof(uint16_t); }   // (= 1536 << lclp: also a kFastSpill unit's slab)

// `order[0..n)` lists the unit indices this launch decodes (one 64-thread block each).
// spill_lclp: kLitSpill only -- the largest lc + lp among the launch's units (block b's table is at d_scratch + b * (0x300 << spill_lclp)).
hipError_t launch_generic(LitClass cls, const milzma_unit* d_units, const uint32_t* d_order, uint32_t n,
                          const uint8_t* d_in, uint8_t* d_out, milzma_result* d_results, uint16_t* d_scratch,
                          uint32_t spill_lclp, hipStream_t stream);

// The lane-resident-model kernel (symbol loop in gfx950 asm), 16 waves per CU; d_slab: the literal-row slab of class kFastSpill.
hipError_t launch_fast(const milzma_unit* d_units, const uint32_t* d_order, uint32_t n, const uint8_t* d_in, uint8_t* d_out,
                       milzma_result* d_results, hipStream_t stream, uint32_t lds_pad, uint32_t* d_flag,
                       const uint8_t* d_slab = nullptr, uint32_t slab_bytes = 0);
// d_slab (kFastSpill): slab_bytes per unit of the BATCH (indexed by unit, like the results), every probability 0x400 before the launch

// Time-sliced form of the same kernel for launches that are not a whole number of chip-fulls: as many persistent waves as the chip
// holds take the units from a queue (d_queue: slice_queue_bytes(cap) bytes; cap = n + every yield there can be), decode `quantum`
// bytes of output per turn and -- while other units wait (always_park: in any case, a testing mode) -- park the unit's state in
// d_ctxmem (slice_ctx_bytes() per unit) and take the unit that has waited longest.
uint32_t fast_resident_blocks(uint32_t lds_pad);
size_t slice_ctx_bytes();  // per unit, the same for both instantiations
size_t slice_queue_bytes(uint32_t cap);
hipError_t launch_fast_sliced(const milzma_unit* d_units, const uint32_t* d_order, uint32_t n, const uint8_t* d_in, uint8_t* d_out,
                              milzma_result* d_results, hipStream_t stream, uint32_t lds_pad, uint32_t* d_flag, void* d_queue,
                              uint32_t cap, uint32_t quantum, bool always_park, void* d_ctxmem, bool grow, uint32_t feed, uint32_t span_bytes = 0,
                              uint32_t n_spans = 0, uint32_t* progress = nullptr, uint8_t* host_out = nullptr, uint32_t* in_ready = nullptr,
                              const uint64_t* host_ptrs = nullptr, const uint8_t* d_slab = nullptr, uint32_t slab_bytes = 0);
// host_ptrs (device array, one entry per unit of the batch, or null): the unit's own destination in host memory (0: none)
// in_ready (streamed launches that read their input from host memory): a word in such memory the host sets once every unit's input
// is complete; before that only the first stream_lead_bytes(in_len) bytes of each unit are guaranteed
uint32_t stream_lead_bytes(uint32_t in_len);
// Streamed launches: the waves write their output to host_out (page-locked host memory the device can reach, same offsets as d_out)
// span by span; progress: n_spans counters in such memory, counter s = units whose span s has arrived (see SliceQueue)
// grow: growable output (milzma_decode_units_ex) -- a unit that runs out of room is parked in d_ctxmem with status OUT_FULL /
// err_a = MILZMA_PARKED; d_order entries with bit 31 set resume such a unit (parked by an earlier launch with the same d_ctxmem)
// feed: fed input (MILZMA_DECODE_FEED) -- a unit that comes within 20 bytes of the end of its input view is parked the same way with
// status NEED_INPUT; resumed, its reader moves to the start of the view its descriptor then names

// every probability (u16) of the literal-row slabs of the units d_order[0 .. n) = 0x400: d_slab + unit * slab_bytes, slab_bytes each.  Only
// THOSE units' rows: other units of the batch may be parked with their trained rows in the same slab (bit 31 of an entry is ignored).
hipError_t launch_slab_init(uint8_t* d_slab, uint32_t slab_bytes, const uint32_t* d_order, uint32_t n, hipStream_t stream);

// d_offs: 3 n device words -- src_off[n], dst_off[n], len[n]
hipError_t launch_move_units(const uint8_t* d_src, uint8_t* d_dst, const uint64_t* d_offs, uint32_t n, hipStream_t stream);

// Partial CRCs (64 chunks per unit) of the units' decoded output; see crc_units.hip.h.
struct CrcParts;
constexpr size_t kCrcPartsBytes = 64 * 4 + 64 * 8 + 8;
hipError_t launch_crc_units(const milzma_unit* d_units, uint32_t n, const uint8_t* d_out, const milzma_result* d_results,
                            void* d_parts, hipStream_t stream);

}  // namespace milzma
