// pmc_calib.hip -- known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on the access
// shapes the LZMA decode kernel uses (MI355X_MICROARCH.md: "other access widths and WRITE_SIZE are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Built and run on the GPU box only:
//   hipcc --offload-arch=gfx950 -O2 experiments/pmc_calib.hip -o /tmp/pmc_calib && rocprofv3 --pmc WRITE_SIZE ... -- /tmp/pmc_calib
// Kernels (4096 waves of 64 lanes, one wave per block, each wave owns a 1 MiB region of a 4 GiB buffer):
//   calib_store_lane0   each wave: 131072 single-byte stores from lane 0 to consecutive addresses   (a literal run)
//   calib_store_64      each wave: 16384 stores of 64 consecutive bytes, one per lane, start skewed   (a match's store)
//   calib_load_64       each wave: 16384 loads of 64 consecutive bytes at pseudo-random offsets        (a match's load)
//   calib_copy_x4       streaming copy, 16 B per lane                                                  (the guide's reference shape)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr uint32_t kWaves = 4096, kRegion = 1u << 20;

__global__ void calib_store_lane0(uint8_t* buf, uint32_t n) {
  uint8_t* p = buf + size_t(blockIdx.x) * kRegion;
  if (threadIdx.x == 0)
    for (uint32_t i = 0; i < n; i++) p[i] = uint8_t(i);
}
__global__ void calib_store_64(uint8_t* buf, uint32_t n) {
  uint8_t* p = buf + size_t(blockIdx.x) * kRegion + 7;
  for (uint32_t i = 0; i + 1 < n; i++) p[size_t(i) * 64 + threadIdx.x] = uint8_t(i + threadIdx.x);
}
__global__ void calib_load_64(const uint8_t* buf, uint32_t n, uint32_t* sink) {
  const uint8_t* p = buf + size_t(blockIdx.x) * kRegion;
  uint32_t acc = 0, x = blockIdx.x * 2654435761u + 12345u;
  for (uint32_t i = 0; i < n; i++) {
    x = x * 1664525u + 1013904223u;
    const uint32_t off = (x >> 8) % (kRegion - 128);
    acc += p[off + threadIdx.x];
  }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}
__global__ void calib_copy_x4(const uint4* src, uint4* dst, size_t n16) {
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += size_t(gridDim.x) * blockDim.x) dst[i] = src[i];
}

int main() {
  uint8_t *a, *b;
  uint32_t* sink;
  const size_t bytes = size_t(kWaves) * kRegion;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
  hipMemset(a, 1, bytes);
  hipMemset(b, 2, bytes);
  hipDeviceSynchronize();
  calib_store_lane0<<<kWaves, 64>>>(a, 131072);
  hipDeviceSynchronize();
  calib_store_64<<<kWaves, 64>>>(b, 16384);
  hipDeviceSynchronize();
  calib_load_64<<<kWaves, 64>>>(b, 16384, sink);
  hipDeviceSynchronize();
  calib_copy_x4<<<8192, 256>>>(reinterpret_cast<const uint4*>(a), reinterpret_cast<uint4*>(b), bytes / 16);
  hipDeviceSynchronize();
  printf("known bytes: store_lane0 %zu, store_64 %zu, load_64 %zu (requested), copy_x4 %zu read + %zu written\n",
         size_t(kWaves) * 131072, size_t(kWaves) * 16383 * 64, size_t(kWaves) * 16384 * 64, bytes, bytes);
  return 0;
}
