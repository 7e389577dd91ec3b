// Does v_cvt_pk_u8_f32 truncate like the C cast?  Exhaustive over every float in [0, 256):
// the classify pass (csrc/avr_device.h, table_index_pair) relies on
//   cvt_pk_u8_f32(x, 0, 0) == (unsigned)(int)x   for x in [0, 255].
// build: hipcc -O2 --offload-arch=gfx950 -o tools/ubench/cvt_pk_u8 tools/ubench/cvt_pk_u8.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void check(uint32_t first_bits, uint32_t n, unsigned long long* mismatches,
                      uint32_t* first_bad) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const uint32_t bits = first_bits + idx;
  const float x = __uint_as_float(bits);
  const uint32_t want = static_cast<uint32_t>(static_cast<int>(x));
  const uint32_t low = __builtin_amdgcn_cvt_pk_u8_f32(x, 0u, 0u);
  const uint32_t high = __builtin_amdgcn_cvt_pk_u8_f32(x, 3u, 0x00ABCDEFu);
  if (low != want || high != ((want << 24) | 0x00ABCDEFu)) {
    atomicAdd(mismatches, 1ull);
    atomicMin(first_bad, bits);
  }
}

int main() {
  unsigned long long* mismatches;
  uint32_t* first_bad;
  (void)hipMalloc(&mismatches, 8);
  (void)hipMalloc(&first_bad, 4);
  (void)hipMemset(mismatches, 0, 8);
  (void)hipMemset(first_bad, 0xff, 4);
  const uint32_t end_bits = 0x43800000u;  // 256.0f
  const uint32_t chunk = 1u << 28;
  for (uint32_t at = 0; at < end_bits; at += chunk) {
    const uint32_t n = (end_bits - at < chunk) ? (end_bits - at) : chunk;
    hipLaunchKernelGGL(check, dim3((n + 255) / 256), dim3(256), 0, 0, at, n, mismatches, first_bad);
  }
  unsigned long long bad = 0;
  uint32_t bad_bits = 0;
  (void)hipMemcpy(&bad, mismatches, 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(&bad_bits, first_bad, 4, hipMemcpyDeviceToHost);
  std::printf("floats checked: %u, mismatches vs (int) cast: %llu", end_bits, bad);
  if (bad) {
    float f;
    __builtin_memcpy(&f, &bad_bits, 4);
    std::printf(" (first at %.9g, bits 0x%08x)", f, bad_bits);
  }
  std::printf("\n");
  return bad ? 1 : 0;
}
