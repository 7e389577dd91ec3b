// What random line-sized reads cost a saturating HBM stream beside them, by the size of the
// contiguous piece each read fetches (profiles/r5_corun_gap/README.md, section 3).
//   S: streams 2.95 GB of f64 once (16 B per lane, non-temporal) and writes 1/8 of that as a stream
//      -- the classify pass's traffic
//   R: reads `total` bytes of a 1 GB buffer in contiguous pieces of `piece` bytes at pseudo-random
//      piece-aligned places (one wave per piece per trip, 16 B per lane... 4 B per lane for the
//      smallest), with `spin` dependent multiply-adds between two pieces so that it is paced by
//      arithmetic like the march
// prints: S alone, R alone, both side by side on two streams (ms).
// build: hipcc --offload-arch=gfx950 -O2 tools/microbench/hbm_mix.hip -o tools/microbench/_build/hbm_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void stream_kernel(const d2_t* __restrict__ in, size_t n, u2_t* __restrict__ out) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride * 4) {
    d2_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t at = i + static_cast<size_t>(k) * stride;
      v[k] = at < n ? __builtin_nontemporal_load(in + at) : d2_t{0.0, 0.0};
    }
    u2_t packed = {0u, 0u};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      packed.x += static_cast<unsigned>(v[k].x * 255.0);
      packed.y += static_cast<unsigned>(v[k].y * 255.0);
    }
    // 2 bytes out per 16 in
    if ((threadIdx.x & 3) == 0) __builtin_nontemporal_store(packed, out + (i >> 2));
  }
}

__device__ __forceinline__ unsigned hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// one wave fetches `piece` contiguous bytes per trip: piece / 64 bytes per lane
template <int LANE_BYTES>
__global__ __launch_bounds__(256) void random_kernel(const unsigned char* __restrict__ buffer, size_t bytes,
                                                     int trips, int spin, unsigned seed, float* sink,
                                                     unsigned long long* clocks) {
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const unsigned lane = threadIdx.x & 63;
  constexpr size_t kPiece = static_cast<size_t>(LANE_BYTES) * 64;
  const size_t pieces = bytes / kPiece;
  float acc = static_cast<float>(lane);
  for (int t = 0; t < trips; ++t) {
    const size_t piece = hash(wave * 7919u + static_cast<unsigned>(t) * 104729u + seed) % pieces;
    const unsigned char* at = buffer + piece * kPiece + static_cast<size_t>(lane) * LANE_BYTES;
    if (LANE_BYTES == 2) acc += *reinterpret_cast<const unsigned short*>(at);
    if (LANE_BYTES == 4) acc += *reinterpret_cast<const unsigned*>(at);
    if (LANE_BYTES == 8) { const uint2 v = *reinterpret_cast<const uint2*>(at); acc += v.x + v.y; }
    if (LANE_BYTES == 16) { const uint4 v = *reinterpret_cast<const uint4*>(at); acc += v.x + v.y + v.z + v.w; }
    for (int s = 0; s < spin; ++s) acc = acc * 1.0000001f + 0.5f;
  }
  if (acc == 12345.678f) *sink = acc;
  if (lane == 0 && (wave & 63u) == 0) {  // shader clock cycles and 100 MHz ticks this wave lived
    atomicAdd(clocks, clock64() - c0);
    atomicAdd(clocks + 1, wall_clock64() - w0);
  }
}

int main(int argc, char** argv) {
  const size_t stream_bytes = 2950ull << 20;
  const double total_mb = argc > 1 ? std::atof(argv[1]) : 300.0;   // bytes R fetches per launch
  const int spin = argc > 2 ? std::atoi(argv[2]) : 400;
  const int waves = argc > 3 ? std::atoi(argv[3]) : 256 * 7 * 4;    // resident like the march
  // (1 MB: every piece is an L2 hit -- the control for what R's arithmetic alone costs S)
  const size_t random_bytes = (argc > 4 ? static_cast<size_t>(std::atoi(argv[4])) : 1024) << 20;
  d2_t* in; u2_t* out; unsigned char* buffer; float* sink;
  CK(hipMalloc(&in, stream_bytes)); CK(hipMalloc(&out, stream_bytes / 8 + 4096)); CK(hipMalloc(&buffer, random_bytes));
  CK(hipMalloc(&sink, 4));
  unsigned long long* clocks; CK(hipMalloc(&clocks, 16));
  CK(hipMemset(in, 0, stream_bytes)); CK(hipMemset(buffer, 1, random_bytes));
  hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  int least, greatest; CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, greatest));
  hipEvent_t e0, e1, f0, f1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
  const size_t n16 = stream_bytes / 16;
  auto launch_s = [&](hipStream_t s) { hipLaunchKernelGGL(stream_kernel, dim3(256 * 8), dim3(256), 0, s, in, n16, out); };
  unsigned seed = 1;
  auto launch_r = [&](hipStream_t s, int lane_bytes) {
    const size_t piece = static_cast<size_t>(lane_bytes) * 64;
    const int trips = static_cast<int>(total_mb * 1e6 / (static_cast<double>(piece) * waves)) + 1;
    const dim3 grid(waves / 4), block(256);
    ++seed;
    switch (lane_bytes) {
      case 2: hipLaunchKernelGGL(random_kernel<2>, grid, block, 0, s, buffer, random_bytes, trips, spin, seed, sink, clocks); break;
      case 4: hipLaunchKernelGGL(random_kernel<4>, grid, block, 0, s, buffer, random_bytes, trips, spin, seed, sink, clocks); break;
      case 8: hipLaunchKernelGGL(random_kernel<8>, grid, block, 0, s, buffer, random_bytes, trips, spin, seed, sink, clocks); break;
      default: hipLaunchKernelGGL(random_kernel<16>, grid, block, 0, s, buffer, random_bytes, trips, spin, seed, sink, clocks); break;
    }
  };
  auto timed = [&](bool with_s, int lane_bytes, float* ms_s, float* ms_r) {
    const int reps = 10;
    CK(hipDeviceSynchronize());
    if (with_s) CK(hipEventRecord(e0, a));
    if (lane_bytes) CK(hipEventRecord(f0, b));
    for (int r = 0; r < reps; ++r) {
      if (with_s) launch_s(a);
      if (lane_bytes) launch_r(b, lane_bytes);
    }
    if (with_s) CK(hipEventRecord(e1, a));
    if (lane_bytes) CK(hipEventRecord(f1, b));
    CK(hipDeviceSynchronize());
    *ms_s = *ms_r = 0.0f;
    if (with_s) { CK(hipEventElapsedTime(ms_s, e0, e1)); *ms_s /= reps; }
    if (lane_bytes) { CK(hipEventElapsedTime(ms_r, f0, f1)); *ms_r /= reps; }
  };
  float s_alone, r_alone, s_both, r_both, unused;
  timed(true, 0, &s_alone, &unused); timed(true, 0, &s_alone, &unused);
  std::printf("R fetches %.0f MB per launch out of %zu MB, %d waves, %d multiply-adds between pieces; S alone %.3f ms (%.2f TB/s read)\n",
              total_mb, random_bytes >> 20, waves, spin, s_alone, stream_bytes / s_alone / 1e9);
  for (int lane_bytes : {2, 4, 8, 16}) {
    unsigned long long host[2];
    CK(hipMemset(clocks, 0, 16));
    timed(false, lane_bytes, &unused, &r_alone);
    CK(hipMemcpy(host, clocks, 16, hipMemcpyDeviceToHost));
    const double mhz_alone = host[1] ? 100.0 * static_cast<double>(host[0]) / static_cast<double>(host[1]) : 0.0;
    CK(hipMemset(clocks, 0, 16));
    timed(true, lane_bytes, &s_both, &r_both);
    CK(hipMemcpy(host, clocks, 16, hipMemcpyDeviceToHost));
    const double mhz_both = host[1] ? 100.0 * static_cast<double>(host[0]) / static_cast<double>(host[1]) : 0.0;
    std::printf("piece %5d B: R alone %.3f ms (shader clock %.0f MHz); side by side S %.3f ms, R %.3f ms (%.0f MHz)\n",
                lane_bytes * 64, r_alone, mhz_alone, s_both, r_both, mhz_both);
  }
  return 0;
}
