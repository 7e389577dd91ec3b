// How much HBM bandwidth a stream kernel keeps when a compute kernel holds most of a CU's wave
// slots, by how the stream keeps its bytes in flight (profiles/r5_corun_gap/README.md):
//   regs<D>   D x 4 x 16 B per lane in flight in VGPRs (D = 1 is the classify pass of rounds 1-4)
//   dma<K>    K KiB per wave in flight through LDS-DMA (global_load_lds_dwordx4: no VGPRs held)
// beside R: `waves_per_cu` x 256 waves of dependent multiply-adds (no memory at all).
// S runs in workgroups of 256 threads, `s_wgs_per_cu` of them per CU at most (an LDS pad caps it).
// build: hipcc --offload-arch=gfx950 -O2 tools/microbench/stream_residency.hip -o tools/microbench/_build/stream_residency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef double d2_t __attribute__((ext_vector_type(2)));
extern __shared__ char dyn_lds[];

// every workgroup streams `per_wg` bytes: DEPTH x 4 loads of 16 B per lane in flight
template <int DEPTH>
__global__ __launch_bounds__(256) void stream_regs(const d2_t* __restrict__ in, size_t per_wg16, float* sink) {
  const d2_t* base = in + static_cast<size_t>(blockIdx.x) * per_wg16;
  float acc = 0.0f;
  for (size_t i = threadIdx.x; i < per_wg16; i += 256 * 4 * DEPTH) {
    d2_t v[4 * DEPTH];
#pragma unroll
    for (int k = 0; k < 4 * DEPTH; ++k) {
      const size_t at = i + static_cast<size_t>(k) * 256;
      v[k] = __builtin_nontemporal_load(base + (at < per_wg16 ? at : threadIdx.x));
    }
#pragma unroll
    for (int k = 0; k < 4 * DEPTH; ++k) acc += static_cast<float>(v[k].x * 255.0) + static_cast<float>(v[k].y * 255.0);
  }
  if (acc == 12345.678f) *sink = acc;
}

// LDS-DMA: each wave keeps PIECES x 1 KiB in flight in a ring of its own in LDS
template <int PIECES>
__global__ __launch_bounds__(256) void stream_dma(const d2_t* __restrict__ in, size_t per_wg16, float* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // wave-uniform ring base; lane l of a piece lands at base + piece * 1024 + l * 16
  __attribute__((address_space(3))) char* ring =
      (__attribute__((address_space(3))) char*)dyn_lds + wave * (PIECES * 1024);
  const d2_t* base = in + static_cast<size_t>(blockIdx.x) * per_wg16 + static_cast<size_t>(wave) * (per_wg16 / 4);
  const size_t n = per_wg16 / 4;  // 16-byte units of this wave
  float acc = 0.0f;
  for (size_t i = 0; i < n; i += 64 * PIECES) {
#pragma unroll
    for (int k = 0; k < PIECES; ++k) {
      const size_t at = i + static_cast<size_t>(k) * 64 + lane;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(base + (at < n ? at : static_cast<size_t>(lane))),
          (__attribute__((address_space(3))) void*)(ring + k * 1024), 16, 0, 2 /* nt */);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < PIECES; ++k) {
      const d2_t v = *reinterpret_cast<__attribute__((address_space(3))) d2_t*>(ring + k * 1024 + lane * 16);
      acc += static_cast<float>(v.x * 255.0) + static_cast<float>(v.y * 255.0);
    }
  }
  if (acc == 12345.678f) *sink = acc;
}

__global__ __launch_bounds__(256) void compute_kernel(int spin, float* sink) {
  float acc = static_cast<float>(threadIdx.x);
  for (int s = 0; s < spin; ++s) acc = acc * 1.0000001f + 0.5f;
  if (acc == 12345.678f) *sink = acc;
}

int main(int argc, char** argv) {
  const size_t bytes = 2944ull << 20;
  const int r_wgs_per_cu = argc > 1 ? std::atoi(argv[1]) : 7;      // R: workgroups (4 waves) per CU
  const int s_wgs_per_cu = argc > 2 ? std::atoi(argv[2]) : 1;      // S: resident workgroups per CU
  d2_t* in; float* sink;
  CK(hipMalloc(&in, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMemset(in, 0, bytes));
  hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  int least, greatest; CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, greatest));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int s_wgs = 256 * s_wgs_per_cu * 8;  // eight rounds of workgroups
  const size_t per_wg16 = bytes / 16 / s_wgs / 1024 * 1024;
  // R long enough to cover S: ~3 ms of multiply-adds
  const int spin = 1 << 20;
  auto run = [&](const char* name, auto launch_s, size_t lds) {
    float alone = 0, beside = 0;
    for (int with_r = 0; with_r < 2; ++with_r) {
      CK(hipDeviceSynchronize());
      if (with_r && r_wgs_per_cu > 0) {
        // R's workgroups pad their LDS so that exactly r_wgs_per_cu fit a CU and nothing else does
        hipLaunchKernelGGL(compute_kernel, dim3(256 * r_wgs_per_cu), dim3(256), 0, b, spin, sink);
        // (give R a head start: it should be resident before S arrives)
        for (volatile int w = 0; w < 2000000; ++w) {}
      }
      CK(hipEventRecord(e0, a));
      launch_s(lds);
      CK(hipEventRecord(e1, a));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(with_r ? &beside : &alone, e0, e1));
      CK(hipDeviceSynchronize());
    }
    const double gb = static_cast<double>(per_wg16) * 16 * s_wgs / 1e9;
    std::printf("%-8s alone %.3f ms (%.2f TB/s)   beside R (%d x 4 waves per CU) %.3f ms (%.2f TB/s)\n", name, alone,
                gb / alone, r_wgs_per_cu, beside, gb / beside);
  };
  // LDS per S workgroup: caps S at s_wgs_per_cu per CU (160 KiB / n, minus a little)
  const size_t cap_lds = (160 * 1024 / s_wgs_per_cu - 1024) & ~size_t{1023};
  std::printf("S: %d workgroups of 256 per CU at most (%zu KiB of LDS each), %.2f GB\n", s_wgs_per_cu, cap_lds >> 10,
              static_cast<double>(per_wg16) * 16 * s_wgs / 1e9);
  run("regs1", [&](size_t lds) { hipLaunchKernelGGL(stream_regs<1>, dim3(s_wgs), dim3(256), lds, a, in, per_wg16, sink); }, cap_lds);
  run("regs2", [&](size_t lds) { hipLaunchKernelGGL(stream_regs<2>, dim3(s_wgs), dim3(256), lds, a, in, per_wg16, sink); }, cap_lds);
  run("regs4", [&](size_t lds) { hipLaunchKernelGGL(stream_regs<4>, dim3(s_wgs), dim3(256), lds, a, in, per_wg16, sink); }, cap_lds);
  run("dma4", [&](size_t lds) { hipLaunchKernelGGL(stream_dma<4>, dim3(s_wgs), dim3(256), lds, a, in, per_wg16, sink); }, cap_lds);
  run("dma8", [&](size_t lds) { hipLaunchKernelGGL(stream_dma<8>, dim3(s_wgs), dim3(256), lds, a, in, per_wg16, sink); }, cap_lds);
  run("dma16", [&](size_t lds) { hipLaunchKernelGGL(stream_dma<16>, dim3(s_wgs), dim3(256), lds, a, in, per_wg16, sink); }, cap_lds);
  if (cap_lds >= 32 * 4 * 1024) {
    run("dma32", [&](size_t lds) { hipLaunchKernelGGL(stream_dma<32>, dim3(s_wgs), dim3(256), lds, a, in, per_wg16, sink); }, cap_lds);
  }
  return 0;
}
