// What the packets between two kernels of a stream cost on this GPU (diagnostics for the frame
// driver: the 20-37 us between two marches).  Each case queues `iters` times a kernel that spins
// for a fixed time followed by the packets named, and reports (elapsed / iters - spin).
//   hipcc --offload-arch=gfx950 -O2 -o tools/microbench/_build/packet_gap tools/microbench/packet_gap.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#define OK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
      std::exit(1);                                                                \
    }                                                                              \
  } while (0)

__global__ void spin_kernel(long long ticks, float* sink, int writes) {
  const long long begin = wall_clock64();
  while (wall_clock64() - begin < ticks) {
  }
  // dirty some cache lines, as a real kernel does
  for (int i = 0; i < writes; ++i) {
    sink[(static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) * writes + i] = 1.0f;
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 400;
  const double spin_us = argc > 2 ? std::atof(argv[2]) : 100.0;
  const int grid = argc > 3 ? std::atoi(argv[3]) : 2048;
  OK(hipSetDevice(0));
  int rate_khz = 0;
  OK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
  const long long ticks = static_cast<long long>(spin_us * 1e-3 * rate_khz);
  const int writes = argc > 4 ? std::atoi(argv[4]) : 16;
  float* sink = nullptr;
  OK(hipMalloc(&sink, static_cast<size_t>(grid) * 256 * writes * sizeof(float)));
  int least = 0, greatest = 0;
  OK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  hipStream_t s1, s2, s3;
  OK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, greatest));
  OK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, least));
  OK(hipStreamCreateWithPriority(&s3, hipStreamNonBlocking, greatest));
  const unsigned plain = hipEventDisableTiming;
  const unsigned unfenced = hipEventDisableTiming | hipEventDisableSystemFence;
  std::vector<hipEvent_t> ev(8), evn(8);
  for (auto& e : ev) OK(hipEventCreateWithFlags(&e, plain));
  for (auto& e : evn) OK(hipEventCreateWithFlags(&e, unfenced));
  hipEvent_t done_elsewhere, done_elsewhere_n;
  OK(hipEventCreateWithFlags(&done_elsewhere, plain));
  OK(hipEventCreateWithFlags(&done_elsewhere_n, unfenced));
  OK(hipEventRecord(done_elsewhere, s3));
  OK(hipEventRecord(done_elsewhere_n, s3));
  OK(hipDeviceSynchronize());
  int* flag = nullptr;
  OK(hipMalloc(&flag, 64));
  OK(hipMemset(flag, 0, 64));

  auto kernel = [&](hipStream_t s) {
    hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, s, ticks, sink, writes);
  };
  auto run = [&](const char* name, int kernels_per_iter, const std::function<void(int)>& body) {
    for (int i = 0; i < 20; ++i) body(i);
    OK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) body(i);
    OK(hipDeviceSynchronize());
    const double us =
        std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%-68s %7.2f us per kernel over the spin\n", name,
                us / iters / kernels_per_iter - spin_us);
    std::fflush(stdout);
  };

  run("kernel only", 1, [&](int) { kernel(s1); });
  run("kernel + record", 1, [&](int i) {
    kernel(s1);
    OK(hipEventRecord(ev[i % 8], s1));
  });
  run("kernel + record (no system fence)", 1, [&](int i) {
    kernel(s1);
    OK(hipEventRecord(evn[i % 8], s1));
  });
  run("kernel + 2 records", 1, [&](int i) {
    kernel(s1);
    OK(hipEventRecord(ev[i % 4], s1));
    OK(hipEventRecord(ev[4 + i % 4], s1));
  });
  run("kernel + wait on an event another stream finished long ago", 1, [&](int) {
    kernel(s1);
    OK(hipStreamWaitEvent(s1, done_elsewhere, 0));
  });
  run("kernel + record + wait(done) + record  [the classify stream's pattern]", 1, [&](int i) {
    kernel(s1);
    OK(hipEventRecord(ev[i % 4], s1));
    OK(hipStreamWaitEvent(s1, done_elsewhere, 0));
    OK(hipEventRecord(ev[4 + i % 4], s1));
  });
  run("same, events without system fence", 1, [&](int i) {
    kernel(s1);
    OK(hipEventRecord(evn[i % 4], s1));
    OK(hipStreamWaitEvent(s1, done_elsewhere_n, 0));
    OK(hipEventRecord(evn[4 + i % 4], s1));
  });
  run("kernel + hipStreamWriteValue32", 1, [&](int i) {
    kernel(s1);
    OK(hipStreamWriteValue32(s1, flag, i, 0));
  });
  run("ping-pong between two streams (record, wait): per kernel", 2, [&](int i) {
    kernel(s1);
    OK(hipEventRecord(ev[i % 4], s1));
    OK(hipStreamWaitEvent(s2, ev[i % 4], 0));
    kernel(s2);
    OK(hipEventRecord(ev[4 + i % 4], s2));
    OK(hipStreamWaitEvent(s1, ev[4 + i % 4], 0));
  });
  run("ping-pong, events without system fence", 2, [&](int i) {
    kernel(s1);
    OK(hipEventRecord(evn[i % 4], s1));
    OK(hipStreamWaitEvent(s2, evn[i % 4], 0));
    kernel(s2);
    OK(hipEventRecord(evn[4 + i % 4], s2));
    OK(hipStreamWaitEvent(s1, evn[4 + i % 4], 0));
  });
  // a graph of the same chain: kernel -> kernel -> ... (8 per launch)
  {
    hipGraph_t graph;
    hipGraphExec_t exec;
    OK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < 8; ++k) kernel(s1);
    OK(hipStreamEndCapture(s1, &graph));
    OK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    run("graph of 8 chained kernels: per kernel", 8, [&](int) { OK(hipGraphLaunch(exec, s1)); });
    // two branches per stage (fork / join inside the graph), as classify || march
    hipGraph_t forked;
    hipGraphExec_t forked_exec;
    OK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < 4; ++k) {
      OK(hipEventRecord(ev[0], s1));
      OK(hipStreamWaitEvent(s2, ev[0], 0));
      kernel(s1);
      kernel(s2);
      OK(hipEventRecord(ev[1], s2));
      OK(hipStreamWaitEvent(s1, ev[1], 0));
    }
    OK(hipStreamEndCapture(s1, &forked));
    OK(hipGraphInstantiate(&forked_exec, forked, nullptr, nullptr, 0));
    run("graph of 4 stages of two parallel kernels: per stage", 4,
        [&](int) { OK(hipGraphLaunch(forked_exec, s1)); });
  }
  return 0;
}
