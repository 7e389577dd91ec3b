// CPU only, built with -fsanitize=thread: frame plans made and tightened concurrently on four
// threads -- what avr_renderer_prepare does beside a frame's own planning -- must equal the plans
// made one after the other, without a data race report (the plan code keeps its scratch per
// thread).  Compiles the host sources of the library directly: no HIP.
//   plan_threads_test      prints "ok ...", exit code 0; ThreadSanitizer's report and 66 otherwise
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "../../amrvolumerenderer_amd/csrc/avr_internal.h"
#include "../../amrvolumerenderer_amd/csrc/avr_plan.h"

int main() {
  const int n = 4 * 4 * 4 * 2;
  std::vector<avr_box> boxes(n);
  std::vector<int32_t> owner(n);
  int b = 0;
  for (int level = 0; level < 2; ++level) {
    const double size = level == 0 ? 0.25 : 0.125;
    const double origin = level == 0 ? 0.0 : 0.25;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int k = 0; k < 4; ++k) {
      avr_box& box = boxes[b];
      std::memset(&box, 0, sizeof(box));
      box.min_corner[0] = origin + i * size; box.min_corner[1] = origin + j * size; box.min_corner[2] = origin + k * size;
      box.max_corner[0] = box.min_corner[0] + size; box.max_corner[1] = box.min_corner[1] + size; box.max_corner[2] = box.min_corner[2] + size;
      box.dims[0] = box.dims[1] = box.dims[2] = 8;
      owner[b] = b % 4;
      ++b;
    }
  }
  avr_paint_params params{};
  params.width = 640; params.height = 480; params.scalar_range[1] = 1.0f; params.box_transparency = 0.8f;
  params.reference_sample_distance = 0.01f; params.bounds_max[0] = params.bounds_max[1] = params.bounds_max[2] = 1.0;
  std::vector<int64_t> serial(48), threaded(48);
  auto job = [&](int v, int64_t* out) {
    avr_camera cam{};
    // (every third camera looks from INSIDE the volume: boxes behind the eye, corners that project
    // far outside any integer range)
    const float radius = (v % 3 == 2) ? 0.12f : 2.0f;
    cam.eye[0] = 0.5f + radius * std::cos(0.13f * v); cam.eye[1] = 0.6f; cam.eye[2] = 0.5f + radius * std::sin(0.13f * v);
    cam.look_at[0] = cam.look_at[1] = cam.look_at[2] = (v % 3 == 2) ? 0.9f : 0.5f; cam.up[1] = 1.0f;
    cam.fov_y_degrees = 45.0f; cam.near_plane = 0.1f; cam.far_plane = 100.0f;
    avr_frame_plan plan;
    avr::build_frame_plan(boxes.data(), owner.data(), n, 4, v % 4, nullptr, params, cam, 1, 8, &plan);
    avr::tighten_frame_plan(boxes.data(), n, &plan);
    *out = plan.info.send_floats * 1000003 + plan.info.recv_floats;
  };
  for (int v = 0; v < 48; ++v) job(v, &serial[v]);
  std::vector<std::thread> threads;
  for (int t = 0; t < 4; ++t) threads.emplace_back([&, t] { for (int v = t; v < 48; v += 4) job(v, &threaded[v]); });
  for (auto& t : threads) t.join();
  int bad = 0;
  for (int v = 0; v < 48; ++v) bad += serial[v] != threaded[v];
  std::printf("%s (%d mismatches, first layout %lld)\n", bad ? "MISMATCH" : "ok", bad, (long long)serial[0]);
  return bad != 0;
}
