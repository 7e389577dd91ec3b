// Compile-and-run check of include/avr_reference_api.hpp against stand-ins that carry exactly
// the member names the reference's types have (volume::AmrBox, amrex::Array4, amrex::Box,
// volume::CameraParameters, volume::ScalarTransform, ImageRGBAFloatColorDepthSort,
// LayeredVolumeImage).  Driven by tests/test_cxx_adapter.py, which compares the outputs with
// the oracle.
//   adapter_test paint   cells.bin nx ny nz ghost W H out.bin
//   adapter_test compose layers.bin n_layers n_pixels hints.bin out.bin
//   adapter_test frame   scene.bin n_ranks W H transparency antialiasing frames out_image.bin out_rgb8.bin
//     scene.bin: int32 n_boxes, then per box 6 doubles (corners), 3 int32 dims, int32 owner,
//     nx*ny*nz doubles.  All ranks of the frame are played in this one process, one host thread
//     per rank as MPI would run one process per rank: every rank owns an avr::FrameDriver (the
//     pipelined C++ frame driver, avr_renderer) and the ranks are connected by the in-process
//     rehearsal communicator; `frames` frames are rendered back to back without synchronising.
//   adapter_test compose_ranks layers.bin n_layers n_ranks W H hints.bin owners.bin group.bin out.bin
//     Compositor::compose of a LayeredVolumeImage per rank (avr::HipDirectSend), threads as ranks.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <future>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/avr_reference_api.hpp"
#include "../../include/avr_hip_debug.h"  // the search histories the coordinated-search check compares

namespace standin {

struct RealVect {  // amrex::RealVect
  double v[3];
  double operator[](int i) const { return v[i]; }
};
struct IntVect {  // amrex::IntVect
  int v[3];
  int operator[](int i) const { return v[i]; }
};
struct Box {  // amrex::Box
  IntVect lo;
  IntVect smallEnd() const { return lo; }
};
struct Array4 {  // amrex::Array4<Real const>: x fastest, begin = first index of the fab
  const double* p;
  long jstride, kstride, nstride;
  IntVect begin;
  const double* ptr(int i, int j, int k, int n) const {
    return p + (i - begin[0]) + (j - begin[1]) * jstride + (k - begin[2]) * kstride + n * nstride;
  }
};
struct AmrBox {  // volume::AmrBox
  RealVect minCorner, maxCorner;
  IntVect cellDimensions;
  Box validBox;
  Array4 values;
  int component = 0;
};
struct VolumeBounds {
  RealVect minCorner, maxCorner;
};
struct ScalarTransform {  // volume::ScalarTransform
  bool logScaleInput = false;
  bool normalizeToUnitRange = false;
  double positiveFloor = 0.0, processedMin = 0.0, processedMax = 1.0, inverseProcessedSpan = 1.0,
         normalizationMin = 0.0, normalizationMax = 1.0, inverseNormalizationSpan = 1.0;
};
struct CameraParameters {  // volume::CameraParameters
  RealVect eye, lookAt, up;
  float fovYDegrees, nearPlane, farPlane;
};
struct ColorMapControlPoint {
  float value, red, green, blue, alpha;
};
using ColorMap = std::vector<ColorMapControlPoint>;

// Common/Image.hpp: the polymorphic base every image derives from (width, height, pixel region).
struct Image {
  int width, height, regionBegin, regionEnd;
  Image(int w, int h, int begin, int end) : width(w), height(h), regionBegin(begin), regionEnd(end) {}
  virtual ~Image() = default;
  int getWidth() const { return width; }
  int getHeight() const { return height; }
  int getRegionBegin() const { return regionBegin; }
  int getRegionEnd() const { return regionEnd; }
  int getNumberOfPixels() const { return regionEnd - regionBegin; }
  // createNew(regionBegin, regionEnd): same type, width and height, new region
  virtual std::unique_ptr<Image> createNew(int begin, int end) const = 0;
};
// Common/ImageColorOnly.hpp: ColorType / ColorVecSize / getColorBuffer() as the reference has them
template <typename Features>
struct ColorOnlyImage : Image {
  using ColorType = typename Features::ColorType;
  static constexpr int ColorVecSize = Features::ColorVecSize;
  std::vector<ColorType> buffer;
  ColorOnlyImage(int w, int h) : ColorOnlyImage(w, h, 0, w * h) {}
  ColorOnlyImage(int w, int h, int begin, int end)
      : Image(w, h, begin, end), buffer(static_cast<size_t>(end - begin) * ColorVecSize) {}
  ColorType* getColorBuffer() { return buffer.data(); }
  const ColorType* getColorBuffer() const { return buffer.data(); }
  std::unique_ptr<Image> createNew(int begin, int end) const override {
    return std::make_unique<ColorOnlyImage<Features>>(width, height, begin, end);
  }
};
struct DepthSortFeatures {  // ImageRGBAFloatColorDepthSortFeatures
  using ColorType = float;
  static constexpr int ColorVecSize = 5;
};
struct FloatFeatures {  // ImageRGBAFloatColorOnlyFeatures
  using ColorType = float;
  static constexpr int ColorVecSize = 4;
};
struct UByteFeatures {  // ImageRGBAUByteColorOnlyFeatures
  using ColorType = unsigned int;
  static constexpr int ColorVecSize = 1;
};
using DepthSortImage = ColorOnlyImage<DepthSortFeatures>;  // ImageRGBAFloatColorDepthSort
using FloatImage = ColorOnlyImage<FloatFeatures>;          // ImageRGBAFloatColorOnly
using UByteImage = ColorOnlyImage<UByteFeatures>;          // ImageRGBAUByteColorOnly
// LayeredVolumeImage: an Image that is also a LayeredImageInterface (getLayer hands out `Image*`)
struct Layered : Image {
  std::vector<std::unique_ptr<DepthSortImage>> layers;
  std::vector<float> hints;
  Layered(int w, int h) : Image(w, h, 0, w * h) {}
  int getLayerCount() const { return static_cast<int>(layers.size()); }
  Image* getLayer(int i) { return layers[static_cast<size_t>(i)].get(); }
  float getLayerDepthHint(int i) const { return hints[static_cast<size_t>(i)]; }
  std::unique_ptr<Image> createEmptyLayer(int begin, int end) const {
    return std::make_unique<DepthSortImage>(width, height, begin, end);
  }
  std::unique_ptr<Image> createNew(int, int) const override {
    throw std::runtime_error("LayeredVolumeImage does not support createNew");
  }
};

// The control plane of the ranks-as-threads rehearsal: what MPI_Comm_rank / MPI_Bcast /
// MPI_Allgather(v) / MPI_Group_translate_ranks are to the reference's host code.
struct ThreadWorld {
  int n = 1;
  std::mutex mutex;
  std::condition_variable cv;
  int waiting = 0;
  unsigned long generation = 0;
  std::vector<char> bytes;
  std::vector<int> ints;
  std::vector<float> floats;
  void barrier() {
    std::unique_lock<std::mutex> lock(mutex);
    const unsigned long mine = generation;
    if (++waiting == n) {
      waiting = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lock, [&] { return generation != mine; });
    }
  }
};
struct ThreadControl {
  ThreadWorld* world;
  int my_rank;
  int rank() const { return my_rank; }
  int size() const { return world->n; }
  void broadcast(void* data, int n_bytes, int root) {
    if (my_rank == root) world->bytes.assign(static_cast<char*>(data), static_cast<char*>(data) + n_bytes);
    world->barrier();
    std::memcpy(data, world->bytes.data(), static_cast<size_t>(n_bytes));
    world->barrier();
  }
  void allgather_int(int value, int* out) {
    {
      std::lock_guard<std::mutex> lock(world->mutex);
      world->ints.resize(static_cast<size_t>(world->n));
      world->ints[static_cast<size_t>(my_rank)] = value;
    }
    world->barrier();
    std::copy(world->ints.begin(), world->ints.end(), out);
    world->barrier();
  }
  void allgatherv_float(const float* in, int count, float* out, const int* counts, const int* displs) {
    int total = 0;
    for (int r = 0; r < world->n; ++r) total += counts[r];
    {
      std::lock_guard<std::mutex> lock(world->mutex);
      world->floats.resize(static_cast<size_t>(total > 0 ? total : 1));
      std::copy(in, in + count, world->floats.begin() + displs[my_rank]);
    }
    world->barrier();
    std::copy(world->floats.begin(), world->floats.begin() + total, out);
    world->barrier();
  }
  void allgather_bytes(const void* mine, void* all, int n_bytes) {
    {
      std::lock_guard<std::mutex> lock(world->mutex);
      world->bytes.resize(static_cast<size_t>(world->n) * static_cast<size_t>(n_bytes));
      std::memcpy(world->bytes.data() + static_cast<size_t>(my_rank) * static_cast<size_t>(n_bytes), mine,
                  static_cast<size_t>(n_bytes));
    }
    world->barrier();
    std::memcpy(all, world->bytes.data(), static_cast<size_t>(world->n) * static_cast<size_t>(n_bytes));
    world->barrier();
  }
  std::vector<int> group_ranks(const std::vector<int>& group) { return group; }
};

}  // namespace standin

template <typename T>
static std::vector<T> read_file(const char* path, size_t count) {
  std::vector<T> data(count);
  FILE* f = std::fopen(path, "rb");
  if (!f || std::fread(data.data(), sizeof(T), count, f) != count) {
    std::fprintf(stderr, "cannot read %s\n", path);
    std::exit(2);
  }
  std::fclose(f);
  return data;
}

static void write_file(const char* path, const float* data, size_t count) {
  FILE* f = std::fopen(path, "wb");
  if (!f || std::fwrite(data, sizeof(float), count, f) != count) {
    std::fprintf(stderr, "cannot write %s\n", path);
    std::exit(2);
  }
  std::fclose(f);
}

int main(int argc, char** argv) {
  try {
    const std::string mode = argc > 1 ? argv[1] : "";
    avr::Context context(0);
    if (mode == "paint" && argc == 10) {
      const int nx = std::atoi(argv[3]), ny = std::atoi(argv[4]), nz = std::atoi(argv[5]);
      const int ghost = std::atoi(argv[6]);  // the fab is larger than the valid box by `ghost`
      const int W = std::atoi(argv[7]), H = std::atoi(argv[8]);
      const int fx = nx + 2 * ghost, fy = ny + 2 * ghost, fz = nz + 2 * ghost;
      const std::vector<double> fab = read_file<double>(argv[2], static_cast<size_t>(fx) * fy * fz);
      standin::AmrBox box;
      box.minCorner = {{0.1, 0.2, -0.3}};
      box.maxCorner = {{0.8, 0.65, 0.8}};
      box.cellDimensions = {{nx, ny, nz}};
      box.validBox.lo = {{7, -3, 11}};  // arbitrary index origin of the valid region
      box.values = {fab.data(), fx, static_cast<long>(fx) * fy, static_cast<long>(fx) * fy * fz,
                    {{7 - ghost, -3 - ghost, 11 - ghost}}};
      standin::VolumeBounds bounds{{{-0.05, -0.05, -0.35}}, {{1.05, 1.05, 1.05}}};
      standin::ScalarTransform transform;
      transform.normalizeToUnitRange = true;
      standin::CameraParameters camera{{{2.2, 1.6, 2.9}}, {{0.5, 0.5, 0.5}}, {{0.0, 1.0, 0.0}},
                                       45.0f, 0.1f, 20.0f};
      const standin::ColorMap map = {{0.0f, 0.0f, 0.0f, 0.2f, 0.0f},
                                     {0.4f, 0.9f, 0.8f, 0.1f, 0.3f},
                                     {1.0f, 1.0f, 1.0f, 1.0f, 0.9f}};
      standin::DepthSortImage image(W, H);
      avr::VolumePainter painter(context);
      painter.paint(box, bounds, transform, std::make_pair(0.0f, 1.0f), 0, 1, 0.4f, 1, 0.01f, image,
                    camera, &map);
      write_file(argv[9], image.getColorBuffer(), image.buffer.size());
      return 0;
    }
    if (mode == "compose" && argc == 7) {
      const int n_layers = std::atoi(argv[3]);
      const int n_pixels = std::atoi(argv[4]);
      const std::vector<float> all =
          read_file<float>(argv[2], static_cast<size_t>(n_layers) * n_pixels * 5);
      standin::Layered layered(n_pixels, 1);
      layered.hints = read_file<float>(argv[5], static_cast<size_t>(n_layers));
      for (int l = 0; l < n_layers; ++l) {
        auto img = std::make_unique<standin::DepthSortImage>(n_pixels, 1);
        std::memcpy(img->getColorBuffer(), all.data() + static_cast<size_t>(l) * n_pixels * 5,
                    sizeof(float) * static_cast<size_t>(n_pixels) * 5);
        layered.layers.push_back(std::move(img));
      }
      const std::vector<float> out =
          avr::compose_single_rank<standin::DepthSortImage>(context, layered, n_pixels);
      write_file(argv[6], out.data(), out.size());
      return 0;
    }
    if (mode == "frame" && argc >= 11 && argc <= 13) {
      // (a twelfth argument "bytes": no float image is asked for -- the frames' RGB8 pieces then
      // travel to rank 0 with the next frame's grouped round, the last frame's with synchronize();
      // a thirteenth "rccl" / "rccl_inband": the ranks' communicators are the RCCL flavour, made
      // from a unique id over the threads' control plane as the reference's host would make them
      // over MPI -- with AVR_RCCL_LIBRARY naming tests/cxx/libmock_rccl.so, since RCCL itself
      // refuses two ranks on one device; "rccl_inband" drops the caller's control plane, so that
      // the plan agreement and the co-run windows travel as tiny grouped rounds in band)
      const bool want_image = !(argc >= 12 && std::string(argv[11]) == "bytes");
      const std::string flavour = argc == 13 ? argv[12] : "local";
      const int n_ranks = std::atoi(argv[3]);
      const int W = std::atoi(argv[4]), H = std::atoi(argv[5]);
      const float transparency = static_cast<float>(std::atof(argv[6]));
      const int antialiasing = std::atoi(argv[7]);
      const int frames = std::atoi(argv[8]);
      FILE* f = std::fopen(argv[2], "rb");
      if (!f) throw std::runtime_error("cannot read scene");
      int32_t n_boxes = 0;
      if (std::fread(&n_boxes, 4, 1, f) != 1) throw std::runtime_error("short scene file");
      std::vector<avr_box> boxes(static_cast<size_t>(n_boxes));
      std::vector<int32_t> owner(static_cast<size_t>(n_boxes));
      std::vector<std::unique_ptr<avr::DeviceBuffer<double>>> cells;
      for (int b = 0; b < n_boxes; ++b) {
        avr_box& box = boxes[static_cast<size_t>(b)];
        std::memset(&box, 0, sizeof(box));
        double corners[6];
        int32_t meta[4];
        if (std::fread(corners, 8, 6, f) != 6 || std::fread(meta, 4, 4, f) != 4) {
          throw std::runtime_error("short scene file");
        }
        for (int c = 0; c < 3; ++c) {
          box.min_corner[c] = corners[c];
          box.max_corner[c] = corners[3 + c];
          box.dims[c] = meta[c];
        }
        owner[static_cast<size_t>(b)] = meta[3];
        const size_t n = static_cast<size_t>(meta[0]) * meta[1] * meta[2];
        std::vector<double> host(n);
        if (std::fread(host.data(), 8, n, f) != n) throw std::runtime_error("short scene file");
        auto dev = std::make_unique<avr::DeviceBuffer<double>>(n);
        dev->upload(host.data(), n);
        box.cells = dev->data();
        box.jstride = meta[0];
        box.kstride = static_cast<int64_t>(meta[0]) * meta[1];
        cells.push_back(std::move(dev));
      }
      std::fclose(f);
      avr_scalar_transform transform;
      std::memset(&transform, 0, sizeof(transform));
      transform.normalize_to_unit_range = 1;
      transform.inverse_normalization_span = 1.0;
      const double bmin[3] = {-0.05, -0.05, -0.05}, bmax[3] = {1.05, 1.05, 1.05};
      avr_camera camera;
      std::memset(&camera, 0, sizeof(camera));
      const double eye[3] = {2.2, 1.6, 2.9}, look[3] = {0.5, 0.5, 0.5}, up[3] = {0.0, 1.0, 0.0};
      for (int c = 0; c < 3; ++c) {
        camera.eye[c] = eye[c];
        camera.look_at[c] = look[c];
        camera.up[c] = up[c];
      }
      camera.fov_y_degrees = 45.0f;
      camera.near_plane = 0.1f;
      camera.far_plane = 20.0f;
      avr_render_params render;
      std::memset(&render, 0, sizeof(render));
      render.width = W;
      render.height = H;
      render.box_transparency = transparency;
      render.antialiasing = antialiasing;
      render.use_visibility_graph = 1;
      render.draw_bounds = 1;

      // one host thread per rank, connected by the in-process rehearsal communicator
      std::vector<std::unique_ptr<avr::Communicator>> comms;
      if (flavour == "local") comms = avr::Communicator::local(n_ranks);
      standin::ThreadWorld world;
      world.n = n_ranks;
      const size_t n_pixels = static_cast<size_t>(W) * H;
      avr::DeviceBuffer<float> image(n_pixels * 5 + 1);
      avr::DeviceBuffer<unsigned char> bytes(n_pixels * 3 + 1);
      std::vector<std::string> errors(static_cast<size_t>(n_ranks));
      // the co-run candidate every rank's driver held in every frame: ranks of several search as
      // one system, so the rows must be equal (avr_renderer_set_corun_history)
      std::vector<std::vector<int16_t>> held(static_cast<size_t>(n_ranks));
      std::vector<std::thread> threads;
      for (int r = 0; r < n_ranks; ++r) {
        threads.emplace_back([&, r] {
          try {
            avr::hip_ok(hipSetDevice(0), "hipSetDevice");
            standin::ThreadControl control{&world, r};
            std::unique_ptr<avr::Communicator> own;
            avr::Communicator* comm = nullptr;
            if (n_ranks > 1 && flavour == "local") {
              comm = comms[static_cast<size_t>(r)].get();
            } else if (n_ranks > 1) {
              own = std::make_unique<avr::Communicator>(control, 0);  // collective: unique id, broadcast, init
              if (flavour == "rccl_inband") avr::check(avr_comm_set_control(own->get(), nullptr, nullptr));
              comm = own.get();
            }
            avr::FrameDriver driver(0, r, n_ranks, comm, boxes, owner, transform, bmin, bmax);
            avr::check(avr_renderer_set_corun_history(driver.get(), frames));
            // pipelined: no synchronisation between the frames.  Every third frame before the
            // last looks from elsewhere: the camera then comes back to a cached plan (whose
            // exchange layout the driver tightens on that second use).
            avr_camera elsewhere = camera;
            elsewhere.eye[0] = camera.eye[0] - 0.9f;
            elsewhere.eye[1] = camera.eye[1] + 0.4f;
            auto away = [&](int frame) { return (frame % 3 == 1) && frame + 1 < frames; };
            for (int frame = 0; frame < frames; ++frame) {
              // the next frame's plan is made on another thread while this frame is queued
              // (avr_renderer_prepare): it must be the plan render() would have made itself
              std::future<void> planned;
              if (frame + 1 < frames) {
                const avr_camera next = away(frame + 1) ? elsewhere : camera;
                planned = std::async(std::launch::async,
                                     [&driver, &render, next] { driver.prepare(render, next); });
              }
              driver.render(render, away(frame) ? elsewhere : camera, r == 0 ? bytes.data() : nullptr,
                            want_image, (r == 0 && want_image) ? image.data() : nullptr);
              if (planned.valid()) planned.get();
            }
            driver.synchronize();
            int recorded = 0;
            held[static_cast<size_t>(r)].resize(static_cast<size_t>(frames));
            avr::check(avr_renderer_corun_history(driver.get(), held[static_cast<size_t>(r)].data(), frames,
                                                  &recorded));
            if (recorded != frames) throw std::runtime_error("co-run history is incomplete");
          } catch (const std::exception& e) {
            errors[static_cast<size_t>(r)] = e.what();
          }
        });
      }
      for (std::thread& t : threads) t.join();
      for (const std::string& e : errors) {
        if (!e.empty()) throw std::runtime_error("rank failed: " + e);
      }
      for (int r = 1; r < n_ranks; ++r) {
        if (held[static_cast<size_t>(r)] != held[0]) {
          throw std::runtime_error("rank " + std::to_string(r) +
                                   " held other co-run candidates than rank 0: the search is not coordinated");
        }
      }
      {
        std::vector<int16_t> distinct(held[0]);
        std::sort(distinct.begin(), distinct.end());
        distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
        std::printf("corun candidates held: %zu distinct over %d frames\n", distinct.size(), frames);
      }
      std::vector<float> host_image(n_pixels * 5);
      std::vector<unsigned char> host_bytes(n_pixels * 3);
      image.download(host_image.data(), host_image.size());
      bytes.download(host_bytes.data(), host_bytes.size());
      write_file(argv[9], host_image.data(), host_image.size());
      FILE* out = std::fopen(argv[10], "wb");
      if (!out || std::fwrite(host_bytes.data(), 1, host_bytes.size(), out) != host_bytes.size()) {
        throw std::runtime_error("cannot write rgb8");
      }
      std::fclose(out);
      return 0;
    }
    if (mode == "compose_ranks" && argc == 11) {
      const int n_layers = std::atoi(argv[3]);
      const int n_ranks = std::atoi(argv[4]);
      const int W = std::atoi(argv[5]), H = std::atoi(argv[6]);
      const size_t n_pixels = static_cast<size_t>(W) * H;
      const std::vector<float> all = read_file<float>(argv[2], static_cast<size_t>(n_layers) * n_pixels * 5);
      const std::vector<float> hints = read_file<float>(argv[7], static_cast<size_t>(n_layers));
      const std::vector<int32_t> owners = read_file<int32_t>(argv[8], static_cast<size_t>(n_layers));
      const std::vector<int32_t> group32 = read_file<int32_t>(argv[9], static_cast<size_t>(n_ranks));
      const std::vector<int> group(group32.begin(), group32.end());
      standin::ThreadWorld world;
      world.n = n_ranks;
      std::vector<float> result(n_pixels * 5, -1.0f);
      std::vector<std::string> errors(static_cast<size_t>(n_ranks));
      std::vector<std::thread> threads;
      for (int r = 0; r < n_ranks; ++r) {
        threads.emplace_back([&, r] {
          try {
            avr::hip_ok(hipSetDevice(0), "hipSetDevice");
            standin::ThreadControl control{&world, r};
            // (in the reference tree the communicator is RCCL over xGMI, built from MPI_COMM_WORLD;
            //  here the ranks share one GPU, so they are wired with the in-process communicator)
            standin::Layered layered(W, H);  // geometry.localBoxes' layers of this rank, in order
            for (int l = 0; l < n_layers; ++l) {
              if (owners[static_cast<size_t>(l)] != r) continue;
              auto img = std::make_unique<standin::DepthSortImage>(W, H);
              std::memcpy(img->getColorBuffer(), all.data() + static_cast<size_t>(l) * n_pixels * 5,
                          sizeof(float) * n_pixels * 5);
              layered.layers.push_back(std::move(img));
              layered.hints.push_back(hints[static_cast<size_t>(l)]);
            }
            static std::vector<std::unique_ptr<avr::Communicator>> comms;
            static std::once_flag once;
            std::call_once(once, [&] { comms = avr::Communicator::local(n_ranks); });
            world.barrier();
            avr::HipDirectSend<standin::ThreadControl> compositor(control, 0, comms[static_cast<size_t>(r)].get());
            // the agreement check of the compose's plan travels over the CALLER's control plane
            // (MPI_Allgather in the reference's host), as with a communicator built from it
            comms[static_cast<size_t>(r)]->use_control(control);
            struct Unhook {
              avr_comm* comm;
              ~Unhook() { (void)avr_comm_set_control(comm, nullptr, nullptr); }
            } unhook{comms[static_cast<size_t>(r)]->get()};
            // the caller holds an `Image*` (Compositor::compose's argument): the layered image is
            // found by dynamic_cast as in DirectSendBase.cpp:288-298.  Three frames: after the
            // first the pooled buffers have their size and nothing is allocated any more.
            standin::Image* local = &layered;
            std::unique_ptr<standin::Image> piece;
            size_t after_first = 0;
            for (int frame = 0; frame < 3; ++frame) {
              piece = compositor.compose<standin::DepthSortImage, standin::Layered>(local, group, 0);
              if (frame == 0) after_first = compositor.allocations();
            }
            if (compositor.allocations() != after_first) {
              throw std::runtime_error("the compositor allocated after its first frame");
            }
            auto* pixels = static_cast<standin::DepthSortImage*>(piece.get());
            std::memcpy(result.data() + static_cast<size_t>(piece->getRegionBegin()) * 5,
                        pixels->getColorBuffer(),
                        sizeof(float) * static_cast<size_t>(piece->getNumberOfPixels()) * 5);
          } catch (const std::exception& e) {
            errors[static_cast<size_t>(r)] = e.what();
          }
        });
      }
      for (std::thread& t : threads) t.join();
      for (const std::string& e : errors) {
        if (!e.empty()) throw std::runtime_error("rank failed: " + e);
      }
      write_file(argv[10], result.data(), result.size());
      return 0;
    }
    if (mode == "compose_image" && argc == 9) {
      // classic direct send of ONE plain image per rank (DirectSendBase.cpp:257-281):
      //   adapter_test compose_image <images.bin> <kind> <n_ranks> <W> <H> <group.bin> <out.bin>
      // images.bin: n_ranks images of W*H pixels of the kind's pixel type; out: the gathered result
      const int kind = std::atoi(argv[3]);
      const int n_ranks = std::atoi(argv[4]);
      const int W = std::atoi(argv[5]), H = std::atoi(argv[6]);
      const size_t n_pixels = static_cast<size_t>(W) * H;
      const size_t words = (kind == 0) ? 5 : (kind == 1) ? 4 : 1;  // 32-bit words per pixel
      const std::vector<uint32_t> all =
          read_file<uint32_t>(argv[2], static_cast<size_t>(n_ranks) * n_pixels * words);
      const std::vector<int32_t> group32 = read_file<int32_t>(argv[7], static_cast<size_t>(n_ranks));
      const std::vector<int> group(group32.begin(), group32.end());
      standin::ThreadWorld world;
      world.n = n_ranks;
      std::vector<uint32_t> result(n_pixels * words, 0xdeadbeefu);
      std::vector<std::string> errors(static_cast<size_t>(n_ranks));
      std::vector<std::thread> threads;
      auto comms = avr::Communicator::local(n_ranks);
      auto run = [&](int r, auto image_tag) {
        using ImageT = decltype(image_tag);
        avr::hip_ok(hipSetDevice(0), "hipSetDevice");
        standin::ThreadControl control{&world, r};
        ImageT image(W, H);
        std::memcpy(image.getColorBuffer(), all.data() + static_cast<size_t>(r) * n_pixels * words,
                    n_pixels * words * 4);
        avr::HipDirectSend<standin::ThreadControl> compositor(control, 0, comms[static_cast<size_t>(r)].get());
        standin::Image* local = &image;  // not layered: the plugin takes the classic path
        std::unique_ptr<standin::Image> piece;
        size_t after_first = 0;
        for (int frame = 0; frame < 2; ++frame) {
          piece = compositor.template compose<ImageT, standin::Layered>(local, group, 0);
          if (frame == 0) after_first = compositor.allocations();
        }
        if (compositor.allocations() != after_first) {
          throw std::runtime_error("the compositor allocated after its first frame");
        }
        std::memcpy(result.data() + static_cast<size_t>(piece->getRegionBegin()) * words,
                    static_cast<ImageT*>(piece.get())->getColorBuffer(),
                    static_cast<size_t>(piece->getNumberOfPixels()) * words * 4);
      };
      for (int r = 0; r < n_ranks; ++r) {
        threads.emplace_back([&, r] {
          try {
            if (kind == 0) run(r, standin::DepthSortImage(1, 1));
            else if (kind == 1) run(r, standin::FloatImage(1, 1));
            else run(r, standin::UByteImage(1, 1));
          } catch (const std::exception& e) {
            errors[static_cast<size_t>(r)] = e.what();
          }
        });
      }
      for (std::thread& t : threads) t.join();
      for (const std::string& e : errors) {
        if (!e.empty()) throw std::runtime_error("rank failed: " + e);
      }
      FILE* out = std::fopen(argv[8], "wb");
      if (!out || std::fwrite(result.data(), 4, result.size(), out) != result.size()) {
        throw std::runtime_error("cannot write the result");
      }
      std::fclose(out);
      return 0;
    }
    std::fprintf(stderr, "usage: adapter_test paint|compose|frame|compose_ranks|compose_image ...\n");
    return 2;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "adapter_test: %s\n", e.what());
    return 1;
  }
}
