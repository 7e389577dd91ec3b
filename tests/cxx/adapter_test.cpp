// Compile-and-run check of include/avr_reference_api.hpp against stand-ins that carry exactly
// the member names the reference's types have (volume::AmrBox, amrex::Array4, amrex::Box,
// volume::CameraParameters, volume::ScalarTransform, ImageRGBAFloatColorDepthSort,
// LayeredVolumeImage).  Driven by tests/test_cxx_adapter.py, which compares the outputs with
// the oracle.
//   adapter_test paint   cells.bin nx ny nz ghost W H out.bin
//   adapter_test compose layers.bin n_layers n_pixels hints.bin out.bin
//   adapter_test frame   scene.bin n_ranks W H transparency out_image.bin out_rgb8.bin
//     scene.bin: int32 n_boxes, then per box 6 doubles (corners), 3 int32 dims, int32 owner,
//     nx*ny*nz doubles.  All ranks of the frame are played in this one process: every rank has
//     its own avr::RankFrame, and the all-to-all is the device-to-device copies below.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/avr_reference_api.hpp"

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

struct DepthSortImage {  // ImageRGBAFloatColorDepthSort: 5 floats per pixel on the host
  int width, height;
  std::vector<float> buffer;
  DepthSortImage(int w, int h) : width(w), height(h), buffer(static_cast<size_t>(w) * h * 5) {}
  int getWidth() const { return width; }
  int getHeight() const { return height; }
  int getNumberOfPixels() const { return width * height; }
  float* getColorBuffer() { return buffer.data(); }
  const float* getColorBuffer() const { return buffer.data(); }
};
struct Layered {  // LayeredVolumeImage / LayeredImageInterface
  std::vector<std::unique_ptr<DepthSortImage>> layers;
  std::vector<float> hints;
  int getLayerCount() const { return static_cast<int>(layers.size()); }
  DepthSortImage* getLayer(int i) { return layers[static_cast<size_t>(i)].get(); }
  float getLayerDepthHint(int i) const { return hints[static_cast<size_t>(i)]; }
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
      standin::Layered layered;
      layered.hints = read_file<float>(argv[5], static_cast<size_t>(n_layers));
      for (int l = 0; l < n_layers; ++l) {
        auto img = std::make_unique<standin::DepthSortImage>(n_pixels, 1);
        std::memcpy(img->getColorBuffer(), all.data() + static_cast<size_t>(l) * n_pixels * 5,
                    sizeof(float) * static_cast<size_t>(n_pixels) * 5);
        layered.layers.push_back(std::move(img));
      }
      const std::vector<float> out = avr::compose_single_rank(context, layered, n_pixels);
      write_file(argv[6], out.data(), out.size());
      return 0;
    }
    if (mode == "frame" && argc == 9) {
      const int n_ranks = std::atoi(argv[3]);
      const int W = std::atoi(argv[4]), H = std::atoi(argv[5]);
      const float transparency = static_cast<float>(std::atof(argv[6]));
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
      avr_paint_params params;
      std::memset(&params, 0, sizeof(params));
      params.width = W;
      params.height = H;
      params.scalar_range[0] = 0.0f;
      params.scalar_range[1] = 1.0f;
      params.box_transparency = transparency;
      double bmin[3] = {-0.05, -0.05, -0.05}, bmax[3] = {1.05, 1.05, 1.05};
      for (int c = 0; c < 3; ++c) {
        params.bounds_min[c] = bmin[c];
        params.bounds_max[c] = bmax[c];
      }
      avr::check(avr_reference_sample_distance(boxes.data(), n_boxes, bmin, bmax,
                                               &params.reference_sample_distance));
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

      std::vector<std::unique_ptr<avr::RankFrame>> ranks;
      for (int r = 0; r < n_ranks; ++r) {
        ranks.push_back(std::make_unique<avr::RankFrame>(context, boxes, owner, r, n_ranks, transform));
        ranks.back()->plan(params, camera);
        ranks.back()->paint();
      }
      context.synchronize();
      // the all-to-all: block for peer d in rank s's send buffer -> block from s in d's recv buffer
      for (int s = 0; s < n_ranks; ++s) {
        int64_t send_at = 0;
        for (int d = 0; d < n_ranks; ++d) {
          const int64_t count = ranks[static_cast<size_t>(s)]->send_splits()[static_cast<size_t>(d)];
          int64_t recv_at = 0;
          for (int k = 0; k < s; ++k) recv_at += ranks[static_cast<size_t>(d)]->recv_splits()[static_cast<size_t>(k)];
          if (ranks[static_cast<size_t>(d)]->recv_splits()[static_cast<size_t>(s)] != count) {
            throw std::runtime_error("send / recv splits disagree");
          }
          if (count > 0) {
            avr::hip_ok(hipMemcpy(ranks[static_cast<size_t>(d)]->recv_buffer() + recv_at,
                                  ranks[static_cast<size_t>(s)]->paint_buffer() + send_at,
                                  static_cast<size_t>(count) * sizeof(float), hipMemcpyDeviceToDevice),
                        "hipMemcpy(exchange)");
          }
          send_at += count;
        }
      }
      // fold + Gather (pieces concatenated by region begin) + overlay + bytes
      const size_t n_pixels = static_cast<size_t>(W) * H;
      avr::DeviceBuffer<float> full(n_pixels * 5 + 1);
      avr::DeviceBuffer<unsigned char> full8(n_pixels * 3 + 1);
      double tight_min[3], tight_max[3];
      avr::check(avr_tight_bounds(boxes.data(), n_boxes, bmin, bmax, tight_min, tight_max));
      for (int r = 0; r < n_ranks; ++r) {
        avr::DeviceBuffer<float> piece;
        ranks[static_cast<size_t>(r)]->fold(&piece, nullptr);
        const avr_frame_plan_info& info = ranks[static_cast<size_t>(r)]->info();
        const size_t n = static_cast<size_t>(info.piece_end - info.piece_begin);
        // each rank overlays its own piece and converts it to bytes (pixels are independent)
        avr::check(avr_bbox_overlay(context.get(), tight_min, tight_max, &camera, 1, W, H,
                                    info.piece_begin, info.piece_end, piece.data(),
                                    full8.data() + static_cast<size_t>(info.piece_begin) * 3));
        context.synchronize();
        if (n > 0) {
          avr::hip_ok(hipMemcpy(full.data() + static_cast<size_t>(info.piece_begin) * 5, piece.data(),
                                n * 5 * sizeof(float), hipMemcpyDeviceToDevice), "hipMemcpy(gather)");
        }
      }
      std::vector<float> image(n_pixels * 5);
      std::vector<unsigned char> bytes(n_pixels * 3);
      full.download(image.data(), image.size());
      full8.download(bytes.data(), bytes.size());
      write_file(argv[7], image.data(), image.size());
      FILE* out = std::fopen(argv[8], "wb");
      if (!out || std::fwrite(bytes.data(), 1, bytes.size(), out) != bytes.size()) {
        throw std::runtime_error("cannot write rgb8");
      }
      std::fclose(out);
      return 0;
    }
    std::fprintf(stderr, "usage: adapter_test paint|compose|frame ...\n");
    return 2;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "adapter_test: %s\n", e.what());
    return 1;
  }
}
