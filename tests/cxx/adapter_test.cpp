// Compile-and-run check of include/avr_reference_api.hpp against stand-ins that carry exactly
// the member names the reference's types have (volume::AmrBox, amrex::Array4, amrex::Box,
// volume::CameraParameters, volume::ScalarTransform, ImageRGBAFloatColorDepthSort,
// LayeredVolumeImage).  Driven by tests/test_cxx_adapter.py, which compares the outputs with
// the oracle.
//   adapter_test paint   cells.bin nx ny nz ghost W H out.bin
//   adapter_test compose layers.bin n_layers n_pixels hints.bin out.bin
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
    std::fprintf(stderr, "usage: adapter_test paint|compose ...\n");
    return 2;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "adapter_test: %s\n", e.what());
    return 1;
  }
}
