// Test infrastructure (oracle/): a second driver -- this repository's own code -- around the
// REFERENCE's own image classes, compiled from their sources where they lie (oracle/ref_compose/
// build.sh).  It runs, for the three image types the hot path names,
//     Image::blend (ImageColorOnly<Features>::blend, Common/ImageColorOnly.hpp:119-199, with
//     Features::blend of ImageRGBAFloatColorDepthSort.hpp:13-27, ImageRGBAFloatColorOnly.hpp:19-26,
//     ImageRGBAUByteColorOnly.hpp:19-34) on two images with arbitrary regions, and
//     setColor / getColor of the ubyte image (encodeColor / decodeColor,
//     Common/ImageRGBAUByteColorOnly.cpp:16-39, Common/Color.hpp:66-91)
// and writes the results.  Only used to generate tests/golden/ref_blend.npz.
//
//   ref_blend <in.bin> <out.bin>
// in.bin:  int32 n_cases; per case int32 kind (0 depth-sort f32x5, 1 rgba f32x4, 2 rgba u8x4,
//          3 = encode / decode of the ubyte image), width, height, tb, te, bb, be; then the top
//          image's (te - tb) pixels and the bottom image's (be - bb) pixels (kind 3: (te - tb)
//          colours of 4 floats, no bottom).
// out.bin: per case int32 begin, end; then (end - begin) pixels of the blended image (kind 3: the
//          encoded uint32 of every colour, then the 4 floats getColor decodes from it).
#include <mpi.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#include <Common/ImageRGBAFloatColorDepthSort.hpp>
#include <Common/ImageRGBAFloatColorOnly.hpp>
#include <Common/ImageRGBAUByteColorOnly.hpp>

namespace {

void read_exact(FILE* in, void* data, size_t bytes) {
  if (bytes != 0 && std::fread(data, 1, bytes, in) != bytes) throw std::runtime_error("short input");
}

template <class ImageT>
void blend_case(FILE* in, FILE* out, int width, int height, int tb, int te, int bb, int be) {
  using ColorType = typename ImageT::ColorType;
  constexpr int kVec = ImageT::ColorVecSize;
  ImageT top(width, height, tb, te), bottom(width, height, bb, be);
  read_exact(in, top.getColorBuffer(), sizeof(ColorType) * kVec * static_cast<size_t>(te - tb));
  read_exact(in, bottom.getColorBuffer(), sizeof(ColorType) * kVec * static_cast<size_t>(be - bb));
  std::unique_ptr<Image> blended = top.blend(bottom);
  auto* result = dynamic_cast<ImageT*>(blended.get());
  if (result == nullptr) throw std::runtime_error("blend returned another image type");
  const int32_t region[2] = {result->getRegionBegin(), result->getRegionEnd()};
  std::fwrite(region, 4, 2, out);
  std::fwrite(result->getColorBuffer(), sizeof(ColorType) * kVec,
              static_cast<size_t>(region[1] - region[0]), out);
}

}  // namespace

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);  // (the image classes link MPI; nothing is communicated)
  int status = 0;
  try {
    if (argc < 3) throw std::runtime_error("usage: ref_blend in.bin out.bin");
    FILE* in = std::fopen(argv[1], "rb");
    FILE* out = std::fopen(argv[2], "wb");
    if (in == nullptr || out == nullptr) throw std::runtime_error("cannot open the files");
    int32_t n_cases = 0;
    read_exact(in, &n_cases, 4);
    for (int c = 0; c < n_cases; ++c) {
      int32_t head[7];
      read_exact(in, head, sizeof(head));
      const int kind = head[0], width = head[1], height = head[2];
      const int tb = head[3], te = head[4], bb = head[5], be = head[6];
      if (kind == 0) {
        blend_case<ImageRGBAFloatColorDepthSort>(in, out, width, height, tb, te, bb, be);
      } else if (kind == 1) {
        blend_case<ImageRGBAFloatColorOnly>(in, out, width, height, tb, te, bb, be);
      } else if (kind == 2) {
        blend_case<ImageRGBAUByteColorOnly>(in, out, width, height, tb, te, bb, be);
      } else if (kind == 3) {
        const int n = te - tb;
        std::vector<float> colours(static_cast<size_t>(n) * 4);
        read_exact(in, colours.data(), colours.size() * 4);
        ImageRGBAUByteColorOnly image(width, height, tb, te);
        std::vector<float> decoded(colours.size());
        for (int i = 0; i < n; ++i) {
          const float* p = &colours[static_cast<size_t>(i) * 4];
          image.setColor(i, Color(p[0], p[1], p[2], p[3]));
          const Color back = image.getColor(i);
          for (int k = 0; k < 4; ++k) decoded[static_cast<size_t>(i) * 4 + k] = back.Components[k];
        }
        const int32_t region[2] = {tb, te};
        std::fwrite(region, 4, 2, out);
        std::fwrite(image.getColorBuffer(), 4, static_cast<size_t>(n), out);
        std::fwrite(decoded.data(), 4, decoded.size(), out);
      } else {
        throw std::runtime_error("unknown case kind");
      }
    }
    std::fclose(in);
    std::fclose(out);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "ref_blend: %s\n", e.what());
    status = 1;
  }
  MPI_Finalize();
  return status;
}
