// Test infrastructure (oracle/): a driver -- this repository's own code -- around the REFERENCE's own
// compositor, compiled from its sources where they lie under /root/reference (oracle/ref_compose/
// build.sh; nothing of the reference is copied into the repository).  It runs
//     DirectSendBase::compose(LayeredVolumeImage*, group, MPI_COMM_WORLD)      DirectSendBase.cpp:285-458
//     ImageFull::Gather(0, MPI_COMM_WORLD)                                     ImageColorOnly.hpp:220-270
// under mpiexec on synthetic layers exactly as VolumeRenderer::renderSingleTrial does between
// VolumeRenderer.cpp:1225 and :1294, and writes what rank 0 gathered.  Only used to generate the
// golden vectors of tests/golden/ (tests/golden/make_ref_compose.py); never by a product path.
//
//   ref_compose <layers.bin> <out.bin> [reverse|forward [out.ppm]]
//   ref_compose classic <kind> <images.bin> <out.bin> [reverse]
// classic:    the NON-layered path of DirectSendBase::compose (DirectSendBase.cpp:257-281, 300-311): every
//             rank holds ONE image of kind 0 (depth-sort f32x5), 1 (rgba f32x4) or 2 (rgba u8x4);
//             images.bin: int32 W, H, n_ranks, then every rank's W*H pixels in rank order; out.bin as
//             below.  (With more than two ranks the reference blends whichever neighbours have arrived
//             -- the association of the float blends is then its one degree of freedom; two ranks
//             have one blend and one answer.)
// layers.bin: int32 W, H, n_layers; then per layer int32 owner, float depth hint, W*H*5 floats
//             (premultiplied r, g, b, a, depth).  A rank takes the layers it owns in file order
//             (their local index = geometry.localBoxes order).
// out.bin:    int32 n_ranks, then per rank int32 regionBegin, regionEnd of the piece it returned
//             from compose; then W*H*5 floats of the gathered image (written by rank 0).
// reverse:    hand compose the ranks in reversed group order (the visibility order's freedom).
// out.ppm:    rank 0 also writes the gathered image with the reference's SavePPM (SavePPM.cpp:17-36:
//             Color::GetComponentAsByte, rows top-down) -- the 8-bit output of the frame.
#include <mpi.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <Common/ImageFull.hpp>
#include <Common/ImageRGBAFloatColorDepthSort.hpp>
#include <Common/ImageRGBAFloatColorOnly.hpp>
#include <Common/ImageRGBAUByteColorOnly.hpp>
#include <Common/LayeredVolumeImage.hpp>
#include <Common/SavePPM.hpp>
#include <DirectSend/Base/DirectSendBase.hpp>

namespace {

MPI_Group ordered_group(int n_ranks, bool reverse) {
  MPI_Group world = MPI_GROUP_NULL, group = MPI_GROUP_NULL;
  MPI_Comm_group(MPI_COMM_WORLD, &world);
  std::vector<int> order(static_cast<size_t>(n_ranks));
  for (int k = 0; k < n_ranks; ++k) order[static_cast<size_t>(k)] = reverse ? n_ranks - 1 - k : k;
  MPI_Group_incl(world, n_ranks, order.data(), &group);
  MPI_Group_free(&world);
  return group;
}

// The classic direct send of one plain image per rank, gathered and written by rank 0.
template <class ImageT>
void classic(const char* in_path, const char* out_path, bool reverse, int rank, int n_ranks) {
  using ColorType = typename ImageT::ColorType;
  constexpr int kVec = ImageT::ColorVecSize;
  FILE* in = std::fopen(in_path, "rb");
  if (in == nullptr) throw std::runtime_error("cannot read the images");
  int32_t head[3];
  if (std::fread(head, 4, 3, in) != 3 || head[2] != n_ranks) throw std::runtime_error("bad header");
  const int width = head[0], height = head[1];
  const size_t words = static_cast<size_t>(width) * height * kVec;
  ImageT image(width, height);
  std::fseek(in, static_cast<long>(12 + static_cast<size_t>(rank) * words * sizeof(ColorType)), SEEK_SET);
  if (std::fread(image.getColorBuffer(), sizeof(ColorType), words, in) != words) {
    throw std::runtime_error("short image");
  }
  std::fclose(in);
  MPI_Group group = ordered_group(n_ranks, reverse);
  DirectSendBase compositor;
  std::unique_ptr<Image> composited = compositor.compose(&image, group, MPI_COMM_WORLD);
  MPI_Group_free(&group);
  auto* full = dynamic_cast<ImageFull*>(composited.get());
  if (full == nullptr) throw std::runtime_error("compose did not return a full image");
  int32_t region[2] = {full->getRegionBegin(), full->getRegionEnd()};
  std::vector<int32_t> regions(static_cast<size_t>(n_ranks) * 2);
  MPI_Gather(region, 2, MPI_INT, regions.data(), 2, MPI_INT, 0, MPI_COMM_WORLD);
  std::unique_ptr<ImageFull> gathered = full->Gather(0, MPI_COMM_WORLD);
  if (rank == 0) {
    auto* whole = dynamic_cast<ImageT*>(gathered.get());
    if (whole == nullptr || whole->getNumberOfPixels() != width * height) {
      throw std::runtime_error("the gathered image is not the whole image");
    }
    FILE* out = std::fopen(out_path, "wb");
    if (out == nullptr) throw std::runtime_error("cannot write the result");
    const int32_t n = n_ranks;
    std::fwrite(&n, 4, 1, out);
    std::fwrite(regions.data(), 4, regions.size(), out);
    std::fwrite(whole->getColorBuffer(), sizeof(ColorType), words, out);
    std::fclose(out);
  }
}

}  // namespace

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  int rank = 0, n_ranks = 1;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &n_ranks);
  int status = 0;
  try {
    if (argc >= 5 && std::string(argv[1]) == "classic") {
      const int kind = std::atoi(argv[2]);
      const bool backwards = argc > 5 && std::string(argv[5]) == "reverse";
      if (kind == 0) {
        classic<ImageRGBAFloatColorDepthSort>(argv[3], argv[4], backwards, rank, n_ranks);
      } else if (kind == 1) {
        classic<ImageRGBAFloatColorOnly>(argv[3], argv[4], backwards, rank, n_ranks);
      } else if (kind == 2) {
        classic<ImageRGBAUByteColorOnly>(argv[3], argv[4], backwards, rank, n_ranks);
      } else {
        throw std::runtime_error("unknown image kind");
      }
      MPI_Finalize();
      return 0;
    }
    if (argc < 3) throw std::runtime_error("usage: ref_compose layers.bin out.bin [reverse]");
    const bool reverse = argc > 3 && std::string(argv[3]) == "reverse";
    FILE* in = std::fopen(argv[1], "rb");
    if (in == nullptr) throw std::runtime_error("cannot read the layers");
    int32_t head[3];
    if (std::fread(head, 4, 3, in) != 3) throw std::runtime_error("short header");
    const int width = head[0], height = head[1], n_layers = head[2];
    const size_t floats = static_cast<size_t>(width) * height * 5;
    std::vector<std::unique_ptr<ImageRGBAFloatColorDepthSort>> layers;
    std::vector<float> hints;
    std::vector<float> pixels(floats);
    for (int l = 0; l < n_layers; ++l) {
      int32_t owner = 0;
      float hint = 0.0f;
      if (std::fread(&owner, 4, 1, in) != 1 || std::fread(&hint, 4, 1, in) != 1 ||
          std::fread(pixels.data(), 4, floats, in) != floats) {
        throw std::runtime_error("short layer");
      }
      if (owner != rank) continue;
      auto image = std::make_unique<ImageRGBAFloatColorDepthSort>(width, height);
      std::memcpy(image->getColorBuffer(), pixels.data(), floats * sizeof(float));
      layers.push_back(std::move(image));
      hints.push_back(hint);
    }
    std::fclose(in);
    auto prototype = std::make_unique<ImageRGBAFloatColorDepthSort>(width, height);
    LayeredVolumeImage layered(width, height, std::move(layers), std::move(hints), std::move(prototype));

    MPI_Group group = ordered_group(n_ranks, reverse);
    DirectSendBase compositor;
    std::unique_ptr<Image> composited = compositor.compose(&layered, group, MPI_COMM_WORLD);
    MPI_Group_free(&group);
    auto* full = dynamic_cast<ImageFull*>(composited.get());
    if (full == nullptr) throw std::runtime_error("compose did not return a full image");
    int32_t region[2] = {full->getRegionBegin(), full->getRegionEnd()};
    std::vector<int32_t> regions(static_cast<size_t>(n_ranks) * 2);
    MPI_Gather(region, 2, MPI_INT, regions.data(), 2, MPI_INT, 0, MPI_COMM_WORLD);
    std::unique_ptr<ImageFull> gathered = full->Gather(0, MPI_COMM_WORLD);
    if (rank == 0) {
      auto* image = dynamic_cast<ImageRGBAFloatColorDepthSort*>(gathered.get());
      if (image == nullptr || image->getNumberOfPixels() != width * height) {
        throw std::runtime_error("the gathered image is not the whole depth-sort image");
      }
      FILE* out = std::fopen(argv[2], "wb");
      if (out == nullptr) throw std::runtime_error("cannot write the result");
      const int32_t n = n_ranks;
      std::fwrite(&n, 4, 1, out);
      std::fwrite(regions.data(), 4, regions.size(), out);
      std::fwrite(image->getColorBuffer(), 4, floats, out);
      std::fclose(out);
      if (argc > 4 && !SavePPM(*image, argv[4])) throw std::runtime_error("SavePPM failed");
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "ref_compose (rank %d): %s\n", rank, e.what());
    status = 1;
  }
  MPI_Finalize();
  return status;
}
