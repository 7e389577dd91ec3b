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
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <Common/ImageFull.hpp>
#include <Common/ImageRGBAFloatColorDepthSort.hpp>
#include <Common/LayeredVolumeImage.hpp>
#include <Common/SavePPM.hpp>
#include <DirectSend/Base/DirectSendBase.hpp>

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  int rank = 0, n_ranks = 1;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &n_ranks);
  int status = 0;
  try {
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

    MPI_Group world = MPI_GROUP_NULL, group = MPI_GROUP_NULL;
    MPI_Comm_group(MPI_COMM_WORLD, &world);
    std::vector<int> order(static_cast<size_t>(n_ranks));
    for (int k = 0; k < n_ranks; ++k) order[static_cast<size_t>(k)] = reverse ? n_ranks - 1 - k : k;
    MPI_Group_incl(world, n_ranks, order.data(), &group);

    DirectSendBase compositor;
    std::unique_ptr<Image> composited = compositor.compose(&layered, group, MPI_COMM_WORLD);
    MPI_Group_free(&group);
    MPI_Group_free(&world);
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
