// Scene scalar statistics and histogram: the two cell scans of the
// reference's scene construction, as HBM-streaming kernels over the scene's boxes.
//
//   scalar_stats_kernel   reduceLocalScalarStats, VolumeRenderer/SceneBuilder.cpp:53-97
//                         (min, max, min positive, finite count over all cells)
//   histogram_kernel      the binning of ComputeSceneHistogram, SceneBuilder.cpp:495-532
//
// Both use the classify pass's decomposition: one workgroup = 4 k-planes x 4 j-rows x 128
// cells of one box, rows read coalesced (the boxes keep their Array4 strides).  Results are
// exact and order-independent (min / max / integer counts).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "avr_device.h"
#include "avr_internal.h"

namespace avr {

namespace {

constexpr int kThreads = 256;
constexpr int kMaxLdsBins = 4096;
constexpr uint32_t kTilesPerGroup = 16;  // histogram: tiles (2048 cells each) per workgroup

struct TileCoords {
  const BoxDev* box;
  int chunk, bj, bk;
};

// Which box / tile does workgroup `tile` belong to (binary search over the prefix sums).
__device__ __forceinline__ TileCoords locate_tile(const BoxDev* boxes, const uint32_t* tile_begin,
                                                  int n_boxes, uint32_t tile) {
  int lo = 0, hi = n_boxes;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_begin[mid] <= tile) {
      lo = mid;
    } else {
      hi = mid;
    }
  }
  TileCoords t;
  t.box = &boxes[lo];
  const int bricks_y = (t.box->ny + kBrickY - 1) >> 2;
  const int chunks = (t.box->nx + kClassifyChunk - 1) / kClassifyChunk;
  uint32_t local = tile - tile_begin[lo];
  t.chunk = static_cast<int>(local % static_cast<uint32_t>(chunks));
  local /= static_cast<uint32_t>(chunks);
  t.bj = static_cast<int>(local % static_cast<uint32_t>(bricks_y));
  t.bk = static_cast<int>(local / static_cast<uint32_t>(bricks_y));
  return t;
}

// Calls visit(raw) for every valid cell of the workgroup's tile.
template <typename F>
__device__ __forceinline__ void for_each_cell(const TileCoords& tile, F&& visit) {
  const BoxDev& box = *tile.box;
  const double __attribute__((address_space(1)))* cells =
      (const double __attribute__((address_space(1)))*)box.cells;
  const uint32_t jstride = static_cast<uint32_t>(box.jstride);
  const uint32_t kstride = static_cast<uint32_t>(box.kstride);
  const int t = static_cast<int>(threadIdx.x);
  const bool paired = ((reinterpret_cast<uintptr_t>(box.cells) & 15u) == 0) &&
                      ((jstride & 1u) == 0) && ((kstride & 1u) == 0);
  if (paired) {
    typedef double double2_t __attribute__((ext_vector_type(2)));
    const int i = tile.chunk * kClassifyChunk + (t & 63) * 2;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int row = pass * 4 + (t >> 6);
      const int j = tile.bj * kBrickY + (row & 3);
      const int k = tile.bk * kBrickZ + (row >> 2);
      if (i < box.nx && j < box.ny && k < box.nz) {
        const uint32_t at = static_cast<uint32_t>(i) + static_cast<uint32_t>(j) * jstride +
                            static_cast<uint32_t>(k) * kstride;
        if (i + 1 < box.nx) {
          const double2_t raw = *(const double2_t __attribute__((address_space(1)))*)(cells + at);
          visit(raw.x);
          visit(raw.y);
        } else {
          visit(cells[at]);
        }
      }
    }
  } else {
    const int i = tile.chunk * kClassifyChunk + (t & 127);
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int row = pass * 2 + (t >> 7);
      const int j = tile.bj * kBrickY + (row & 3);
      const int k = tile.bk * kBrickZ + (row >> 2);
      if (i < box.nx && j < box.ny && k < box.nz) {
        visit(cells[static_cast<uint32_t>(i) + static_cast<uint32_t>(j) * jstride +
                    static_cast<uint32_t>(k) * kstride]);
      }
    }
  }
}

struct Stats {
  double lo, hi, lo_positive;
  long long finite;
};

__device__ __forceinline__ Stats merge(const Stats& a, const Stats& b) {
  Stats m;
  m.lo = (b.lo < a.lo) ? b.lo : a.lo;
  m.hi = (b.hi > a.hi) ? b.hi : a.hi;
  m.lo_positive = (b.lo_positive < a.lo_positive) ? b.lo_positive : a.lo_positive;
  m.finite = a.finite + b.finite;
  return m;
}

__device__ __forceinline__ Stats shuffle_xor(const Stats& s, int mask) {
  Stats o;
  o.lo = __shfl_xor(s.lo, mask, 64);
  o.hi = __shfl_xor(s.hi, mask, 64);
  o.lo_positive = __shfl_xor(s.lo_positive, mask, 64);
  o.finite = __shfl_xor(s.finite, mask, 64);
  return o;
}

// partial[workgroup] = statistics of the tiles that workgroup walked (grid-stride).
__global__ __launch_bounds__(kThreads) void scalar_stats_kernel(
    const BoxDev* __restrict__ boxes, const uint32_t* __restrict__ tile_begin, const int n_boxes,
    const uint32_t n_tiles, Stats* __restrict__ partial) {
  __shared__ Stats wave_stats[kThreads / 64];
  const double inf = __builtin_huge_val();
  Stats mine = {inf, -inf, inf, 0};
  for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const TileCoords tile = locate_tile(boxes, tile_begin, n_boxes, t);
    for_each_cell(tile, [&](double raw) {
      // non-finite cells contribute {inf, -inf, inf, 0} (SceneBuilder.cpp:84-86)
      if (__builtin_isfinite(raw)) {
        mine.lo = (raw < mine.lo) ? raw : mine.lo;
        mine.hi = (raw > mine.hi) ? raw : mine.hi;
        mine.lo_positive = (raw > 0.0 && raw < mine.lo_positive) ? raw : mine.lo_positive;
        mine.finite += 1;
      }
    });
  }
  for (int mask = 32; mask > 0; mask >>= 1) mine = merge(mine, shuffle_xor(mine, mask));
  const int lane = static_cast<int>(threadIdx.x) & 63;
  const int wave = static_cast<int>(threadIdx.x) >> 6;
  if (lane == 0) wave_stats[wave] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    Stats total = wave_stats[0];
    for (int w = 1; w < kThreads / 64; ++w) total = merge(total, wave_stats[w]);
    partial[blockIdx.x] = total;
  }
}

// out[0] = merge of partial[0..n)
__global__ __launch_bounds__(kThreads) void reduce_stats_kernel(const Stats* __restrict__ partial,
                                                                const uint32_t n,
                                                                Stats* __restrict__ out) {
  __shared__ Stats wave_stats[kThreads / 64];
  const double inf = __builtin_huge_val();
  Stats mine = {inf, -inf, inf, 0};
  for (uint32_t i = threadIdx.x; i < n; i += kThreads) mine = merge(mine, partial[i]);
  for (int mask = 32; mask > 0; mask >>= 1) mine = merge(mine, shuffle_xor(mine, mask));
  if ((threadIdx.x & 63) == 0) wave_stats[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    Stats total = wave_stats[0];
    for (int w = 1; w < kThreads / 64; ++w) total = merge(total, wave_stats[w]);
    out[0] = total;
  }
}

// Bin index of one cell (SceneBuilder.cpp:511-530).  SIMPLE = no log scaling, normalise on:
// the transform's clamp to [0,1] is applied after the f64 -> f32 cast (rounding is monotone and
// 0, 1 are exact, only the sign of a zero can differ and no later step depends on it) and the
// clamps become single median instructions.
template <bool SIMPLE>
__device__ __forceinline__ int histogram_bin(double raw, const FrameConsts& fc, float range_min,
                                             float range_max, float inverse_width,
                                             int bin_count) {
  if (SIMPLE) {
    double v = __builtin_isfinite(raw) ? raw : 0.0;
    v = (v - fc.norm_min) * fc.inv_norm_span;
    float value = __builtin_amdgcn_fmed3f(static_cast<float>(v), 0.0f, 1.0f);
    value = __builtin_amdgcn_fmed3f(value, range_min, range_max);
    const float normalized =
        __builtin_amdgcn_fmed3f((value - range_min) * inverse_width, 0.0f, 1.0f);
    const int index = static_cast<int>(normalized * static_cast<float>(bin_count));
    return (index >= bin_count) ? bin_count - 1 : index;
  }
  float value = apply_scalar_transform(raw, fc);
  if (value < range_min) {
    value = range_min;
  } else if (value > range_max) {
    value = range_max;
  }
  float normalized = (value - range_min) * inverse_width;
  if (normalized < 0.0f) {
    normalized = 0.0f;
  } else if (normalized > 1.0f) {
    normalized = 1.0f;
  }
  int index = static_cast<int>(normalized * static_cast<float>(bin_count));
  if (index >= bin_count) {
    index = bin_count - 1;
  } else if (index < 0) {
    index = 0;
  }
  return index;
}

// LDS == true: the workgroup counts the tiles it walks (grid-stride) into an LDS histogram
// (bin_count <= kMaxLdsBins) and adds its non-zero bins to the global 64-bit counters once --
// a few thousand global atomics per launch instead of one per bin and tile, which serialise on
// the same few addresses.  Otherwise straight global atomics.
template <bool LDS, bool SIMPLE>
__global__ __launch_bounds__(kThreads) void histogram_kernel(
    const FrameConsts fc, const BoxDev* __restrict__ boxes,
    const uint32_t* __restrict__ tile_begin, const int n_boxes, const uint32_t n_tiles,
    const float range_min, const float range_max, const float inverse_width, const int bin_count,
    unsigned long long* __restrict__ counts) {
  extern __shared__ unsigned int local_counts[];
  if (LDS) {
    for (int b = threadIdx.x; b < bin_count; b += kThreads) local_counts[b] = 0u;
    __syncthreads();
  }
  // kTilesPerGroup consecutive tiles per workgroup: enough workgroups in flight to hide the
  // per-tile latency chain, few enough that the final flush stays cheap
  const uint32_t first = blockIdx.x * kTilesPerGroup;
  const uint32_t last = (first + kTilesPerGroup < n_tiles) ? first + kTilesPerGroup : n_tiles;
  for (uint32_t t = first; t < last; ++t) {
    const TileCoords tile = locate_tile(boxes, tile_begin, n_boxes, t);
    for_each_cell(tile, [&](double raw) {
      const int bin =
          histogram_bin<SIMPLE>(raw, fc, range_min, range_max, inverse_width, bin_count);
      if (LDS) {
        atomicAdd(&local_counts[bin], 1u);
      } else {
        atomicAdd(&counts[bin], 1ull);
      }
    });
  }
  if (LDS) {
    __syncthreads();
    for (int b = threadIdx.x; b < bin_count; b += kThreads) {
      const unsigned int c = local_counts[b];
      if (c != 0u) atomicAdd(&counts[b], static_cast<unsigned long long>(c));
    }
  }
}

int check(const char* what) {
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string(what) + ": " + hipGetErrorString(err));
    return AVR_ERR_RUNTIME;
  }
  return AVR_OK;
}

}  // namespace

static_assert(sizeof(Stats) == 32, "Stats is 3 doubles + one 64-bit count");

// workgroups of the grid-stride scans: 8 per CU of the 256 CUs, fewer for small scenes
uint32_t scan_workgroups(uint32_t n_tiles) {
  return n_tiles < kScanWorkgroups ? n_tiles : kScanWorkgroups;
}

int launch_scalar_stats(const BoxDev* boxes_dev, const uint32_t* tile_begin_dev, int n_boxes,
                        uint32_t n_tiles, void* partial_dev, void* out_dev, void* stream_v) {
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  const uint32_t groups = scan_workgroups(n_tiles);
  if (groups > 0) {
    hipLaunchKernelGGL(scalar_stats_kernel, dim3(groups), dim3(kThreads), 0, stream, boxes_dev,
                       tile_begin_dev, n_boxes, n_tiles, static_cast<Stats*>(partial_dev));
    const int status = check("scalar_stats_kernel");
    if (status != AVR_OK) return status;
  }
  hipLaunchKernelGGL(reduce_stats_kernel, dim3(1), dim3(kThreads), 0, stream,
                     static_cast<const Stats*>(partial_dev), groups, static_cast<Stats*>(out_dev));
  return check("reduce_stats_kernel");
}

int launch_histogram(const FrameConsts& fc, const BoxDev* boxes_dev,
                     const uint32_t* tile_begin_dev, int n_boxes, uint32_t n_tiles,
                     float range_min, float range_max, int bin_count, uint64_t* counts_dev,
                     void* stream_v) {
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (n_tiles == 0) return AVR_OK;
  const float inverse_width = 1.0f / (range_max - range_min);
  unsigned long long* counts = reinterpret_cast<unsigned long long*>(counts_dev);
  const uint32_t groups = (n_tiles + kTilesPerGroup - 1) / kTilesPerGroup;
  const bool simple = !fc.log_scale && fc.normalize && range_min <= range_max;
  const size_t lds = (bin_count <= kMaxLdsBins) ? static_cast<size_t>(bin_count) * 4 : 0;
#define AVR_HISTOGRAM(LDS, SIMPLE)                                                              \
  hipLaunchKernelGGL((histogram_kernel<LDS, SIMPLE>), dim3(groups), dim3(kThreads), lds, stream, \
                     fc, boxes_dev, tile_begin_dev, n_boxes, n_tiles, range_min, range_max,      \
                     inverse_width, bin_count, counts)
  if (bin_count <= kMaxLdsBins) {
    if (simple) AVR_HISTOGRAM(true, true); else AVR_HISTOGRAM(true, false);
  } else {
    if (simple) AVR_HISTOGRAM(false, true); else AVR_HISTOGRAM(false, false);
  }
#undef AVR_HISTOGRAM
  return check("histogram_kernel");
}

}  // namespace avr
