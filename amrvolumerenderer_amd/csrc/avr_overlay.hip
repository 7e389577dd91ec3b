// Wireframe overlay of the scene's tight bounds:
// renderBoundingBoxLayer, VolumeRenderer/VolumeRenderer.cpp:139-335.
//
// The reference walks the 12 edges in order and, for every pixel of an edge's screen
// rectangle, blends a white sample with an analytic coverage over the image.  Pixels are
// independent, so here one thread owns a pixel and applies the edges that reach it in the same
// order -- the per-pixel arithmetic and blend sequence are unchanged.  The projection of the 8
// corners and the edge rectangles are computed on the host (avr_host.cpp, plan_overlay).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "avr_internal.h"

namespace avr {

namespace {

__device__ __forceinline__ float clamp01(float v) {  // std::clamp(v, 0.0f, 1.0f)
  return (v < 0.0f) ? 0.0f : ((1.0f < v) ? 1.0f : v);
}

__device__ __forceinline__ uint32_t as_byte(float c) {  // Color::GetComponentAsByte
  const int tv = static_cast<int>(c * 256.f);
  return static_cast<uint32_t>((tv < 0) ? 0 : (tv > 255) ? 255 : tv);
}

// image: pixels [pixel_begin, pixel_end) of a width x height depth-sort image (5 floats each).
__global__ void overlay_kernel(const OverlayPlan plan, const int width, const int64_t pixel_begin,
                               const int64_t pixel_end, const PieceMapDev pieces, const int piece,
                               float* __restrict__ image, uint8_t* __restrict__ rgb8) {
  const int64_t n = pixel_end - pixel_begin;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q < n;
       q += stride) {
    const int64_t p = pixel_begin + q;
    const int px = static_cast<int>(p % width);
    int py = static_cast<int>(p / width);
    // a piece of row bands holds its rows in order: piece row -> image row
    if (pieces.layout == kPiecesRowBands) py = image_row_of(pieces, piece, py);
    float* buffer = image + q * 5;
    float r = buffer[0], g = buffer[1], b = buffer[2], a = buffer[3], depth = buffer[4];
    bool touched = false;
    const float sample_x = static_cast<float>(px) + 0.5f;
    const float sample_y = static_cast<float>(py) + 0.5f;
    for (int e = 0; e < plan.n_edges; ++e) {
      const OverlayEdge& edge = plan.edges[e];
      if (px < edge.x_begin || px > edge.x_end || py < edge.y_begin || py > edge.y_end) continue;
      float coverage;
      if (edge.point) {
        coverage = 1.0f;  // degenerate edge: one sample at the rounded corner (:281-288)
      } else {
        const float apx = sample_x - edge.sx;
        const float apy = sample_y - edge.sy;
        float t = (apx * edge.dx + apy * edge.dy) / edge.len_sq;
        t = clamp01(t);
        const float closest_x = edge.sx + (edge.ex - edge.sx) * t;
        const float closest_y = edge.sy + (edge.ey - edge.sy) * t;
        const float dist_x = sample_x - closest_x;
        const float dist_y = sample_y - closest_y;
        const float distance = sqrtf(dist_x * dist_x + dist_y * dist_y);
        coverage = clamp01((plan.pixel_radius + 0.5f - distance) * 0.6f);
      }
      if (coverage <= 0.0f) continue;
      // blendSample (:236-258): white line colour, premultiplied
      const float src_alpha = coverage;
      const float src = 1.0f * src_alpha;
      r = src + r * (1.0f - src_alpha);
      g = src + g * (1.0f - src_alpha);
      b = src + b * (1.0f - src_alpha);
      a = src_alpha + a * (1.0f - src_alpha);
      depth = -3.402823466e+38f;  // numeric_limits<float>::lowest()
      touched = true;
    }
    if (touched) {
      buffer[0] = r;
      buffer[1] = g;
      buffer[2] = b;
      buffer[3] = a;
      buffer[4] = depth;
    }
    if (rgb8 != nullptr) {
      uint8_t* out = rgb8 + q * 3;
      out[0] = static_cast<uint8_t>(as_byte(r));
      out[1] = static_cast<uint8_t>(as_byte(g));
      out[2] = static_cast<uint8_t>(as_byte(b));
    }
  }
}

}  // namespace

int launch_overlay(const OverlayPlan& plan, int width, int64_t pixel_begin, int64_t pixel_end,
                   const PieceMapDev* pieces, int piece, float* image, uint8_t* rgb8,
                   void* stream_v) {
  const int64_t n = pixel_end - pixel_begin;
  if (n <= 0) return AVR_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(overlay_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_v), plan, width, pixel_begin, pixel_end,
                     pieces != nullptr ? *pieces : PieceMapDev{}, piece, image, rgb8);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("overlay_kernel: ") + hipGetErrorString(err));
    return AVR_ERR_RUNTIME;
  }
  return AVR_OK;
}

}  // namespace avr
