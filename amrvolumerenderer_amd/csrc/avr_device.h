// Device helpers shared by the kernels of the ray-march path (avr_kernels.hip) and of the
// scene statistics (avr_scene_stats.hip).  Included inside translation units only.
#ifndef AVR_DEVICE_H
#define AVR_DEVICE_H

#include <hip/hip_runtime.h>

#include "avr_internal.h"

namespace avr {
namespace {

// applyScalarTransform (Common/VolumeTypes.hpp:33-67), double arithmetic.
__device__ __forceinline__ float apply_scalar_transform(double raw, const FrameConsts& fc) {
  double v = __builtin_isfinite(raw) ? raw : 0.0;
  if (fc.log_scale) {
    if (!(v > 0.0)) {
      v = fc.positive_floor;
    } else if (v < fc.positive_floor) {
      v = fc.positive_floor;
    }
    v = log(v);
  }
  if (fc.normalize) {
    v = (v - fc.norm_min) * fc.inv_norm_span;
    if (v < 0.0) {
      v = 0.0;
    } else if (v > 1.0) {
      v = 1.0;
    }
  }
  return static_cast<float>(v);
}

// saturateSoftTail (Common/VolumePainter.cpp:75-105) with rolloffEnd = 1.
__device__ __forceinline__ float saturate_soft_tail(float value, float clip_start) {
  const float clamped_end = (clip_start < 1.0f) ? 1.0f : clip_start;
  float cv = value;
  if (cv < 0.0f) {
    cv = 0.0f;
  } else if (cv > clamped_end) {
    cv = clamped_end;
  }
  if (!(clamped_end > clip_start + 1e-5f)) return cv;
  if (!(cv > clip_start)) return cv;
  if (!(cv < clamped_end)) return clamped_end;
  const float n = (cv - clip_start) / (clamped_end - clip_start);
  const float smooth = n + n * n - n * n * n;
  return clip_start + (clamped_end - clip_start) * smooth;
}

// Transfer-function table index of one cell value.
//   SIMPLE = the standard API path: no log scaling, normalise on, no soft
//   clip, scalarRange {0,1}.  There scalar = float(clamp_d((v - min) * inv, 0, 1)) and
//   normalized = (scalar - 0) * 1 = scalar; clamping after the float cast gives the same value
//   (rounding is monotone and 0, 1 are exact; only the sign of a zero can differ, which cannot
//   change int(scalar * 255)), and int(scalar * 255) is already in [0, 255].
template <bool SIMPLE>
__device__ __forceinline__ int table_index(double raw, const FrameConsts& fc) {
  if (SIMPLE) {
    double v = __builtin_isfinite(raw) ? raw : 0.0;
    v = (v - fc.norm_min) * fc.inv_norm_span;
    const float scalar = __builtin_amdgcn_fmed3f(static_cast<float>(v), 0.0f, 1.0f);
    return static_cast<int>(scalar * 255.0f);
  }
  float scalar = apply_scalar_transform(raw, fc);
  if (fc.apply_clip) scalar = saturate_soft_tail(scalar, fc.clip_start);
  float normalized = (scalar - fc.range_min) * fc.inverse_range;
  normalized = (normalized < 0.0f) ? 0.0f : normalized;
  normalized = (normalized > 1.0f) ? 1.0f : normalized;
  int idx = static_cast<int>(normalized * 255.0f);
  idx = (idx < 0) ? 0 : idx;
  idx = (idx > kTableSize - 1) ? (kTableSize - 1) : idx;
  return idx;
}

// Table indices of two cells, index(a) in byte 0 and index(b) in byte 1.  On the SIMPLE path the
// non-finite test selects the (wave-uniform) scaled value of 0.0 after the arithmetic instead of
// replacing the f64 operand before it: the same value, one select fewer per cell.
// (v_cvt_pk_u8_f32 would convert and place the byte in one instruction, but it rounds to nearest
// where the reference's int cast truncates: tools/ubench/cvt_pk_u8.hip.)
template <bool SIMPLE>
__device__ __forceinline__ uint32_t table_index_pair(double a, double b, const FrameConsts& fc) {
  if (SIMPLE) {
    const float at_zero =
        __builtin_amdgcn_fmed3f(static_cast<float>((0.0 - fc.norm_min) * fc.inv_norm_span), 0.0f,
                                1.0f) * 255.0f;
    float sa = __builtin_amdgcn_fmed3f(static_cast<float>((a - fc.norm_min) * fc.inv_norm_span),
                                       0.0f, 1.0f) * 255.0f;
    float sb = __builtin_amdgcn_fmed3f(static_cast<float>((b - fc.norm_min) * fc.inv_norm_span),
                                       0.0f, 1.0f) * 255.0f;
    sa = __builtin_isfinite(a) ? sa : at_zero;
    sb = __builtin_isfinite(b) ? sb : at_zero;
    return static_cast<uint32_t>(static_cast<int>(sa)) |
           (static_cast<uint32_t>(static_cast<int>(sb)) << 8);
  }
  return static_cast<uint32_t>(table_index<false>(a, fc)) |
         (static_cast<uint32_t>(table_index<false>(b, fc)) << 8);
}

}  // namespace
}  // namespace avr

#endif
