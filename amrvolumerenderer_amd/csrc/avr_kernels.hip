// HIP kernels of the MI355X (gfx950 / CDNA4) volume-rendering hot path.
//
// Built with -ffp-contract=off: every float operation below is a single IEEE binary32
// operation in source order, which is what the reference's CPU build executes
// ("no FMA contraction").  Division and sqrt are the correctly rounded
// forms (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
//
// Kernels:
//   classify_kernel     streams every f64 cell of the frame's boxes once and stores its
//                       transfer-function table index as a byte in 8x4x4 bricklets
//                       (the per-cell part of VolumePainter.cpp:870-883)
//   render_runs_kernel  one workgroup per (16x16-pixel tile, run), one thread per pixel: marches
//                       the run's boxes in global layer order, folds them with the depth-sort
//                       blend in registers and stores the run's layer in DirectSend "send layout"
//                       (VolumePainter.cpp:735-955 + VolumeRenderer.cpp:1201-1219 +
//                        DirectSendBase.cpp:413-426)
//   fold_plan_kernel    receiver side: blends the received run blocks of a pixel piece in global
//                       order, optional RGB8 (DirectSendBase.cpp:400-446, Color.hpp:66-91)
//   blend_*             Features::blend of the three image types (element-wise)
//   fold_runs_kernel    left fold over dense run layers
//   downsample / quantize / encode / decode   frame tail
//   upload_kernel       per-call descriptors, pinned host block -> HBM
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>

#include "avr_internal.h"
#include "avr_device.h"

namespace avr {

namespace {

constexpr int kBlockThreads = 256;  // workgroup = kTile x kTile pixels = 4 waves of 8x8
// Samples per trip of the march's interior loop: a deep loop first (that many loads in flight per
// wave, that many samples' registers), then the loop of four.  Eight: config-4's march alone 0.657
// -> 0.603 ms and the pipelined frame 0.931 -> 0.874 (round 5, after the side-by-side counters showed
// that a SIMD's wave slots bind, not its registers: profiles/r5_corun_counters/); 6: 0.618 / 0.895,
// 10: 0.630 / 0.888, 12: 0.633 / 0.884, 16: 0.712 / 0.919, 4 (no deep loop): 0.657 / 0.931.
// Power-of-two spacings (every BASELINE configuration) and reciprocal ones alike; degenerate
// spacings (the exact divide throughout) take the loop of four.
#ifndef AVR_MARCH_GROUP
#define AVR_MARCH_GROUP 8
#endif

#define AVR_INF __builtin_huge_valf()

struct Layer5 {
  float r, g, b, a, d;
};

// Morton decode of the even bits.
__device__ __forceinline__ unsigned compact_bits(unsigned v) {
  v &= 0x55555555u;
  v = (v | (v >> 1)) & 0x33333333u;
  v = (v | (v >> 2)) & 0x0f0f0f0fu;
  v = (v | (v >> 4)) & 0x00ff00ffu;
  v = (v | (v >> 8)) & 0x0000ffffu;
  return v;
}

// Features::blend of ImageRGBAFloatColorDepthSort (ImageRGBAFloatColorDepthSort.hpp:13-27).
__device__ __forceinline__ Layer5 blend_depthsort(const Layer5& top, const Layer5& bottom) {
  const bool top_is_front = top.d <= bottom.d;
  const Layer5 front = top_is_front ? top : bottom;
  const Layer5 back = top_is_front ? bottom : top;
  const float t = 1.0f - front.a;
  Layer5 out;
  out.r = front.r + back.r * t;
  out.g = front.g + back.g * t;
  out.b = front.b + back.b * t;
  out.a = front.a + back.a * t;
  out.d = (bottom.d < top.d) ? bottom.d : top.d;  // std::min(topDepth, bottomDepth)
  return out;
}

struct Ray {
  float ox, oy, oz;
  float dx, dy, dz;
};

// Ray of pixel (px, py): VolumePainter.cpp:741-766.
__device__ __forceinline__ Ray make_ray(int px, int py, const FrameConsts& fc) {
  const float ndc_x = (static_cast<float>(px) + 0.5f) * fc.inv_width * 2.0f - 1.0f;
  const float ndc_y = (static_cast<float>(py) + 0.5f) * fc.inv_height * 2.0f - 1.0f;
  const float plane_x = ndc_x * fc.tan_half_fov * fc.aspect;
  const float plane_y = ndc_y * fc.tan_half_fov;
  Ray ray;
  ray.dx = fc.fwd[0] + plane_x * fc.right[0] + plane_y * fc.up[0];
  ray.dy = fc.fwd[1] + plane_x * fc.right[1] + plane_y * fc.up[1];
  ray.dz = fc.fwd[2] + plane_x * fc.right[2] + plane_y * fc.up[2];
  const float len_sq = ray.dx * ray.dx + ray.dy * ray.dy + ray.dz * ray.dz;
  // host amrex::Math::rsqrt(x) is 1/sqrt(x): length = 1 / (1 / sqrt)
  const float len = (len_sq > 0.0f) ? (1.0f / (1.0f / sqrtf(len_sq))) : 0.0f;
  if (len > 0.0f) {
    const float inv = 1.0f / len;
    ray.dx *= inv;
    ray.dy *= inv;
    ray.dz *= inv;
  }
  ray.ox = fc.eye[0];
  ray.oy = fc.eye[1];
  ray.oz = fc.eye[2];
  return ray;
}

// One axis of the slab test (updateBounds, VolumePainter.cpp:775-796).  inv_dir = 1/direction
// is the same value for every box, so it is computed once per pixel.
__device__ __forceinline__ void slab_axis(float origin, float direction, float inv_dir,
                                          float min_val, float max_val, float& tmin,
                                          float& tmax) {
  if (fabsf(direction) < 1e-8f) {
    if (origin < min_val || origin > max_val) {
      tmin = AVR_INF;
      tmax = -AVR_INF;
    }
    return;
  }
  float t1 = (min_val - origin) * inv_dir;
  float t2 = (max_val - origin) * inv_dir;
  if (t1 > t2) {
    const float tmp = t1;
    t1 = t2;
    t2 = tmp;
  }
  tmin = (tmin > t1) ? tmin : t1;
  tmax = (tmax < t2) ? tmax : t2;
}

// Byte offset of cell (i, j, k) inside a box's classified volume: 128-byte bricklets of
// 8 x 4 x 4 cells (x fastest inside), so that the cells a bundle of neighbouring rays touches over
// several steps share cache lines in all three directions (an x-fastest row would only help
// along x: measured 2x slower).  The bricklets themselves are ordered z fastest, then y, then x:
//   offset = (i&7) + 8*(j&3) + 32*(k&3) + 128*((k>>2) + bz*((j>>2) + by*(i>>3)))
// and because a bricklet is 4 cells of 32 bytes deep in z, 32*(k&3) + 128*(k>>2) = 32*k: the z
// term needs no split at all --
//   offset = i + 8*j + 32*k + y_pitch*(j>>2) + x_pitch*(i>>3)
// with y_pitch = 128*bz - 32 and x_pitch = 128*bz*by - 8, both < 2^24 (host check), i.e. two
// shifts, two shift-adds and two 24-bit multiply-adds (the march is bound by the number of
// vector instructions; an x-fastest brick order needs two more).
__device__ __forceinline__ uint32_t bricklet_offset(int i, int j, int k, uint32_t y_pitch,
                                                    uint32_t x_pitch) {
  const uint32_t ui = static_cast<uint32_t>(i), uj = static_cast<uint32_t>(j),
                 uk = static_cast<uint32_t>(k);
  uint32_t offset;
  asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(offset) : "v"(uj), "v"(ui));
  asm("v_lshl_add_u32 %0, %1, 5, %0" : "+v"(offset) : "v"(uk));
  asm("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(offset) : "v"(uj >> 2), "s"(y_pitch));
  asm("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(offset) : "v"(ui >> 3), "s"(x_pitch));
  return offset;
}

// Cell indices of an inside sample: the reference's clamp(int(floor((pos - min) / d)), 0, n - 1)
// per axis (VolumePainter.cpp:846-867).  `fx` = pos - min >= 0 for an inside sample, so
// truncation equals floor and only the upper clamp can bind on the multiply paths; the
// exact-divide path restates the reference literally.
// (qx, qy, qz) = (pos - min) * (1 / spacing), already computed by the caller.  CLAMP = false:
// the caller knows the truncated quotients are <= n - 1 (see the interior loop of march_box).
// Two floats handled by one packed instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: each
// half is an ordinary IEEE binary32 operation, so results equal the scalar expressions bit for bit).
typedef float float_pair __attribute__((ext_vector_type(2)));

template <int MODE, bool CLAMP, bool STATS>
__device__ __forceinline__ uint32_t offset_from_quotients(const BoxDev& box, uint32_t row_pitch,
                                                          uint32_t plane_pitch, float qx, float qy,
                                                          float qz, float fx, float fy, float fz,
                                                          unsigned& near_hits) {
  int i = static_cast<int>(qx);
  int j = static_cast<int>(qy);
  int k = static_cast<int>(qz);
  if (CLAMP) {
    i = (i > box.nx - 1) ? box.nx - 1 : i;
    j = (j > box.ny - 1) ? box.ny - 1 : j;
    k = (k > box.nz - 1) ? box.nz - 1 : k;
  }
  if (MODE == kReciprocal) {
    // q is within 2^-22 * q of the correctly rounded quotient; the floors can differ only
    // when q sits this close to an integer ("Exact index without the divide")
    const float ex = fabsf(qx - rintf(qx));
    const float ey = fabsf(qy - rintf(qy));
    const float ez = fabsf(qz - rintf(qz));
    if (fminf(fminf(ex, ey), ez) <= box.near_tol) {
      if (STATS) ++near_hits;
      int ei = static_cast<int>(floorf(fx / box.dx));
      int ej = static_cast<int>(floorf(fy / box.dy));
      int ek = static_cast<int>(floorf(fz / box.dz));
      i = (ei < 0) ? 0 : ((ei >= box.nx) ? box.nx - 1 : ei);
      j = (ej < 0) ? 0 : ((ej >= box.ny) ? box.ny - 1 : ej);
      k = (ek < 0) ? 0 : ((ek >= box.nz) ? box.nz - 1 : ek);
    }
  }
  return bricklet_offset(i, j, k, row_pitch, plane_pitch);
}

template <int MODE, bool STATS>
__device__ __forceinline__ uint32_t cell_offset(const BoxDev& box, uint32_t row_pitch,
                                                uint32_t plane_pitch, float fx, float fy,
                                                float fz, unsigned& near_hits) {
  if (MODE == kExactDivide) {  // the reference, literally
    int ei = static_cast<int>(floorf(fx / box.dx));
    int ej = static_cast<int>(floorf(fy / box.dy));
    int ek = static_cast<int>(floorf(fz / box.dz));
    ei = (ei < 0) ? 0 : ((ei >= box.nx) ? box.nx - 1 : ei);
    ej = (ej < 0) ? 0 : ((ej >= box.ny) ? box.ny - 1 : ej);
    ek = (ek < 0) ? 0 : ((ek >= box.nz) ? box.nz - 1 : ek);
    return bricklet_offset(ei, ej, ek, row_pitch, plane_pitch);
  }
  return offset_from_quotients<MODE, true, STATS>(box, row_pitch, plane_pitch, fx * box.inv_dx,
                                                  fy * box.inv_dy, fz * box.inv_dz, fx, fy, fz,
                                                  near_hits);
}

// The march of one ray through one box (VolumePainter.cpp:811-921 + host epilogue :939-955).
// Returns the layer pixel the reference would store for this box.
template <bool STATS, int MODE>
__device__ __forceinline__ Layer5 march_box(const BoxDev& box, const FrameConsts& fc,
                                            const uint8_t* __restrict__ classified,
                                            const float4* __restrict__ table, const Ray& ray,
                                            float tmin, float tmax, unsigned& fetches,
                                            unsigned& near_hits) {
  const float min_x = box.minc[0], min_y = box.minc[1], min_z = box.minc[2];
  const float max_x = box.maxc[0], max_y = box.maxc[1], max_z = box.maxc[2];
  const float step = box.sample_dist;
  const uint8_t __attribute__((address_space(1)))* cells =
      (const uint8_t __attribute__((address_space(1)))*)(classified + box.cls_offset);
  const uint32_t bricks_z = static_cast<uint32_t>(box.nz + kBrickZ - 1) >> 2;
  const uint32_t bricks_y = static_cast<uint32_t>(box.ny + kBrickY - 1) >> 2;
  const uint32_t row_pitch = bricks_z * kBrickBytes - 32u;              // y_pitch of bricklet_offset
  const uint32_t plane_pitch = bricks_y * bricks_z * kBrickBytes - 8u;  // x_pitch

  float distance = tmin + box.mesh_eps;
  if (distance < 0.0f) distance = box.mesh_eps;

  float acc_r = 0.0f, acc_g = 0.0f, acc_b = 0.0f, acc_a = 0.0f;

#define AVR_INSIDE(x, y, z) \
  (!((x) < min_x || (x) > max_x || (y) < min_y || (y) > max_y || (z) < min_z || (z) > max_z))
#define AVR_ACCUMULATE(sample)                          \
  do {                                                  \
    const float alpha_ = (sample).w * (1.0f - acc_a);   \
    acc_r += (sample).x * alpha_;                       \
    acc_g += (sample).y * alpha_;                       \
    acc_b += (sample).z * alpha_;                       \
    acc_a += alpha_;                                    \
  } while (0)

  // ---- interior steps without the per-sample inside test ------------------------------------
  // pos(d) = fl(o + fl(dir * d)) is monotone in d on every axis (products and sums of floats
  // round monotonically), so if the positions at `distance` and at `safe_end` are both inside
  // the box, every sampled position with distance <= d < safe_end is inside as well and the
  // reference's inside test (:821-825, :838) is known to pass.  Steps from safe_end on (and
  // whole rays whose end points fail the test, e.g. grazing rays) take the general loop below.
  const float safe_end = tmax - step;
  bool interior = false;
  if (distance < safe_end) {
    const float ax = ray.ox + ray.dx * distance, ay = ray.oy + ray.dy * distance,
                az = ray.oz + ray.dz * distance;
    const float bx = ray.ox + ray.dx * safe_end, by = ray.oy + ray.dy * safe_end,
                bz = ray.oz + ray.dz * safe_end;
    interior = AVR_INSIDE(ax, ay, az) && AVR_INSIDE(bx, by, bz);
    if (MODE != kExactDivide) {
      // (pos - min) * inv_d is monotone in pos as well, so the truncated quotients of all
      // interior samples lie between those of the two end points: if both are <= n - 1 the
      // upper clamp (:846-867) never binds inside the loop.  (A ray whose entry point rounds
      // onto the max face takes the general loop.)
      const int ia = static_cast<int>((ax - min_x) * box.inv_dx), ib = static_cast<int>((bx - min_x) * box.inv_dx);
      const int ja = static_cast<int>((ay - min_y) * box.inv_dy), jb = static_cast<int>((by - min_y) * box.inv_dy);
      const int ka = static_cast<int>((az - min_z) * box.inv_dz), kb = static_cast<int>((bz - min_z) * box.inv_dz);
      interior = interior && ia < box.nx && ib < box.nx && ja < box.ny && jb < box.ny &&
                 ka < box.nz && kb < box.nz;
    }
  }
  if (interior) {
    // Four steps per trip while all four lie below safe_end: the four cell bytes and the four
    // table entries are requested before the first is consumed (four memory round trips in
    // flight per wave).  Remaining steps below safe_end fall through to the general loop.
    //
    // Everything per sample is written with the full-rate vector instructions of gfx950
    // (v_mul / v_add / v_sub / v_fma_f32, 2.25 cycles per wave instruction measured with
    // tools/ubench/valu_rates; v_pk_*_f32, conversions, shifts-left, multiply-adds and compares
    // take 4.15): the march is bound by vector issue, so the sum of those costs is its time.
    const float inv_x = box.inv_dx, inv_y = box.inv_dy, inv_z = box.inv_dz;
    // loop-invariant operand pairs of the packed fused quotient (kept in registers: a packed
    // instruction reads at most one scalar-register operand)
    const float_pair ix2 = {inv_x, inv_x}, iy2 = {inv_y, inv_y}, iz2 = {inv_z, inv_z};
    float_pair nx2 = {box.nmin_inv[0], box.nmin_inv[0]}, ny2 = {box.nmin_inv[1], box.nmin_inv[1]},
               nz2 = {box.nmin_inv[2], box.nmin_inv[2]};
    if (MODE == kPow2Multiply) asm volatile("" : "+v"(nx2), "+v"(ny2), "+v"(nz2));
#if AVR_MARCH_GROUP > 4
    // kDeep (8, 12 ...) steps per trip while all of them lie below safe_end: several groups' worth
    // of cell bytes and table entries in flight per wave.  Side by side with the classify pass a
    // SIMD's eight wave slots bind, not its registers, and a wave's throughput is one memory round
    // trip per trip of this loop (profiles/r5_corun_counters/): more loads per trip is what a wave
    // can still give.  Every sample is accumulated by the same operations in the same order as
    // below; what is left takes the loop of four, then the general loop.
    if (MODE == kPow2Multiply || MODE == kReciprocal) {
      constexpr int kDeep = AVR_MARCH_GROUP;
      static_assert(kDeep % 2 == 0, "the positions are computed two at a time");
      for (;;) {
        float d[kDeep];
        d[0] = distance;
#pragma unroll
        for (int i = 1; i < kDeep; ++i) d[i] = d[i - 1] + step;
        if (!(d[kDeep - 1] < safe_end)) break;
        uint32_t off[kDeep];
#pragma unroll
        for (int pair = 0; pair < kDeep / 2; ++pair) {
          const float_pair dd = {d[2 * pair], d[2 * pair + 1]};
          if (MODE == kPow2Multiply) {
            const float_pair qx = __builtin_elementwise_fma(ray.ox + ray.dx * dd, ix2, nx2);
            const float_pair qy = __builtin_elementwise_fma(ray.oy + ray.dy * dd, iy2, ny2);
            const float_pair qz = __builtin_elementwise_fma(ray.oz + ray.dz * dd, iz2, nz2);
            off[2 * pair] = bricklet_offset(static_cast<int>(qx.x), static_cast<int>(qy.x),
                                            static_cast<int>(qz.x), row_pitch, plane_pitch);
            off[2 * pair + 1] = bricklet_offset(static_cast<int>(qx.y), static_cast<int>(qy.y),
                                                static_cast<int>(qz.y), row_pitch, plane_pitch);
          } else {
            // the reference's two roundings, (pos - min) then * RN(1/d), as in the loop of four
            const float_pair fx = (ray.ox + ray.dx * dd) - min_x;
            const float_pair fy = (ray.oy + ray.dy * dd) - min_y;
            const float_pair fz = (ray.oz + ray.dz * dd) - min_z;
            const float_pair qx = fx * inv_x, qy = fy * inv_y, qz = fz * inv_z;
            off[2 * pair] = offset_from_quotients<MODE, false, STATS>(
                box, row_pitch, plane_pitch, qx.x, qy.x, qz.x, fx.x, fy.x, fz.x, near_hits);
            off[2 * pair + 1] = offset_from_quotients<MODE, false, STATS>(
                box, row_pitch, plane_pitch, qx.y, qy.y, qz.y, fx.y, fy.y, fz.y, near_hits);
          }
        }
        int idx[kDeep];
#pragma unroll
        for (int i = 0; i < kDeep; ++i) idx[i] = cells[off[i]];
        float4 sample[kDeep];
#pragma unroll
        for (int i = 0; i < kDeep; ++i) sample[i] = table[idx[i]];
        float_pair rg = {acc_r, acc_g}, ba = {acc_b, acc_a};
#pragma unroll
        for (int i = 0; i < kDeep; ++i) {
          const float w_ = sample[i].w * (1.0f - ba.y);
          const float_pair color_ = {sample[i].x, sample[i].y};
          const float_pair zw_ = {sample[i].z * w_, w_};
          float_pair weighted_;
          asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]"
              : "=v"(weighted_)
              : "v"(color_), "v"(zw_));
          ba = ba + zw_;
          rg = rg + weighted_;
          asm volatile("" : "+v"(rg));
        }
        if (__builtin_amdgcn_ballot_w64(ba.y >= 1.0f) == 0) {
          acc_r = rg.x;
          acc_g = rg.y;
          acc_b = ba.x;
          acc_a = ba.y;
          distance = d[kDeep - 1] + step;
          if (STATS) fetches += static_cast<unsigned>(kDeep);
        } else {
          // some lane's ray terminates inside this trip: per-sample selects, as in the loop of four
          // (sample k + 1 only if accumA < 1 still holds after sample k; the loop's own condition
          // held before the first)
          unsigned taken = 0u;
          float next = distance;
#pragma unroll
          for (int i = 0; i < kDeep; ++i) {
            const bool running_ = acc_a < 1.0f;
            const float alpha_ = sample[i].w * (1.0f - acc_a);
            const float r_ = acc_r + sample[i].x * alpha_;
            const float g_ = acc_g + sample[i].y * alpha_;
            const float b_ = acc_b + sample[i].z * alpha_;
            const float a_ = acc_a + alpha_;
            acc_r = running_ ? r_ : acc_r;
            acc_g = running_ ? g_ : acc_g;
            acc_b = running_ ? b_ : acc_b;
            acc_a = running_ ? a_ : acc_a;
            next = running_ ? (d[i] + step) : next;
            taken += running_ ? 1u : 0u;
          }
          distance = next;
          if (STATS) fetches += taken;
          if (!(acc_a < 1.0f)) break;  // the reference's loop condition (:837)
        }
      }
    }
#endif
    for (;;) {
      const float d1 = distance;
      const float d2 = d1 + step;
      const float d3 = d2 + step;
      const float d4 = d3 + step;
      if (!(d4 < safe_end)) break;
      uint32_t off1, off2, off3, off4;
      if (MODE == kExactDivide) {
        off1 = cell_offset<MODE, STATS>(box, row_pitch, plane_pitch,
                                        (ray.ox + ray.dx * d1) - min_x,
                                        (ray.oy + ray.dy * d1) - min_y,
                                        (ray.oz + ray.dz * d1) - min_z, near_hits);
        off2 = cell_offset<MODE, STATS>(box, row_pitch, plane_pitch,
                                        (ray.ox + ray.dx * d2) - min_x,
                                        (ray.oy + ray.dy * d2) - min_y,
                                        (ray.oz + ray.dz * d2) - min_z, near_hits);
        off3 = cell_offset<MODE, STATS>(box, row_pitch, plane_pitch,
                                        (ray.ox + ray.dx * d3) - min_x,
                                        (ray.oy + ray.dy * d3) - min_y,
                                        (ray.oz + ray.dz * d3) - min_z, near_hits);
        off4 = cell_offset<MODE, STATS>(box, row_pitch, plane_pitch,
                                        (ray.ox + ray.dx * d4) - min_x,
                                        (ray.oy + ray.dy * d4) - min_y,
                                        (ray.oz + ray.dz * d4) - min_z, near_hits);
      } else if (MODE == kPow2Multiply) {
        // q = (pos - min) * 2^k as ONE fused operation: the exact value pos * 2^k - min * 2^k is
        // rounded once, which is RN(pos - min) * 2^k (see fold_is_exact in avr_host.cpp).  Two
        // samples per packed instruction (v_pk_mul / v_pk_add / v_pk_fma_f32: each half is an
        // ordinary IEEE operation): the march is bound by the NUMBER of vector instructions.
        const float_pair d12 = {d1, d2}, d34 = {d3, d4};
        const float_pair qx12 = __builtin_elementwise_fma(ray.ox + ray.dx * d12, ix2, nx2);
        const float_pair qy12 = __builtin_elementwise_fma(ray.oy + ray.dy * d12, iy2, ny2);
        const float_pair qz12 = __builtin_elementwise_fma(ray.oz + ray.dz * d12, iz2, nz2);
        const float_pair qx34 = __builtin_elementwise_fma(ray.ox + ray.dx * d34, ix2, nx2);
        const float_pair qy34 = __builtin_elementwise_fma(ray.oy + ray.dy * d34, iy2, ny2);
        const float_pair qz34 = __builtin_elementwise_fma(ray.oz + ray.dz * d34, iz2, nz2);
        off1 = bricklet_offset(static_cast<int>(qx12.x), static_cast<int>(qy12.x),
                               static_cast<int>(qz12.x), row_pitch, plane_pitch);
        off2 = bricklet_offset(static_cast<int>(qx12.y), static_cast<int>(qy12.y),
                               static_cast<int>(qz12.y), row_pitch, plane_pitch);
        off3 = bricklet_offset(static_cast<int>(qx34.x), static_cast<int>(qy34.x),
                               static_cast<int>(qz34.x), row_pitch, plane_pitch);
        off4 = bricklet_offset(static_cast<int>(qx34.y), static_cast<int>(qy34.y),
                               static_cast<int>(qz34.y), row_pitch, plane_pitch);
      } else {
        // the reference's two roundings, (pos - min) then * RN(1/d), two samples per instruction
        const float_pair d12 = {d1, d2}, d34 = {d3, d4};
        const float_pair fx12 = (ray.ox + ray.dx * d12) - min_x, fx34 = (ray.ox + ray.dx * d34) - min_x;
        const float_pair fy12 = (ray.oy + ray.dy * d12) - min_y, fy34 = (ray.oy + ray.dy * d34) - min_y;
        const float_pair fz12 = (ray.oz + ray.dz * d12) - min_z, fz34 = (ray.oz + ray.dz * d34) - min_z;
        const float_pair qx12 = fx12 * inv_x, qx34 = fx34 * inv_x;
        const float_pair qy12 = fy12 * inv_y, qy34 = fy34 * inv_y;
        const float_pair qz12 = fz12 * inv_z, qz34 = fz34 * inv_z;
        off1 = offset_from_quotients<MODE, false, STATS>(box, row_pitch, plane_pitch, qx12.x,
                                                         qy12.x, qz12.x, fx12.x, fy12.x, fz12.x,
                                                         near_hits);
        off2 = offset_from_quotients<MODE, false, STATS>(box, row_pitch, plane_pitch, qx12.y,
                                                         qy12.y, qz12.y, fx12.y, fy12.y, fz12.y,
                                                         near_hits);
        off3 = offset_from_quotients<MODE, false, STATS>(box, row_pitch, plane_pitch, qx34.x,
                                                         qy34.x, qz34.x, fx34.x, fy34.x, fz34.x,
                                                         near_hits);
        off4 = offset_from_quotients<MODE, false, STATS>(box, row_pitch, plane_pitch, qx34.y,
                                                         qy34.y, qz34.y, fx34.y, fy34.y, fz34.y,
                                                         near_hits);
      }
      const int idx1 = cells[off1];
      const int idx2 = cells[off2];
      const int idx3 = cells[off3];
      const int idx4 = cells[off4];
      const float4 s1 = table[idx1];
      const float4 s2 = table[idx2];
      const float4 s3 = table[idx3];
      const float4 s4 = table[idx4];
      // accumA never decreases and never exceeds 1 (table alphas lie in [0, 1]:
      // a + s.w * (1 - a) <= 1 also after rounding), and once it is 1 it stays 1 (the weight is
      // then s.w * 0), so "some sample of this group found accumA >= 1" (the reference's loop exit,
      // :837) is exactly a4 >= 1: one comparison and one wave ballot per group.  All four samples
      // are accumulated unconditionally into copies, committed when no lane saturated.
      // (r, g) and (b, a) are accumulated as pairs: per sample  t = 1 - a;  w = s.w * t;
      // (r, g) += (s.x, s.y) * w;  (b, a) += (s.z * w, w)  -- six instructions, each half the
      // reference's own operation.
      float_pair rg = {acc_r, acc_g}, ba = {acc_b, acc_a};
      // (the weight w is the HIGH half of the pair (z * w, w) that is added to (b, a); the packed
      // multiply of (x, y) takes it from there -- op_sel: both halves of the second operand read
      // its high half -- instead of from a copy in the low half of another register pair: one
      // v_mov_b32 per sample less, 4 % of the march's vector instructions)
#define AVR_STEP(sample)                                                          \
      {                                                                           \
        const float w_ = (sample).w * (1.0f - ba.y);                              \
        const float_pair color_ = {(sample).x, (sample).y};                       \
        const float_pair zw_ = {(sample).z * w_, w_};                             \
        float_pair weighted_;                                                     \
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]"                \
            : "=v"(weighted_)                                                     \
            : "v"(color_), "v"(zw_));                                             \
        ba = ba + zw_;                                                            \
        rg = rg + weighted_;                                                      \
        asm volatile("" : "+v"(rg));                                              \
      }
      AVR_STEP(s1)
      AVR_STEP(s2)
      AVR_STEP(s3)
      AVR_STEP(s4)
#undef AVR_STEP
      if (__builtin_amdgcn_ballot_w64(ba.y >= 1.0f) == 0) {
        // the normal case: no lane of the wave saturates inside this group
        acc_r = rg.x;
        acc_g = rg.y;
        acc_b = ba.x;
        acc_a = ba.y;
        distance = d4 + step;
        if (STATS) fetches += 4u;
      } else {
        // some lane's ray terminates inside this group: per-sample selects (sample k + 1 is
        // accumulated only if the reference's loop condition accumA < 1 still holds after
        // sample k; done with selects, not branches, so that the loads above stay unconditional)
        unsigned taken = 1u;
        float next = d2;
        AVR_ACCUMULATE(s1);
#define AVR_ACCUMULATE_IF_RUNNING(sample, following)          \
        {                                                     \
          const bool running_ = acc_a < 1.0f;                 \
          const float alpha_ = (sample).w * (1.0f - acc_a);   \
          const float r_ = acc_r + (sample).x * alpha_;       \
          const float g_ = acc_g + (sample).y * alpha_;       \
          const float b_ = acc_b + (sample).z * alpha_;       \
          const float a_ = acc_a + alpha_;                    \
          acc_r = running_ ? r_ : acc_r;                      \
          acc_g = running_ ? g_ : acc_g;                      \
          acc_b = running_ ? b_ : acc_b;                      \
          acc_a = running_ ? a_ : acc_a;                      \
          next = running_ ? (following) : next;               \
          taken += running_ ? 1u : 0u;                        \
        }
        // once accumA reaches 1 it stays >= 1 (alpha >= 0), so later samples are rejected too
        AVR_ACCUMULATE_IF_RUNNING(s2, d3)
        AVR_ACCUMULATE_IF_RUNNING(s3, d4)
        AVR_ACCUMULATE_IF_RUNNING(s4, d4 + step)
#undef AVR_ACCUMULATE_IF_RUNNING
        distance = next;
        if (STATS) fetches += taken;
        if (!(acc_a < 1.0f)) break;  // the reference's loop condition (:837)
      }
    }
  }

  // ---- general loop --------------------------------------------------------------------------
  float pos_x = ray.ox + ray.dx * distance;
  float pos_y = ray.oy + ray.dy * distance;
  float pos_z = ray.oz + ray.dz * distance;

  // The reference's skip loop (:830-835) and the "continue" branch of the main loop (:838-844)
  // do the same thing -- advance without sampling while the position is outside -- so one loop
  // with an inside test reproduces both.
  while (distance < tmax && acc_a < 1.0f) {
    if (AVR_INSIDE(pos_x, pos_y, pos_z)) {
      const uint32_t offset = cell_offset<MODE, STATS>(box, row_pitch, plane_pitch, pos_x - min_x,
                                                       pos_y - min_y, pos_z - min_z, near_hits);
      // the cell's transfer-function table index, computed from the f64 cell value by the
      // classify pass of this frame (same arithmetic as VolumePainter.cpp:870-883)
      const int idx = cells[offset];
      if (STATS) ++fetches;
      const float4 sample = table[idx];
      AVR_ACCUMULATE(sample);
    }
    distance += step;
    pos_x = ray.ox + ray.dx * distance;
    pos_y = ray.oy + ray.dy * distance;
    pos_z = ray.oz + ray.dz * distance;
  }
#undef AVR_INSIDE
#undef AVR_ACCUMULATE

  // device-side clamp (:902-905) then the host epilogue's std::clamp to [0,1] (:944-947)
  acc_r = (acc_r > 1.0f) ? 1.0f : acc_r;
  acc_g = (acc_g > 1.0f) ? 1.0f : acc_g;
  acc_b = (acc_b > 1.0f) ? 1.0f : acc_b;
  acc_a = (acc_a > 1.0f) ? 1.0f : acc_a;
  Layer5 out;
  out.r = (acc_r < 0.0f) ? 0.0f : acc_r;
  out.g = (acc_g < 0.0f) ? 0.0f : acc_g;
  out.b = (acc_b < 0.0f) ? 0.0f : acc_b;
  out.a = (acc_a < 0.0f) ? 0.0f : acc_a;

  float depth = AVR_INF;
  if (acc_a > 0.0f) {  // entry depth along the view axis (:912-920)
    const float ex = ray.ox + ray.dx * tmin;
    const float ey = ray.oy + ray.dy * tmin;
    const float ez = ray.oz + ray.dz * tmin;
    depth = (ex - fc.eye[0]) * fc.fwd[0] + (ey - fc.eye[1]) * fc.fwd[1] +
            (ez - fc.eye[2]) * fc.fwd[2];
  }
  if (!__builtin_isfinite(depth) || out.a <= 0.0f) depth = AVR_INF;  // (:950-952)
  out.d = depth;
  return out;
}

// ONLY_MODE >= 0: every box of the launch uses that IndexMode (the common case: one scene, one
// kind of spacing), so only that march variant is compiled in; -1 dispatches per box.
// <= 80 SGPRs: 256-thread workgroups are admitted per CU up to floor(800 / (ceil(sgpr/16)*16 + 16))
// (MI355X_MICROARCH.md, "Residency"), i.e. 8 per CU only up to 80 SGPRs, 6 at 98+.
// SPEC: the speculative frame's bookkeeping is compiled in (one vector register more: six waves per
// SIMD instead of seven, so it is its own instantiation; never together with STATS).
template <bool STATS, int ONLY_MODE, bool SPEC>
__device__ __forceinline__ void
render_runs_body(
    const FrameConsts& fc, const BoxDev* __restrict__ boxes,
    const uint8_t* __restrict__ classified, const float* __restrict__ tables,
    const int n_tables, const int32_t* __restrict__ order, const int4* __restrict__ order_rects,
    const int32_t* __restrict__ run_end,
    const int n_runs, const int n_pieces, const RunRectDev* __restrict__ run_rects,
    const RunBlockDev* __restrict__ run_blocks, const RunSpanDev* __restrict__ run_spans,
    const int band_shift,  // >= 0: pieces are bands of 2^band_shift rows dealt round-robin
    const int tiles_x, const int tiles_y,
    const MarchItemDev* __restrict__ items, float* __restrict__ out,
    unsigned long long* samples_out, unsigned long long* counters,
    // A frame marched in several launches (depth-ordered chunks of the global layer order, so that
    // a chunk is marched while the next one is still being classified): this launch takes the
    // positions [pos_begin, pos_end) of `order`; resume != 0: the run accumulator starts from what
    // the launch before stored (the left fold of DirectSendBase.cpp:413-426 is cut, not re-associated:
    // the same blends in the same order on the same five floats).
    const int pos_begin, const int pos_end, const int resume,
    // non-null (a frame in depth-ordered chunks with occlusion culling, avr_render_plan_culled):
    // visible_out[position] is set to 1 for every box behind this launch's that some ray may still
    // sample -- see the end of the kernel; the caller cleared it.
    uint8_t* __restrict__ visible_out,
    // non-null: a SPECULATIVE frame (MarchSpecDev, avr_internal.h) -- this frame's classify pass left
    // out the boxes no ray sampled in an earlier frame of the same plan.  The march meets a box it
    // needs and finds it unclassified: it flags the box, leaves it out and goes on (with a less
    // opaque accumulator it can only meet MORE boxes than the true frame does, so the flags cover
    // everything the true frame needs); a gated second classify + march, queued behind this launch,
    // then redo the frame with those boxes classified -- and do nothing when no flag was raised.
    const MarchSpecDev* __restrict__ spec) {
  extern __shared__ float4 lds_tables[];  // n_tables x 256 RGBA entries
  if (SPEC && spec != nullptr) {
    const uint32_t* const gate = spec->gate;
    if (gate != nullptr) {
      if (*gate == 0) return;  // (uniform over the grid)
      // the repair march redoes only the workgroups whose first pass met an unclassified box (the
      // same grid, the same work items): every other layer pixel is final already
      const uint8_t* const dirty = spec->dirty_blocks;
      if (dirty != nullptr && dirty[blockIdx.x] == 0) return;
    }
  }

  // ---- XCD-aware work assignment ------------------------------------------------------------
  // Work item = one super-tile (2 x 2 workgroups in Morton order) of one run.  Workgroups are
  // dealt round-robin over the 8 XCDs (block b -> XCD b % 8) and every XCD takes whole items, so
  // the workgroups resident on one XCD at a time cover compact patches of the screen and re-use
  // the same bricklets from that XCD's L2; items come in the host's cost order, most expensive
  // first, so the long rays start early and the cheap tiles fill the tail.
  const unsigned b = blockIdx.x;
  const unsigned xcd = b % kXcds;
  const unsigned within = b / kXcds;
  const MarchItemDev item = items[(within / kSuperTileTiles) * kXcds + xcd];
  if (item.slot == kNoMarchItem) return;
  const unsigned seq = item.slot * kSuperTileTiles + (within % kSuperTileTiles);
  const int tile_x = static_cast<int>(compact_bits(seq));
  const int tile_y = static_cast<int>(compact_bits(seq >> 1));
  if (tile_x >= tiles_x || tile_y >= tiles_y) return;
  const int run = static_cast<int>(item.run);
  // Outside the run's screen rectangle nothing is stored (and no box of the run can be hit).
  const RunRectDev rect = run_rects[run];
  if (rect.x1 < tile_x * kTile || rect.x0 > tile_x * kTile + kTile - 1 ||
      rect.y1 < tile_y * kTile || rect.y0 > tile_y * kTile + kTile - 1) {
    return;
  }

  // ---- stage the transfer-function tables in LDS (one per AMR sampling level) -------------
  {
    const float4* src = reinterpret_cast<const float4*>(tables);
    const int total = n_tables * kTableSize;
    for (int e = threadIdx.x; e < total; e += kBlockThreads) lds_tables[e] = src[e];
  }
  __syncthreads();

  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const int lane = static_cast<int>(threadIdx.x) & 63;
  constexpr int kWaveW = 8, kWaveH = 8;
  const int wave_x0 = tile_x * kTile + (wave & 1) * 8;  // wave-uniform 8x8 sub-tile
  const int wave_y0 = tile_y * kTile + (wave >> 1) * 8;
  // lane -> pixel: every 16 consecutive lanes (the unit the texture addresser works on) cover a
  // compact 4 x 4 pixel patch of the wave's 8 x 8 tile, not an 8 x 2 strip: fewer distinct
  // bricklets (cache lines) per address-processing group
  const int px = wave_x0 + (lane & 3) + ((lane >> 2) & 4);
  const int py = wave_y0 + ((lane >> 2) & 3) + ((lane >> 3) & 4);
  const bool live = (px < fc.width) && (py < fc.height);
  const int64_t n_pixels = static_cast<int64_t>(fc.width) * fc.height;
  const int64_t p = static_cast<int64_t>(py) * fc.width + px;

  const Ray ray = make_ray(px, py, fc);
  const float inv_dx = 1.0f / ray.dx;  // as updateBounds computes it per box (:786)
  const float inv_dy = 1.0f / ray.dy;
  const float inv_dz = 1.0f / ray.dz;

  unsigned fetches = 0;
  // diagnostics of the STATS build (avr_context_set_march_counters): samples that took the exact
  // divide because the reciprocal product sat near an integer, and samples per index mode
  unsigned near_hits = 0, mode_fetches[3] = {0u, 0u, 0u};
  {
    // Where this pixel of the run's layer is stored: only inside the run's screen rectangle, in
    // the block of the DirectSend piece the pixel belongs to (nullptr: nowhere).
    auto layer_pixel = [&]() -> float* {
      if (!(live && px >= rect.x0 && px <= rect.x1 && py >= rect.y0 && py <= rect.y1)) return nullptr;
      int64_t piece;
      int row = py;  // the pixel's row in its piece's numbering (contiguous pieces: the image row)
      if (band_shift >= 0) {  // kPiecesRowBands
        const unsigned band = static_cast<unsigned>(py) >> band_shift;
        const unsigned cycle = band / static_cast<unsigned>(n_pieces);
        piece = band - cycle * static_cast<unsigned>(n_pieces);
        row = static_cast<int>((cycle << band_shift) + (static_cast<unsigned>(py) & ((1u << band_shift) - 1u)));
      } else {
        const int64_t piece_size = n_pixels / n_pieces;  // getPieceRange, DirectSendBase.cpp:59-74
        piece = (piece_size > 0) ? (p / piece_size) : (n_pieces - 1);
        if (piece > n_pieces - 1) piece = n_pieces - 1;
      }
      const RunBlockDev block = run_blocks[static_cast<int64_t>(run) * n_pieces + piece];
      if (block.span_base >= 0) {
        // tightened plan: the row stores only the run's conservative extent on screen; a pixel
        // outside it is empty by construction (counted, should the construction ever be wrong)
        const RunSpanDev span = run_spans[block.span_base + (row - block.first_row)];
        if (px < static_cast<int>(span.x0) || px > static_cast<int>(span.x1)) return nullptr;
        return out + block.offset + span.rel + static_cast<int64_t>(px - static_cast<int>(span.x0)) * 5;
      }
      return out + block.offset +
             (static_cast<int64_t>(row - block.first_row) * (rect.x1 - rect.x0 + 1) + (px - rect.x0)) * 5;
    };
    const int run_begin = (run > 0) ? run_end[run - 1] : 0;
    const int end = (run_end[run] < pos_end) ? run_end[run] : pos_end;
    Layer5 acc = {0.0f, 0.0f, 0.0f, 0.0f, AVR_INF};  // cleared layer pixel: exact blend identity
    if (resume != 0) {
      const float* src = layer_pixel();
      if (src != nullptr) {
        acc.r = src[0];
        acc.g = src[1];
        acc.b = src[2];
        acc.a = src[3];
        acc.d = src[4];
      }
    }
    // The boxes whose conservative screen rectangle misses the wave's 8 x 8 pixels are culled 64 at
    // a time: lane l tests the rectangle of position base + l (one coalesced 16-byte load per lane
    // from the rectangles in global layer order), a ballot leaves the candidates as a bit mask, and
    // only those are visited, in order.  (One box per trip -- two dependent scalar loads, four
    // compares and a branch for every box of the run, of which a tile meets a tenth -- was 10 % of
    // config-4's march and most of config-5's, whose tiles see 1856 boxes.)
    for (int base = (run_begin > pos_begin) ? run_begin : pos_begin; base < end; base += 64) {
      bool candidate = false;
      if (base + lane < end) {
        const int4 r = order_rects[base + lane];
        candidate = !(r.z < wave_x0 || r.x > wave_x0 + (kWaveW - 1) || r.w < wave_y0 ||
                      r.y > wave_y0 + (kWaveH - 1));
      }
      unsigned long long pending = __builtin_amdgcn_ballot_w64(candidate);
      unsigned long long unclassified = 0;  // (speculative frames: candidates the classify pass left out)
      if (SPEC && spec != nullptr) {
        const uint8_t* const covered = spec->classified;
        if (covered != nullptr) {
          const bool missing = candidate && covered[base + lane] == 0;
          unclassified = __builtin_amdgcn_ballot_w64(missing);
        }
      }
      while (pending != 0) {
      const int bit = __builtin_ctzll(pending);
      const int position = base + bit;
      pending &= pending - 1;
      const BoxDev& box = boxes[order[position]];
      float tmin = -AVR_INF;
      float tmax = AVR_INF;
      slab_axis(ray.ox, ray.dx, inv_dx, box.minc[0], box.maxc[0], tmin, tmax);
      slab_axis(ray.oy, ray.dy, inv_dy, box.minc[1], box.maxc[1], tmin, tmax);
      slab_axis(ray.oz, ray.dz, inv_dz, box.minc[2], box.maxc[2], tmin, tmax);
      bool hit = live && (tmax >= tmin);
      if (hit && acc.a == 1.0f) {
        // The run accumulator is opaque: if it is also in front of this box's entry point the
        // blend returns the accumulator unchanged (front + back * (1 - 1)), whatever the box
        // holds, so the march is skipped without changing bits.
        const float ex = ray.ox + ray.dx * tmin;
        const float ey = ray.oy + ray.dy * tmin;
        const float ez = ray.oz + ray.dz * tmin;
        const float entry_depth = (ex - fc.eye[0]) * fc.fwd[0] + (ey - fc.eye[1]) * fc.fwd[1] +
                                  (ez - fc.eye[2]) * fc.fwd[2];
        if (acc.d <= entry_depth) hit = false;
      }
      if (!__builtin_amdgcn_ballot_w64(hit)) continue;  // whole wave missed or terminated
      if (SPEC && spec != nullptr) {
        if ((unclassified >> bit) & 1ull) {
          if (lane == 0) {
            spec->missed[position] = 1;
            if (spec->dirty_blocks_out != nullptr) spec->dirty_blocks_out[blockIdx.x] = 1;
            atomicAdd(spec->miss_count, 1u);
            if (spec->host_miss_flag != nullptr) {
              __hip_atomic_store(spec->host_miss_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
          }
          continue;
        }
        uint8_t* const visited = spec->visited;
        if (visited != nullptr && lane == 0) visited[position] = 1;
      }
      if (hit) {
        const float4* table = lds_tables + box.lut * kTableSize;
        Layer5 layer;
        const unsigned before = fetches;
        int mode = ONLY_MODE;
        if (ONLY_MODE == kPow2Multiply) {
          layer = march_box<STATS, kPow2Multiply>(box, fc, classified, table, ray, tmin, tmax,
                                                  fetches, near_hits);
        } else if (ONLY_MODE == kReciprocal) {
          layer = march_box<STATS, kReciprocal>(box, fc, classified, table, ray, tmin, tmax,
                                                fetches, near_hits);
        } else if (box.index_mode == kPow2Multiply) {  // wave-uniform
          mode = kPow2Multiply;
          layer = march_box<STATS, kPow2Multiply>(box, fc, classified, table, ray, tmin, tmax,
                                                  fetches, near_hits);
        } else if (box.index_mode == kReciprocal) {
          mode = kReciprocal;
          layer = march_box<STATS, kReciprocal>(box, fc, classified, table, ray, tmin, tmax,
                                                fetches, near_hits);
        } else {
          mode = kExactDivide;
          layer = march_box<STATS, kExactDivide>(box, fc, classified, table, ray, tmin, tmax,
                                                 fetches, near_hits);
        }
        if (STATS) mode_fetches[mode] += fetches - before;
        acc = blend_depthsort(acc, layer);
      }
      }
    }
    // ---- which of the boxes BEHIND this launch's can still be sampled ---------------------------
    // The march skips a box at a pixel whose run accumulator is opaque and in front of the box's
    // entry point (above: the blend would return the accumulator unchanged).  A box that EVERY ray
    // skips is never read -- so it need not be classified (the classify launch of the next chunk
    // leaves out the boxes whose flag stays 0: the f64 cells of occluded boxes are not even read;
    // the reference's default boxTransparency = 0 saturates most rays in the first boxes,
    // VolumePainter.cpp:837).  Exactly the march's own test, evaluated now for the boxes to come,
    // in order, with one more bit per pixel: once a box is NOT skipped at a pixel it is marched and
    // may change the accumulator there (even an opaque one: a layer in front of it gives
    // a + 1 * (1 - a), which need not round to 1), so every later box the pixel's ray hits counts
    // as visible.  By induction a box flagged invisible finds, at every pixel that hits it, the
    // accumulator this launch left -- and is skipped.
    if (visible_out != nullptr) {
      bool open = false;  // an earlier box to come is marched at this pixel
      // (the run's boxes behind this launch's; a run that only starts behind them: all of its boxes)
      for (int position = (end > run_begin) ? end : run_begin; position < run_end[run]; ++position) {
        const BoxDev& box = boxes[order[position]];
        if (box.rect[2] < wave_x0 || box.rect[0] > wave_x0 + (kWaveW - 1) || box.rect[3] < wave_y0 ||
            box.rect[1] > wave_y0 + (kWaveH - 1)) {
          continue;
        }
        float tmin = -AVR_INF;
        float tmax = AVR_INF;
        slab_axis(ray.ox, ray.dx, inv_dx, box.minc[0], box.maxc[0], tmin, tmax);
        slab_axis(ray.oy, ray.dy, inv_dy, box.minc[1], box.maxc[1], tmin, tmax);
        slab_axis(ray.oz, ray.dz, inv_dz, box.minc[2], box.maxc[2], tmin, tmax);
        bool visible = live && (tmax >= tmin);
        if (visible && !open && acc.a == 1.0f) {
          const float ex = ray.ox + ray.dx * tmin;
          const float ey = ray.oy + ray.dy * tmin;
          const float ez = ray.oz + ray.dz * tmin;
          const float entry_depth = (ex - fc.eye[0]) * fc.fwd[0] + (ey - fc.eye[1]) * fc.fwd[1] +
                                    (ez - fc.eye[2]) * fc.fwd[2];
          if (acc.d <= entry_depth) visible = false;
        }
        open = open || visible;
        if (__builtin_amdgcn_ballot_w64(visible) != 0 && lane == 0) visible_out[position] = 1;
      }
    }
    // (a resumed launch stores every pixel again, reached or not: keeping "was it reached" per lane
    // costs the register that decides between 7 and 6 waves per SIMD, and as a flag a scalar mask
    // that every trip of the box loop merges -- config-5: + 15 % on the march)
    {
      float* const dst = layer_pixel();
      if (STATS && counters != nullptr && dst == nullptr && acc.a != 0.0f && live && px >= rect.x0 &&
          px <= rect.x1 && py >= rect.y0 && py <= rect.y1) {
        atomicAdd(counters + 4, 1ull);  // a non-empty pixel outside its row's span: never
      }
      if (dst != nullptr) {
        dst[0] = acc.r;
        dst[1] = acc.g;
        dst[2] = acc.b;
        dst[3] = acc.a;
        dst[4] = acc.d;
      }
    }
  }

  if (STATS && samples_out != nullptr) {
    unsigned long long total = fetches;
    for (int offset = 32; offset > 0; offset >>= 1) {
      total += __shfl_down(total, offset, 64);
    }
    if (lane == 0 && total != 0) atomicAdd(samples_out, total);
    if (counters != nullptr) {
      // counters[0] near-integer fallbacks, [1] kExactDivide, [2] kReciprocal, [3] kPow2Multiply
      const unsigned values[4] = {near_hits, mode_fetches[kExactDivide], mode_fetches[kReciprocal],
                                  mode_fetches[kPow2Multiply]};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        unsigned long long sum = values[c];
        for (int offset = 32; offset > 0; offset >>= 1) sum += __shfl_down(sum, offset, 64);
        if (lane == 0 && sum != 0) atomicAdd(counters + c, sum);
      }
    }
  }
}

#define AVR_MARCH_PARAMETERS                                                                       \
    const FrameConsts fc, const BoxDev* __restrict__ boxes, const uint8_t* __restrict__ classified, \
    const float* __restrict__ tables, const int n_tables, const int32_t* __restrict__ order,       \
    const int4* __restrict__ order_rects, const int32_t* __restrict__ run_end, const int n_runs,   \
    const int n_pieces, const RunRectDev* __restrict__ run_rects,                                  \
    const RunBlockDev* __restrict__ run_blocks, const RunSpanDev* __restrict__ run_spans,          \
    const int band_shift, const int tiles_x, const int tiles_y,                                    \
    const MarchItemDev* __restrict__ items, float* __restrict__ out,                               \
    unsigned long long* samples_out, unsigned long long* counters, const int pos_begin,            \
    const int pos_end, const int resume, uint8_t* __restrict__ visible_out,                        \
    const MarchSpecDev* __restrict__ spec
#define AVR_MARCH_ARGUMENTS                                                                        \
    fc, boxes, classified, tables, n_tables, order, order_rects, run_end, n_runs, n_pieces,        \
    run_rects, run_blocks, run_spans, band_shift, tiles_x, tiles_y, items, out, samples_out,       \
    counters, pos_begin, pos_end, resume, visible_out, spec

template <bool STATS, int ONLY_MODE, bool SPEC = false>
__global__ __launch_bounds__(kBlockThreads, AVR_MARCH_GROUP >= 16 ? 3 : AVR_MARCH_GROUP >= 8 ? 4 : AVR_MARCH_GROUP >= 6 ? 5 : 6) void render_runs_kernel(AVR_MARCH_PARAMETERS) {
  render_runs_body<STATS, ONLY_MODE, SPEC>(AVR_MARCH_ARGUMENTS);
}

// The gated second march of a speculative frame under a name of its own (a profile then tells the
// marches that ran from the repair launches that found nothing to do).
template <int ONLY_MODE>
__global__ __launch_bounds__(kBlockThreads, AVR_MARCH_GROUP >= 16 ? 3 : AVR_MARCH_GROUP >= 8 ? 4 : AVR_MARCH_GROUP >= 6 ? 5 : 6) void render_runs_repair_kernel(AVR_MARCH_PARAMETERS) {
  render_runs_body<false, ONLY_MODE, true>(AVR_MARCH_ARGUMENTS);
}
#undef AVR_MARCH_PARAMETERS
#undef AVR_MARCH_ARGUMENTS

// Descriptor upload: pinned host block (read over PCIe through its device mapping) -> HBM.
__global__ void upload_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    dst[i] = src[i];
  }
}

// Test hook (avr_debug_stall_stream): one wave that keeps its stream busy for a BOUNDED time --
// the constant 100 MHz wall clock, and an iteration cap that ends it whatever the clock says -- so
// that the deadline of the host waits can be exercised on a real stream without ever hanging one.
__global__ void stall_kernel(unsigned long long ticks, unsigned int max_iterations) {
  const unsigned long long begin = wall_clock64();
  for (unsigned int i = 0; i < max_iterations; ++i) {
    if (wall_clock64() - begin >= ticks) break;
    __builtin_amdgcn_s_sleep(127);
  }
}

// ---- classify pass ---------------------------------------------------------------------------
// Streams every f64 cell of the frame's boxes once (coalesced rows, the HBM-bound part of the
// frame) and stores its transfer-function table index -- the value VolumePainter.cpp:870-883
// derives per SAMPLE is a pure function of the cell, so it is derived once per CELL here -- as
// one byte in 8 x 4 x 4 bricklets.  One workgroup = 4 k-planes x 4 j-rows x 128 cells of x:
// 16 rows of 1 KiB in, 16 complete bricklets (2 KiB, contiguous) out through LDS.
constexpr int kStagedStride = 34;  // dwords per staged bricklet (32 + 2: conflict-free LDS writes)
constexpr int kRawBufferFlags = 0x00020000;  // word 3 of a raw (untyped) gfx9 buffer resource
// Cache policy of the cell loads (buffer_load aux bits: 1 = sc0, 2 = nt, 16 = sc1): every cell is
// read once per frame, so non-temporal.  Measured beside the march, the policy does not matter
// (0, nt, sc1, nt|sc1, sc0|sc1|nt: 1.049-1.063 ms per frame, profiles/experiments_rounds_1_to_3.md section 3).
constexpr int kStreamingLoad = 2;

// One tile (4 planes x 4 rows x 128 cells) of the classify pass; `staged`: 16 bricklets of LDS.
template <bool SIMPLE>
__device__ __forceinline__ void classify_tile(
    const FrameConsts& fc, const BoxDev* __restrict__ boxes,
    const uint32_t* __restrict__ tile_begin, const int n_boxes, uint8_t* __restrict__ classified,
    const int stream_stores,
    // non-null: this launch classifies the boxes box_list[0 .. n_boxes) (one depth-ordered chunk of
    // the frame; tile_begin is the chunk's prefix sum); null: boxes[0 .. n_boxes)
    const int32_t* __restrict__ box_list,
    // non-null: visible[i] == 0 means no ray samples box box_list[i] this frame (the march's flags,
    // render_runs_kernel): it is not classified
    const uint8_t* __restrict__ visible, const uint32_t tile, uint32_t* const staged) {
  // which box does this workgroup belong to (wave-uniform binary search over the prefix sums)
  int lo = 0, hi = n_boxes;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_begin[mid] <= tile) {
      lo = mid;
    } else {
      hi = mid;
    }
  }
  if (visible != nullptr && visible[lo] == 0) return;  // (wave-uniform; before any barrier)
  const BoxDev& box = boxes[box_list != nullptr ? box_list[lo] : lo];
  const int nx = box.nx, ny = box.ny, nz = box.nz;
  const int bricks_x = (nx + kBrickX - 1) >> 3;
  const int bricks_y = (ny + kBrickY - 1) >> 2;
  const int chunks = (nx + kClassifyChunk - 1) / kClassifyChunk;
  uint32_t local = tile - tile_begin[lo];
  const int chunk = static_cast<int>(local % static_cast<uint32_t>(chunks));
  local /= static_cast<uint32_t>(chunks);
  const int bj = static_cast<int>(local % static_cast<uint32_t>(bricks_y));
  const int bk = static_cast<int>(local / static_cast<uint32_t>(bricks_y));

  const double __attribute__((address_space(1)))* cells =
      (const double __attribute__((address_space(1)))*)box.cells;
  const uint32_t jstride = static_cast<uint32_t>(box.jstride);
  const uint32_t kstride = static_cast<uint32_t>(box.kstride);
  const int t = static_cast<int>(threadIdx.x);
  // 16-byte loads (two cells per lane) need every row to start 16-byte aligned
  const bool paired = ((reinterpret_cast<uintptr_t>(box.cells) & 15u) == 0) &&
                      ((jstride & 1u) == 0) && ((kstride & 1u) == 0) && jstride < (1u << 27);
  typedef double double2_t __attribute__((ext_vector_type(2)));
  // ---- the tile that lies wholly inside its box (nearly all of them: every tile of a box whose
  // sides are multiples of 128 x 4 x 4 cells): no lane or plane is masked, so the pass is straight
  // line code -- per pair of cells two f64 subtractions and multiplications, two conversions, two
  // scalings, two truncations, one pack, and ONE wave-wide test for a non-finite cell instead of a
  // select per cell (the vector instructions of this pass are a fifth of the frame's, and the
  // frame is bound by their issue: DESIGN.md section 4).
  const bool whole_tile = SIMPLE && paired && (chunk + 1) * kClassifyChunk <= nx &&
                          (bj + 1) * kBrickY <= ny && (bk + 1) * kBrickZ <= nz;
  if (whole_tile) {
    const int jj = t >> 6;
    const int xi = (t & 63) * 2;
    const uint32_t lane_bytes = (static_cast<uint32_t>(chunk * kClassifyChunk + xi) +
                                 static_cast<uint32_t>(bj * kBrickY + jj) * jstride) * 8u;
    const __amdgpu_buffer_rsrc_t resource = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double*>(box.cells), 0, 0x7fffffff, kRawBufferFlags);
    double2_t raw[4];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const uint32_t plane_bytes = static_cast<uint32_t>(bk * kBrickZ + pass) * kstride * 8u;
      raw[pass] = __builtin_bit_cast(
          double2_t, __builtin_amdgcn_raw_buffer_load_b128(resource, lane_bytes, plane_bytes, kStreamingLoad));
    }
    char* const staged_at = reinterpret_cast<char*>(staged) + (xi >> 3) * (kStagedStride * 4) + jj * 8 + (xi & 7);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const double a = raw[pass].x, b = raw[pass].y;
      float sa = __builtin_amdgcn_fmed3f(static_cast<float>((a - fc.norm_min) * fc.inv_norm_span), 0.0f, 1.0f) * 255.0f;
      float sb = __builtin_amdgcn_fmed3f(static_cast<float>((b - fc.norm_min) * fc.inv_norm_span), 0.0f, 1.0f) * 255.0f;
      // sanitizeScalarSample (VolumeTypes.hpp:33-38): a non-finite cell counts as 0.0 -- rare, so
      // tested for the whole wave at once
      const bool odd = !__builtin_isfinite(a) || !__builtin_isfinite(b);
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(odd) != 0, 0)) {
        const float at_zero =
            __builtin_amdgcn_fmed3f(static_cast<float>((0.0 - fc.norm_min) * fc.inv_norm_span), 0.0f, 1.0f) * 255.0f;
        sa = __builtin_isfinite(a) ? sa : at_zero;
        sb = __builtin_isfinite(b) ? sb : at_zero;
      }
      const uint32_t two = static_cast<uint32_t>(static_cast<int>(sa)) | (static_cast<uint32_t>(static_cast<int>(sb)) << 8);
      *reinterpret_cast<uint16_t*>(staged_at + pass * 32) = static_cast<uint16_t>(two);
    }
  } else if (paired && nx >= 2) {
    // One j-row per wave, one k-plane per pass, two x-cells (one 16-byte load) per lane.  The
    // lane's byte offset inside the plane is the same in all four passes, so the addresses are a
    // wave-uniform base per pass plus one 32-bit lane offset (global_load ... saddr: no vector
    // address arithmetic per pass).  To keep the loads unconditional, the last cell of an odd row
    // is read as the second half of the pair before it (an 8-byte aligned 16-byte load), lanes
    // outside the box read the chunk's first pair and planes past the box re-read the last plane.
    const int jj = t >> 6;
    const int xi = (t & 63) * 2;  // first of this lane's two cells inside the 128-cell chunk
    const int i0 = chunk * kClassifyChunk;
    const int i = i0 + xi;
    const int j = bj * kBrickY + jj;
    const bool valid = i < nx && j < ny;
    const bool whole = valid && (i + 1 < nx);
    const uint32_t lead = (i0 + 1 < nx) ? 0u : 1u;  // the chunk's first cell is the odd last one
    const uint32_t lane_cell =
        valid ? (static_cast<uint32_t>(xi) + static_cast<uint32_t>(jj) * jstride + lead -
                 (whole ? 0u : 1u))
              : 0u;
    // buffer addressing: the box's cells as the resource, the tile's first pair + the lane's
    // offset in the vector offset, the plane in the scalar offset (host: spans stay below 2^28
    // cells, so every byte offset fits 31 bits) -- no vector address arithmetic per pass
    const uint32_t row0 = (static_cast<uint32_t>(i0) - lead) +
                          static_cast<uint32_t>(bj * kBrickY) * jstride;
    const uint32_t lane_bytes = (row0 + lane_cell) * 8u;
    const __amdgpu_buffer_rsrc_t resource = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double*>(box.cells), 0, 0x7fffffff, kRawBufferFlags);
    // All four planes' loads are issued before the first is used: four round trips in flight per
    // lane, so that a few resident workgroups already keep HBM busy -- what the classify pass
    // gets when it shares the CUs with the march of the previous frame.
    double2_t raw[4];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int k = bk * kBrickZ + pass;
      const uint32_t plane_bytes = static_cast<uint32_t>(k < nz ? k : nz - 1) * kstride * 8u;
      // every f64 cell is read exactly once per frame: a streaming load keeps it from evicting
      // the classified bricklets the co-resident march gathers through the same L1 / L2
      raw[pass] = __builtin_bit_cast(
          double2_t, __builtin_amdgcn_raw_buffer_load_b128(resource, lane_bytes, plane_bytes,
                                                           kStreamingLoad));
    }
    // bricklet row (pass * 4 + jj) of bricklet xi / 8, bytes xi % 8 and xi % 8 + 1
    char* const staged_at = reinterpret_cast<char*>(staged) + (xi >> 3) * (kStagedStride * 4) +
                            jj * 8 + (xi & 7);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      uint32_t two = 0;
      if (bk * kBrickZ + pass < nz) {  // wave-uniform
        two = table_index_pair<SIMPLE>(raw[pass].x, raw[pass].y, fc);
        if (__builtin_expect(!whole, 0)) two = valid ? (two >> 8) : 0u;
      }
      *reinterpret_cast<uint16_t*>(staged_at + pass * 32) = static_cast<uint16_t>(two);
    }
  } else {
    const int xi = t & 127;
    const int i = chunk * kClassifyChunk + xi;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int row = pass * 2 + (t >> 7);
      const int j = bj * kBrickY + (row & 3);
      const int k = bk * kBrickZ + (row >> 2);
      uint32_t idx = 0;
      if (i < nx && j < ny && k < nz) {
        const double raw = __builtin_nontemporal_load(
            cells + (static_cast<uint32_t>(i) + static_cast<uint32_t>(j) * jstride +
                     static_cast<uint32_t>(k) * kstride));
        idx = static_cast<uint32_t>(table_index<SIMPLE>(raw, fc));
      }
      uint32_t packed = idx << (8 * (t & 3));
      packed |= static_cast<uint32_t>(
          __builtin_amdgcn_update_dpp(0, static_cast<int>(packed), 0xB1, 0xF, 0xF, false));
      packed |= static_cast<uint32_t>(
          __builtin_amdgcn_update_dpp(0, static_cast<int>(packed), 0x4E, 0xF, 0xF, false));
      if ((t & 3) == 0) staged[(xi >> 3) * kStagedStride + row * 2 + ((xi & 7) >> 2)] = packed;
    }
  }
  __syncthreads();
  // up to 16 complete bricklets, x-neighbours: in the z-fastest brick order of bricklet_offset
  // they lie bricks_y * bricks_z lines apart, each written as one whole 128-byte line
  const int first_brick_x = chunk * (kClassifyChunk / kBrickX);
  const int bricks_here = (bricks_x - first_brick_x < 16) ? (bricks_x - first_brick_x) : 16;
  const int bricks_z = (nz + kBrickZ - 1) >> 2;
  if (t * 8 < bricks_here * kBrickBytes) {
    // wave-uniform line of the chunk's first bricklet + a 32-bit lane offset (the host keeps
    // bricks_y * bricks_z * 128 below 2^24, avr_host.cpp plan_frame)
    const uint64_t brick0 = (static_cast<uint64_t>(first_brick_x) * static_cast<uint64_t>(bricks_y) +
                             static_cast<uint64_t>(bj)) * static_cast<uint64_t>(bricks_z) +
                            static_cast<uint64_t>(bk);
    const uint32_t x_pitch = static_cast<uint32_t>(bricks_y * bricks_z) * kBrickBytes;
    const uint32_t lane_off = static_cast<uint32_t>(__umul24(static_cast<unsigned>(t >> 4), x_pitch)) +
                              static_cast<uint32_t>(t & 15) * 8u;
    const uint2 v =
        *reinterpret_cast<const uint2*>(&staged[(t >> 4) * kStagedStride + (t & 15) * 2]);
    uint2* const target =
        reinterpret_cast<uint2*>(classified + box.cls_offset + brick0 * kBrickBytes + lane_off);
    // Written through to memory as a stream (sc0 sc1 nt): the bricklets are whole 128-byte lines
    // nobody reads before the next frame's march, and a plain store leaves them dirty in L2 until
    // the kernel's end -- whose write-back then sits between this classify pass and the next
    // kernel of every queue.  One box, fixed reserve: plain 0.9705 / 0.967 ms per frame, nt
    // 0.9686 / 0.9632, sc0 sc1 0.9638 / 0.9624, sc0 sc1 nt 0.9597 / 0.9584.  (The march's 20-byte
    // layer stores want the opposite: nt 0.973 against 0.967, sc0 sc1 nt 1.079.)
    // -- when the march that reads them is a frame away (side by side: the next frame's classify
    // pass runs beside this frame's march) and the volume is more than the memory-side cache
    // holds (avr_capi.cpp).  Where a frame's march follows its classify pass on
    // the same stream (back to back, paired: the ranks of eight), the bricklets it gathers first
    // are still in the caches if they were stored plainly: a rank of eight 0.143 ms plain, 0.149
    // streamed (the march's L2 hit rate is 63.4 % either way, tools/share_cache_pmc.sh: what helps
    // sits below L2, in the memory-side cache, or is the write traffic itself).
    if (stream_stores != 0) {  // (wave-uniform)
      asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" : : "v"(target), "v"(v) : "memory");
    } else {
      *target = v;
    }
  }
}

template <bool SIMPLE>
__global__ __launch_bounds__(kBlockThreads) void classify_kernel(
    const FrameConsts fc, const BoxDev* __restrict__ boxes,
    const uint32_t* __restrict__ tile_begin, const int n_boxes, uint8_t* __restrict__ classified,
    const int stream_stores, const int32_t* __restrict__ box_list,
    const uint8_t* __restrict__ visible) {
  __shared__ uint32_t staged[16 * kStagedStride];  // 16 bricklets
  classify_tile<SIMPLE>(fc, boxes, tile_begin, n_boxes, classified, stream_stores, box_list, visible,
                        blockIdx.x, staged);
}

// The boxes a speculative frame's march found missing (render_runs_kernel, MarchSpecDev): a small
// grid that walks the listed boxes' tiles and does nothing at all unless *gate != 0 -- the usual
// case, which must not cost the dispatch of a workgroup per tile.
template <bool SIMPLE>
__global__ __launch_bounds__(kBlockThreads) void classify_gated_kernel(
    const FrameConsts fc, const BoxDev* __restrict__ boxes,
    const uint32_t* __restrict__ tile_begin, const int n_boxes, uint8_t* __restrict__ classified,
    const int32_t* __restrict__ box_list, const uint8_t* __restrict__ visible,
    const uint32_t* __restrict__ gate, const uint32_t n_tiles) {
  __shared__ uint32_t staged[16 * kStagedStride];
  if (*gate == 0) return;  // (uniform over the grid)
  for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    classify_tile<SIMPLE>(fc, boxes, tile_begin, n_boxes, classified, 0, box_list, visible, tile, staged);
    __syncthreads();  // (the next tile re-uses the staging area)
  }
}

// ---- element-wise image algebra ------------------------------------------------------------

__global__ void blend_depthsort_kernel(const float* __restrict__ top,
                                       const float* __restrict__ bottom, float* __restrict__ out,
                                       int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < n;
       p += stride) {
    const float* t = top + p * 5;
    const float* b = bottom + p * 5;
    const Layer5 tl = {t[0], t[1], t[2], t[3], t[4]};
    const Layer5 bl = {b[0], b[1], b[2], b[3], b[4]};
    const Layer5 o = blend_depthsort(tl, bl);
    float* d = out + p * 5;
    d[0] = o.r;
    d[1] = o.g;
    d[2] = o.b;
    d[3] = o.a;
    d[4] = o.d;
  }
}

// ImageRGBAFloatColorOnlyFeatures::blend (ImageRGBAFloatColorOnly.hpp:19-26)
__device__ __forceinline__ float4 blend_rgba_f32(const float4 top, const float4 bottom) {
  const float t = 1.0f - top.w;
  float4 o;
  o.x = top.x + bottom.x * t;
  o.y = top.y + bottom.y * t;
  o.z = top.z + bottom.z * t;
  o.w = top.w + bottom.w * t;
  return o;
}

__global__ void blend_rgba_f32_kernel(const float4* __restrict__ top,
                                      const float4* __restrict__ bottom, float4* __restrict__ out,
                                      int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < n;
       p += stride) {
    out[p] = blend_rgba_f32(top[p], bottom[p]);
  }
}

// ImageRGBAUByteColorOnlyFeatures::blend (ImageRGBAUByteColorOnly.hpp:19-34): the bottom
// component is scaled in float, truncated to uint8, and the uint8 sum wraps (no saturation).
__device__ __forceinline__ uint32_t blend_rgba_u8(uint32_t top, uint32_t bottom) {
  const float bottom_scale = 1.0f - static_cast<float>(top >> 24) / 255.0f;
  uint32_t out = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint32_t tc = (top >> (8 * c)) & 0xffu;
    const uint32_t bc = (bottom >> (8 * c)) & 0xffu;
    const uint32_t scaled =
        static_cast<uint32_t>(static_cast<int>(static_cast<float>(bc) * bottom_scale)) & 0xffu;
    out |= ((tc + scaled) & 0xffu) << (8 * c);
  }
  return out;
}

__global__ void blend_rgba_u8_kernel(const uint32_t* __restrict__ top,
                                     const uint32_t* __restrict__ bottom,
                                     uint32_t* __restrict__ out, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < n;
       p += stride) {
    out[p] = blend_rgba_u8(top[p], bottom[p]);
  }
}

// ImageColorOnly<F>::blend with regions (Common/ImageColorOnly.hpp:119-199).  One thread per
// output pixel q in [min(tb,bb), max(te,be)): copy where only one image covers q, blend where
// both do.  KIND: 0 depth-sort, 1 float, 2 ubyte.
template <int KIND>
__global__ void blend_regions_kernel(const void* __restrict__ top_v, int64_t tb, int64_t te,
                                     const void* __restrict__ bottom_v, int64_t bb, int64_t be,
                                     void* __restrict__ out_v) {
  const int64_t ob = (tb < bb) ? tb : bb;
  const int64_t oe = (te > be) ? te : be;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t q = ob + static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q < oe;
       q += stride) {
    const bool in_top = (q >= tb) && (q < te);
    const bool in_bottom = (q >= bb) && (q < be);
    if (KIND == 0) {
      const float* t = static_cast<const float*>(top_v) + (q - tb) * 5;
      const float* b = static_cast<const float*>(bottom_v) + (q - bb) * 5;
      float* d = static_cast<float*>(out_v) + (q - ob) * 5;
      Layer5 o;
      if (in_top && in_bottom) {
        const Layer5 tl = {t[0], t[1], t[2], t[3], t[4]};
        const Layer5 bl = {b[0], b[1], b[2], b[3], b[4]};
        o = blend_depthsort(tl, bl);
      } else if (in_top) {
        o = {t[0], t[1], t[2], t[3], t[4]};
      } else if (in_bottom) {
        o = {b[0], b[1], b[2], b[3], b[4]};
      } else {
        continue;
      }
      d[0] = o.r;
      d[1] = o.g;
      d[2] = o.b;
      d[3] = o.a;
      d[4] = o.d;
    } else if (KIND == 1) {
      const float4* t = static_cast<const float4*>(top_v) + (q - tb);
      const float4* b = static_cast<const float4*>(bottom_v) + (q - bb);
      float4* d = static_cast<float4*>(out_v) + (q - ob);
      if (in_top && in_bottom) {
        *d = blend_rgba_f32(*t, *b);
      } else if (in_top) {
        *d = *t;
      } else if (in_bottom) {
        *d = *b;
      }
    } else {
      const uint32_t* t = static_cast<const uint32_t*>(top_v) + (q - tb);
      const uint32_t* b = static_cast<const uint32_t*>(bottom_v) + (q - bb);
      uint32_t* d = static_cast<uint32_t*>(out_v) + (q - ob);
      if (in_top && in_bottom) {
        *d = blend_rgba_u8(*t, *b);
      } else if (in_top) {
        *d = *t;
      } else if (in_bottom) {
        *d = *b;
      }
    }
  }
}

// Color::GetComponentAsByte (Common/Color.hpp:86-90)
__device__ __forceinline__ uint32_t component_as_byte(float c) {
  const int tv = static_cast<int>(c * 256.f);
  return static_cast<uint32_t>((tv < 0) ? 0 : (tv > 255) ? 255 : tv);
}

// The same for ARBITRARY floats (encodeColor of a caller's colour, Common/ImageRGBAUByteColorOnly.cpp:
// 16-27).  int(c * 256.f) is undefined in C++ when the product does not fit an int; on the CPU the
// reference runs on (x86-64: cvttss2si) it is INT_MIN for NaN and for |product| >= 2^31, hence byte
// 0 -- where this GPU's conversion saturates (3e9 -> 255).  Found by the vectors made with the
// reference's own object code (tests/golden/ref_blend.npz); the frame's own colours (<= 1 after the
// march's clamp) never get there, so the hot kernels keep the plain conversion.
__device__ __forceinline__ uint32_t component_as_byte_of_any_float(float c) {
  const float t = c * 256.f;
  const int tv = (t >= -2147483648.f && t < 2147483648.f) ? static_cast<int>(t) : (-2147483647 - 1);
  return static_cast<uint32_t>((tv < 0) ? 0 : (tv > 255) ? 255 : tv);
}

__global__ void encode_u8_kernel(const float4* __restrict__ rgba, uint32_t* __restrict__ out,
                                 int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < n;
       p += stride) {
    const float4 c = rgba[p];
    out[p] = component_as_byte_of_any_float(c.x) | (component_as_byte_of_any_float(c.y) << 8) |
             (component_as_byte_of_any_float(c.z) << 16) | (component_as_byte_of_any_float(c.w) << 24);
  }
}

// Color::SetComponentFromByte (Common/Color.hpp:69-77)
__global__ void decode_u8_kernel(const uint32_t* __restrict__ in, float4* __restrict__ rgba,
                                 int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < n;
       p += stride) {
    const uint32_t e = in[p];
    float4 c;
    c.x = static_cast<float>(e & 0xffu) / 255.f;
    c.y = static_cast<float>((e >> 8) & 0xffu) / 255.f;
    c.z = static_cast<float>((e >> 16) & 0xffu) / 255.f;
    c.w = static_cast<float>(e >> 24) / 255.f;
    rgba[p] = c;
  }
}

// Receiver-side fold over runs in global order (DirectSendBase.cpp:441-445): the first run's
// slice is taken as is, every further slice is blended underneath.
__global__ void fold_runs_kernel(const float* const* __restrict__ slices, int n_slices,
                                 float* __restrict__ out, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < n;
       p += stride) {
    Layer5 acc = {0.0f, 0.0f, 0.0f, 0.0f, AVR_INF};
    for (int s = 0; s < n_slices; ++s) {
      const float* src = slices[s] + p * 5;
      const Layer5 layer = {src[0], src[1], src[2], src[3], src[4]};
      acc = (s == 0) ? layer : blend_depthsort(acc, layer);
    }
    float* d = out + p * 5;
    d[0] = acc.r;
    d[1] = acc.g;
    d[2] = acc.b;
    d[3] = acc.a;
    d[4] = acc.d;
  }
}

// Receiver side of the layered compose for one DirectSend piece (DirectSendBase.cpp:400-446):
// per pixel, the runs that cover it are blended in global order.  Starting from the cleared
// pixel instead of the first run's pixel gives the same bits (exact identity of the blend).
//
// One workgroup = up to 256 consecutive pixels of one image row.  The runs are scanned in chunks
// of 256 (one run per thread): those whose rectangle touches the workgroup's row segment are
// compacted -- order preserved -- into an LDS list with their source row address, and the pixels
// then blend only the listed runs.
struct FoldEntry {
  int32_t x0, x1;
  int64_t base;  // float offset of pixel x = 0 of this row inside the run's block
};

template <bool OWN>
__global__ __launch_bounds__(256) void fold_plan_kernel(
    const int width, const int64_t piece_begin, const int64_t piece_end, const int n_runs,
    const RunRectDev* __restrict__ rects, const RunBlockDev* __restrict__ blocks,
    const RunSpanDev* __restrict__ spans, const float* __restrict__ recv,
    float* __restrict__ out_piece,
    uint8_t* __restrict__ out_rgb8, const int first_row, const int chunks_per_row,
    const PieceMapDev pieces, const int piece, const int64_t own_begin, const int64_t own_end,
    const int64_t own_delta, const int n_segments, const int flip_height) {
  __shared__ FoldEntry list[256];
  __shared__ int wave_count[4];
  const int tid = static_cast<int>(threadIdx.x);
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const bool bands = pieces.layout == kPiecesRowBands;
  // A workgroup folds one 256-pixel segment of a row after the other (launch_fold_plan caps the
  // grid): the fold is queued where a frame's march ends and the next pair of paint kernels
  // starts, and a grid of one workgroup per segment (16384 at 2048^2) took the whole GPU for
  // itself just then.  One rank, config-4, fixed classify reserve, one box: 16384 workgroups
  // 0.997 ms per frame, 2048 0.991, 512 0.987, 256 0.980, 64 0.985.
  for (int seg = static_cast<int>(blockIdx.x); seg < n_segments; seg += static_cast<int>(gridDim.x)) {
  // `block_row`: the row in the numbering the blocks use (piece rows); `row`: the image row
  const int block_row = first_row + seg / chunks_per_row;
  const int row = bands ? image_row_of(pieces, piece, block_row) : block_row;
  const int seg_x0 = (seg % chunks_per_row) * 256;
  const int seg_x1 = min(seg_x0 + 255, width - 1);
  const int px = seg_x0 + tid;
  // position among the piece's pixels: its rows in order, or the image's pixel range
  const int64_t p = static_cast<int64_t>(block_row) * width + px;
  const bool live = (px < width) && (p >= piece_begin) && (p < piece_end);

  Layer5 acc = {0.0f, 0.0f, 0.0f, 0.0f, AVR_INF};
  for (int chunk = 0; chunk < n_runs; chunk += 256) {
    const int g = chunk + tid;
    bool touches = false;
    FoldEntry entry = {0, -1, 0};
    if (g < n_runs) {
      const RunRectDev rect = rects[g];
      touches = rect.x0 <= seg_x1 && rect.x1 >= seg_x0 && rect.y0 <= row && rect.y1 >= row;
      if (touches) {
        const RunBlockDev block = blocks[g];
        if (block.span_base >= 0) {  // tightened plan: this row's stored extent
          const RunSpanDev span = spans[block.span_base + (block_row - block.first_row)];
          entry.x0 = static_cast<int32_t>(span.x0);
          entry.x1 = static_cast<int32_t>(span.x1);
          touches = entry.x0 <= seg_x1 && entry.x1 >= seg_x0;
          entry.base = block.offset + span.rel - static_cast<int64_t>(entry.x0) * 5;
        } else {
          entry.x0 = rect.x0;
          entry.x1 = rect.x1;
          entry.base = block.offset + (static_cast<int64_t>(block_row - block.first_row) *
                                           (rect.x1 - rect.x0 + 1) - rect.x0) * 5;
        }
        // a block of the rank's own runs may still lie where the march stored it (FoldLaunch):
        // the same floats, own_delta away from where the receive layout has them
        if (OWN && block.offset >= own_begin && block.offset < own_end) entry.base += own_delta;
      }
    }
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(touches);
    if (lane == 0) wave_count[wave] = __popcll(mask);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < 4; ++w) {
      const int c = wave_count[w];
      before += (w < wave) ? c : 0;
      total += c;
    }
    if (touches) list[before + __popcll(mask & ((1ull << lane) - 1ull))] = entry;
    __syncthreads();
    for (int e = 0; e < total; ++e) {
      const FoldEntry run = list[e];  // same address for all lanes: LDS broadcast
      if (live && px >= run.x0 && px <= run.x1) {
        const float* src = recv + run.base + static_cast<int64_t>(px) * 5;
        const Layer5 layer = {src[0], src[1], src[2], src[3], src[4]};
        acc = blend_depthsort(acc, layer);
      }
    }
    __syncthreads();  // the list is rewritten by the next chunk
  }
  if (!live) continue;  // (after the chunk loop's closing barrier: the list is free again)
  const int64_t q = p - piece_begin;
  if (out_piece != nullptr) {
    float* d = out_piece + q * 5;
    d[0] = acc.r;
    d[1] = acc.g;
    d[2] = acc.b;
    d[3] = acc.a;
    d[4] = acc.d;
  }
  if (out_rgb8 != nullptr) {
    // flip_height > 0 (one rank, the piece is the image): the bytes go straight to the output
    // file's rows, top-down (SavePPM.cpp:25) -- avr_assemble_rows' flip in the same pass
    const int64_t at = (flip_height > 0)
                           ? static_cast<int64_t>(flip_height - 1 - row) * width + px
                           : q;
    uint8_t* b = out_rgb8 + at * 3;
    b[0] = static_cast<uint8_t>(component_as_byte(acc.r));
    b[1] = static_cast<uint8_t>(component_as_byte(acc.g));
    b[2] = static_cast<uint8_t>(component_as_byte(acc.b));
  }
  }  // segments
}

// downsampleImage (VolumeRenderer.cpp:479-528): sums in dy-major, dx-minor order.
__global__ void downsample_kernel(const float* __restrict__ src, int tw, int th, int block,
                                  float* __restrict__ dst) {
  const int64_t n = static_cast<int64_t>(tw) * th;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int src_w = tw * block;
  const float inv_samples = 1.0f / static_cast<float>(block * block);
  for (int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q < n;
       q += stride) {
    const int x = static_cast<int>(q % tw);
    const int y = static_cast<int>(q / tw);
    float sr = 0.0f, sg = 0.0f, sb = 0.0f, sa = 0.0f;
    for (int dy = 0; dy < block; ++dy) {
      const float* row = src + (static_cast<int64_t>(y * block + dy) * src_w + x * block) * 5;
      for (int dx = 0; dx < block; ++dx) {
        sr += row[dx * 5 + 0];
        sg += row[dx * 5 + 1];
        sb += row[dx * 5 + 2];
        sa += row[dx * 5 + 3];
      }
    }
    float* d = dst + q * 5;
    d[0] = sr * inv_samples;
    d[1] = sg * inv_samples;
    d[2] = sb * inv_samples;
    d[3] = sa * inv_samples;
    d[4] = AVR_INF;
  }
}

// SavePPM pixel bytes (SavePPM.cpp:17-36): RGB8, output rows top-down.
__global__ void quantize_kernel(const float* __restrict__ src, int w, int h, int stride_f,
                                uint8_t* __restrict__ dst) {
  const int64_t n = static_cast<int64_t>(w) * h;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q < n;
       q += stride) {
    const int x = static_cast<int>(q % w);
    const int out_row = static_cast<int>(q / w);
    const int y = h - 1 - out_row;
    const float* s = src + (static_cast<int64_t>(y) * w + x) * stride_f;
    uint8_t* d = dst + q * 3;
    d[0] = static_cast<uint8_t>(component_as_byte(s[0]));
    d[1] = static_cast<uint8_t>(component_as_byte(s[1]));
    d[2] = static_cast<uint8_t>(component_as_byte(s[2]));
  }
}

// Rows of a byte image in reverse order (image origin bottom-left -> file rows top-down,
// Common/SavePPM.cpp:25, Common/SavePNG.cpp:64-71).
template <typename T>
__global__ void flip_rows_kernel(const T* __restrict__ src, int64_t row_items, int h,
                                 T* __restrict__ dst) {
  const int64_t n = row_items * h;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q < n;
       q += stride) {
    const int64_t row = q / row_items;
    dst[q] = src[(h - 1 - row) * row_items + (q - row * row_items)];
  }
}

// Gathered piece-major rows -> image rows (row bands), optionally upside down: destination row
// y_out holds image row y = flip ? h - 1 - y_out : y_out, which is piece row j of piece k and
// sits after the rows of the pieces before k in the gathered buffer.
// own (may be null): the rows of piece own_piece are read from there -- the root's own piece where
// its fold wrote it -- instead of from the gathered buffer (one device copy less per frame).
template <typename T>
__global__ void assemble_rows_kernel(const T* __restrict__ src, int64_t row_items,
                                     const PieceMapDev pieces, int flip, T* __restrict__ dst,
                                     const T* __restrict__ own, int own_piece) {
  const int h = pieces.height;
  const int64_t n = row_items * h;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q < n;
       q += stride) {
    const int64_t out_row = q / row_items;
    const int y = flip ? (h - 1 - static_cast<int>(out_row)) : static_cast<int>(out_row);
    int64_t src_row = y;
    const T* from = src;
    if (pieces.layout == kPiecesRowBands) {
      const int k = piece_of_row(pieces, y);
      src_row = piece_row_of(pieces, y);
      if (own != nullptr && k == own_piece) {
        from = own;
      } else {
        for (int before = 0; before < k; ++before) src_row += piece_row_count(pieces, before);
      }
    } else if (own != nullptr) {
      // contiguous pieces: piece k = pixels [k * piece_size, ...) of the image, the last to the end
      const int64_t pixel = static_cast<int64_t>(y) * pieces.width;
      const int64_t own_begin = pieces.piece_size * own_piece;
      const int64_t own_end = (own_piece < pieces.n_pieces - 1) ? own_begin + pieces.piece_size
                                                               : static_cast<int64_t>(pieces.width) * pieces.height;
      // (rows are whole only if the piece boundaries fall on rows: the caller checks)
      if (pixel >= own_begin && pixel < own_end) {
        from = own;
        src_row = y - own_begin / pieces.width;
      }
    }
    dst[q] = from[src_row * row_items + (q - out_row * row_items)];
  }
}

int grid_for(int64_t n, int block) {
  int64_t blocks = (n + block - 1) / block;
  const int64_t cap = 256 * 8;  // 256 CUs x 8 blocks, grid-stride the rest
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return static_cast<int>(blocks);
}

int check_launch(const char* what) {
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string(what) + ": " + hipGetErrorString(err));
    return AVR_ERR_RUNTIME;
  }
  return AVR_OK;
}

}  // namespace

int launch_upload(const void* host_mapped, void* dev, size_t bytes, void* stream_v) {
  const size_t n = (bytes + 15) / 16;
  if (n == 0) return AVR_OK;
  const unsigned blocks = static_cast<unsigned>(std::min<size_t>((n + 255) / 256, 1024));
  hipLaunchKernelGGL(upload_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_v),
                     static_cast<const uint4*>(host_mapped), static_cast<uint4*>(dev), n);
  return check_launch("upload_kernel");
}

int launch_stall(int milliseconds, void* stream_v) {
  const unsigned long long ticks = static_cast<unsigned long long>(milliseconds) * 100000ull;
  // (an iteration sleeps ~8 K cycles = 3-4 us: the cap ends the kernel after a few times the
  // requested time even if the clock never moved)
  const unsigned int cap = static_cast<unsigned int>(milliseconds) * 1000u;
  hipLaunchKernelGGL(stall_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream_v), ticks, cap);
  return check_launch("stall_kernel");
}

int launch_classify(const RenderLaunch& L, void* stream_v) {
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (L.n_classify_tiles == 0) return AVR_OK;
  const FrameConsts& fc = L.consts;
  // the standard API path (normalise on, scalarRange {0,1}, no log, no soft clip)
  const bool simple = !fc.log_scale && fc.normalize && !fc.apply_clip && fc.range_min == 0.0f &&
                      fc.inverse_range == 1.0f;
  const size_t pad = L.classify_lds_pad;  // occupancy cap beside the march (avr_renderer)
  // (a chunk of the frame: n_classify_boxes entries of box_list_dev under the chunk's own prefix)
  const int n_listed = L.box_list_dev != nullptr ? L.n_classify_boxes : L.n_boxes;
  if (L.classify_gate != nullptr) {
    // (a speculative frame's repair pass: nothing to do as a rule, so not a workgroup per tile)
    const unsigned grid = std::min<unsigned>(L.n_classify_tiles, 2048u);
    if (simple) {
      hipLaunchKernelGGL(classify_gated_kernel<true>, dim3(grid), dim3(kBlockThreads), 0, stream,
                         L.consts, L.boxes_dev, L.tile_begin_dev, n_listed, L.classified,
                         L.box_list_dev, L.visible_in, L.classify_gate, L.n_classify_tiles);
    } else {
      hipLaunchKernelGGL(classify_gated_kernel<false>, dim3(grid), dim3(kBlockThreads), 0, stream,
                         L.consts, L.boxes_dev, L.tile_begin_dev, n_listed, L.classified,
                         L.box_list_dev, L.visible_in, L.classify_gate, L.n_classify_tiles);
    }
    return check_launch("classify_gated_kernel");
  }
  if (simple) {
    hipLaunchKernelGGL(classify_kernel<true>, dim3(L.n_classify_tiles), dim3(kBlockThreads), pad,
                       stream, L.consts, L.boxes_dev, L.tile_begin_dev, n_listed, L.classified,
                       L.classify_stream_stores, L.box_list_dev, L.visible_in);
  } else {
    hipLaunchKernelGGL(classify_kernel<false>, dim3(L.n_classify_tiles), dim3(kBlockThreads), pad,
                       stream, L.consts, L.boxes_dev, L.tile_begin_dev, n_listed, L.classified,
                       L.classify_stream_stores, L.box_list_dev, L.visible_in);
  }
  return check_launch("classify_kernel");
}

int launch_march(const RenderLaunch& L, void* stream_v) {
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  const int tiles_x = (L.consts.width + kTile - 1) / kTile;
  const int tiles_y = (L.consts.height + kTile - 1) / kTile;
  const unsigned blocks = L.n_items * kSuperTileTiles;
  if (blocks == 0) return AVR_OK;
  size_t lds_bytes = static_cast<size_t>(L.n_tables) * kTableSize * sizeof(float4);
  if (L.workgroups_per_cu > 0) {
    // Occupancy cap through the LDS allocation: n workgroups of 160 KiB / n minus a share of the
    // 10 KiB left for the co-resident kernel's own LDS (classify_kernel stages 2 KiB per workgroup).
    const size_t share = (150 * 1024 / static_cast<size_t>(L.workgroups_per_cu)) & ~size_t{1023};
    lds_bytes = std::max(lds_bytes, share);
  }
  const bool stats = L.samples_out != nullptr;
  int band_shift = -1;
  if (L.pieces.layout == kPiecesRowBands) {
    band_shift = 0;
    while ((1 << band_shift) < L.pieces.band_rows) ++band_shift;
    if ((1 << band_shift) != L.pieces.band_rows) {
      set_error("render_runs_kernel: band_rows must be a power of two");
      return AVR_ERR_INVALID_ARGUMENT;
    }
  }
#define AVR_LAUNCH(STATS, ONLY)                                                                 \
  hipLaunchKernelGGL((render_runs_kernel<STATS, ONLY>), dim3(blocks), dim3(kBlockThreads),      \
                     lds_bytes, stream, L.consts, L.boxes_dev, L.classified, L.tables_dev,      \
                     L.n_tables, L.order_dev, reinterpret_cast<const int4*>(L.order_rects_dev),  \
                     L.run_end_dev, L.n_runs, L.n_pieces,                                        \
                     L.run_rects_dev, L.run_blocks_dev, L.run_spans_dev, band_shift, tiles_x,    \
                     tiles_y,                                                                    \
                     L.items_dev,                                                                \
                     L.out_layers, L.samples_out, L.counters, L.pos_begin,                       \
                     (L.pos_end < 0 ? L.n_order : L.pos_end), L.resume, L.visible_out, L.spec_dev)
  if (L.spec_dev != nullptr) {
    if (stats) {
      set_error("render_runs_kernel: a speculative frame cannot count samples");
      return AVR_ERR_INVALID_ARGUMENT;
    }
#define AVR_LAUNCH_SPEC(ONLY)                                                                    \
  hipLaunchKernelGGL((L.spec_is_repair ? render_runs_repair_kernel<ONLY>                           \
                                       : render_runs_kernel<false, ONLY, true>),                  \
                     dim3(blocks), dim3(kBlockThreads),                                            \
                     lds_bytes, stream, L.consts, L.boxes_dev, L.classified, L.tables_dev,       \
                     L.n_tables, L.order_dev, reinterpret_cast<const int4*>(L.order_rects_dev),  \
                     L.run_end_dev, L.n_runs, L.n_pieces,                                        \
                     L.run_rects_dev, L.run_blocks_dev, L.run_spans_dev, band_shift, tiles_x,    \
                     tiles_y,                                                                    \
                     L.items_dev,                                                                \
                     L.out_layers, L.samples_out, L.counters, L.pos_begin,                       \
                     (L.pos_end < 0 ? L.n_order : L.pos_end), L.resume, L.visible_out, L.spec_dev)
    if (L.only_mode == kPow2Multiply) {
      AVR_LAUNCH_SPEC(kPow2Multiply);
    } else if (L.only_mode == kReciprocal) {
      AVR_LAUNCH_SPEC(kReciprocal);
    } else {
      AVR_LAUNCH_SPEC(-1);
    }
#undef AVR_LAUNCH_SPEC
    return check_launch("render_runs_kernel (speculative)");
  }
  if (L.only_mode == kPow2Multiply) {
    if (stats) AVR_LAUNCH(true, kPow2Multiply); else AVR_LAUNCH(false, kPow2Multiply);
  } else if (L.only_mode == kReciprocal) {
    if (stats) AVR_LAUNCH(true, kReciprocal); else AVR_LAUNCH(false, kReciprocal);
  } else {
    if (stats) AVR_LAUNCH(true, -1); else AVR_LAUNCH(false, -1);
  }
#undef AVR_LAUNCH
  return check_launch("render_runs_kernel");
}

int launch_blend(int kind, const void* top, const void* bottom, void* out, int64_t n,
                 void* stream_v) {
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (n <= 0) return AVR_OK;
  const int grid = grid_for(n, 256);
  if (kind == 0) {
    hipLaunchKernelGGL(blend_depthsort_kernel, dim3(grid), dim3(256), 0, stream,
                       static_cast<const float*>(top), static_cast<const float*>(bottom),
                       static_cast<float*>(out), n);
  } else if (kind == 1) {
    hipLaunchKernelGGL(blend_rgba_f32_kernel, dim3(grid), dim3(256), 0, stream,
                       static_cast<const float4*>(top), static_cast<const float4*>(bottom),
                       static_cast<float4*>(out), n);
  } else {
    hipLaunchKernelGGL(blend_rgba_u8_kernel, dim3(grid), dim3(256), 0, stream,
                       static_cast<const uint32_t*>(top), static_cast<const uint32_t*>(bottom),
                       static_cast<uint32_t*>(out), n);
  }
  return check_launch("blend kernel");
}

int launch_blend_regions(int kind, const void* top, int64_t tb, int64_t te, const void* bottom,
                         int64_t bb, int64_t be, void* out, void* stream_v) {
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  const int64_t ob = tb < bb ? tb : bb;
  const int64_t oe = te > be ? te : be;
  const int64_t n = oe - ob;
  if (n <= 0) return AVR_OK;
  const int grid = grid_for(n, 256);
  if (kind == 0) {
    hipLaunchKernelGGL(blend_regions_kernel<0>, dim3(grid), dim3(256), 0, stream, top, tb, te,
                       bottom, bb, be, out);
  } else if (kind == 1) {
    hipLaunchKernelGGL(blend_regions_kernel<1>, dim3(grid), dim3(256), 0, stream, top, tb, te,
                       bottom, bb, be, out);
  } else {
    hipLaunchKernelGGL(blend_regions_kernel<2>, dim3(grid), dim3(256), 0, stream, top, tb, te,
                       bottom, bb, be, out);
  }
  return check_launch("blend_regions_kernel");
}

int launch_encode_u8(const float* rgba, uint32_t* out, int64_t n, void* stream_v) {
  if (n <= 0) return AVR_OK;
  hipLaunchKernelGGL(encode_u8_kernel, dim3(grid_for(n, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_v), reinterpret_cast<const float4*>(rgba), out,
                     n);
  return check_launch("encode_u8_kernel");
}

int launch_decode_u8(const uint32_t* in, float* rgba, int64_t n, void* stream_v) {
  if (n <= 0) return AVR_OK;
  hipLaunchKernelGGL(decode_u8_kernel, dim3(grid_for(n, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_v), in, reinterpret_cast<float4*>(rgba), n);
  return check_launch("decode_u8_kernel");
}

// workgroups of a fold launch at most unless the caller says otherwise (fold_plan_kernel): 8 per CU
constexpr int64_t kFoldWorkgroups = 2048;

int launch_fold_plan(const FoldLaunch& L, void* stream_v) {
  const int64_t n = L.piece_end - L.piece_begin;
  if (n <= 0) return AVR_OK;
  // (row bands: piece_begin is 0 and the rows are the piece's own)
  const int first_row = static_cast<int>(L.piece_begin / L.width);
  const int last_row = static_cast<int>((L.piece_end - 1) / L.width);
  const int chunks_per_row = (L.width + 255) / 256;
  const int64_t blocks = static_cast<int64_t>(last_row - first_row + 1) * chunks_per_row;
  if (blocks > 0x7fffffffLL) {
    set_error("fold_plan_kernel: image too large");
    return AVR_ERR_INVALID_ARGUMENT;
  }
  auto kernel = (L.own_end > L.own_begin) ? fold_plan_kernel<true> : fold_plan_kernel<false>;
  const int64_t grid = std::min<int64_t>(blocks, L.max_workgroups > 0 ? L.max_workgroups : kFoldWorkgroups);
  hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_v), L.width, L.piece_begin, L.piece_end,
                     L.n_runs, L.run_rects_dev, L.run_blocks_dev, L.run_spans_dev, L.recv,
                     L.out_piece, L.out_rgb8,
                     first_row, chunks_per_row, L.pieces, L.piece, L.own_begin, L.own_end,
                     L.own_delta, static_cast<int>(blocks), L.flip_height);
  return check_launch("fold_plan_kernel");
}

int launch_fold_runs(const float* const* slices_dev, int n_slices, float* out, int64_t n,
                     void* stream_v) {
  if (n <= 0) return AVR_OK;
  hipLaunchKernelGGL(fold_runs_kernel, dim3(grid_for(n, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_v), slices_dev, n_slices, out, n);
  return check_launch("fold_runs_kernel");
}

int launch_downsample(const float* src, int tw, int th, int block, float* dst, void* stream_v) {
  const int64_t n = static_cast<int64_t>(tw) * th;
  if (n <= 0) return AVR_OK;
  hipLaunchKernelGGL(downsample_kernel, dim3(grid_for(n, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_v), src, tw, th, block, dst);
  return check_launch("downsample_kernel");
}

int launch_flip_rows(const uint8_t* src, int64_t row_bytes, int h, uint8_t* dst, void* stream_v) {
  if (row_bytes <= 0 || h <= 0) return AVR_OK;
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  const bool wide = (row_bytes % 16 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) &&
                    ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0);
  if (wide) {
    const int64_t items = row_bytes / 16;
    hipLaunchKernelGGL(flip_rows_kernel<uint4>, dim3(grid_for(items * h, 256)), dim3(256), 0, stream,
                       reinterpret_cast<const uint4*>(src), items, h, reinterpret_cast<uint4*>(dst));
  } else {
    hipLaunchKernelGGL(flip_rows_kernel<uint8_t>, dim3(grid_for(row_bytes * h, 256)), dim3(256), 0,
                       stream, src, row_bytes, h, dst);
  }
  return check_launch("flip_rows_kernel");
}

int launch_assemble_rows(const PieceMapDev& pieces, const uint8_t* src, int64_t row_bytes, int flip,
                         uint8_t* dst, void* stream_v, const uint8_t* own, int own_piece) {
  const int h = pieces.height;
  if (row_bytes <= 0 || h <= 0) return AVR_OK;
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (own != nullptr && pieces.layout != kPiecesRowBands &&
      (pieces.piece_size % pieces.width != 0 || own_piece < 0)) {
    set_error("assemble_rows: the own piece can only be read in place if the pieces are whole rows");
    return AVR_ERR_INVALID_ARGUMENT;
  }
  const uintptr_t alignment = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) |
                              reinterpret_cast<uintptr_t>(own);
  const bool wide = (row_bytes % 16 == 0) && ((alignment & 15u) == 0);
  // (a small grid: the pass runs beside the paint kernels of the next frames on the root rank and
  // should take bandwidth, which it barely needs, not workgroup slots)
  if (wide) {
    const int64_t items = row_bytes / 16;
    hipLaunchKernelGGL(assemble_rows_kernel<uint4>, dim3(std::min(grid_for(items * h, 256), 256)), dim3(256), 0,
                       stream, reinterpret_cast<const uint4*>(src), items, pieces, flip,
                       reinterpret_cast<uint4*>(dst), reinterpret_cast<const uint4*>(own), own_piece);
  } else if (row_bytes % 4 == 0 && (alignment & 3u) == 0) {
    const int64_t items = row_bytes / 4;
    hipLaunchKernelGGL(assemble_rows_kernel<uint32_t>, dim3(grid_for(items * h, 256)), dim3(256), 0,
                       stream, reinterpret_cast<const uint32_t*>(src), items, pieces, flip,
                       reinterpret_cast<uint32_t*>(dst), reinterpret_cast<const uint32_t*>(own),
                       own_piece);
  } else {
    hipLaunchKernelGGL(assemble_rows_kernel<uint8_t>, dim3(grid_for(row_bytes * h, 256)), dim3(256),
                       0, stream, src, row_bytes, pieces, flip, dst, own, own_piece);
  }
  return check_launch("assemble_rows_kernel");
}

int launch_quantize(const float* src, int w, int h, int stride, uint8_t* dst, void* stream_v) {
  const int64_t n = static_cast<int64_t>(w) * h;
  if (n <= 0) return AVR_OK;
  hipLaunchKernelGGL(quantize_kernel, dim3(grid_for(n, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_v), src, w, h, stride, dst);
  return check_launch("quantize_kernel");
}

}  // namespace avr
