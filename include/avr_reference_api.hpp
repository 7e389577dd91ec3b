// avr_reference_api.hpp -- header-only C++ adapters that put the C ABI (avr_hip.h) behind the
// reference's own operator interfaces for the hot path:
//
//   VolumePainter::paint(...)            Common/VolumePainter.hpp:15-32
//   Compositor::compose(...) semantics   Common/Compositor.hpp:19-40 (+ LayeredImageInterface,
//                                        Common/LayeredImageInterface.hpp:9-28)
//   Image blend                          Common/ImageColorOnly.hpp:119-199
//
// The adapters are templates over the reference's own types (amrex::RealVect, amrex::Array4,
// volume::AmrBox, volume::CameraParameters, ImageRGBAFloatColorDepthSort, ...): they only use
// the members the reference's code uses, so inside the reference tree they bind to the real
// types, and in this repository's tests to small stand-ins with the same member names
// (tests/cxx/adapter_test.cpp).  Errors are reported the way the reference reports them:
// std::invalid_argument / std::runtime_error.
//
// Cell data may be host or device memory: device pointers are used in place, host pointers are
// staged to HBM for the call (the reference's CPU build keeps MultiFab data on the host).
#ifndef AVR_REFERENCE_API_HPP
#define AVR_REFERENCE_API_HPP

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "avr_hip.h"

namespace avr {

inline void check(int status) {
  if (status == AVR_OK) return;
  const std::string message = avr_last_error();
  if (status == AVR_ERR_INVALID_ARGUMENT) throw std::invalid_argument(message);
  throw std::runtime_error(message);
}

inline void hip_ok(hipError_t err, const char* what) {
  if (err != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(err));
}

// One per rank (= per GPU); replaces the function-local static painter/compositor instances of
// VolumeRenderer/VolumeRenderer.cpp:909-927.
class Context {
 public:
  explicit Context(int device = 0) { check(avr_context_create(device, &ctx_)); }
  ~Context() { avr_context_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  avr_context* get() const { return ctx_; }
  void synchronize() const { check(avr_context_synchronize(ctx_)); }

 private:
  avr_context* ctx_ = nullptr;
};

// RAII device buffer.
template <typename T>
class DeviceBuffer {
 public:
  DeviceBuffer() = default;
  explicit DeviceBuffer(std::size_t count) { resize(count); }
  ~DeviceBuffer() { release(); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  void resize(std::size_t count) {
    release();
    if (count > 0) hip_ok(hipMalloc(reinterpret_cast<void**>(&ptr_), count * sizeof(T)), "hipMalloc");
    count_ = count;
  }
  void upload(const T* host, std::size_t count) {
    if (count > count_) resize(count);
    if (count) hip_ok(hipMemcpy(ptr_, host, count * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy");
  }
  void download(T* host, std::size_t count) const {
    if (count) hip_ok(hipMemcpy(host, ptr_, count * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy");
  }
  T* data() const { return ptr_; }
  std::size_t size() const { return count_; }

 private:
  void release() {
    if (ptr_ != nullptr) (void)hipFree(ptr_);
    ptr_ = nullptr;
    count_ = 0;
  }
  T* ptr_ = nullptr;
  std::size_t count_ = 0;
};

inline bool is_device_pointer(const void* ptr) {
  hipPointerAttribute_t attr;
  std::memset(&attr, 0, sizeof(attr));
  if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
    (void)hipGetLastError();  // unregistered host memory reports an error: not a device pointer
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// ---- conversions from the reference's POD types (Common/VolumeTypes.hpp:21-100) -------------

template <class CameraT>
avr_camera to_camera(const CameraT& camera) {
  avr_camera out{};
  for (int c = 0; c < 3; ++c) {
    out.eye[c] = static_cast<double>(camera.eye[c]);
    out.look_at[c] = static_cast<double>(camera.lookAt[c]);
    out.up[c] = static_cast<double>(camera.up[c]);
  }
  out.fov_y_degrees = camera.fovYDegrees;
  out.near_plane = camera.nearPlane;
  out.far_plane = camera.farPlane;
  return out;
}

template <class TransformT>
avr_scalar_transform to_transform(const TransformT& transform) {
  avr_scalar_transform out{};
  out.log_scale_input = transform.logScaleInput ? 1 : 0;
  out.normalize_to_unit_range = transform.normalizeToUnitRange ? 1 : 0;
  out.positive_floor = static_cast<double>(transform.positiveFloor);
  out.normalization_min = static_cast<double>(transform.normalizationMin);
  out.inverse_normalization_span = static_cast<double>(transform.inverseNormalizationSpan);
  return out;
}

// volume::AmrBox -> avr_box.  `values` is an amrex::Array4<Real const>: p, jstride, kstride,
// nstride, begin; the address of values(validBox.smallEnd(), component) is taken with ptr().
template <class AmrBoxT>
avr_box to_box(const AmrBoxT& box) {
  avr_box out{};
  for (int c = 0; c < 3; ++c) {
    out.min_corner[c] = static_cast<double>(box.minCorner[c]);
    out.max_corner[c] = static_cast<double>(box.maxCorner[c]);
    out.dims[c] = box.cellDimensions[c];
  }
  const auto lo = box.validBox.smallEnd();
  out.cells = (out.dims[0] > 0 && out.dims[1] > 0 && out.dims[2] > 0)
                  ? box.values.ptr(lo[0], lo[1], lo[2], box.component)
                  : nullptr;
  out.jstride = static_cast<int64_t>(box.values.jstride);
  out.kstride = static_cast<int64_t>(box.values.kstride);
  return out;
}

template <class ColorMapT>
std::vector<avr_colormap_point> to_colormap(const ColorMapT* color_map) {
  std::vector<avr_colormap_point> out;
  if (color_map == nullptr) return out;
  for (const auto& p : *color_map) out.push_back({p.value, p.red, p.green, p.blue, p.alpha});
  return out;
}

// Cells of one box made device-resident for a call (in place when they already are).
class ResidentCells {
 public:
  explicit ResidentCells(avr_box* box) {
    if (box->cells == nullptr || is_device_pointer(box->cells)) return;
    // host Array4: copy the addressed span (strided rows stay strided)
    const std::size_t span = static_cast<std::size_t>(box->dims[0] - 1) +
                             static_cast<std::size_t>(box->dims[1] - 1) * box->jstride +
                             static_cast<std::size_t>(box->dims[2] - 1) * box->kstride + 1;
    staged_.upload(box->cells, span);
    box->cells = staged_.data();
  }

 private:
  DeviceBuffer<double> staged_;
};

// ---- VolumePainter (Common/VolumePainter.hpp:15-32) -----------------------------------------
// Same argument list as the reference.  `image` is the reference's
// ImageRGBAFloatColorDepthSort (host buffer of 5 floats per pixel reachable through
// getColorBuffer(), getWidth(), getHeight()): every pixel is written, as in the reference.
class VolumePainter {
 public:
  explicit VolumePainter(Context& context) : context_(context) {}

  template <class AmrBoxT, class BoundsT, class TransformT, class ImageT, class CameraT,
            class ColorMapT>
  void paint(const AmrBoxT& box, const BoundsT& bounds, const TransformT& scalarTransform,
             const std::pair<float, float>& scalarRange, int /*rank*/, int /*numProcs*/,
             float boxTransparency, int /*antialiasing*/, float referenceSampleDistance,
             ImageT& image, const CameraT& camera, const ColorMapT* colorMap) {
    const int width = image.getWidth();
    const int height = image.getHeight();
    if (width <= 0 || height <= 0) return;  // VolumePainter.cpp:618-622
    avr_box cbox = to_box(box);
    ResidentCells resident(&cbox);
    const avr_scalar_transform transform = to_transform(scalarTransform);
    const avr_camera cam = to_camera(camera);
    const std::vector<avr_colormap_point> map = to_colormap(colorMap);
    avr_paint_params params{};
    params.width = width;
    params.height = height;
    params.scalar_range[0] = scalarRange.first;
    params.scalar_range[1] = scalarRange.second;
    params.box_transparency = boxTransparency;
    params.reference_sample_distance = referenceSampleDistance;
    for (int c = 0; c < 3; ++c) {
      params.bounds_min[c] = static_cast<double>(bounds.minCorner[c]);
      params.bounds_max[c] = static_cast<double>(bounds.maxCorner[c]);
    }
    params.colormap = map.empty() ? nullptr : map.data();
    params.colormap_count = static_cast<int32_t>(map.size());

    const std::size_t floats = static_cast<std::size_t>(width) * height * 5;
    if (layer_.size() < floats) layer_.resize(floats);
    check(avr_paint_box(context_.get(), &cbox, &transform, &params, &cam, layer_.data(), nullptr));
    context_.synchronize();
    layer_.download(image.getColorBuffer(), floats);
  }

 private:
  Context& context_;
  DeviceBuffer<float> layer_;
};

// ---- image blend (Common/ImageColorOnly.hpp:119-199) ------------------------------------------
// kind: 0 = ImageRGBAFloatColorDepthSort, 1 = ImageRGBAFloatColorOnly, 2 = ImageRGBAUByteColorOnly.
// Host buffers in, host buffer out (top covers [tb,te), bottom [bb,be)); returns the region.
inline std::pair<int64_t, int64_t> blend_regions_host(Context& context, int kind, const void* top,
                                                      int64_t tb, int64_t te, const void* bottom,
                                                      int64_t bb, int64_t be,
                                                      std::vector<unsigned char>* out) {
  const std::size_t px = (kind == 0) ? 20 : (kind == 1) ? 16 : 4;
  const int64_t ob = tb < bb ? tb : bb, oe = te > be ? te : be;
  DeviceBuffer<unsigned char> dtop(px * static_cast<std::size_t>(te - tb) + 1),
      dbottom(px * static_cast<std::size_t>(be - bb) + 1),
      dout(px * static_cast<std::size_t>(oe - ob) + 1);
  dtop.upload(static_cast<const unsigned char*>(top), px * static_cast<std::size_t>(te - tb));
  dbottom.upload(static_cast<const unsigned char*>(bottom), px * static_cast<std::size_t>(be - bb));
  check(avr_blend_regions(context.get(), kind, dtop.data(), tb, te, dbottom.data(), bb, be,
                          dout.data()));
  context.synchronize();
  out->resize(px * static_cast<std::size_t>(oe - ob));
  dout.download(out->data(), out->size());
  return {ob, oe};
}

// ---- single-rank Compositor::compose for a LayeredImageInterface ------------------------------
// (Common/Compositor.hpp:19-40, DirectSendBase.cpp:316-458 with one rank: the layers are sorted
// by depth hint and left-folded.)  `layers` exposes getLayerCount(), getLayer(i) (an image with
// getColorBuffer()/getNumberOfPixels()) and getLayerDepthHint(i), as LayeredVolumeImage does.
// Multi-rank compositing goes through the frame plan (avr_frame_plan_*) and RCCL.
template <class LayeredT>
std::vector<float> compose_single_rank(Context& context, LayeredT& layers, int64_t n_pixels) {
  const int count = layers.getLayerCount();
  std::vector<float> hints(static_cast<std::size_t>(count));
  std::vector<int32_t> owner(static_cast<std::size_t>(count), 0), local(static_cast<std::size_t>(count));
  for (int i = 0; i < count; ++i) {
    hints[static_cast<std::size_t>(i)] = layers.getLayerDepthHint(i);
    local[static_cast<std::size_t>(i)] = i;
  }
  std::vector<int32_t> order(static_cast<std::size_t>(count > 0 ? count : 1)),
      run_end(static_cast<std::size_t>(count > 0 ? count : 1));
  int n_runs = 0;
  check(avr_layer_order(hints.data(), owner.data(), local.data(), count, order.data(),
                        run_end.data(), &n_runs));
  const std::size_t floats = static_cast<std::size_t>(n_pixels) * 5;
  std::vector<std::unique_ptr<DeviceBuffer<float>>> device_layers;
  std::vector<const float*> slices;
  for (int l = 0; l < count; ++l) {
    auto buffer = std::make_unique<DeviceBuffer<float>>(floats);
    buffer->upload(layers.getLayer(order[static_cast<std::size_t>(l)])->getColorBuffer(), floats);
    slices.push_back(buffer->data());
    device_layers.push_back(std::move(buffer));
  }
  DeviceBuffer<float> result(floats + 1);
  check(avr_fold_runs_depthsort(context.get(), slices.data(), count, result.data(), n_pixels));
  context.synchronize();
  std::vector<float> out(floats);
  result.download(out.data(), floats);
  return out;
}

// ---- one rank's share of renderSingleTrial (VolumeRenderer/VolumeRenderer.cpp:1200-1314) --------
// The frame of INTEGRATION.md section 3 as a class: the caller supplies the exchange (RCCL
// ncclSend/ncclRecv group, hipMemcpyPeerAsync, or MPI_Alltoallv on device pointers) between
// paint() and fold().  `all_boxes` is the replicated metadata of every box (cells may be null for
// boxes of other ranks), `owner[b]` its rank; this rank's boxes must carry device pointers.
class RankFrame {
 public:
  RankFrame(Context& context, std::vector<avr_box> all_boxes, std::vector<int32_t> owner, int rank,
            int n_ranks, const avr_scalar_transform& transform)
      : context_(context), all_boxes_(std::move(all_boxes)), owner_(std::move(owner)), rank_(rank),
        n_ranks_(n_ranks) {
    std::vector<avr_box> local;
    for (std::size_t b = 0; b < all_boxes_.size(); ++b) {
      if (owner_[b] == rank_) local.push_back(all_boxes_[b]);
    }
    check(avr_scene_create(context_.get(), local.data(), static_cast<int>(local.size()), &transform,
                           &scene_));
    check(avr_visibility_graph_create(all_boxes_.data(), owner_.data(),
                                      static_cast<int>(all_boxes_.size()), n_ranks_, &visibility_));
  }
  ~RankFrame() {
    if (plan_ != nullptr) avr_frame_plan_destroy(plan_);
    if (visibility_ != nullptr) avr_visibility_graph_destroy(visibility_);
    if (scene_ != nullptr) avr_scene_destroy(scene_);
  }
  RankFrame(const RankFrame&) = delete;
  RankFrame& operator=(const RankFrame&) = delete;

  // buildVisibilityOrderedGroup + the layer order / run grouping / exchange layout that
  // composeLayered derives from its allgathers (DirectSendBase.cpp:329-410).
  const avr_frame_plan_info& plan(const avr_paint_params& params, const avr_camera& camera,
                                  bool use_visibility_graph = true) {
    if (plan_ != nullptr) avr_frame_plan_destroy(plan_);
    plan_ = nullptr;
    std::vector<int32_t> group(static_cast<std::size_t>(n_ranks_));
    const float aspect = static_cast<float>(params.width) / static_cast<float>(params.height);
    int succeeded = 1;
    check(avr_visibility_order(visibility_, &camera, aspect, use_visibility_graph ? 1 : 0, nullptr,
                               group.data(), &succeeded, nullptr));
    check(avr_frame_plan_create(all_boxes_.data(), owner_.data(), static_cast<int>(all_boxes_.size()),
                                n_ranks_, rank_, group.data(), &params, &camera, &plan_));
    check(avr_frame_plan_get_info(plan_, &info_));
    send_splits_.assign(static_cast<std::size_t>(n_ranks_), 0);
    recv_splits_.assign(static_cast<std::size_t>(n_ranks_), 0);
    check(avr_frame_plan_splits(plan_, send_splits_.data(), recv_splits_.data()));
    if (send_.size() < static_cast<std::size_t>(info_.send_floats) + 1) {
      send_.resize(static_cast<std::size_t>(info_.send_floats) + 1);
    }
    if (recv_.size() < static_cast<std::size_t>(info_.recv_floats) + 1) {
      recv_.resize(static_cast<std::size_t>(info_.recv_floats) + 1);
    }
    return info_;
  }
  // classify + march of the local boxes, owner-side run fold: the all-to-all's send buffer
  float* paint() {
    check(avr_render_plan(context_.get(), scene_, plan_, send_.data(), nullptr));
    return send_.data();
  }
  float* paint_buffer() { return send_.data(); }
  float* recv_buffer() { return recv_.data(); }
  const std::vector<int64_t>& send_splits() const { return send_splits_; }  // floats per peer
  const std::vector<int64_t>& recv_splits() const { return recv_splits_; }
  const avr_frame_plan_info& info() const { return info_; }
  // receiver-side fold of this rank's pixel piece (+ RGB8) from the received blocks
  void fold(DeviceBuffer<float>* piece, DeviceBuffer<unsigned char>* rgb8) {
    const std::size_t n = static_cast<std::size_t>(info_.piece_end - info_.piece_begin);
    if (piece->size() < n * 5 + 1) piece->resize(n * 5 + 1);
    if (rgb8 != nullptr && rgb8->size() < n * 3 + 1) rgb8->resize(n * 3 + 1);
    check(avr_fold_plan(context_.get(), plan_, recv_.data(), piece->data(),
                        rgb8 != nullptr ? rgb8->data() : nullptr));
  }

 private:
  Context& context_;
  std::vector<avr_box> all_boxes_;
  std::vector<int32_t> owner_;
  int rank_, n_ranks_;
  avr_scene* scene_ = nullptr;
  avr_visibility_graph* visibility_ = nullptr;
  avr_frame_plan* plan_ = nullptr;
  avr_frame_plan_info info_{};
  std::vector<int64_t> send_splits_, recv_splits_;
  DeviceBuffer<float> send_, recv_;
};

}  // namespace avr

#endif  // AVR_REFERENCE_API_HPP
