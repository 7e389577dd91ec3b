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
#include <type_traits>
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

// ---- rank communicator ---------------------------------------------------------------------------
// RAII over avr_comm.  `Control` is the caller's control plane for the three tiny host collectives
// the compositor needs besides the GPU exchange; inside the reference tree it is MPI
// (INTEGRATION.md shows the ten-line struct), in this repository's tests a thread barrier:
//   int  rank() const;  int size() const;
//   void broadcast(void* data, int bytes, int root);                       // MPI_Bcast
//   void allgather_int(int value, int* out);                               // MPI_Allgather
//   void allgatherv_float(const float* in, int n, float* out, const int* counts,
//                         const int* displs);                              // MPI_Allgatherv
//   std::vector<int> group_ranks(GroupT group);   // ranks of the ordered group, MPI_Group_translate_ranks
class Communicator {
 public:
  Communicator() = default;
  explicit Communicator(avr_comm* adopted) : comm_(adopted) {}
  // One RCCL communicator per rank (= per GPU); collective over `control`.
  template <class Control>
  Communicator(Control& control, int device) {
    char id[AVR_COMM_ID_BYTES];
    std::memset(id, 0, sizeof(id));
    if (control.rank() == 0) check(avr_comm_unique_id(id));
    control.broadcast(id, AVR_COMM_ID_BYTES, 0);
    check(avr_comm_create(device, id, control.rank(), control.size(), &comm_));
  }
  // n connected in-process communicators (one GPU, one host thread per rank): rehearsal only.
  static std::vector<std::unique_ptr<Communicator>> local(int n_ranks) {
    std::vector<avr_comm*> raw(static_cast<std::size_t>(n_ranks), nullptr);
    check(avr_comm_create_local(n_ranks, raw.data()));
    std::vector<std::unique_ptr<Communicator>> out;
    for (avr_comm* c : raw) out.push_back(std::make_unique<Communicator>(c));
    return out;
  }
  ~Communicator() { avr_comm_destroy(comm_); }
  Communicator(const Communicator&) = delete;
  Communicator& operator=(const Communicator&) = delete;
  avr_comm* get() const { return comm_; }
  int rank() const { return avr_comm_rank(comm_); }
  int size() const { return avr_comm_size(comm_); }

 private:
  avr_comm* comm_ = nullptr;
};

// ---- Compositor plugin (Common/Compositor.hpp:19-40) -------------------------------------------
// HipDirectSend::compose has the reference's call shape
//     std::unique_ptr<Image> compose(Image* localImage, MPI_Group group, MPI_Comm communicator)
// and DirectSendBase's behaviour (DirectSend/Base/DirectSendBase.cpp:285-458): a localImage that
// offers the LayeredImageInterface (getLayerCount / getLayer / getLayerDepthHint /
// createEmptyLayer, Common/LayeredImageInterface.hpp:9-28) is composited by global depth order
// with same-owner runs folded on the owner; the result is this rank's piece
// [k * floor(P/N), (k+1) * floor(P/N)) (k = position in the ordered group) of the fully composited
// image, as a new image of the layers' type.  The layers are host images as in the reference
// (ImageRGBAFloatColorDepthSort: getColorBuffer(), 5 floats per pixel); they are staged to HBM,
// exchanged over RCCL and folded there.  This is the drop-in for code that already has painted
// layers; a renderer that lets this library paint as well uses FrameDriver below and never
// materialises per-box layers.
template <class Control>
class HipDirectSend {
 public:
  // one RCCL communicator per rank, built over the control plane (collective)
  HipDirectSend(Control& control, int device)
      : control_(control), context_(device), owned_(std::make_unique<Communicator>(control, device)),
        comm_(owned_.get()) {}
  // ... or an existing communicator (e.g. the in-process rehearsal one)
  HipDirectSend(Control& control, int device, Communicator* comm)
      : control_(control), context_(device), comm_(comm) {}

  // ConcreteT: the layers' image type (the reference's ImageRGBAFloatColorDepthSort; getLayer
  // and createEmptyLayer hand out its base class `Image` there).  LayeredT: an image that is
  // also a LayeredImageInterface (the reference's LayeredVolumeImage).
  template <class ConcreteT, class LayeredT, class GroupT, class CommT>
  auto compose(LayeredT* localImage, GroupT group, CommT /*communicator*/)
      -> decltype(localImage->createEmptyLayer(0, 0)) {
    if (localImage == nullptr) throw std::invalid_argument("compose: null image");
    const int n_ranks = control_.size(), rank = control_.rank();
    const int local_count = localImage->getLayerCount();
    // MPI_Allgather of the layer counts, MPI_Allgatherv of the depth hints (:329-361)
    std::vector<int> counts(static_cast<std::size_t>(n_ranks)), displs(static_cast<std::size_t>(n_ranks));
    control_.allgather_int(local_count, counts.data());
    int total = 0;
    for (int r = 0; r < n_ranks; ++r) {
      displs[static_cast<std::size_t>(r)] = total;
      total += counts[static_cast<std::size_t>(r)];
    }
    std::vector<float> mine(static_cast<std::size_t>(local_count > 0 ? local_count : 1));
    for (int i = 0; i < local_count; ++i) mine[static_cast<std::size_t>(i)] = localImage->getLayerDepthHint(i);
    std::vector<float> hints(static_cast<std::size_t>(total > 0 ? total : 1));
    control_.allgatherv_float(mine.data(), local_count, hints.data(), counts.data(), displs.data());
    std::vector<int32_t> owner(static_cast<std::size_t>(total > 0 ? total : 1));
    for (int r = 0; r < n_ranks; ++r) {
      for (int i = 0; i < counts[static_cast<std::size_t>(r)]; ++i) {
        owner[static_cast<std::size_t>(displs[static_cast<std::size_t>(r)] + i)] = r;
      }
    }
    const std::vector<int> ordered = control_.group_ranks(group);
    std::vector<int32_t> group_order(ordered.begin(), ordered.end());
    const int width = localImage->getWidth(), height = localImage->getHeight();
    avr_frame_plan* plan = nullptr;
    check(avr_layered_plan_create(hints.data(), owner.data(), total, n_ranks, rank, group_order.data(),
                                  width, height, &plan));
    std::unique_ptr<avr_frame_plan, void (*)(avr_frame_plan*)> plan_guard(plan, avr_frame_plan_destroy);
    avr_frame_plan_info info{};
    check(avr_frame_plan_get_info(plan, &info));
    // stage the local layers, fold the local runs into the send layout
    const std::size_t floats = static_cast<std::size_t>(width) * height * 5;
    std::vector<std::unique_ptr<DeviceBuffer<float>>> staged;
    std::vector<const float*> pointers;
    for (int i = 0; i < local_count; ++i) {
      auto buffer = std::make_unique<DeviceBuffer<float>>(floats);
      buffer->upload(static_cast<ConcreteT*>(localImage->getLayer(i))->getColorBuffer(), floats);
      pointers.push_back(buffer->data());
      staged.push_back(std::move(buffer));
    }
    DeviceBuffer<float> send(static_cast<std::size_t>(info.send_floats) + 1),
        recv(static_cast<std::size_t>(info.recv_floats) + 1);
    check(avr_pack_layers(context_.get(), plan, pointers.data(), local_count, send.data()));
    check(avr_exchange(context_.get(), plan, comm_->get(), send.data(), recv.data()));
    const std::size_t piece_pixels = static_cast<std::size_t>(info.piece_end - info.piece_begin);
    DeviceBuffer<float> piece(piece_pixels * 5 + 1);
    check(avr_fold_plan(context_.get(), plan, recv.data(), piece.data(), nullptr));
    context_.synchronize();
    // the result image: this rank's pixel range, created like the reference's empty layer
    auto result = localImage->createEmptyLayer(static_cast<int>(info.piece_begin),
                                               static_cast<int>(info.piece_end));
    piece.download(static_cast<ConcreteT*>(result.get())->getColorBuffer(), piece_pixels * 5);
    return result;
  }

 private:
  Control& control_;
  Context context_;
  std::unique_ptr<Communicator> owned_;
  Communicator* comm_ = nullptr;
};

// ---- frame driver: one rank's share of renderSingleTrial, pipelined (avr_renderer) --------------
// all_boxes / owner: the replicated metadata of every box (this rank's boxes with device cell
// pointers).  render() returns immediately; outputs are complete after synchronize().
class FrameDriver {
 public:
  FrameDriver(int device, int rank, int n_ranks, Communicator* comm, const std::vector<avr_box>& all_boxes,
              const std::vector<int32_t>& owner, const avr_scalar_transform& transform,
              const double bounds_min[3], const double bounds_max[3], float range_min = 0.0f,
              float range_max = 1.0f, const std::vector<avr_colormap_point>& colormap = {}) {
    const float range[2] = {range_min, range_max};
    check(avr_renderer_create(device, rank, n_ranks, comm != nullptr ? comm->get() : nullptr,
                              all_boxes.data(), owner.data(), static_cast<int>(all_boxes.size()),
                              &transform, bounds_min, bounds_max, range,
                              colormap.empty() ? nullptr : colormap.data(),
                              static_cast<int>(colormap.size()), &renderer_));
  }
  ~FrameDriver() { avr_renderer_destroy(renderer_); }
  FrameDriver(const FrameDriver&) = delete;
  FrameDriver& operator=(const FrameDriver&) = delete;
  avr_renderer* get() const { return renderer_; }
  // rgb8_out / image_out: device buffers on rank 0 (W*H*3 bytes rows top-down; W*H*5 floats), else null
  // want_image must be the same on every rank (it adds a gather of the float pieces)
  void render(const avr_render_params& params, const avr_camera& camera, unsigned char* rgb8_out,
              bool want_image = false, float* image_out = nullptr,
              const int32_t* group_order = nullptr, uint64_t* samples_out = nullptr) {
    check(avr_renderer_render(renderer_, &params, &camera, group_order, nullptr, samples_out,
                              want_image ? 1 : 0, rgb8_out, image_out));
  }
  void synchronize() { check(avr_renderer_synchronize(renderer_)); }

 private:
  avr_renderer* renderer_ = nullptr;
};

}  // namespace avr

#endif  // AVR_REFERENCE_API_HPP
