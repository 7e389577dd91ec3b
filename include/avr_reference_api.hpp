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
template <class ConcreteT, class LayeredT>
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
    // (getLayer hands out the base class `Image`, Common/LayeredImageInterface.hpp:17-20)
    buffer->upload(static_cast<ConcreteT*>(layers.getLayer(order[static_cast<std::size_t>(l)]))
                       ->getColorBuffer(),
                   floats);
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
//   void allgather_bytes(const void* mine, void* all, int bytes);          // MPI_Allgather, MPI_BYTE
//   std::vector<int> group_ranks(GroupT group);   // ranks of the ordered group, MPI_Group_translate_ranks
// allgather_bytes also becomes the communicator's own control plane (avr_comm_set_control): the
// agreement check of every new frame plan and the decisions of the frame driver's co-run search
// travel over it, so `control` must outlive the Communicator.
class Communicator {
 public:
  Communicator() = default;
  explicit Communicator(avr_comm* adopted) : comm_(adopted) {}
  // One RCCL communicator per rank (= per GPU); collective over `control`.
  template <class Control>
  Communicator(Control& control, int device) {
    // [id | ok flag]: the broadcast takes place whatever happened on rank 0 -- the other ranks are
    // already waiting in it, and a rank 0 that threw first would leave them there
    char id[AVR_COMM_ID_BYTES + 1];
    std::memset(id, 0, sizeof(id));
    std::string failure;
    if (control.rank() == 0) {
      if (avr_comm_unique_id(id) == AVR_OK) {
        id[AVR_COMM_ID_BYTES] = 1;
      } else {
        failure = avr_last_error();
      }
    }
    control.broadcast(id, AVR_COMM_ID_BYTES + 1, 0);
    if (id[AVR_COMM_ID_BYTES] != 1) {
      throw std::runtime_error(failure.empty() ? "rank 0 could not create the communicator id"
                                               : failure);
    }
    check(avr_comm_create(device, id, control.rank(), control.size(), &comm_));
    use_control(control);
  }
  // The small host-side agreements of an N-rank frame go through the caller's control plane.
  template <class Control>
  void use_control(Control& control) {
    check(avr_comm_set_control(
        comm_,
        [](void* user, const void* mine, void* all, int bytes) -> int {
          try {
            static_cast<Control*>(user)->allgather_bytes(mine, all, bytes);
            return 0;
          } catch (...) {
            return 1;
          }
        },
        &control));
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
// and DirectSendBase's behaviour (DirectSend/Base/DirectSendBase.cpp:285-458):
//  * a localImage that offers the LayeredImageInterface (getLayerCount / getLayer /
//    getLayerDepthHint / createEmptyLayer, Common/LayeredImageInterface.hpp:9-28) is composited by
//    global depth order with same-owner runs folded on the owner (composeLayered, :316-458);
//  * any other image takes the classic direct send (:257-281, 76-255): every rank's image is cut
//    into pieces, piece k goes to the rank at position k of the group, which blends what it
//    receives in group order (the lower position on top).
// Either way the result is this rank's piece [k * floor(P/N), (k+1) * floor(P/N)) (k = position in
// the ordered group) of the fully composited image, as a new image of the input's type.
//
// The images are host images as in the reference (ImageColorOnly<Features>: getColorBuffer());
// they are staged to HBM, exchanged over RCCL and blended there.  A layer whose getColorBuffer()
// already IS device memory is used in place.  Every buffer the compositor needs -- the device
// staging of the layers, a two-slot pinned upload ring, send / receive / piece buffers -- is
// pooled in the object and only ever grows: after the first frame of a given shape a compose
// allocates nothing (allocations() counts; tests/cxx/adapter_test.cpp asserts it).  Uploads are
// asynchronous: while layer i travels to the GPU, layer i+1 is copied into the other pinned slot.
// This is the drop-in for code that already has painted layers; a renderer that lets this library
// paint as well uses FrameDriver below and never materialises per-box layers.

// Grow-only buffers; every (re)allocation is counted by the owner.
template <typename T>
class PooledDeviceBuffer {
 public:
  PooledDeviceBuffer() = default;
  ~PooledDeviceBuffer() {
    if (ptr_ != nullptr) (void)hipFree(ptr_);
  }
  PooledDeviceBuffer(const PooledDeviceBuffer&) = delete;
  PooledDeviceBuffer& operator=(const PooledDeviceBuffer&) = delete;
  // at least `count` elements (contents are not kept when it grows)
  T* reserve(std::size_t count, std::size_t* allocations) {
    if (count > capacity_) {
      if (ptr_ != nullptr) (void)hipFree(ptr_);
      ptr_ = nullptr;
      capacity_ = 0;
      hip_ok(hipMalloc(reinterpret_cast<void**>(&ptr_), count * sizeof(T)), "hipMalloc(pool)");
      capacity_ = count;
      ++*allocations;
    }
    return ptr_;
  }

 private:
  T* ptr_ = nullptr;
  std::size_t capacity_ = 0;
};

class PooledPinnedBuffer {
 public:
  PooledPinnedBuffer() = default;
  ~PooledPinnedBuffer() {
    if (ptr_ != nullptr) (void)hipHostFree(ptr_);
    if (free_ != nullptr) (void)hipEventDestroy(free_);
  }
  PooledPinnedBuffer(const PooledPinnedBuffer&) = delete;
  PooledPinnedBuffer& operator=(const PooledPinnedBuffer&) = delete;
  void* reserve(std::size_t bytes, std::size_t* allocations) {
    if (bytes > capacity_) {
      wait();
      if (ptr_ != nullptr) (void)hipHostFree(ptr_);
      ptr_ = nullptr;
      capacity_ = 0;
      hip_ok(hipHostMalloc(&ptr_, bytes, hipHostMallocDefault), "hipHostMalloc(pool)");
      capacity_ = bytes;
      ++*allocations;
    }
    return ptr_;
  }
  // the copy out of this slot has been queued on `stream`: the slot is free when it has run
  void in_flight(hipStream_t stream) {
    if (free_ == nullptr) hip_ok(hipEventCreateWithFlags(&free_, hipEventDisableTiming), "hipEventCreate");
    hip_ok(hipEventRecord(free_, stream), "hipEventRecord");
    busy_ = true;
  }
  void wait() {
    if (busy_) hip_ok(hipEventSynchronize(free_), "hipEventSynchronize");
    busy_ = false;
  }

 private:
  void* ptr_ = nullptr;
  std::size_t capacity_ = 0;
  hipEvent_t free_ = nullptr;
  bool busy_ = false;
};

// Which Features::blend an image type carries (Common/ImageColorOnly.hpp:36-38: ColorType and
// ColorVecSize are public members of every ImageColorOnly<Features>).
template <class ConcreteT>
constexpr int blend_kind_of() {
  using Color = typename ConcreteT::ColorType;
  if (std::is_same<Color, float>::value && ConcreteT::ColorVecSize == 5) return 0;  // depth sort
  if (std::is_same<Color, float>::value && ConcreteT::ColorVecSize == 4) return 1;  // RGBA float
  if (sizeof(Color) == 4 && ConcreteT::ColorVecSize == 1) return 2;                  // RGBA ubyte
  return -1;
}

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

  // (re)allocations of pooled buffers so far: constant once the frames' shape has been seen
  std::size_t allocations() const { return allocations_; }

  // Compositor::compose.  ConcreteT: the image type that carries the pixels -- the layers' type
  // for a layered image (the reference's ImageRGBAFloatColorDepthSort; getLayer and
  // createEmptyLayer hand out its base class `Image` there), the image's own type otherwise.
  // ImageT: what the caller holds (the reference's `Image`); LayeredT: the layered image class it
  // may turn out to be (the reference's LayeredVolumeImage, found by dynamic_cast as in
  // DirectSendBase.cpp:288-298).
  template <class ConcreteT, class LayeredT, class ImageT, class GroupT, class CommT>
  std::unique_ptr<ImageT> compose(ImageT* localImage, GroupT group, CommT communicator) {
    if (localImage == nullptr) throw std::invalid_argument("compose: null image");
    if (auto* layered = dynamic_cast<LayeredT*>(localImage)) {
      if constexpr (blend_kind_of<ConcreteT>() == 0) {
        return composeLayered<ConcreteT>(layered, group, communicator);
      } else {
        throw std::runtime_error("HipDirectSend: a layered image holds depth-sorted RGBA float layers");
      }
    }
    auto* plain = dynamic_cast<ConcreteT*>(localImage);
    if (plain == nullptr) {
      throw std::runtime_error("HipDirectSend: the image is neither layered nor of the expected type");
    }
    return composeImage(plain, group, communicator);
  }

  // composeLayered (DirectSendBase.cpp:316-458) of a LayeredImageInterface.
  template <class ConcreteT, class LayeredT, class GroupT, class CommT>
  auto composeLayered(LayeredT* localImage, GroupT group, CommT /*communicator*/)
      -> decltype(localImage->createEmptyLayer(0, 0)) {
    if (localImage == nullptr) throw std::invalid_argument("compose: null image");
    static_assert(blend_kind_of<ConcreteT>() == 0,
                  "layered compositing blends depth-sorted RGBA float layers");
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
    // every rank built its plan from the same allgathered hints: checked before anything is
    // queued, because an exchange whose two sides disagree on a block size never ends (the
    // reference learns the sizes from a metadata message per transfer, Common/Image.cpp:62-90)
    if (n_ranks > 1) check(avr_frame_plan_agree(plan, comm_->get(), context_.get(), 0));
    // the local layers in HBM (device-resident ones in place), then the owner-side run fold into
    // the send layout
    const std::size_t floats = static_cast<std::size_t>(width) * height * 5;
    hipStream_t stream = static_cast<hipStream_t>(avr_context_stream(context_.get()));
    std::vector<const float*> pointers(static_cast<std::size_t>(local_count), nullptr);
    std::size_t staged = 0;
    for (int i = 0; i < local_count; ++i) {
      const float* pixels = static_cast<ConcreteT*>(localImage->getLayer(i))->getColorBuffer();
      if (is_device_pointer(pixels)) {
        pointers[static_cast<std::size_t>(i)] = pixels;
      } else {
        ++staged;
      }
    }
    float* staging = layers_.reserve(staged * floats, &allocations_);
    std::size_t slot = 0, at = 0;
    for (int i = 0; i < local_count; ++i) {
      if (pointers[static_cast<std::size_t>(i)] != nullptr) continue;
      const float* pixels = static_cast<ConcreteT*>(localImage->getLayer(i))->getColorBuffer();
      float* twin = staging + at * floats;
      upload(pixels, twin, floats * sizeof(float), &slot, stream);
      pointers[static_cast<std::size_t>(i)] = twin;
      ++at;
    }
    float* send = send_.reserve(static_cast<std::size_t>(info.send_floats) + 1, &allocations_);
    float* recv = recv_.reserve(static_cast<std::size_t>(info.recv_floats) + 1, &allocations_);
    check(avr_pack_layers(context_.get(), plan, pointers.data(), local_count, send));
    // (the rank's block for itself is not copied: the fold reads it from the send buffer)
    check(avr_exchange_peers(context_.get(), plan, comm_->get(), send, recv));
    const std::size_t piece_pixels = static_cast<std::size_t>(info.piece_end - info.piece_begin);
    float* piece = piece_.reserve(piece_pixels * 5 + 1, &allocations_);
    check(avr_fold_plan_own(context_.get(), plan, recv, send, piece, nullptr));
    // the result image: this rank's pixel range, created like the reference's empty layer
    auto result = localImage->createEmptyLayer(static_cast<int>(info.piece_begin),
                                               static_cast<int>(info.piece_end));
    download(piece, static_cast<ConcreteT*>(result.get())->getColorBuffer(),
             piece_pixels * 5 * sizeof(float), stream);
    return result;
  }

  // The classic direct send of one image per rank (DirectSendBase.cpp:257-281 with the receive
  // group = the whole group, :300-311): piece k of every rank's image meets on the rank at group
  // position k and is blended there in group order, the lower position on top
  // (ProcessIncomingImages, :179-255, blends whichever neighbours have arrived; for the float
  // types that association order is the reference's one degree of freedom -- here it is always
  // the left fold).
  template <class ConcreteT, class GroupT, class CommT>
  auto composeImage(ConcreteT* localImage, GroupT group, CommT /*communicator*/)
      -> decltype(localImage->createNew(0, 0)) {
    constexpr int kind = blend_kind_of<ConcreteT>();
    static_assert(kind >= 0, "no Features::blend of this image type is implemented on the GPU");
    using Color = typename ConcreteT::ColorType;
    constexpr std::size_t pixel_bytes = sizeof(Color) * ConcreteT::ColorVecSize;
    if (localImage->getRegionBegin() != 0 ||
        localImage->getRegionEnd() != localImage->getWidth() * localImage->getHeight()) {
      throw std::invalid_argument("HipDirectSend: the local image must hold the whole frame");
    }
    const int n_ranks = control_.size(), rank = control_.rank();
    const std::vector<int> ordered = control_.group_ranks(group);
    if (static_cast<int>(ordered.size()) != n_ranks) {
      throw std::invalid_argument("HipDirectSend: the group must hold every rank of the communicator");
    }
    std::vector<int32_t> group_order(ordered.begin(), ordered.end());
    int position = 0;
    while (position < n_ranks && group_order[static_cast<std::size_t>(position)] != rank) ++position;
    const int64_t n_pixels = static_cast<int64_t>(localImage->getWidth()) * localImage->getHeight();
    int64_t begin = 0, end = 0;
    check(avr_piece_range(n_pixels, position, n_ranks, &begin, &end));
    const std::size_t piece_bytes = static_cast<std::size_t>(end - begin) * pixel_bytes;
    hipStream_t stream = static_cast<hipStream_t>(avr_context_stream(context_.get()));
    const Color* pixels = localImage->getColorBuffer();
    const void* image = pixels;
    if (!is_device_pointer(pixels)) {
      char* twin = image_.reserve(static_cast<std::size_t>(n_pixels) * pixel_bytes + 1, &allocations_);
      std::size_t slot = 0;
      upload(pixels, twin, static_cast<std::size_t>(n_pixels) * pixel_bytes, &slot, stream);
      image = twin;
    }
    char* slices = slices_.reserve(piece_bytes * static_cast<std::size_t>(n_ranks) + 1, &allocations_);
    check(avr_exchange_pieces(context_.get(), comm_->get(), group_order.data(), n_pixels,
                              static_cast<int>(pixel_bytes), image, slices));
    // left fold in group order: ((slice 0 over slice 1) over slice 2) ...
    const char* top = slices;
    char* scratch[2] = {fold_[0].reserve(piece_bytes + 1, &allocations_),
                        fold_[1].reserve(piece_bytes + 1, &allocations_)};
    for (int j = 1; j < n_ranks && end > begin; ++j) {
      char* out = scratch[j & 1];
      check(avr_blend_regions(context_.get(), kind, top, begin, end, slices + piece_bytes * j, begin,
                              end, out));
      top = out;
    }
    auto result = localImage->createNew(static_cast<int>(begin), static_cast<int>(end));
    download(top, static_cast<ConcreteT*>(result.get())->getColorBuffer(), piece_bytes, stream);
    return result;
  }

 private:
  // host -> HBM through the two-slot pinned ring, in chunks of at most 64 MiB: the CPU fills one
  // slot while the DMA engine drains the other
  void upload(const void* host, void* device, std::size_t bytes, std::size_t* slot, hipStream_t stream) {
    constexpr std::size_t kChunk = std::size_t{64} << 20;
    const char* src = static_cast<const char*>(host);
    char* dst = static_cast<char*>(device);
    for (std::size_t done = 0; done < bytes;) {
      const std::size_t n = (bytes - done < kChunk) ? bytes - done : kChunk;
      PooledPinnedBuffer& pinned = pinned_[*slot & 1];
      pinned.wait();
      void* block = pinned.reserve(n < kChunk && bytes > kChunk ? kChunk : n, &allocations_);
      std::memcpy(block, src + done, n);
      hip_ok(hipMemcpyAsync(dst + done, block, n, hipMemcpyHostToDevice, stream), "hipMemcpyAsync");
      pinned.in_flight(stream);
      ++*slot;
      done += n;
    }
  }
  void download(const void* device, void* host, std::size_t bytes, hipStream_t stream) {
    if (bytes != 0) {
      hip_ok(hipMemcpyAsync(host, device, bytes, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
    }
    hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
  }

  Control& control_;
  Context context_;
  std::unique_ptr<Communicator> owned_;
  Communicator* comm_ = nullptr;
  std::size_t allocations_ = 0;
  PooledDeviceBuffer<float> layers_, send_, recv_, piece_;
  PooledDeviceBuffer<char> image_, slices_, fold_[2];
  PooledPinnedBuffer pinned_[2];
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
  // Plans a frame ahead of time (host geometry only; avr_renderer_prepare): for a camera path
  // that never repeats, call it for frame f + 1 on another thread -- std::async -- while frame f
  // is being queued by render().
  void prepare(const avr_render_params& params, const avr_camera& camera,
               const int32_t* group_order = nullptr) {
    check(avr_renderer_prepare(renderer_, &params, &camera, group_order));
  }
  void synchronize() { check(avr_renderer_synchronize(renderer_)); }
  // How many of the frames rendered so far have their outputs written once the compositing stream
  // has passed what is queued now (ranks of several: the last frame's bytes travel with the next
  // frame's round, avr_renderer_set_deferred_gather) -- for a caller that orders its own stream
  // after avr_renderer_stream(get(), 2) instead of calling the collective synchronize().
  uint64_t outputs_complete() const {
    uint64_t complete = 0;
    check(avr_renderer_outputs_complete(renderer_, &complete, nullptr));
    return complete;
  }
  // NULL, or what did not finish within the deadline (avr_set_frame_timeout_ms): the renderer is
  // then failed for good -- report it (MPI_Abort in the reference's main, main.cpp:27-33) and exit.
  const char* failure() const { return avr_renderer_failure(renderer_); }

 private:
  avr_renderer* renderer_ = nullptr;
};

}  // namespace avr

#endif  // AVR_REFERENCE_API_HPP
