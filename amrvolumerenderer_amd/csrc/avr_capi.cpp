// C ABI of the MI355X volume-rendering path (include/avr_hip.h).  Thin: validates, runs the
// host prologue (avr_host.cpp), stages the per-frame descriptors and launches the kernels
// (avr_kernels.hip) on the context's stream.  No CPU compute fallback exists behind it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <exception>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/avr_hip_debug.h"
#include "avr_internal.h"
#include "avr_plan.h"

namespace avr {

static_assert((hipEventDisableTiming | hipEventDisableSystemFence) == (0x2u | 0x20000000u),
              "avr::ordering_event_flags spells the flags out (avr_internal.h has no HIP header)");

namespace {
thread_local std::string g_last_error;
}

void set_error(const std::string& message) { g_last_error = message; }

namespace {

class HipFailure : public std::runtime_error {
 public:
  using std::runtime_error::runtime_error;
};

void hip_check(hipError_t err, const char* what) {
  if (err != hipSuccess) {
    throw HipFailure(std::string(what) + ": " + hipGetErrorString(err));
  }
}

}  // namespace

// Every host wait on device work has a deadline (AVR_FRAME_TIMEOUT_MS, default 30 s; 0 = none): the
// frame of a rank of several contains collectives, and a peer that died, or whose calls differ from
// this rank's, would otherwise leave the wait -- and the job -- hanging.  The reference's exchange
// either completes or errors (MPI_Waitany / MPI_Waitall, DirectSendBase.cpp:206-220, 277); so does
// this one: the wait throws, the C ABI call returns AVR_ERR_RUNTIME naming what did not finish.
namespace {
std::atomic<int> g_timeout_override{-1};
thread_local int t_collective_depth = 0;
}
void set_frame_timeout_ms(int ms) { g_timeout_override.store(ms < 0 ? -1 : ms); }
CollectiveScope::CollectiveScope(bool collective) : on_(collective) {
  if (on_) ++t_collective_depth;
}
CollectiveScope::~CollectiveScope() {
  if (on_) --t_collective_depth;
}
// A deadline somebody asked for (avr_set_frame_timeout_ms, AVR_FRAME_TIMEOUT_MS) holds for every
// wait; the DEFAULT of 30 s only where a collective is involved (a renderer of several ranks, a
// communicator's own calls): a single-rank context waiting for a deep queue of long frames, or
// for a GPU it shares with other processes, waits as long as it takes.
int frame_timeout_ms() {
  const int forced = g_timeout_override.load(std::memory_order_relaxed);
  if (forced >= 0) return forced;
  static const int value = [] {
    const char* text = std::getenv("AVR_FRAME_TIMEOUT_MS");
    if (text == nullptr || text[0] == '\0') return -1;
    const long parsed = std::strtol(text, nullptr, 10);
    return static_cast<int>(std::min<long>(std::max<long>(parsed, 0), 3600000));
  }();
  if (value >= 0) return value;
  return t_collective_depth > 0 ? 30000 : 0;
}

// Host waits poll the event instead of blocking in hipEventSynchronize / hipStreamSynchronize:
// a blocking wait that outlasts the runtime's spin phase sleeps on an interrupt, and waking from
// it was measured to take milliseconds on this platform -- by then a pipelined renderer's queue
// has run dry (5-7 ms stalls every few frames at 1.3 ms per frame, tools/host_stalls.py).
void wait_event_deadline(void* event_v, const char* what) {
  hipEvent_t event = static_cast<hipEvent_t>(event_v);
  const int limit_ms = frame_timeout_ms();
  std::chrono::steady_clock::time_point deadline{};
  for (unsigned spins = 0;; ++spins) {
    const hipError_t status = hipEventQuery(event);
    if (status == hipSuccess) return;
    if (status != hipErrorNotReady) hip_check(status, what);
    (void)hipGetLastError();  // hipErrorNotReady is not an error here
    if (spins > 64) std::this_thread::yield();
    if (limit_ms > 0 && (spins & 255u) == 255u) {
      const auto now = std::chrono::steady_clock::now();
      if (deadline == std::chrono::steady_clock::time_point{}) {
        deadline = now + std::chrono::milliseconds(limit_ms);
      } else if (now > deadline) {
        throw DeadlineExceeded(std::string(what) + " did not finish within " +
                               std::to_string(limit_ms) + " ms (AVR_FRAME_TIMEOUT_MS)");
      }
    }
  }
}

// Waits until everything queued on `stream` so far has finished.
void wait_stream_deadline(void* stream_v, const char* what) {
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  hipEvent_t event = nullptr;
  hip_check(hipEventCreateWithFlags(&event, hipEventDisableTiming), "hipEventCreate");
  const hipError_t recorded = hipEventRecord(event, stream);
  if (recorded != hipSuccess) {
    (void)hipEventDestroy(event);
    hip_check(recorded, what);
  }
  try {
    wait_event_deadline(event, what);
  } catch (const DeadlineExceeded&) {
    throw;  // the event is still pending on a stream that does not move: it is leaked, not destroyed
  } catch (...) {
    (void)hipEventDestroy(event);
    throw;
  }
  (void)hipEventDestroy(event);
}

namespace {

void wait_event(hipEvent_t event, const char* what) { wait_event_deadline(event, what); }
void wait_stream(hipStream_t stream, const char* what) { wait_stream_deadline(stream, what); }

// Per-call descriptors (box table, transfer-function tables, run tables, ...) travel to the
// device in ONE asynchronous copy per call: they are packed into a pinned host block and copied
// to its device twin by a small kernel.  A ring of blocks lets the host run several calls ahead of the GPU
// without waiting (a block is re-used only after the copy that read it has completed).
class StagingRing {
 public:
  // Nine: a speculative frame (avr_renderer.cpp) stages three batches on the march's context -- the
  // checking march, the gated classify pass, the gated march -- whose flag buffers rotate with the
  // frame's slot, so the batches repeat with period nine: every slot then always sees the same
  // batch (not copied again, below), and the host may run three such frames ahead.  (With four the
  // host waited 0.38 ms per frame here for a block of the frame before.)
  static constexpr int kSlots = 9;
  static constexpr size_t kAlign = 256;

  ~StagingRing() { release(); }

  void release() {
    for (Slot& slot : slots_) {
      if (slot.dev != nullptr) (void)hipFree(slot.dev);
      if (slot.host != nullptr) (void)hipHostFree(slot.host);
      if (slot.done != nullptr) (void)hipEventDestroy(slot.done);
      slot = Slot{};
    }
  }

  // The owner keeps the host from running ahead by other means (the frame driver's host-side
  // back-pressure): a skipped copy then leaves no packet at all on the consumer stream.
  void set_lean(bool lean) { lean_ = lean; }

  // Starts a batch with room for `bytes` in `items` arrays.
  // How many batches the host may run ahead of the consumer (1 .. kSlots; default 4: what the
  // ring was when it had four blocks -- the frame driver's feedback loops, the co-run search among
  // them, are timed for that lead; a speculating march context takes all nine).
  void set_ahead(int batches) { ahead_ = std::min(std::max(batches, 1), kSlots); }

  void begin(size_t bytes, int items) {
    if (ahead_ < kSlots) {
      Slot& behind = slots_[(next_ + kSlots - ahead_) % kSlots];
      if (behind.pending) {
        wait_event(behind.done, "hipEventQuery(staging)");
        behind.pending = false;
      }
    }
    current_ = &slots_[next_];
    next_ = (next_ + 1) % kSlots;
    if (current_->done == nullptr) {
      hip_check(hipEventCreateWithFlags(&current_->done, avr::ordering_event_flags()), "hipEventCreate");
    }
    if (current_->pending) {
      wait_event(current_->done, "hipEventQuery(staging)");
      current_->pending = false;
    }
    const size_t need = bytes + static_cast<size_t>(items + 1) * kAlign;
    if (need > current_->capacity) {
      if (current_->dev != nullptr) (void)hipFree(current_->dev);
      if (current_->host != nullptr) (void)hipHostFree(current_->host);
      current_->dev = current_->host = nullptr;
      current_->capacity = 0;
      current_->shadow.clear();
      size_t cap = 1 << 16;
      while (cap < need) cap *= 2;
      hip_check(hipMalloc(&current_->dev, cap), "hipMalloc(staging)");
      hip_check(hipHostMalloc(&current_->host, cap, hipHostMallocMapped), "hipHostMalloc(staging)");
      hip_check(hipHostGetDevicePointer(&current_->host_mapped, current_->host, 0),
                "hipHostGetDevicePointer(staging)");
      current_->capacity = cap;
    }
    used_ = 0;
  }

  // Adds one array to the batch; returns where it will be on the device.
  template <typename T>
  const T* add(const T* src, size_t count) {
    used_ = (used_ + kAlign - 1) / kAlign * kAlign;
    const size_t bytes = count * sizeof(T);
    if (used_ + bytes > current_->capacity) throw std::runtime_error("staging batch overflow");
    if (bytes != 0) std::memcpy(static_cast<char*>(current_->host) + used_, src, bytes);
    const T* device = reinterpret_cast<const T*>(static_cast<char*>(current_->dev) + used_);
    used_ += bytes;
    return device;
  }

  // One copy for the whole batch, ordered on `stream` before the kernels that read it.  The
  // copy is a kernel reading the pinned block through its device mapping: hipMemcpyAsync from
  // pinned memory was measured to block the host until the stream had drained, every few frames
  // (5-7 ms with five 1.3 ms frames queued; tools/host_stalls.py), and a launch never does.
  void commit(hipStream_t stream) {
    // A batch that equals what this slot's device twin already holds is not copied again: while
    // camera and parameters repeat, a slot sees the same batch every kSlots calls, and the copy
    // kernel with its event are two packets less between two kernels of the consumer stream (a
    // rank of eight: 27 us from one march to the next, of a 0.19 ms frame).  The twin is intact:
    // begin() has waited for the slot's previous copy, and nothing else writes it.
    static const bool always_copy = std::getenv("AVR_ALWAYS_UPLOAD") != nullptr;  // A/B only
    if (!always_copy && used_ != 0 && used_ == current_->shadow.size() &&
        std::memcmp(current_->host, current_->shadow.data(), used_) == 0) {
      // The slot's event is what keeps the host at most kSlots batches ahead of the consumer;
      // unless the owner bounds the frames in flight itself (set_lean), it is still recorded.
      if (!lean_) {
        hip_check(hipEventRecord(current_->done, stream), "hipEventRecord(staging)");
        current_->pending = true;
      }
      return;
    }
    if (used_ != 0) {
      const int status = launch_upload(current_->host_mapped, current_->dev, used_, stream);
      if (status != AVR_OK) throw HipFailure(g_last_error);
    }
    hip_check(hipEventRecord(current_->done, stream), "hipEventRecord(staging)");
    current_->pending = true;
    current_->shadow.assign(static_cast<const char*>(current_->host),
                            static_cast<const char*>(current_->host) + used_);
  }

  // Blocks until every committed batch has been copied (before the context's stream changes).
  void drain() {
    for (Slot& slot : slots_) {
      if (slot.pending) {
        wait_event(slot.done, "hipEventQuery(staging)");
        slot.pending = false;
      }
    }
  }

 private:
  struct Slot {
    void* dev = nullptr;
    void* host = nullptr;
    void* host_mapped = nullptr;  // device address of `host`
    size_t capacity = 0;
    hipEvent_t done = nullptr;      // the copy has run: the pinned block may be refilled
    bool pending = false;
    std::vector<char> shadow;       // what the device twin holds (host copy of the last batch copied)
  };
  bool lean_ = false;
  int ahead_ = 4;
  Slot slots_[kSlots];
  Slot* current_ = nullptr;
  int next_ = 0;
  size_t used_ = 0;
};

}  // namespace
}  // namespace avr

struct avr_scene {
  avr_context* ctx = nullptr;
  std::vector<avr_box> boxes;
  avr_scalar_transform transform{};
  // classified volumes (allocated on first use, grow-only; a frame of the same scene never
  // reallocates): two let the classify pass of frame i+1 overlap the march of frame i, the third
  // lets it run ahead so that neither stream waits for the other's launch (avr_renderer)
  void* classified[AVR_CLASSIFIED_SLOTS] = {};
  size_t classified_capacity[AVR_CLASSIFIED_SLOTS] = {};
  int device = 0;
  // optional re-use of a slot's classified volume across frames (avr_scene_set_classification_cache):
  // what the classify pass of the slot's current contents depended on
  bool cache_classification = false;
  std::vector<uint64_t> classified_key[AVR_CLASSIFIED_SLOTS];

  ~avr_scene() {
    for (void* buffer : classified) {
      if (buffer != nullptr) (void)hipFree(buffer);
    }
  }
  uint8_t* classified_slot(int slot, size_t bytes, hipStream_t stream) {
    if (bytes > classified_capacity[slot]) {
      avr::wait_stream(stream, "classified_slot");
      if (classified[slot] != nullptr) (void)hipFree(classified[slot]);
      classified[slot] = nullptr;
      classified_capacity[slot] = 0;
      classified_key[slot].clear();
      avr::hip_check(hipMalloc(&classified[slot], bytes), "hipMalloc(classified)");
      classified_capacity[slot] = bytes;
    }
    return static_cast<uint8_t*>(classified[slot]);
  }
};

struct avr_context {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  avr::StagingRing staging;
  avr_scene scratch_scene;         // classified storage of avr_paint_box
  std::vector<avr::MarchItemDev> march_items;  // host scratch of render()
  std::vector<int32_t> order_rects;            // ... (the boxes' screen rectangles in layer order)
  int priority = 0;                            // 1: own stream in the highest priority class
  int march_workgroups_per_cu = 0;             // 0 = uncapped
  uint32_t classify_lds_pad = 0;               // avr_context_set_classify_lds_reserve
  bool classify_stream_stores = false;         // context_set_classify_stream_stores
  bool fold_whole_grid = false;                // context_set_fold_whole_grid
  uint64_t* march_counters = nullptr;          // diagnostics (avr_context_set_march_counters)
};


namespace {

// Runs `body`, mapping exceptions to status codes (no exception crosses the ABI).
template <typename F>
int guarded(F&& body) {
  try {
    return body();
  } catch (const std::invalid_argument& e) {
    avr::set_error(e.what());
    return AVR_ERR_INVALID_ARGUMENT;
  } catch (const std::bad_alloc&) {
    avr::set_error("out of host memory");
    return AVR_ERR_OUT_OF_MEMORY;
  } catch (const std::exception& e) {
    avr::set_error(e.what());
    return AVR_ERR_RUNTIME;
  } catch (...) {
    avr::set_error("unknown failure");
    return AVR_ERR_RUNTIME;
  }
}

void require(bool condition, const char* message) {
  if (!condition) throw std::invalid_argument(message);
}

void bind_device(avr_context* ctx) {
  require(ctx != nullptr, "null context");
  avr::hip_check(hipSetDevice(ctx->device), "hipSetDevice");
  if (ctx->stream == nullptr) {  // no external stream was supplied: create the context's own
    if (ctx->priority != 0) {
      int least = 0, greatest = 0;  // numerically lower = more urgent
      avr::hip_check(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
      avr::hip_check(hipStreamCreateWithPriority(&ctx->own_stream, hipStreamNonBlocking, greatest),
                     "hipStreamCreateWithPriority");
    } else {
      avr::hip_check(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking),
                     "hipStreamCreate");
    }
    ctx->stream = ctx->own_stream;
  }
}

enum Phase { kClassify = 1, kMarch = 2 };

// A frame cut into depth-ordered chunks of its global layer order (avr_classify_plan_chunked /
// avr_march_plan_chunked): chunk k is classified by launch k of the classify call, which records
// events[k] behind it, and marched by launch k of the march call, which waits for events[k] first.
struct FrameChunks {
  int count = 1;
  hipEvent_t const* events = nullptr;  // count events, or null (both calls on one stream)
  bool first_alone = false;            // classify: chunk 0 has the GPU to itself (no LDS reserve)
  // avr_render_plan_culled: ONE call queues classify 0, march 0, classify 1, march 1 ... on the
  // context's stream; march k flags the boxes behind chunk k that a ray may still sample in
  // visibility[k * n_order + position] (count * n_order bytes, cleared by the call), classify
  // k + 1 leaves out the boxes of its chunk whose flag is 0.
  uint8_t* visibility = nullptr;
  // A speculative frame (avr_classify_plan_flagged / avr_march_plan_speculative; count == 1):
  // the classify pass takes only the positions of the layer order whose flag is set (and, with a
  // gate, does nothing at all unless *gate != 0: the repair pass), the march checks and records.
  const uint8_t* classify_flags = nullptr;
  const uint32_t* classify_gate = nullptr;
  // ... or the positions themselves, known on the host (avr_classify_plan_positions): a launch of
  // exactly their tiles
  const int32_t* classify_positions = nullptr;
  int n_classify_positions = 0;
  const avr_speculation* speculation = nullptr;
};

// Positions [bounds[k], bounds[k + 1]) of the global layer order for chunk k: equal shares of the
// classify pass's work (tiles = cells), so that classifying chunk k + 1 takes about as long as
// any other; a function of the plan alone -- the classify and the march call cut alike.
std::vector<int> chunk_bounds(const avr::FramePlan& plan, const int32_t* box_order, int n_order,
                              int n_chunks) {
  std::vector<int> bounds(static_cast<size_t>(n_chunks) + 1, n_order);
  bounds[0] = 0;
  auto tiles_of = [&](int position) -> uint64_t {
    const size_t b = static_cast<size_t>(box_order[position]);
    return plan.classify_tile_begin[b + 1] - plan.classify_tile_begin[b];
  };
  uint64_t total = 0;
  for (int i = 0; i < n_order; ++i) total += tiles_of(i);
  uint64_t seen = 0;
  int k = 1;
  for (int i = 0; i < n_order && k < n_chunks; ++i) {
    seen += tiles_of(i);
    while (k < n_chunks && seen * static_cast<uint64_t>(n_chunks) >= total * static_cast<uint64_t>(k)) {
      bounds[static_cast<size_t>(k++)] = i + 1;
    }
  }
  return bounds;
}

// One frame's device work for a list of boxes: the classify pass and/or the march.
// `classified` must hold plan.classified_bytes bytes; `cached` (optional) carries the host
// Upper bound of the march's workgroups: a work item per super-tile of the screen and run, padded
// to whole rounds of the XCDs, four workgroups each (build_march_items, avr_host.cpp).
int64_t march_workgroups_bound(int width, int height, int n_runs) {
  const int64_t span = avr::kTile * avr::kSuperTileSide;
  const int64_t super_tiles = ((width + span - 1) / span) * ((height + span - 1) / span);
  const int64_t items = (super_tiles * std::max(n_runs, 0) + avr::kXcds - 1) / avr::kXcds * avr::kXcds;
  return items * avr::kSuperTileTiles;
}

// prologue from the classify call of a frame to its march call.
int render(avr_context* ctx, int phases, const avr_box* boxes, int n_boxes,
           const avr_scalar_transform& transform, const avr_paint_params& params,
           const avr_camera& camera, const int32_t* box_order, int n_order,
           const int32_t* run_end, int n_runs, int n_pieces,
           const std::vector<avr::RunRectDev>& run_rects,
           const std::vector<avr::RunBlockDev>& run_blocks,
           const std::vector<avr::RunSpanDev>* run_spans, const avr::PieceMapDev& pieces,
           avr_scene* scene, int slot,
           float* out_layers, uint64_t* samples_out, avr::FramePlan* cached,
           const FrameChunks& chunks = FrameChunks{}) {
  require(n_runs >= 0 && n_order >= 0 && n_pieces >= 1, "invalid run description");
  require(chunks.count >= 1 && chunks.count <= AVR_MAX_FRAME_CHUNKS, "invalid chunk count");
  require(chunks.count == 1 || phases == kClassify || phases == kMarch || chunks.visibility != nullptr,
          "a chunked frame is classified and marched by separate calls (or by avr_render_plan_culled)");
  require(slot >= 0 && slot < AVR_CLASSIFIED_SLOTS, "classified slot out of range");
  avr::FramePlan local;
  avr::FramePlan& plan = cached ? *cached : local;
  if (plan.boxes.size() != static_cast<size_t>(n_boxes) || !plan.ready) {
    avr::plan_frame(boxes, n_boxes, transform, params, camera, &plan);
    plan.march_items_ready = false;
  }
  if (n_runs == 0) return AVR_OK;
  require(plan.n_tables <= avr::kMaxLdsTables,
          "too many distinct sampling levels for the LDS transfer-function cache");

  avr::RenderLaunch launch;
  launch.consts = plan.consts;
  launch.n_boxes = n_boxes;
  launch.n_classify_tiles = plan.classify_tile_begin.back();
  launch.classify_lds_pad = ctx->classify_lds_pad;
  // (streamed only when the classified volume is more than the 256 MB memory-side cache holds:
  // config-2's 134 MB, stored plainly, are still there when its march comes a frame later --
  // 0.378 ms per frame against 0.465 streamed; config-4's 370 MB are not, 0.967 -> 0.958)
  launch.classify_stream_stores =
      (ctx->classify_stream_stores && plan.classified_bytes > (256ull << 20)) ? 1 : 0;
  launch.classified = scene->classified_slot(slot, plan.classified_bytes, ctx->stream);

  if ((phases & kClassify) && scene->cache_classification) {
    // everything classify_kernel reads besides the cells themselves
    std::vector<uint64_t> key;
    auto bits = [](double v) {
      uint64_t u;
      std::memcpy(&u, &v, sizeof(u));
      return u;
    };
    const avr::FrameConsts& fc = plan.consts;
    key.reserve(12 + plan.boxes.size() * 4);
    for (double v : {static_cast<double>(fc.range_min), static_cast<double>(fc.inverse_range),
                     static_cast<double>(fc.clip_start), fc.positive_floor, fc.norm_min,
                     fc.inv_norm_span}) {
      key.push_back(bits(v));
    }
    key.push_back(static_cast<uint64_t>(fc.apply_clip) | (static_cast<uint64_t>(fc.log_scale) << 8) |
                  (static_cast<uint64_t>(fc.normalize) << 16));
    for (const avr::BoxDev& dev : plan.boxes) {
      key.push_back(reinterpret_cast<uint64_t>(dev.cells));
      key.push_back((static_cast<uint64_t>(static_cast<uint32_t>(dev.jstride)) << 32) |
                    static_cast<uint32_t>(dev.kstride));
      key.push_back((static_cast<uint64_t>(dev.nx) << 42) | (static_cast<uint64_t>(dev.ny) << 21) |
                    static_cast<uint64_t>(dev.nz));
      key.push_back(dev.cls_offset);
    }
    if (key == scene->classified_key[slot]) {
      phases &= ~kClassify;  // the slot already holds exactly this classification
      if (phases == 0) return AVR_OK;
    } else {
      scene->classified_key[slot] = std::move(key);
    }
  }

  // scratch, reused across frames; a frame plan's prologue keeps its own (they depend on the
  // plan alone)
  const bool keep_items = cached != nullptr;
  std::vector<avr::MarchItemDev>& items = keep_items ? plan.march_items : ctx->march_items;
  size_t bytes = plan.boxes.size() * sizeof(avr::BoxDev);
  if (phases & kClassify) bytes += plan.classify_tile_begin.size() * sizeof(uint32_t);
  if (phases & kMarch) {
    require(out_layers != nullptr, "null output image");
    require(n_runs == 0 || (run_end != nullptr), "null run_end");
    require(n_order == 0 || (box_order != nullptr), "null box_order");
    require(run_rects.size() == static_cast<size_t>(n_runs) &&
                run_blocks.size() == static_cast<size_t>(n_runs) * n_pieces,
            "run tables do not match the runs");
    if (!(keep_items && plan.march_items_ready)) {
      int previous = 0;
      for (int r = 0; r < n_runs; ++r) {
        require(run_end[r] >= previous && run_end[r] <= n_order, "run_end must be non-decreasing");
        previous = run_end[r];
      }
      require(run_end[n_runs - 1] == n_order, "runs must cover box_order");
      for (int i = 0; i < n_order; ++i) {
        require(box_order[i] >= 0 && box_order[i] < n_boxes, "box_order entry out of range");
      }
      avr::build_march_items(plan, box_order, run_end, n_runs, run_rects, &items);
      plan.march_items_ready = keep_items;
    }
    bytes += plan.tables.size() * sizeof(float) + static_cast<size_t>(n_order + n_runs) * 4 +
             static_cast<size_t>(n_order) * 16 +
             items.size() * sizeof(avr::MarchItemDev) +
             run_rects.size() * sizeof(avr::RunRectDev) + run_blocks.size() * sizeof(avr::RunBlockDev) +
             (run_spans != nullptr ? run_spans->size() * sizeof(avr::RunSpanDev) : 0);
  }
  // chunked: the positions of every chunk, and for the classify call the chunks' box lists (the
  // local boxes in global layer order ARE the lists: chunk k is box_order[bounds[k] .. bounds[k+1]))
  // with one prefix sum of classify workgroups per chunk
  const bool chunked = chunks.count > 1;
  std::vector<int> bounds;
  std::vector<uint32_t> chunk_tile_begin;  // chunk k: entries bounds[k] + k .. bounds[k + 1] + k
  if (chunked) {
    require(n_order == 0 || box_order != nullptr, "null box_order");
    for (int i = 0; i < n_order; ++i) {
      require(box_order[i] >= 0 && box_order[i] < n_boxes, "box_order entry out of range");
    }
    bounds = chunk_bounds(plan, box_order, n_order, chunks.count);
    if (phases & kClassify) {
      chunk_tile_begin.reserve(static_cast<size_t>(n_order + chunks.count));
      for (int k = 0; k < chunks.count; ++k) {
        uint32_t sum = 0;
        chunk_tile_begin.push_back(0u);
        for (int i = bounds[static_cast<size_t>(k)]; i < bounds[static_cast<size_t>(k) + 1]; ++i) {
          const size_t b = static_cast<size_t>(box_order[i]);
          sum += plan.classify_tile_begin[b + 1] - plan.classify_tile_begin[b];
          chunk_tile_begin.push_back(sum);
        }
      }
      bytes += chunk_tile_begin.size() * sizeof(uint32_t) + static_cast<size_t>(n_order) * 4;
    }
  }
  // a flagged classify pass: the local boxes in layer order as ONE list under its own prefix sum
  std::vector<uint32_t> listed_tile_begin;
  std::vector<int32_t> listed_boxes;
  const bool positioned = (phases & kClassify) && chunks.classify_positions != nullptr;
  const bool flagged = (phases & kClassify) && (chunks.classify_flags != nullptr || positioned);
  if (flagged) {
    require(chunks.count == 1, "a flagged classify pass is not cut into chunks");
    require(n_order == 0 || box_order != nullptr, "null box_order");
    require(!(positioned && chunks.classify_flags != nullptr), "flags or positions, not both");
    const int n_listed = positioned ? chunks.n_classify_positions : n_order;
    require(n_listed >= 0 && n_listed <= n_order, "too many positions");
    listed_tile_begin.reserve(static_cast<size_t>(n_listed) + 1);
    listed_boxes.reserve(static_cast<size_t>(n_listed));
    uint32_t sum = 0;
    listed_tile_begin.push_back(0u);
    int previous = -1;
    for (int i = 0; i < n_listed; ++i) {
      const int position = positioned ? chunks.classify_positions[i] : i;
      require(position > previous && position < n_order, "positions must ascend within the layer order");
      previous = position;
      require(box_order[position] >= 0 && box_order[position] < n_boxes, "box_order entry out of range");
      const size_t b = static_cast<size_t>(box_order[position]);
      sum += plan.classify_tile_begin[b + 1] - plan.classify_tile_begin[b];
      listed_tile_begin.push_back(sum);
      listed_boxes.push_back(box_order[position]);
    }
    bytes += listed_tile_begin.size() * sizeof(uint32_t) + listed_boxes.size() * 4;
  }
  if ((phases & kMarch) && chunks.speculation != nullptr) {
    bytes += sizeof(avr::MarchSpecDev) + static_cast<size_t>(n_order);
  }
  avr::StagingRing& staging = ctx->staging;
  staging.begin(bytes, 16);
  launch.boxes_dev = staging.add(plan.boxes.data(), plan.boxes.size());
  const uint32_t* chunk_tile_begin_dev = nullptr;
  const int32_t* chunk_box_list_dev = nullptr;
  if (phases & kClassify) {
    launch.tile_begin_dev =
        staging.add(plan.classify_tile_begin.data(), plan.classify_tile_begin.size());
    if (chunked) {
      chunk_tile_begin_dev = staging.add(chunk_tile_begin.data(), chunk_tile_begin.size());
      chunk_box_list_dev = staging.add(box_order, static_cast<size_t>(n_order));
    }
    if (flagged) {
      launch.tile_begin_dev = staging.add(listed_tile_begin.data(), listed_tile_begin.size());
      launch.box_list_dev = staging.add(listed_boxes.data(), listed_boxes.size());
      launch.n_classify_boxes = static_cast<int>(listed_boxes.size());
      launch.n_classify_tiles = listed_tile_begin.back();
      launch.visible_in = chunks.classify_flags;  // (null with positions: all of them)
      launch.classify_gate = chunks.classify_gate;
    }
  }
  if (phases & kMarch) {
    launch.tables_dev = staging.add(plan.tables.data(), plan.tables.size());
    launch.n_tables = plan.n_tables;
    launch.order_dev = staging.add(box_order, static_cast<size_t>(n_order));
    {
      // the boxes' screen rectangles in that order: what the march's cull reads 64 at a time
      std::vector<int32_t>& rects = ctx->order_rects;
      rects.resize(static_cast<size_t>(n_order) * 4);
      for (int i = 0; i < n_order; ++i) {
        const avr::BoxDev& dev = plan.boxes[static_cast<size_t>(box_order[i])];
        std::copy(dev.rect, dev.rect + 4, rects.begin() + static_cast<std::ptrdiff_t>(i) * 4);
      }
      launch.order_rects_dev = staging.add(rects.data(), rects.size());
    }
    launch.run_end_dev = staging.add(run_end, static_cast<size_t>(n_runs));
    launch.n_order = n_order;
    launch.n_runs = n_runs;
    launch.n_pieces = n_pieces;
    launch.run_rects_dev = staging.add(run_rects.data(), run_rects.size());
    launch.run_blocks_dev = staging.add(run_blocks.data(), run_blocks.size());
    launch.run_spans_dev = (run_spans != nullptr && !run_spans->empty())
                               ? staging.add(run_spans->data(), run_spans->size())
                               : nullptr;
    launch.pieces = pieces;
    launch.out_layers = out_layers;
    launch.samples_out = reinterpret_cast<unsigned long long*>(samples_out);
    launch.counters = reinterpret_cast<unsigned long long*>(ctx->march_counters);
    if (chunks.speculation != nullptr) {
      require(chunks.count == 1 && samples_out == nullptr,
              "a speculative march is one launch and counts no samples");
      const avr_speculation& given = *chunks.speculation;
      require(given.classified == nullptr || given.classified_host == nullptr,
              "the flags are on the device or on the host, not both");
      const bool checks = given.classified != nullptr || given.classified_host != nullptr;
      require(!checks || (given.missed != nullptr && given.miss_count != nullptr),
              "a speculative march that checks flags needs somewhere to report misses");
      avr::MarchSpecDev spec{};
      spec.classified = given.classified_host != nullptr
                            ? staging.add(given.classified_host, static_cast<size_t>(n_order))
                            : given.classified;
      spec.visited = given.visited;
      spec.missed = given.missed;
      spec.miss_count = given.miss_count;
      spec.host_miss_flag = given.host_miss_flag;
      spec.gate = given.gate;
      // (the first pass marks, the gated pass reads: one array, a byte per workgroup of the grid)
      spec.dirty_blocks_out = given.gate == nullptr ? given.dirty_workgroups : nullptr;
      spec.dirty_blocks = given.gate != nullptr ? given.dirty_workgroups : nullptr;
      if (given.dirty_workgroups != nullptr) {
        require(static_cast<int64_t>(items.size()) * avr::kSuperTileTiles <=
                    march_workgroups_bound(params.width, params.height, n_runs),
                "more march workgroups than avr_march_plan_workgroups promised");
      }
      launch.spec_dev = staging.add(&spec, 1);
      launch.spec_is_repair = given.gate != nullptr;
    }
    launch.items_dev = staging.add(items.data(), items.size());
    launch.n_items = static_cast<uint32_t>(items.size());
    launch.workgroups_per_cu = ctx->march_workgroups_per_cu;
    launch.only_mode = plan.boxes.empty() ? -1 : plan.boxes[0].index_mode;
    for (const avr::BoxDev& dev : plan.boxes) {
      if (dev.index_mode != launch.only_mode) launch.only_mode = -1;
    }
  }
  staging.commit(ctx->stream);
  if (chunked) {
    const uint32_t reserve = launch.classify_lds_pad;
    if (chunks.visibility != nullptr) {
      avr::hip_check(hipMemsetAsync(chunks.visibility, 0,
                                    static_cast<size_t>(chunks.count) * static_cast<size_t>(n_order),
                                    ctx->stream), "hipMemsetAsync(visibility)");
    }
    for (int k = 0; k < chunks.count; ++k) {
      const int first = bounds[static_cast<size_t>(k)], last = bounds[static_cast<size_t>(k) + 1];
      if (phases & kClassify) {
        // (what the march launch of the chunk before found still visible of this chunk's boxes)
        launch.visible_in = (chunks.visibility != nullptr && k > 0)
                                ? chunks.visibility + static_cast<size_t>(k - 1) * n_order + first
                                : nullptr;
        if (last > first) {
          launch.box_list_dev = chunk_box_list_dev + first;
          launch.n_classify_boxes = last - first;
          launch.tile_begin_dev = chunk_tile_begin_dev + first + k;
          launch.n_classify_tiles = chunk_tile_begin[static_cast<size_t>(last + k)];
          launch.classify_lds_pad = (k == 0 && chunks.first_alone) ? 0u : reserve;
          const int status = avr::launch_classify(launch, ctx->stream);
          if (status != AVR_OK) return status;
        }
        if (chunks.events != nullptr) {
          avr::hip_check(hipEventRecord(chunks.events[k], ctx->stream), "hipEventRecord(chunk)");
        }
      }
      if (phases & kMarch) {
        launch.visible_out = (chunks.visibility != nullptr && k + 1 < chunks.count)
                                 ? chunks.visibility + static_cast<size_t>(k) * n_order
                                 : nullptr;
        if (chunks.events != nullptr && !(phases & kClassify)) {
          avr::hip_check(hipStreamWaitEvent(ctx->stream, chunks.events[k], 0), "hipStreamWaitEvent(chunk)");
        }
        // (the first launch stores every pixel of the runs' layers; the later ones resume them)
        // (with occlusion culling every chunk's march is launched: it is what flags the boxes of
        // the chunks behind it -- flags left at 0 would mean "nobody samples them")
        if (last > first || k == 0 || chunks.visibility != nullptr) {
          launch.pos_begin = first;
          launch.pos_end = last;
          launch.resume = k > 0 ? 1 : 0;
          const int status = avr::launch_march(launch, ctx->stream);
          if (status != AVR_OK) return status;
        }
      }
    }
    return AVR_OK;
  }
  if (phases & kClassify) {
    const int status = avr::launch_classify(launch, ctx->stream);
    if (status != AVR_OK) return status;
  }
  if (phases & kMarch) return avr::launch_march(launch, ctx->stream);
  return AVR_OK;
}

}  // namespace

namespace avr {
void* context_stream(avr_context* ctx) {
  bind_device(ctx);
  return ctx->stream;
}
void context_set_classify_stream_stores(avr_context* ctx, bool stream) {
  ctx->classify_stream_stores = stream;
}
void context_set_lean_descriptors(avr_context* ctx, bool lean) { ctx->staging.set_lean(lean); }
void context_set_descriptor_lead(avr_context* ctx, int batches) { ctx->staging.set_ahead(batches); }
void context_set_fold_whole_grid(avr_context* ctx, bool whole) { ctx->fold_whole_grid = whole; }
}  // namespace avr

extern "C" {

const char* avr_last_error(void) { return avr::g_last_error.c_str(); }

int avr_abi_version(void) { return AVR_ABI_VERSION; }

int avr_debug_stall_stream(void* hip_stream, int milliseconds) {
  return guarded([&]() -> int {
    require(milliseconds >= 1 && milliseconds <= 2000, "milliseconds must be in [1, 2000]");
    return avr::launch_stall(milliseconds, hip_stream);
  });
}

int avr_set_frame_timeout_ms(int milliseconds) {
  avr::set_frame_timeout_ms(milliseconds);
  return AVR_OK;
}

int avr_context_create(int device_id, avr_context** out_ctx) {
  return guarded([&]() -> int {
    require(out_ctx != nullptr, "null out_ctx");
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
      avr::set_error("no HIP device available (this library has no CPU fallback)");
      return AVR_ERR_NO_DEVICE;
    }
    require(device_id >= 0 && device_id < count, "device_id out of range");
    avr::hip_check(hipSetDevice(device_id), "hipSetDevice");
    auto* ctx = new avr_context();
    ctx->device = device_id;
    ctx->stream = ctx->own_stream;
    *out_ctx = ctx;
    return AVR_OK;
  });
}

int avr_context_create_with_priority(int device_id, int high_priority, avr_context** out_ctx) {
  const int status = avr_context_create(device_id, out_ctx);
  if (status == AVR_OK) (*out_ctx)->priority = high_priority ? 1 : 0;
  return status;
}

void* avr_context_stream(avr_context* ctx) {
  try {
    return avr::context_stream(ctx);
  } catch (const std::exception& e) {
    avr::set_error(e.what());
    return nullptr;
  }
}

void avr_context_destroy(avr_context* ctx) {
  if (ctx == nullptr) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->own_stream != nullptr) (void)hipStreamSynchronize(ctx->own_stream);
  ctx->staging.release();
  if (ctx->own_stream != nullptr) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

int avr_context_set_stream(avr_context* ctx, void* hip_stream) {
  return guarded([&]() -> int {
    require(ctx != nullptr, "null context");
    avr::hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->staging.drain();
    // NULL: back to the context's own stream (created on first use)
    ctx->stream = (hip_stream != nullptr) ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return AVR_OK;
  });
}

int avr_context_set_march_occupancy(avr_context* ctx, int workgroups_per_cu) {
  return guarded([&]() -> int {
    require(ctx != nullptr, "null context");
    require(workgroups_per_cu >= 0 && workgroups_per_cu <= 8, "workgroups_per_cu must be in [0, 8]");
    ctx->march_workgroups_per_cu = (workgroups_per_cu == 8) ? 0 : workgroups_per_cu;
    return AVR_OK;
  });
}

int avr_context_set_classify_lds_reserve(avr_context* ctx, int bytes) {
  return guarded([&]() -> int {
    require(ctx != nullptr, "null context");
    require(bytes >= 0 && bytes <= AVR_CLASSIFY_LDS_RESERVE_MAX,
            "bytes must be in [0, AVR_CLASSIFY_LDS_RESERVE_MAX]");
    ctx->classify_lds_pad = static_cast<uint32_t>(bytes);
    return AVR_OK;
  });
}

int avr_context_set_march_counters(avr_context* ctx, uint64_t* counters_dev) {
  return guarded([&]() -> int {
    require(ctx != nullptr, "null context");
    ctx->march_counters = counters_dev;
    return AVR_OK;
  });
}

int avr_context_synchronize(avr_context* ctx) {
  return guarded([&]() -> int {
    bind_device(ctx);
    avr::wait_stream(ctx->stream, "avr_context_synchronize");
    return AVR_OK;
  });
}

int avr_build_color_table(float alpha_scale, float normalization_factor,
                          const float scalar_range[2], const avr_colormap_point* colormap,
                          int colormap_count, float out_table_host[1024]) {
  return guarded([&]() -> int {
    require(scalar_range != nullptr && out_table_host != nullptr, "null argument");
    require(colormap_count >= 0 && (colormap_count == 0 || colormap != nullptr),
            "invalid color map");
    avr::build_color_table(alpha_scale, normalization_factor, scalar_range, colormap,
                           colormap_count, out_table_host);
    return AVR_OK;
  });
}

int avr_box_sampling(const avr_box* box, const avr_paint_params* params, float* sample_distance,
                     float* normalization_factor, float* alpha_scale) {
  return guarded([&]() -> int {
    require(box != nullptr && params != nullptr && sample_distance != nullptr &&
                normalization_factor != nullptr && alpha_scale != nullptr,
            "null argument");
    avr::box_sampling(*box, *params, sample_distance, normalization_factor, alpha_scale);
    return AVR_OK;
  });
}

int avr_box_depth_hint(const avr_box* box, const avr_camera* camera, float* out_hint) {
  return guarded([&]() -> int {
    require(box != nullptr && camera != nullptr && out_hint != nullptr, "null argument");
    *out_hint = avr::box_depth_hint(*box, *camera);
    return AVR_OK;
  });
}

int avr_reference_sample_distance(const avr_box* boxes, int n_boxes, const double bounds_min[3],
                                  const double bounds_max[3], float* out_distance) {
  return guarded([&]() -> int {
    require(n_boxes >= 0 && (n_boxes == 0 || boxes != nullptr), "invalid box list");
    require(bounds_min != nullptr && bounds_max != nullptr && out_distance != nullptr,
            "null argument");
    *out_distance = avr::reference_sample_distance(boxes, n_boxes, bounds_min, bounds_max);
    return AVR_OK;
  });
}

int avr_layer_order(const float* hints, const int32_t* owner, const int32_t* local_index,
                    int n_layers, int32_t* order_out, int32_t* run_end_out, int* n_runs_out) {
  return guarded([&]() -> int {
    require(n_layers >= 0 && n_runs_out != nullptr, "invalid argument");
    if (n_layers == 0) {
      *n_runs_out = 0;
      return AVR_OK;
    }
    require(hints != nullptr && owner != nullptr && local_index != nullptr &&
                order_out != nullptr && run_end_out != nullptr,
            "null argument");
    *n_runs_out = avr::layer_order(hints, owner, local_index, n_layers, order_out, run_end_out);
    return AVR_OK;
  });
}

int avr_piece_range(int64_t image_size, int piece_index, int num_pieces, int64_t* begin,
                    int64_t* end) {
  return guarded([&]() -> int {
    require(begin != nullptr && end != nullptr, "null argument");
    require(num_pieces >= 1 && piece_index >= 0 && piece_index < num_pieces && image_size >= 0,
            "invalid piece");
    const int64_t piece_size = image_size / num_pieces;
    *begin = piece_size * piece_index;
    *end = (piece_index < num_pieces - 1) ? (*begin + piece_size) : image_size;
    return AVR_OK;
  });
}

int avr_paint_box(avr_context* ctx, const avr_box* box, const avr_scalar_transform* transform,
                  const avr_paint_params* params, const avr_camera* camera, float* out_rgbad,
                  uint64_t* samples_out) {
  return guarded([&]() -> int {
    bind_device(ctx);
    require(box != nullptr && transform != nullptr && params != nullptr && camera != nullptr,
            "null argument");
    const int32_t order[1] = {0};
    const int32_t run_end[1] = {1};
    std::vector<avr::RunRectDev> rects;
    std::vector<avr::RunBlockDev> blocks;
    require(params->width > 0 && params->height > 0, "image width and height must be positive");
    avr::dense_run_tables(params->width, params->height, 1, 1, &rects, &blocks);
    return render(ctx, kClassify | kMarch, box, 1, *transform, *params, *camera, order, 1, run_end,
                  1, 1, rects, blocks, nullptr,
                  avr::make_piece_map(AVR_PIECES_CONTIGUOUS, 1, 1, params->width, params->height),
                  &ctx->scratch_scene, 0, out_rgbad, samples_out, nullptr);
  });
}

int avr_scene_create(avr_context* ctx, const avr_box* boxes, int n_boxes,
                     const avr_scalar_transform* transform, avr_scene** out_scene) {
  return guarded([&]() -> int {
    require(ctx != nullptr && out_scene != nullptr && transform != nullptr, "null argument");
    require(n_boxes >= 0 && (n_boxes == 0 || boxes != nullptr), "invalid box list");
    auto* scene = new avr_scene();
    scene->ctx = ctx;
    scene->boxes.assign(boxes, boxes + n_boxes);
    scene->transform = *transform;
    *out_scene = scene;
    return AVR_OK;
  });
}

void avr_scene_destroy(avr_scene* scene) { delete scene; }

int avr_render_runs(avr_context* ctx, const avr_scene* scene, const avr_paint_params* params,
                    const avr_camera* camera, const int32_t* box_order, int n_order,
                    const int32_t* run_end, int n_runs, int n_pieces, float* out_layers,
                    uint64_t* samples_out) {
  return guarded([&]() -> int {
    bind_device(ctx);
    require(scene != nullptr && params != nullptr && camera != nullptr, "null argument");
    require(n_runs >= 0 && n_pieces >= 1, "invalid run description");
    require(params->width > 0 && params->height > 0, "image width and height must be positive");
    std::vector<avr::RunRectDev> rects;
    std::vector<avr::RunBlockDev> blocks;
    avr::dense_run_tables(params->width, params->height, n_runs, n_pieces, &rects, &blocks);
    return render(ctx, kClassify | kMarch, scene->boxes.data(),
                  static_cast<int>(scene->boxes.size()), scene->transform, *params, *camera,
                  box_order, n_order, run_end, n_runs, n_pieces, rects, blocks, nullptr,
                  avr::make_piece_map(AVR_PIECES_CONTIGUOUS, 1, n_pieces, params->width,
                                      params->height),
                  const_cast<avr_scene*>(scene), 0, out_layers, samples_out, nullptr);
  });
}

int avr_frame_plan_create_pieces(const avr_box* all_boxes, const int32_t* owner, int n_boxes,
                                 int n_ranks, int rank, const int32_t* group_order,
                                 const avr_paint_params* params, const avr_camera* camera,
                                 int piece_layout, int band_rows, avr_frame_plan** out_plan) {
  return guarded([&]() -> int {
    require(out_plan != nullptr && params != nullptr && camera != nullptr, "null argument");
    *out_plan = nullptr;
    require(params->colormap_count >= 0 && (params->colormap_count == 0 || params->colormap),
            "invalid color map");
    auto* plan = new avr_frame_plan();
    try {
      avr::build_frame_plan(all_boxes, owner, n_boxes, n_ranks, rank, group_order, *params, *camera,
                            piece_layout, band_rows, plan);
    } catch (...) {
      delete plan;
      throw;
    }
    *out_plan = plan;
    return AVR_OK;
  });
}

int avr_frame_plan_create(const avr_box* all_boxes, const int32_t* owner, int n_boxes, int n_ranks,
                          int rank, const int32_t* group_order, const avr_paint_params* params,
                          const avr_camera* camera, avr_frame_plan** out_plan) {
  return avr_frame_plan_create_pieces(all_boxes, owner, n_boxes, n_ranks, rank, group_order, params,
                                      camera, AVR_PIECES_CONTIGUOUS, 1, out_plan);
}

int avr_layered_plan_create(const float* hints, const int32_t* owner, int n_layers, int n_ranks,
                            int rank, const int32_t* group_order, int width, int height,
                            avr_frame_plan** out_plan) {
  return guarded([&]() -> int {
    require(out_plan != nullptr, "null argument");
    *out_plan = nullptr;
    auto* plan = new avr_frame_plan();
    try {
      plan->params.width = width;
      plan->params.height = height;
      avr::build_layer_plan(n_layers, hints, owner, nullptr, n_ranks, rank, group_order, width,
                            height, AVR_PIECES_CONTIGUOUS, 1, plan);
    } catch (...) {
      delete plan;
      throw;
    }
    *out_plan = plan;
    return AVR_OK;
  });
}

int avr_pack_layers(avr_context* ctx, const avr_frame_plan* plan, const float* const* local_layers,
                    int n_local_layers, float* send_buffer) {
  return guarded([&]() -> int {
    bind_device(ctx);
    require(plan != nullptr, "null plan");
    require(n_local_layers == plan->info.n_local_boxes, "layer count does not match the plan");
    require(n_local_layers == 0 || local_layers != nullptr, "null layer list");
    require(plan->info.send_floats == 0 || send_buffer != nullptr, "null send buffer");
    require(plan->pieces.layout == avr::kPiecesContiguous,
            "avr_pack_layers needs the reference's contiguous pieces");
    const int n_ranks = plan->info.n_ranks;
    const int64_t width = plan->params.width;
    int start = 0;
    std::vector<const float*> slices;
    for (int r = 0; r < plan->info.n_local_runs; ++r) {
      const int end = plan->local_run_end[static_cast<size_t>(r)];
      const avr::RunRectDev& rect = plan->local_rects[static_cast<size_t>(r)];
      require(rect.x0 == 0 && rect.x1 == width - 1, "avr_pack_layers needs a full-width layered plan");
      for (int piece = 0; piece < n_ranks; ++piece) {
        const size_t at = static_cast<size_t>(r) * n_ranks + static_cast<size_t>(piece);
        const int rows = plan->send_block_rows[at];
        if (rows == 0) continue;
        const int64_t first_pixel = static_cast<int64_t>(plan->send_blocks[at].first_row) * width;
        slices.clear();
        for (int l = start; l < end; ++l) {
          const int32_t local = plan->local_order[static_cast<size_t>(l)];
          require(local >= 0 && local < n_local_layers && local_layers[local] != nullptr,
                  "invalid local layer");
          slices.push_back(local_layers[local] + first_pixel * 5);
        }
        // owner-side fold of the run's layers, in order (DirectSendBase.cpp:413-426), over the
        // block's whole rows
        ctx->staging.begin(slices.size() * sizeof(float*), 1);
        const float* const* slices_dev = ctx->staging.add(slices.data(), slices.size());
        ctx->staging.commit(ctx->stream);
        const int status = avr::launch_fold_runs(slices_dev, static_cast<int>(slices.size()),
                                                 send_buffer + plan->send_blocks[at].offset,
                                                 static_cast<int64_t>(rows) * width, ctx->stream);
        if (status != AVR_OK) return status;
      }
      start = end;
    }
    return AVR_OK;
  });
}

void avr_frame_plan_destroy(avr_frame_plan* plan) { delete plan; }

int avr_frame_plan_get_info(const avr_frame_plan* plan, avr_frame_plan_info* out) {
  return guarded([&]() -> int {
    require(plan != nullptr && out != nullptr, "null argument");
    *out = plan->info;
    return AVR_OK;
  });
}

int avr_frame_plan_splits(const avr_frame_plan* plan, int64_t* send_splits, int64_t* recv_splits) {
  return guarded([&]() -> int {
    require(plan != nullptr && send_splits != nullptr && recv_splits != nullptr, "null argument");
    std::copy(plan->send_splits.begin(), plan->send_splits.end(), send_splits);
    std::copy(plan->recv_splits.begin(), plan->recv_splits.end(), recv_splits);
    return AVR_OK;
  });
}

int avr_frame_plan_layers(const avr_frame_plan* plan, int32_t* layer_box) {
  return guarded([&]() -> int {
    require(plan != nullptr && (plan->layer_box.empty() || layer_box != nullptr), "null argument");
    std::copy(plan->layer_box.begin(), plan->layer_box.end(), layer_box);
    return AVR_OK;
  });
}

int avr_frame_plan_runs(const avr_frame_plan* plan, avr_run_info* runs) {
  return guarded([&]() -> int {
    require(plan != nullptr && (plan->runs.empty() || runs != nullptr), "null argument");
    std::copy(plan->runs.begin(), plan->runs.end(), runs);
    return AVR_OK;
  });
}

int avr_box_footprint(const avr_box* box, const avr_camera* camera, int width, int height,
                      int32_t rect[4], int32_t* row_x0, int32_t* row_x1) {
  return guarded([&]() -> int {
    require(box != nullptr && camera != nullptr && rect != nullptr, "null argument");
    require(width > 0 && height > 0, "image width and height must be positive");
    require((row_x0 == nullptr) == (row_x1 == nullptr), "row_x0 and row_x1 go together");
    avr::box_screen_rect(*box, *camera, width, height, rect);
    if (row_x0 == nullptr) return AVR_OK;
    for (int y = 0; y < height; ++y) {
      row_x0[y] = 0;
      row_x1[y] = -1;
    }
    std::vector<int32_t> x0, x1;
    avr::box_row_spans(*box, *camera, width, height, rect, &x0, &x1);
    for (size_t r = 0; r < x0.size(); ++r) {
      row_x0[rect[1] + static_cast<int>(r)] = x0[r];
      row_x1[rect[1] + static_cast<int>(r)] = x1[r];
    }
    return AVR_OK;
  });
}

int avr_frame_plan_tighten(avr_frame_plan* plan, const avr_box* all_boxes, int n_boxes) {
  return guarded([&]() -> int {
    require(plan != nullptr && n_boxes >= 0 && (n_boxes == 0 || all_boxes != nullptr),
            "invalid argument");
    avr::tighten_frame_plan(all_boxes, n_boxes, plan);
    return AVR_OK;
  });
}

int avr_frame_plan_send_block(const avr_frame_plan* plan, int peer, int local_run, int64_t* offset,
                              int32_t* first_row, int32_t* n_rows) {
  return guarded([&]() -> int {
    require(plan != nullptr && offset != nullptr && first_row != nullptr && n_rows != nullptr,
            "null argument");
    require(peer >= 0 && peer < plan->info.n_ranks && local_run >= 0 &&
                local_run < plan->info.n_local_runs,
            "block index out of range");
    require(!plan->tightened, "a tightened plan has no rectangular blocks");
    const size_t at = static_cast<size_t>(local_run) * plan->info.n_ranks +
                      static_cast<size_t>(plan->piece_of_rank[static_cast<size_t>(peer)]);
    *n_rows = plan->send_block_rows[at];
    *first_row = plan->send_blocks[at].first_row;
    *offset = (*n_rows > 0) ? plan->send_blocks[at].offset : -1;
    return AVR_OK;
  });
}

int avr_frame_plan_recv_block(const avr_frame_plan* plan, int global_run, int64_t* offset,
                              int32_t* first_row, int32_t* n_rows) {
  return guarded([&]() -> int {
    require(plan != nullptr && offset != nullptr && first_row != nullptr && n_rows != nullptr,
            "null argument");
    require(global_run >= 0 && global_run < plan->info.n_runs_total, "run index out of range");
    require(!plan->tightened, "a tightened plan has no rectangular blocks");
    const size_t at = static_cast<size_t>(global_run);
    *n_rows = plan->recv_block_rows[at];
    *first_row = plan->recv_blocks[at].first_row;
    *offset = (*n_rows > 0) ? plan->recv_blocks[at].offset : -1;
    return AVR_OK;
  });
}

static int plan_phase(avr_context* ctx, int phases, const avr_scene* scene,
                      const avr_frame_plan* plan, int slot, float* send_buffer,
                      uint64_t* samples_out, const FrameChunks& chunks = FrameChunks{}) {
  return guarded([&]() -> int {
    bind_device(ctx);
    require(scene != nullptr && plan != nullptr, "null argument");
    require(chunks.count == 1 || !scene->cache_classification,
            "a cached classification is not classified in chunks");
    require((chunks.classify_flags == nullptr && chunks.classify_positions == nullptr) ||
                !scene->cache_classification,
            "a cached classification is not classified by flags");
    require(static_cast<int>(scene->boxes.size()) == plan->info.n_local_boxes,
            "the scene does not hold this rank's boxes of the plan");
    if (plan->info.n_local_runs == 0) return AVR_OK;
    return render(ctx, phases, scene->boxes.data(), static_cast<int>(scene->boxes.size()),
                  scene->transform, plan->params, plan->camera, plan->local_order.data(),
                  static_cast<int>(plan->local_order.size()), plan->local_run_end.data(),
                  plan->info.n_local_runs, plan->info.n_ranks, plan->local_rects, plan->send_blocks,
                  ((phases & kMarch) && plan->tightened) ? &plan->send_spans : nullptr,
                  plan->pieces,
                  const_cast<avr_scene*>(scene), slot, send_buffer, samples_out,
                  &const_cast<avr_frame_plan*>(plan)->prologue, chunks);
  });
}

int avr_scene_set_classification_cache(avr_scene* scene, int enabled) {
  return guarded([&]() -> int {
    require(scene != nullptr, "null scene");
    scene->cache_classification = enabled != 0;
    for (auto& key : scene->classified_key) key.clear();
    return AVR_OK;
  });
}

int avr_scene_invalidate(avr_scene* scene) {
  return guarded([&]() -> int {
    require(scene != nullptr, "null scene");
    for (auto& key : scene->classified_key) key.clear();
    return AVR_OK;
  });
}

int avr_render_plan(avr_context* ctx, const avr_scene* scene, const avr_frame_plan* plan,
                    float* send_buffer, uint64_t* samples_out) {
  return plan_phase(ctx, kClassify | kMarch, scene, plan, 0, send_buffer, samples_out);
}

int avr_classify_plan(avr_context* ctx, const avr_scene* scene, const avr_frame_plan* plan,
                      int slot) {
  return plan_phase(ctx, kClassify, scene, plan, slot, nullptr, nullptr);
}

int avr_march_plan(avr_context* ctx, const avr_scene* scene, const avr_frame_plan* plan, int slot,
                   float* send_buffer, uint64_t* samples_out) {
  return plan_phase(ctx, kMarch, scene, plan, slot, send_buffer, samples_out);
}

int avr_classify_plan_chunked(avr_context* ctx, const avr_scene* scene, const avr_frame_plan* plan,
                              int slot, int n_chunks, void* const* chunk_events, int first_alone) {
  FrameChunks chunks;
  chunks.count = n_chunks;
  chunks.events = reinterpret_cast<hipEvent_t const*>(chunk_events);
  chunks.first_alone = first_alone != 0;
  return plan_phase(ctx, kClassify, scene, plan, slot, nullptr, nullptr, chunks);
}

int avr_march_plan_chunked(avr_context* ctx, const avr_scene* scene, const avr_frame_plan* plan,
                           int slot, float* send_buffer, uint64_t* samples_out, int n_chunks,
                           void* const* chunk_events) {
  FrameChunks chunks;
  chunks.count = n_chunks;
  chunks.events = reinterpret_cast<hipEvent_t const*>(chunk_events);
  return plan_phase(ctx, kMarch, scene, plan, slot, send_buffer, samples_out, chunks);
}

int avr_classify_plan_flagged(avr_context* ctx, const avr_scene* scene, const avr_frame_plan* plan,
                              int slot, const uint8_t* flags, const uint32_t* gate) {
  return guarded([&]() -> int {
    require(flags != nullptr, "null flags");
    FrameChunks chunks;
    chunks.classify_flags = flags;
    chunks.classify_gate = gate;
    return plan_phase(ctx, kClassify, scene, plan, slot, nullptr, nullptr, chunks);
  });
}

int avr_march_plan_workgroups(const avr_frame_plan* plan, int64_t* workgroups) {
  return guarded([&]() -> int {
    require(plan != nullptr && workgroups != nullptr, "null argument");
    *workgroups = march_workgroups_bound(plan->params.width, plan->params.height, plan->info.n_local_runs);
    return AVR_OK;
  });
}

int avr_classify_plan_positions(avr_context* ctx, const avr_scene* scene, const avr_frame_plan* plan,
                                int slot, const int32_t* positions, int n_positions) {
  return guarded([&]() -> int {
    require(positions != nullptr && n_positions >= 0, "null positions");
    if (n_positions == 0) return AVR_OK;
    FrameChunks chunks;
    chunks.classify_positions = positions;
    chunks.n_classify_positions = n_positions;
    return plan_phase(ctx, kClassify, scene, plan, slot, nullptr, nullptr, chunks);
  });
}

int avr_march_plan_speculative(avr_context* ctx, const avr_scene* scene, const avr_frame_plan* plan,
                               int slot, float* send_buffer, const avr_speculation* speculation) {
  return guarded([&]() -> int {
    require(speculation != nullptr, "null speculation");
    FrameChunks chunks;
    chunks.speculation = speculation;
    return plan_phase(ctx, kMarch, scene, plan, slot, send_buffer, nullptr, chunks);
  });
}

int avr_render_plan_culled(avr_context* ctx, const avr_scene* scene, const avr_frame_plan* plan,
                           int slot, float* send_buffer, uint64_t* samples_out, int n_chunks,
                           uint8_t* visibility) {
  return guarded([&]() -> int {
    require(visibility != nullptr && n_chunks >= 2, "a culled frame needs chunks and a visibility buffer");
    FrameChunks chunks;
    chunks.count = n_chunks;
    chunks.visibility = visibility;
    chunks.first_alone = true;
    return plan_phase(ctx, kClassify | kMarch, scene, plan, slot, send_buffer, samples_out, chunks);
  });
}

int avr_fold_plan(avr_context* ctx, const avr_frame_plan* plan, const float* recv_buffer,
                  float* out_piece, uint8_t* out_rgb8) {
  return avr_fold_plan_own(ctx, plan, recv_buffer, nullptr, out_piece, out_rgb8);
}

namespace {
int fold_plan(avr_context* ctx, const avr_frame_plan* plan, const float* recv_buffer,
              const float* own_send_buffer, float* out_piece, uint8_t* out_rgb8, bool to_image);
}

int avr_fold_plan_own(avr_context* ctx, const avr_frame_plan* plan, const float* recv_buffer,
                      const float* own_send_buffer, float* out_piece, uint8_t* out_rgb8) {
  return fold_plan(ctx, plan, recv_buffer, own_send_buffer, out_piece, out_rgb8, false);
}

int avr_fold_plan_image(avr_context* ctx, const avr_frame_plan* plan, const float* recv_buffer,
                        float* out_piece, uint8_t* out_rgb8_image) {
  return fold_plan(ctx, plan, recv_buffer, nullptr, out_piece, out_rgb8_image, true);
}

namespace {
int fold_plan(avr_context* ctx, const avr_frame_plan* plan, const float* recv_buffer,
              const float* own_send_buffer, float* out_piece, uint8_t* out_rgb8, bool to_image) {
  return guarded([&]() -> int {
    bind_device(ctx);
    require(plan != nullptr, "null argument");
    require(!to_image || (plan->info.n_ranks == 1 && out_rgb8 != nullptr),
            "avr_fold_plan_image is for one rank's whole image");
    // (a rank whose piece is empty -- more ranks than pixels, or row bands on a short image --
    // has nothing to fold and may pass empty buffers)
    if (plan->info.piece_end <= plan->info.piece_begin) return AVR_OK;
    require(out_piece != nullptr || out_rgb8 != nullptr, "null argument");
    require(plan->info.recv_floats == 0 || recv_buffer != nullptr, "null receive buffer");
    avr::FoldLaunch launch;
    const bool spans = plan->tightened && !plan->recv_spans.empty();
    ctx->staging.begin(plan->global_rects.size() * sizeof(avr::RunRectDev) +
                           plan->recv_blocks.size() * sizeof(avr::RunBlockDev) +
                           (spans ? plan->recv_spans.size() * sizeof(avr::RunSpanDev) : 0),
                       3);
    launch.run_rects_dev = ctx->staging.add(plan->global_rects.data(), plan->global_rects.size());
    launch.run_blocks_dev = ctx->staging.add(plan->recv_blocks.data(), plan->recv_blocks.size());
    launch.run_spans_dev =
        spans ? ctx->staging.add(plan->recv_spans.data(), plan->recv_spans.size()) : nullptr;
    ctx->staging.commit(ctx->stream);
    launch.pieces = plan->pieces;
    launch.piece = plan->piece_of_rank[static_cast<size_t>(plan->info.rank)];
    launch.width = plan->params.width;
    launch.piece_begin = plan->info.piece_begin;
    launch.piece_end = plan->info.piece_end;
    launch.n_runs = plan->info.n_runs_total;
    launch.recv = recv_buffer;
    launch.out_piece = out_piece;
    launch.out_rgb8 = out_rgb8;
    // One rank folds the whole image while the next frame's paint kernels start, and nothing
    // waits for it: one workgroup per CU keeps it out of their way (0.997 -> 0.980 ms per frame).
    // A rank of several folds its piece on the stream that also carries the exchange and the
    // gather, five kernels per frame: there it should be through quickly.
    // (A frame that found its renderer's pipeline empty has nothing to stay out of the way of,
    // and somebody is waiting for it: the whole grid, 63 -> ~15 us for 2048^2.)
    launch.max_workgroups = (plan->info.n_ranks == 1 && !ctx->fold_whole_grid) ? 256 : 0;
    {
      static const int forced = [] {  // A/B only (tools/ab_env_share.sh)
        const char* text = std::getenv("AVR_FOLD_WORKGROUPS");
        return text != nullptr ? std::atoi(text) : 0;
      }();
      if (forced > 0) launch.max_workgroups = forced;
    }
    launch.flip_height = to_image ? plan->params.height : 0;
    if (own_send_buffer != nullptr) {
      // the rank's block for itself: where the receive layout has it and where the march put it
      const int me = plan->info.rank;
      int64_t send_at = 0, recv_at = 0;
      for (int s = 0; s < me; ++s) {
        send_at += plan->send_splits[static_cast<size_t>(s)];
        recv_at += plan->recv_splits[static_cast<size_t>(s)];
      }
      const int64_t own = plan->recv_splits[static_cast<size_t>(me)];
      if (plan->send_splits[static_cast<size_t>(me)] != own) {
        throw std::runtime_error(
            "frame plan: a rank's block for itself differs between send and receive layout");
      }
      launch.own_begin = recv_at;
      launch.own_end = recv_at + own;
      // (two allocations: the distance is taken between addresses, not between pointers)
      launch.own_delta = (reinterpret_cast<intptr_t>(own_send_buffer + send_at) -
                          reinterpret_cast<intptr_t>(recv_buffer + recv_at)) /
                         static_cast<intptr_t>(sizeof(float));
    }
    return avr::launch_fold_plan(launch, ctx->stream);
  });
}
}  // namespace

int avr_visibility_graph_create(const avr_box* all_boxes, const int32_t* owner, int n_boxes,
                                int n_ranks, avr_visibility_graph** out_graph) {
  return guarded([&]() -> int {
    require(out_graph != nullptr, "null out_graph");
    *out_graph = nullptr;
    require(n_ranks >= 1, "n_ranks must be positive");
    require(n_boxes >= 0 && (n_boxes == 0 || (all_boxes != nullptr && owner != nullptr)),
            "invalid box list");
    for (int b = 0; b < n_boxes; ++b) {
      require(owner[b] >= 0 && owner[b] < n_ranks, "box owner out of range");
    }
    *out_graph = avr::visibility_graph_create(all_boxes, owner, n_boxes, n_ranks);
    return AVR_OK;
  });
}

void avr_visibility_graph_destroy(avr_visibility_graph* graph) {
  avr::visibility_graph_destroy(graph);
}

int avr_visibility_order(avr_visibility_graph* graph, const avr_camera* camera, float aspect,
                         int use_visibility_graph, const char* dot_prefix,
                         int32_t* rank_order_out, int* succeeded_out, int* n_splits_out) {
  return guarded([&]() -> int {
    require(graph != nullptr && camera != nullptr && rank_order_out != nullptr, "null argument");
    if (succeeded_out != nullptr) *succeeded_out = 1;
    if (n_splits_out != nullptr) *n_splits_out = 0;
    if (!use_visibility_graph) {
      const int n = avr::visibility_rank_count(graph);
      for (int r = 0; r < n; ++r) rank_order_out[r] = r;
      return AVR_OK;
    }
    const bool ok = avr::visibility_order(graph, *camera, aspect, dot_prefix, rank_order_out,
                                          n_splits_out);
    if (succeeded_out != nullptr) *succeeded_out = ok ? 1 : 0;
    return AVR_OK;
  });
}

int avr_tight_bounds(const avr_box* all_boxes, int n_boxes, const double fallback_min[3],
                     const double fallback_max[3], double out_min[3], double out_max[3]) {
  return guarded([&]() -> int {
    require(n_boxes >= 0 && (n_boxes == 0 || all_boxes != nullptr), "invalid box list");
    require(fallback_min != nullptr && fallback_max != nullptr && out_min != nullptr &&
                out_max != nullptr, "null argument");
    avr::tight_bounds(all_boxes, n_boxes, fallback_min, fallback_max, out_min, out_max);
    return AVR_OK;
  });
}

int avr_bbox_overlay(avr_context* ctx, const double bounds_min[3], const double bounds_max[3],
                     const avr_camera* camera, int sqrt_antialiasing, int width, int height,
                     int64_t pixel_begin, int64_t pixel_end, float* image, uint8_t* rgb8) {
  return guarded([&]() -> int {
    bind_device(ctx);
    require(bounds_min != nullptr && bounds_max != nullptr && camera != nullptr, "null argument");
    if (width <= 0 || height <= 0) return AVR_OK;  // VolumeRenderer.cpp:145-147
    require(pixel_begin >= 0 && pixel_begin <= pixel_end &&
                pixel_end <= static_cast<int64_t>(width) * height, "invalid pixel range");
    if (pixel_end == pixel_begin) return AVR_OK;
    require(image != nullptr, "null image");
    avr::OverlayPlan plan;
    avr::plan_overlay(bounds_min, bounds_max, *camera, sqrt_antialiasing, width, height, &plan);
    return avr::launch_overlay(plan, width, pixel_begin, pixel_end, nullptr, 0, image, rgb8,
                               ctx->stream);
  });
}

int avr_bbox_overlay_piece(avr_context* ctx, const avr_frame_plan* plan, const double bounds_min[3],
                           const double bounds_max[3], const avr_camera* camera, float* piece,
                           uint8_t* rgb8) {
  return guarded([&]() -> int {
    bind_device(ctx);
    require(plan != nullptr && bounds_min != nullptr && bounds_max != nullptr && camera != nullptr,
            "null argument");
    const int64_t begin = plan->info.piece_begin, end = plan->info.piece_end;
    if (end <= begin) return AVR_OK;
    require(piece != nullptr, "null image");
    avr::OverlayPlan overlay;
    avr::plan_overlay(bounds_min, bounds_max, *camera, 1, plan->params.width, plan->params.height,
                      &overlay);
    return avr::launch_overlay(overlay, plan->params.width, begin, end, &plan->pieces,
                               plan->piece_of_rank[static_cast<size_t>(plan->info.rank)], piece,
                               rgb8, ctx->stream);
  });
}

int avr_assemble_rows(avr_context* ctx, const avr_frame_plan* plan, const void* gathered,
                      int bytes_per_pixel, int flip, void* image) {
  return guarded([&]() -> int {
    bind_device(ctx);
    require(plan != nullptr && bytes_per_pixel > 0, "invalid argument");
    if (plan->info.n_pixels == 0) return AVR_OK;
    require(gathered != nullptr && image != nullptr && gathered != image, "invalid image pointers");
    return avr::launch_assemble_rows(plan->pieces, static_cast<const uint8_t*>(gathered),
                                     static_cast<int64_t>(plan->params.width) * bytes_per_pixel,
                                     flip, static_cast<uint8_t*>(image), ctx->stream);
  });
}

int avr_assemble_rows_own(avr_context* ctx, const avr_frame_plan* plan, const void* gathered,
                          const void* own_piece, int bytes_per_pixel, int flip, void* image) {
  return guarded([&]() -> int {
    bind_device(ctx);
    require(plan != nullptr && bytes_per_pixel > 0, "invalid argument");
    if (plan->info.n_pixels == 0) return AVR_OK;
    require(gathered != nullptr && image != nullptr && gathered != image, "invalid image pointers");
    const int mine = plan->piece_of_rank[static_cast<size_t>(plan->info.rank)];
    return avr::launch_assemble_rows(plan->pieces, static_cast<const uint8_t*>(gathered),
                                     static_cast<int64_t>(plan->params.width) * bytes_per_pixel,
                                     flip, static_cast<uint8_t*>(image), ctx->stream,
                                     static_cast<const uint8_t*>(own_piece), mine);
  });
}

int avr_scene_scalar_stats(avr_context* ctx, const avr_scene* scene, double stats_host[3],
                           int64_t* finite_count_host) {
  return guarded([&]() -> int {
    bind_device(ctx);
    require(scene != nullptr && stats_host != nullptr && finite_count_host != nullptr,
            "null argument");
    avr::FramePlan plan;
    avr::plan_cells(scene->boxes.data(), static_cast<int>(scene->boxes.size()), scene->transform,
                    &plan);
    const uint32_t n_tiles = plan.classify_tile_begin.back();
    void* scratch = nullptr;
    avr::hip_check(hipMalloc(&scratch, (static_cast<size_t>(avr::kScanWorkgroups) + 1) * 32),
                   "hipMalloc");
    struct { double lo, hi, lo_positive; long long finite; } result{};
    int status = AVR_OK;
    try {
      ctx->staging.begin(plan.boxes.size() * sizeof(avr::BoxDev) +
                             plan.classify_tile_begin.size() * sizeof(uint32_t), 2);
      const avr::BoxDev* boxes_dev = ctx->staging.add(plan.boxes.data(), plan.boxes.size());
      const uint32_t* tiles_dev =
          ctx->staging.add(plan.classify_tile_begin.data(), plan.classify_tile_begin.size());
      ctx->staging.commit(ctx->stream);
      char* out_dev = static_cast<char*>(scratch) + static_cast<size_t>(avr::kScanWorkgroups) * 32;
      status = avr::launch_scalar_stats(boxes_dev, tiles_dev, static_cast<int>(plan.boxes.size()),
                                        n_tiles, scratch, out_dev, ctx->stream);
      if (status == AVR_OK) {
        avr::hip_check(hipMemcpyAsync(&result, out_dev, sizeof(result), hipMemcpyDeviceToHost,
                                      ctx->stream), "hipMemcpyAsync");
        avr::wait_stream(ctx->stream, "avr_scene_scalar_stats");
      }
    } catch (...) {
      (void)hipFree(scratch);
      throw;
    }
    (void)hipFree(scratch);
    stats_host[0] = result.lo;
    stats_host[1] = result.hi;
    stats_host[2] = result.lo_positive;
    *finite_count_host = result.finite;
    return status;
  });
}

int avr_scene_transform_from_stats(const double stats[3], int64_t finite_count, int log_scale,
                                   int normalize_to_data_range, avr_scalar_transform* transform,
                                   double processed[2], float processed_range[2],
                                   float scalar_range[2]) {
  return guarded([&]() -> int {
    require(stats != nullptr && transform != nullptr && processed != nullptr &&
                processed_range != nullptr && scalar_range != nullptr, "null argument");
    avr::scene_transform_from_stats(stats, finite_count, log_scale != 0,
                                    normalize_to_data_range != 0, transform, processed,
                                    processed_range, scalar_range);
    return AVR_OK;
  });
}

int avr_scene_histogram(avr_context* ctx, const avr_scene* scene,
                        const avr_scalar_transform* transform, float range_min, float range_max,
                        int bin_count, uint64_t* counts_dev) {
  return guarded([&]() -> int {
    bind_device(ctx);
    require(scene != nullptr && transform != nullptr && counts_dev != nullptr, "null argument");
    require(bin_count > 0, "binCount must be positive");  // SceneBuilder.cpp:448-450
    const float width = range_max - range_min;
    if (!(width > 0.0f) || !std::isfinite(width)) return AVR_OK;  // (:470-472): empty histogram
    avr::FramePlan plan;
    avr::plan_cells(scene->boxes.data(), static_cast<int>(scene->boxes.size()), *transform, &plan);
    ctx->staging.begin(plan.boxes.size() * sizeof(avr::BoxDev) +
                           plan.classify_tile_begin.size() * sizeof(uint32_t), 2);
    const avr::BoxDev* boxes_dev = ctx->staging.add(plan.boxes.data(), plan.boxes.size());
    const uint32_t* tiles_dev =
        ctx->staging.add(plan.classify_tile_begin.data(), plan.classify_tile_begin.size());
    ctx->staging.commit(ctx->stream);
    return avr::launch_histogram(plan.consts, boxes_dev, tiles_dev,
                                 static_cast<int>(plan.boxes.size()),
                                 plan.classify_tile_begin.back(), range_min, range_max, bin_count,
                                 counts_dev, ctx->stream);
  });
}

static int blend_common(avr_context* ctx, int kind, const void* top, const void* bottom, void* out,
                        int64_t n_pixels) {
  return guarded([&]() -> int {
    bind_device(ctx);
    require(n_pixels >= 0, "negative pixel count");
    if (n_pixels == 0) return AVR_OK;
    require(top != nullptr && bottom != nullptr && out != nullptr, "null image");
    return avr::launch_blend(kind, top, bottom, out, n_pixels, ctx->stream);
  });
}

int avr_blend_depthsort_f32x5(avr_context* ctx, const float* top, const float* bottom, float* out,
                              int64_t n_pixels) {
  return blend_common(ctx, 0, top, bottom, out, n_pixels);
}

int avr_blend_rgba_f32x4(avr_context* ctx, const float* top, const float* bottom, float* out,
                         int64_t n_pixels) {
  return blend_common(ctx, 1, top, bottom, out, n_pixels);
}

int avr_blend_rgba_u8x4(avr_context* ctx, const uint32_t* top, const uint32_t* bottom,
                        uint32_t* out, int64_t n_pixels) {
  return blend_common(ctx, 2, top, bottom, out, n_pixels);
}

int avr_blend_regions(avr_context* ctx, int kind, const void* top, int64_t tb, int64_t te,
                      const void* bottom, int64_t bb, int64_t be, void* out) {
  return guarded([&]() -> int {
    bind_device(ctx);
    require(kind >= 0 && kind <= 2, "unknown image kind");
    require(tb <= te && bb <= be, "invalid region");
    // the reference asserts the regions touch or overlap (ImageColorOnly.hpp:129-130)
    require(tb <= be && bb <= te, "regions neither overlap nor touch");
    const int64_t n = (te > be ? te : be) - (tb < bb ? tb : bb);
    if (n == 0) return AVR_OK;
    require(out != nullptr && (te == tb || top != nullptr) && (be == bb || bottom != nullptr),
            "null image");
    return avr::launch_blend_regions(kind, top, tb, te, bottom, bb, be, out, ctx->stream);
  });
}

int avr_encode_rgba_u8(avr_context* ctx, const float* rgba, uint32_t* out, int64_t n_pixels) {
  return guarded([&]() -> int {
    bind_device(ctx);
    require(n_pixels >= 0, "negative pixel count");
    if (n_pixels == 0) return AVR_OK;
    require(rgba != nullptr && out != nullptr, "null image");
    return avr::launch_encode_u8(rgba, out, n_pixels, ctx->stream);
  });
}

int avr_decode_rgba_u8(avr_context* ctx, const uint32_t* in, float* rgba, int64_t n_pixels) {
  return guarded([&]() -> int {
    bind_device(ctx);
    require(n_pixels >= 0, "negative pixel count");
    if (n_pixels == 0) return AVR_OK;
    require(rgba != nullptr && in != nullptr, "null image");
    return avr::launch_decode_u8(in, rgba, n_pixels, ctx->stream);
  });
}

int avr_fold_runs_depthsort(avr_context* ctx, const float* const* slices_host, int n_slices,
                            float* out, int64_t n_pixels) {
  return guarded([&]() -> int {
    bind_device(ctx);
    require(n_slices >= 0 && n_pixels >= 0, "invalid argument");
    if (n_pixels == 0) return AVR_OK;
    require(out != nullptr && (n_slices == 0 || slices_host != nullptr), "null argument");
    for (int s = 0; s < n_slices; ++s) require(slices_host[s] != nullptr, "null slice");
    ctx->staging.begin(static_cast<size_t>(n_slices) * sizeof(float*), 1);
    const float* const* slices_dev = ctx->staging.add(slices_host, static_cast<size_t>(n_slices));
    ctx->staging.commit(ctx->stream);
    return avr::launch_fold_runs(slices_dev, n_slices, out, n_pixels, ctx->stream);
  });
}

int avr_downsample_depthsort(avr_context* ctx, const float* src, int target_w, int target_h,
                             int block, float* dst) {
  return guarded([&]() -> int {
    bind_device(ctx);
    // downsampleImage throws for sqrtAA <= 1 (VolumeRenderer.cpp:483-487)
    require(block > 1, "downsample expects a block size > 1");
    require(target_w >= 0 && target_h >= 0, "invalid target size");
    if (target_w == 0 || target_h == 0) return AVR_OK;
    require(src != nullptr && dst != nullptr, "null image");
    return avr::launch_downsample(src, target_w, target_h, block, dst, ctx->stream);
  });
}

int avr_flip_rows(avr_context* ctx, const uint8_t* src, int64_t row_bytes, int h, uint8_t* dst) {
  return guarded([&]() -> int {
    bind_device(ctx);
    require(row_bytes >= 0 && h >= 0, "invalid image description");
    if (row_bytes == 0 || h == 0) return AVR_OK;
    require(src != nullptr && dst != nullptr && src != dst, "invalid image pointers");
    return avr::launch_flip_rows(src, row_bytes, h, dst, ctx->stream);
  });
}

int avr_quantize_rgb8(avr_context* ctx, const float* src, int w, int h, int stride, uint8_t* dst) {
  return guarded([&]() -> int {
    bind_device(ctx);
    require(w >= 0 && h >= 0 && stride >= 3, "invalid image description");
    if (w == 0 || h == 0) return AVR_OK;
    require(src != nullptr && dst != nullptr, "null image");
    return avr::launch_quantize(src, w, h, stride, dst, ctx->stream);
  });
}

}  // extern "C"
