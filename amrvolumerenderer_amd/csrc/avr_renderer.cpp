// Frame driver of one rank: what VolumeRenderer::renderSingleTrial does between "per-box
// rendering" and the saved image (VolumeRenderer/VolumeRenderer.cpp:1103-1339), re-cut for one
// rank per GPU and pipelined over three HIP streams (include/avr_hip.h, "frame driver").
//
//   reference stage                                   here
//   ------------------------------------------------  -------------------------------------------
//   referenceSampleDistance + MPI_Allreduce (:1138)   host, from the replicated box metadata
//   BuildVisibilityOrderedGroup (:1235-1241)          avr_visibility_order (cached adjacency)
//   allgather of layer counts / hints, sort, runs     avr_frame_plan (replicated metadata)
//     (DirectSendBase.cpp:329-410)
//   per-box paint loop + owner-side run fold          classify pass (stream C) + ONE march launch
//     (:1201-1219, DirectSendBase.cpp:413-426)          (stream M) into the sparse send layout
//   one direct-send round per run (:400-446)          avr_exchange: one RCCL round (stream X)
//   receiver blend chain                              avr_fold_plan (stream X)
//   Gather(0) (:1293)                                 avr_gather (stream X)
//   downsampleImage / bounds overlay / 8 bit          avr_downsample / avr_bbox_overlay /
//     (:479-528, :139-335, SavePPM.cpp)                 avr_quantize on the root (stream X)
//
// Frames are independent, so frame i+1 is classified while frame i is marched and frame i-1 is
// exchanged, folded and gathered: two classified volumes and two send buffers rotate, events
// guard their re-use, and the host never waits for the device inside a frame.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/avr_hip_debug.h"
#include "avr_corun.h"
#include "avr_internal.h"
#include "avr_plan.h"

namespace {

void hip_ok(hipError_t err, const char* what) {
  if (err != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(err));
}

void abi_ok(int status) {
  if (status == AVR_OK) return;
  const std::string message = avr_last_error();
  if (status == AVR_ERR_INVALID_ARGUMENT) throw std::invalid_argument(message);
  // (a wait inside a nested C ABI call that ran into the deadline: the renderer fails for good)
  if (message.find("AVR_FRAME_TIMEOUT_MS") != std::string::npos) throw avr::DeadlineExceeded(message);
  throw std::runtime_error(message);
}

void require(bool condition, const char* message) {
  if (!condition) throw std::invalid_argument(message);
}

// Grow-only device buffer.  Growing frees the old block, so every stream that may still touch it
// is drained first (rare: sizes settle after the first frames; 25 % headroom).
struct DeviceBuffer {
  void* ptr = nullptr;
  size_t capacity = 0;
  ~DeviceBuffer() {
    if (ptr != nullptr) (void)hipFree(ptr);
  }
  template <typename Drain>
  void* reserve(size_t bytes, Drain&& drain) {
    if (bytes > capacity) {
      if (ptr != nullptr) {  // (a first allocation replaces nothing a frame in flight could read)
        drain();
        (void)hipFree(ptr);
      }
      ptr = nullptr;
      capacity = 0;
      const size_t want = bytes + bytes / 4 + 256;
      hip_ok(hipMalloc(&ptr, want), "hipMalloc(frame buffer)");
      capacity = want;
    }
    return ptr;
  }
};

struct PlanKey {
  avr_render_params render{};
  avr_camera camera{};
  std::vector<int32_t> group;  // explicit group order, if any
  // Field by field (bit patterns): avr_camera carries tail padding whose bytes are whatever the
  // caller's stack held, so comparing whole structs could hit on one rank and miss on another.
  bool operator==(const PlanKey& o) const {
    auto same = [](const auto& a, const auto& b) {
      static_assert(sizeof(a) == sizeof(b), "field");
      return std::memcmp(&a, &b, sizeof(a)) == 0;
    };
    return render.width == o.render.width && render.height == o.render.height &&
           same(render.box_transparency, o.render.box_transparency) &&
           render.antialiasing == o.render.antialiasing &&
           render.use_visibility_graph == o.render.use_visibility_graph &&
           render.draw_bounds == o.render.draw_bounds &&
           render.write_visibility_graph == o.render.write_visibility_graph &&
           same(camera.eye, o.camera.eye) && same(camera.look_at, o.camera.look_at) &&
           same(camera.up, o.camera.up) && same(camera.fov_y_degrees, o.camera.fov_y_degrees) &&
           same(camera.near_plane, o.camera.near_plane) &&
           same(camera.far_plane, o.camera.far_plane) && group == o.group;
  }
};

struct FrameEvents {
  hipEvent_t classify_begin = nullptr, classify_end = nullptr, march_begin = nullptr,
             march_end = nullptr;
};

}  // namespace

struct avr_renderer {
  int device = 0, rank = 0, n_ranks = 1;
  avr_comm* comm = nullptr;
  avr_context* march = nullptr;     // stream M (high priority)
  avr_context* compose = nullptr;   // stream X (high priority)
  avr_context* classify = nullptr;  // stream C (default priority)
  avr_context* pair_b = nullptr;    // stream B of the paired layout (the odd frames'; high priority)
  avr_scene* scene = nullptr;
  avr_visibility_graph* visibility = nullptr;
  std::vector<avr_box> all_boxes;
  std::vector<int32_t> owner;
  avr_scalar_transform transform{};
  double bounds_min[3]{}, bounds_max[3]{};
  double tight_min[3]{}, tight_max[3]{};
  float scalar_range[2]{0.0f, 1.0f};
  std::vector<avr_colormap_point> colormap;
  float reference_sample_distance = 0.0f;
  int march_cap = -1;  // -1: default (uncapped)
  bool cache_classification = false;
  bool tighten_exchange = true;  // avr_renderer_set_tighten
  // How the image is dealt to the ranks' pieces (avr_renderer_set_piece_layout): bands of 8 rows
  // (the height of a march wave's tile) dealt round-robin, so that every rank receives and folds
  // the same share of every screen region; one rank has one piece either way.
  int piece_layout = AVR_PIECES_ROW_BANDS;
  int band_rows = 8;
  int overlap_classify = -1;  // -1: default (1 for one rank, 0 otherwise); see avr_renderer_set_overlap
  int host_backpressure = -1;  // -1: default (ranks of several); see avr_renderer_set_host_backpressure
  hipEvent_t paired_previous = nullptr;  // paired layout: the previous frame's classify pass finished
  // Ranks of several (round 4): a new plan is agreed on over the control plane before anything of
  // its first frame is queued (avr_frame_plan_agree), the co-run search is one search of all ranks
  // (CoRunTuner::coordinated), and every host wait has a deadline after which the renderer is
  // failed for good -- the frame either completes or errors (DirectSendBase.cpp:206-220, 277).
  int plan_check = 1;          // avr_renderer_set_plan_check
  int coordinate = -1;         // avr_renderer_set_corun_coordination: -1 = ranks of several
  uint64_t settings_epoch = 1; // bumped by every setter the ranks must agree on
  std::string failed;          // non-empty: what did not finish; every later call returns it
  const char* stage = "idle";  // what the frame being queued was doing (error messages)
  std::vector<int16_t> history;  // candidate of every frame (avr_renderer_set_corun_history)
  size_t history_capacity = 0;

  // Frame plans by (render parameters, camera, group order), most recently used kept: a camera
  // that comes back (an orbit) finds its plan -- and, for N > 1, its tightened exchange layout.
  struct CachedPlan {
    PlanKey key;
    avr_frame_plan* plan = nullptr;
    uint64_t last_used = 0;
  };
  static constexpr size_t kCachedPlans = 32;
  std::vector<CachedPlan> plans;
  uint64_t plan_clock = 0;
  avr_frame_plan* plan = nullptr;  // of the last frame (owned by `plans`)
  bool have_plan = false;
  // avr_renderer_prepare may make a plan on another thread while a frame is being queued:
  // `plans_mutex` guards the cache (and `plan`, which eviction spares), `making_mutex` lets one
  // plan be made at a time (the visibility graph keeps state between calls).
  std::mutex plans_mutex, making_mutex;

  // Changes what plans are made from (scalar range, exchange layout, piece layout): waits for a
  // plan in the making on another thread (avr_renderer_prepare), so that no plan built from the
  // old settings enters the cache after it was cleared.
  template <typename Change>
  void change_plan_inputs(Change&& change) {
    std::lock_guard<std::mutex> making(making_mutex);
    change();
    forget_plans();
  }

  void forget_plans() {
    std::lock_guard<std::mutex> lock(plans_mutex);
    for (CachedPlan& entry : plans) avr_frame_plan_destroy(entry.plan);
    plans.clear();
    plan = nullptr;
    have_plan = false;
    spec.forget();  // (what was sampled under the old settings)
  }

  DeviceBuffer send[AVR_CLASSIFIED_SLOTS], recv, piece, piece_rgb8, full_rgb8, full_image, assembled_image, small_image;
  // Ranks of several: the RGB8 pieces of frame f travel to the root inside the grouped round of
  // frame f + 1 (avr_exchange_peers_gather) -- ONE RCCL launch per frame on the compositing stream
  // instead of two.  The fold of frame f + 1 must not overwrite what that round still sends: two
  // piece buffers alternate.  avr_renderer_synchronize sends what is still pending (collective).
  DeviceBuffer piece_rgb8_odd;
  int deferred_gather = -1;  // avr_renderer_set_deferred_gather: -1 = ranks of several
  struct PendingGather {
    bool valid = false;
    avr::PieceMapDev pieces{};
    std::vector<int64_t> begin, end;  // every rank's piece in the gathered buffer (pixels)
    int own_piece = 0;
    bool own_in_place = true;         // the root's assemble pass reads its own piece where it lies
    const uint8_t* piece = nullptr;   // this rank's RGB8 piece of that frame
    uint8_t* out = nullptr;           // root: that frame's rgb8_out
  } pending;
  // classified volume and send buffer f % 3
  hipEvent_t classified_event[AVR_CLASSIFIED_SLOTS] = {};  // classify pass of the volume finished
  hipEvent_t marched_event[AVR_CLASSIFIED_SLOTS] = {};     // march finished reading the volume
  hipEvent_t composed_event[AVR_CLASSIFIED_SLOTS] = {};    // stream X finished reading send[slot]
  hipEvent_t input_event = nullptr;
  // A frame may be classified and marched in depth-ordered chunks, chunk k marched while chunk
  // k + 1 is classified (avr_classify_plan_chunked) -- built for the frame that finds the pipeline
  // empty (every frame of a caller who waits for each: the reference's Render() returns after ONE
  // frame), whose classify pass and march otherwise run strictly one after the other.  Measured
  // (profiles/r5_latency/): it does not pay on this GPU -- config-4's single frame 1.48 ms in one
  // launch each, 1.50-1.55 in 2-4 chunks at any LDS reserve, 1.7-1.8 in 6-8: the two kernels
  // stretch each other while they overlap (the first march launch takes as long as the three
  // classify launches beside it) and a march in K launches costs 20 % more than in one.  So:
  // frame_chunks -1 / 1 = one launch per kernel (default), k > 1 = every frame in k chunks.
  int frame_chunks = -1;
  int balance = -1;  // avr_renderer_set_corun_balance: 0 = always the full search
  // Occlusion culling between depth-ordered chunks (avr_render_plan_culled; one rank): k >= 2 =
  // every frame in k chunks on one stream, each classify launch leaving out the boxes no ray can
  // still sample; -1 / 0 = never (the default).  Exact and tested, but not faster on the
  // configurations measured (profiles/r5_opaque/): config-4 with the reference's default
  // boxTransparency = 0 goes from 0.70 ms (pipelined pair of kernels) to 0.82 / 0.93 ms in 2 / 4
  // chunks -- with this field and the default map a ray crosses one to two 128-cell boxes before
  // its accumulator rounds to 1, so only half of the classify pass's work is culled (0.52 -> 0.26
  // ms), while a march in four launches costs 0.66 ms instead of 0.38 (every launch walks every
  // tile and evaluates the boxes behind it), on one stream with nothing beside it.
  int occlusion_chunks = -1;
  DeviceBuffer visible_flags[AVR_CLASSIFIED_SLOTS];
  hipEvent_t chunk_event[AVR_CLASSIFIED_SLOTS][AVR_MAX_FRAME_CHUNKS] = {};
  int last_chunks = 1;  // what the last frame did (avr_renderer_corun_state / diagnostics)

  // ---- visibility speculation (avr_classify_plan_positions / avr_march_plan_speculative; one rank) --
  // With the reference's default boxTransparency = 0 the march's skip test keeps config-4's rays out
  // of 118 of its 176 boxes, yet every frame reads their f64 cells.  The driver remembers, per BOX
  // of the rank, the last frame whose march sampled it (observations: a march that records the boxes
  // it samples, the flags copied to the host and read a few frames later -- by box, so they outlive
  // the plan: a moving camera keeps what it learnt).  While at most spec_worth_it of the boxes were
  // sampled in the last kSpecMemory frames, a frame classifies only those (a launch of exactly
  // their tiles), its march checks every box it is about to march against the same set, and two
  // gated launches behind it repair the frame when the set was wrong (the cells changed, the camera
  // turned) -- results never change.  Config-4 opaque: classify pass 0.55 -> 0.18 ms, pipelined
  // frame 0.64 -> 0.43 ms.  A translucent frame (every box sampled) is observed once and then left
  // alone but for one observing frame in kSpecProbeEvery.
  int speculation = -1;  // avr_renderer_set_visibility_speculation: -1 = auto (one rank), 0 = never
  struct Speculating {
    enum State { kObserving, kDeciding, kActive, kRejected, kBackoff };
    State state = kObserving;
    std::vector<int64_t> last_sampled;     // per local box: the last frame that sampled it (-1 never)
    std::vector<int32_t> positions;        // this frame's set, as positions in its layer order
    std::vector<uint8_t> flags;            // ... and as flags by position (staged with the march)
    int64_t frame = 0;                     // frames this struct has seen
    int64_t asleep_until = 0;              // kRejected / kBackoff: the next observing frame
    int next_backoff = 64;                 // after the next run of repairs
    int recent_repairs = 0, recent_frames = 0;
    int64_t last_repair = -1000;           // the frame that learnt of the latest repair
    const void* previous_plan = nullptr;   // of the frame before (only compared: a standing camera)
    // observations in flight: the march's flags by POSITION, that frame's layer order, an event
    struct Observation {
      uint8_t* host = nullptr;             // pinned, device-visible
      uint8_t* host_dev = nullptr;
      size_t capacity = 0;
      hipEvent_t copied = nullptr;
      std::vector<int32_t> order;          // position -> local box of that frame's plan
      int64_t frame = 0;
      bool pending = false;
      bool stale = false;                  // taken under settings that are gone: ignored when it arrives
    };
    static constexpr int kObservations = 4;
    Observation observations[kObservations];
    DeviceBuffer visited[AVR_CLASSIFIED_SLOTS];  // by frame % 3: what that frame's march sampled
    DeviceBuffer missed[AVR_CLASSIFIED_SLOTS];   // ... needed and found unclassified, then a counter
    uint32_t* host_miss = nullptr;         // pinned, device-visible: set by a march that missed
    uint32_t* host_miss_dev = nullptr;
    long active_frames = 0, repaired_frames = 0;
    float sampled_fraction = -1.0f;        // of the rank's boxes, in the last kSpecMemory frames
    void forget() {                        // (another transfer function, other boxes)
      last_sampled.clear();
      state = kObserving;
      sampled_fraction = -1.0f;
      for (Observation& o : observations) o.stale = o.pending;  // (their flags are of the old settings)
    }
    ~Speculating() {
      for (Observation& o : observations) {
        if (o.host != nullptr) (void)hipHostFree(o.host);
        if (o.copied != nullptr) (void)hipEventDestroy(o.copied);
      }
      if (host_miss != nullptr) (void)hipHostFree(host_miss);
    }
  } spec;
  static constexpr int kSpecMemory = 24;       // frames a box stays in the set after it was last sampled
  static constexpr int kSpecProbeEvery = 512;  // kRejected: one observing frame in so many
  static constexpr int kSpecObserveEvery = 8;  // kActive without repairs: one observed frame in so many
  float spec_worth_it = 0.85f;  // (avr_renderer_debug_set_speculation_threshold: tests)
  double spec_min_saving_ms = 0.15;  // (... sets it to 0 with any threshold: small test scenes)
  bool marched_pending[AVR_CLASSIFIED_SLOTS] = {}, composed_pending[AVR_CLASSIFIED_SLOTS] = {};
  unsigned frame = 0;

  // host time spent inside avr_renderer_render, by section (avr_renderer_host_profile)
  double host_seconds[6] = {0, 0, 0, 0, 0, 0};  // plan, classify, march, exchange, fold, gather + tail
  long host_frames = 0;

  bool timing = false;
  hipEvent_t epoch = nullptr;
  std::vector<FrameEvents> timed;
  // CoRunTuner::kBalance: the two kernels' own durations of the frames in flight (a ring of timing
  // events, read back a few frames later, in frame order)
  struct Probe {
    FrameEvents events;
    int candidate = 0;
  };
  static constexpr unsigned kProbes = 8;
  Probe probes[kProbes];
  unsigned probe_head = 0, probe_tail = 0;  // [tail, head) are in flight

  // How the two kernels of a frame share the GPU (CoRunTuner below): the caller's wishes ...
  int share_fixed = -1;   // >= 0: the caller's LDS reserve (avr_renderer_set_classify_share)
  // ... and the tuner's state
  CoRunTuner tuner;
  hipEvent_t window_begin = nullptr, window_end = nullptr;  // timing events on the march stream
  bool last_overlap = false;  // what the last frame did (avr_renderer_corun_state)
  bool last_paired = false;
  int last_reserve = 0;
  bool pipeline_idle = true;  // nothing in flight: the next classify pass has the GPU to itself

  ~avr_renderer() {
    (void)hipSetDevice(device);
    if (!failed.empty()) {
      // A stream of this renderer does not move (a collective whose peer never came): waiting for
      // it, or freeing device memory (which waits for the device), would hang the caller too.
      // Everything on the device is leaked; the process is expected to report the error and exit.
      forget_plans();
      for (DeviceBuffer* buffer : {&send[0], &send[1], &send[2], &recv, &piece, &piece_rgb8, &piece_rgb8_odd,
                                   &full_rgb8, &full_image, &assembled_image, &small_image,
                                   &visible_flags[0], &visible_flags[1], &visible_flags[2]}) {
        buffer->ptr = nullptr;  // (hipFree waits for the device)
      }
      for (int i = 0; i < AVR_CLASSIFIED_SLOTS; ++i) spec.visited[i].ptr = spec.missed[i].ptr = nullptr;
      for (Speculating::Observation& o : spec.observations) {
        o.host = nullptr;  // (hipHostFree waits as well)
        o.copied = nullptr;
      }
      spec.host_miss = nullptr;
      return;
    }
    for (avr_context* ctx : {classify, march, compose, pair_b}) {
      if (ctx != nullptr) (void)avr_context_synchronize(ctx);
    }
    clear_timing();
    for (hipEvent_t ev : {window_begin, window_end}) {
      if (ev != nullptr) (void)hipEventDestroy(ev);
    }
    for (Probe& probe : probes) {
      for (hipEvent_t ev : {probe.events.classify_begin, probe.events.classify_end,
                            probe.events.march_begin, probe.events.march_end}) {
        if (ev != nullptr) (void)hipEventDestroy(ev);
      }
    }
    for (hipEvent_t ev : classified_event) {
      if (ev != nullptr) (void)hipEventDestroy(ev);
    }
    for (hipEvent_t ev : marched_event) {
      if (ev != nullptr) (void)hipEventDestroy(ev);
    }
    for (hipEvent_t ev : composed_event) {
      if (ev != nullptr) (void)hipEventDestroy(ev);
    }
    for (auto& events : chunk_event) {
      for (hipEvent_t ev : events) {
        if (ev != nullptr) (void)hipEventDestroy(ev);
      }
    }
    if (input_event != nullptr) (void)hipEventDestroy(input_event);
    if (epoch != nullptr) (void)hipEventDestroy(epoch);
    forget_plans();
    if (visibility != nullptr) avr_visibility_graph_destroy(visibility);
    if (scene != nullptr) avr_scene_destroy(scene);
    for (avr_context* ctx : {classify, march, compose}) avr_context_destroy(ctx);
    if (pair_b != nullptr) avr_context_destroy(pair_b);
  }

  void clear_timing() {
    for (FrameEvents& e : timed) {
      for (hipEvent_t ev : {e.classify_begin, e.classify_end, e.march_begin, e.march_end}) {
        if (ev != nullptr) (void)hipEventDestroy(ev);
      }
    }
    timed.clear();
  }

  hipStream_t stream_of(avr_context* ctx) { return static_cast<hipStream_t>(avr::context_stream(ctx)); }

  // What the ranks must have set alike (mixed into the plan agreement's digest).
  uint64_t settings_digest() const {
    uint64_t digest = 0xcbf29ce484222325ull;
    for (int64_t value : {static_cast<int64_t>(overlap_classify), static_cast<int64_t>(share_fixed),
                          static_cast<int64_t>(cache_classification), static_cast<int64_t>(tighten_exchange),
                          static_cast<int64_t>(coordinate), static_cast<int64_t>(plan_check),
                          static_cast<int64_t>(deferred_gather)}) {
      digest = (digest ^ static_cast<uint64_t>(value)) * 0x100000001b3ull;
    }
    return digest;
  }

  std::string describe() const {
    char text[256];
    std::snprintf(text, sizeof text,
                  "rank %d of %d, frame %u, stage: %s; co-run: %s, LDS reserve %d, %s after %ld windows",
                  rank, n_ranks, frame, stage,
                  last_paired ? "paired" : last_overlap ? "side by side" : "back to back", last_reserve,
                  tuner.settled() ? "settled" : "searching", tuner.windows);
    return text;
  }

  void drain_all() {
    // (waits with the deadline, stream by stream, so that an error names the one that is stuck)
    const std::pair<avr_context*, const char*> streams[] = {
        {classify, "the classify stream"}, {march, "the march stream"},
        {compose, "the compositing stream (exchange, fold, gather)"}, {pair_b, "the second march stream"}};
    for (const auto& entry : streams) {
      if (entry.first != nullptr) avr::wait_stream_deadline(stream_of(entry.first), entry.second);
    }
    pipeline_idle = true;
    paired_previous = nullptr;  // nothing is in flight: nothing to order the next classify pass after
    probe_tail = probe_head;    // (their frames ran into the drain: not the steady state)
    tuner.drained();
  }
};

namespace {

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

// A call on a renderer: refused once the renderer has failed; a wait that runs into the deadline
// fails it (the message says which stream of which rank in which frame, and how the co-run search
// stood) -- the caller reports it and exits, and its peers run into their own deadlines.
template <typename F>
int guarded_renderer(avr_renderer* r, F&& body) {
  return guarded([&]() -> int {
    if (r == nullptr) throw std::invalid_argument("null renderer");
    if (!r->failed.empty()) throw std::runtime_error(r->failed);
    // (the default deadline belongs to frames with collectives in them: avr_internal.h)
    const avr::CollectiveScope collective(r->n_ranks > 1 || r->comm != nullptr);
    try {
      return body();
    } catch (const avr::DeadlineExceeded& e) {
      r->failed = std::string(e.what()) + " [" + r->describe() + "]";
      throw std::runtime_error(r->failed);
    }
  });
}

// validateRenderParameters (VolumeRenderer.cpp:562-579); returns sqrt(antialiasing)
int validate(const avr_render_params& p) {
  require(p.width > 0 && p.height > 0, "image dimensions must be positive");
  require(p.box_transparency >= 0.0f && p.box_transparency <= 1.0f,
          "box_transparency must be in [0, 1]");
  require(p.antialiasing >= 1, "antialiasing must be >= 1");
  const int root = static_cast<int>(std::lround(std::sqrt(static_cast<double>(p.antialiasing))));
  require(root * root == p.antialiasing, "antialiasing must be a perfect square");
  return root;
}

hipEvent_t make_event(bool timing) {
  hipEvent_t event = nullptr;
  // (hipEventReleaseToDevice on the timing events was measured and changes nothing)
  hip_ok(hipEventCreateWithFlags(&event, timing ? hipEventDefault : avr::ordering_event_flags()),
         "hipEventCreate");
  return event;
}

// The frame plan of (render parameters, camera, group order): from the cache, or made now --
// visibility order, layer plan, and for N > 1 the tightened exchange layout: host geometry only.
// `use`: the caller renders with it (it becomes the renderer's current plan); otherwise it is only
// put into the cache (avr_renderer_prepare, possibly on another thread than the frames').
const avr_frame_plan* plan_for(avr_renderer* r, const avr_render_params& render,
                               const avr_camera& camera, const int32_t* group_order, bool use) {
  const int root = validate(render);
  const int width = render.width, height = render.height;
  PlanKey key;
  key.render = render;
  key.camera = camera;
  if (group_order != nullptr) key.group.assign(group_order, group_order + r->n_ranks);
  auto lookup = [&]() -> avr_frame_plan* {  // plans_mutex held
    if (render.write_visibility_graph) return nullptr;
    for (avr_renderer::CachedPlan& entry : r->plans) {
      if (entry.key == key) {
        entry.last_used = ++r->plan_clock;
        if (use) {
          r->plan = entry.plan;
          r->have_plan = true;
        }
        return entry.plan;
      }
    }
    return nullptr;
  };
  {
    std::lock_guard<std::mutex> lock(r->plans_mutex);
    if (avr_frame_plan* found = lookup()) return found;
  }
  std::lock_guard<std::mutex> making(r->making_mutex);
  {  // (the other thread may have made it meanwhile)
    std::lock_guard<std::mutex> lock(r->plans_mutex);
    if (avr_frame_plan* found = lookup()) return found;
  }
  std::vector<int32_t> order;
  const int32_t* group = group_order;
  if (group == nullptr && r->visibility != nullptr) {
    order.resize(static_cast<size_t>(r->n_ranks));
    // aspect as VolumeRenderer.cpp:1114 computes it (float division of the image size)
    const float aspect = static_cast<float>(width) / static_cast<float>(std::max(height, 1));
    int succeeded = 1;
    abi_ok(avr_visibility_order(r->visibility, &camera, aspect, render.use_visibility_graph,
                                (render.write_visibility_graph && r->rank == 0)
                                    ? "visibility_graph_"
                                    : nullptr,
                                order.data(), &succeeded, nullptr));
    group = order.data();
  }
  avr_paint_params params{};
  params.width = width * root;
  params.height = height * root;
  params.scalar_range[0] = r->scalar_range[0];
  params.scalar_range[1] = r->scalar_range[1];
  params.box_transparency = render.box_transparency;
  params.reference_sample_distance = r->reference_sample_distance;
  std::copy(r->bounds_min, r->bounds_min + 3, params.bounds_min);
  std::copy(r->bounds_max, r->bounds_max + 3, params.bounds_max);
  params.colormap = r->colormap.empty() ? nullptr : r->colormap.data();
  params.colormap_count = static_cast<int32_t>(r->colormap.size());
  avr_frame_plan* fresh = nullptr;
  abi_ok(avr_frame_plan_create_pieces(r->all_boxes.data(), r->owner.data(),
                                      static_cast<int>(r->all_boxes.size()), r->n_ranks, r->rank,
                                      group, &params, &camera, r->piece_layout, r->band_rows,
                                      &fresh));
  if (r->n_ranks > 1 && r->tighten_exchange) {
    // The exchange layout is tightened to the runs' per-row extents when the plan is made
    // (25-50 % fewer bytes on the links from a camera's FIRST frame on; tens of microseconds
    // of host geometry, avr_plan.cpp).  The decision depends on nothing but the renderer's
    // settings, so every rank of the frame takes it alike.
    const int status = avr_frame_plan_tighten(fresh, r->all_boxes.data(),
                                              static_cast<int>(r->all_boxes.size()));
    if (status != AVR_OK) {
      avr_frame_plan_destroy(fresh);
      abi_ok(status);
    }
  }
  std::lock_guard<std::mutex> lock(r->plans_mutex);
  if (r->plans.size() >= avr_renderer::kCachedPlans) {
    // the least recently used one goes -- but never the current frame's (a frame being queued on
    // another thread reads it; otherwise a plan is host data only: every launch copied what it
    // reads into its own descriptors)
    size_t oldest = r->plans.size();
    for (size_t i = 0; i < r->plans.size(); ++i) {
      if (r->plans[i].plan == r->plan && r->have_plan) continue;
      if (oldest == r->plans.size() || r->plans[i].last_used < r->plans[oldest].last_used) {
        oldest = i;
      }
    }
    if (oldest != r->plans.size()) {
      avr_frame_plan_destroy(r->plans[oldest].plan);
      r->plans.erase(r->plans.begin() + static_cast<std::ptrdiff_t>(oldest));
    }
  }
  r->plans.push_back(avr_renderer::CachedPlan{key, fresh, ++r->plan_clock});
  if (use) {
    r->plan = fresh;
    r->have_plan = true;
  }
  return fresh;
}

}  // namespace

extern "C" {

int avr_renderer_create(int device_id, int rank, int n_ranks, avr_comm* comm,
                        const avr_box* all_boxes, const int32_t* owner, int n_boxes,
                        const avr_scalar_transform* transform, const double bounds_min[3],
                        const double bounds_max[3], const float scalar_range[2],
                        const avr_colormap_point* colormap, int colormap_count,
                        avr_renderer** out_renderer) {
  return guarded([&]() -> int {
    require(out_renderer != nullptr && transform != nullptr && bounds_min != nullptr &&
                bounds_max != nullptr && scalar_range != nullptr, "null argument");
    *out_renderer = nullptr;
    require(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "invalid rank");
    require(n_boxes >= 0 && (n_boxes == 0 || (all_boxes != nullptr && owner != nullptr)),
            "invalid box list");
    require(colormap_count >= 0 && (colormap_count == 0 || colormap != nullptr), "invalid color map");
    require(n_ranks == 1 || comm != nullptr, "a communicator is required for more than one rank");
    require(comm == nullptr || (avr_comm_size(comm) == n_ranks && avr_comm_rank(comm) == rank),
            "the communicator belongs to another rank / rank count");
    auto r = std::make_unique<avr_renderer>();
    r->device = device_id;
    r->rank = rank;
    r->n_ranks = n_ranks;
    r->comm = comm;
    r->all_boxes.assign(all_boxes, all_boxes + n_boxes);
    r->owner.assign(owner, owner + n_boxes);
    r->transform = *transform;
    std::copy(bounds_min, bounds_min + 3, r->bounds_min);
    std::copy(bounds_max, bounds_max + 3, r->bounds_max);
    r->scalar_range[0] = scalar_range[0];
    r->scalar_range[1] = scalar_range[1];
    r->colormap.assign(colormap, colormap + colormap_count);
    // The march and the compositing stream are high priority, the classify stream is not: not for
    // the priority itself but because HIP multiplexes streams onto a few hardware queues in
    // creation order and the two priority classes never share one (when classify and march landed
    // on one queue the kernels ran strictly one after the other).
    abi_ok(avr_context_create_with_priority(device_id, 1, &r->march));
    abi_ok(avr_context_create_with_priority(device_id, 1, &r->compose));
    abi_ok(avr_context_create_with_priority(device_id, 0, &r->classify));
    if (const char* bytes = std::getenv("AVR_CLASSIFY_LDS_RESERVE")) {  // experiment: fixed share
      r->share_fixed = std::clamp(std::atoi(bytes), 0, static_cast<int>(AVR_CLASSIFY_LDS_RESERVE_MAX));
    }
    std::vector<avr_box> local;
    for (int b = 0; b < n_boxes; ++b) {
      require(owner[b] >= 0 && owner[b] < n_ranks, "box owner out of range");
      if (owner[b] == rank) {
        require(all_boxes[b].cells != nullptr, "a box of this rank has no cell data");
        local.push_back(all_boxes[b]);
      }
    }
    abi_ok(avr_scene_create(r->march, local.data(), static_cast<int>(local.size()), transform,
                            &r->scene));
    // coarsest min spacing over all ranks == MPI_Allreduce(MAX) of VolumeRenderer.cpp:1166
    abi_ok(avr_reference_sample_distance(all_boxes, n_boxes, bounds_min, bounds_max,
                                         &r->reference_sample_distance));
    // computeTightBounds (:791-848): min / max over all ranks' boxes == over all_boxes
    abi_ok(avr_tight_bounds(all_boxes, n_boxes, bounds_min, bounds_max, r->tight_min, r->tight_max));
    if (n_ranks > 1) {
      abi_ok(avr_visibility_graph_create(all_boxes, owner, n_boxes, n_ranks, &r->visibility));
    }
    hip_ok(hipSetDevice(device_id), "hipSetDevice");
    for (hipEvent_t& ev : r->classified_event) ev = make_event(false);
    for (hipEvent_t& ev : r->marched_event) ev = make_event(false);
    for (hipEvent_t& ev : r->composed_event) ev = make_event(false);
    // (the caller's producer stream is not ours to reason about: this one keeps the system fence)
    hip_ok(hipEventCreateWithFlags(&r->input_event, hipEventDisableTiming), "hipEventCreate");
    *out_renderer = r.release();
    return AVR_OK;
  });
}

void avr_renderer_destroy(avr_renderer* renderer) { delete renderer; }

int avr_renderer_set_options(avr_renderer* r, int march_workgroups_per_cu, int cache_classification) {
  return guarded([&]() -> int {
    require(r != nullptr, "null renderer");
    require(march_workgroups_per_cu >= -1 && march_workgroups_per_cu <= 8,
            "march_workgroups_per_cu must be in [-1, 8]");
    r->march_cap = march_workgroups_per_cu;
    r->cache_classification = cache_classification != 0;
    ++r->settings_epoch;
    abi_ok(avr_scene_set_classification_cache(r->scene, r->cache_classification ? 1 : 0));
    return AVR_OK;
  });
}

int avr_renderer_set_scalar_range(avr_renderer* r, const float scalar_range[2]) {
  return guarded([&]() -> int {
    require(r != nullptr && scalar_range != nullptr, "null argument");
    r->change_plan_inputs([&] {  // the plans carry the paint parameters
      r->scalar_range[0] = scalar_range[0];
      r->scalar_range[1] = scalar_range[1];
    });
    return AVR_OK;
  });
}

int avr_renderer_invalidate(avr_renderer* r) {
  return guarded([&]() -> int {
    require(r != nullptr, "null renderer");
    return avr_scene_invalidate(r->scene);
  });
}

int avr_renderer_set_classify_share(avr_renderer* r, int bytes) {
  return guarded([&]() -> int {
    require(r != nullptr, "null renderer");
    require(bytes >= -1 && bytes <= AVR_CLASSIFY_LDS_RESERVE_MAX,
            "bytes must be -1 or in [0, AVR_CLASSIFY_LDS_RESERVE_MAX]");
    r->share_fixed = bytes;
    ++r->settings_epoch;
    return AVR_OK;
  });
}

int avr_renderer_corun_state(const avr_renderer* r, int* overlap_out, int* reserve_bytes_out,
                             int* settled_out, long* windows_out) {
  return guarded([&]() -> int {  // (still answered by a failed renderer: the caller reports it)
    require(r != nullptr, "null renderer");
    const CoRunTuner& t = r->tuner;
    if (overlap_out != nullptr) *overlap_out = r->last_paired ? 2 : r->last_overlap ? 1 : 0;
    if (reserve_bytes_out != nullptr) *reserve_bytes_out = r->last_reserve;
    if (settled_out != nullptr) *settled_out = t.settled() ? 1 : 0;
    if (windows_out != nullptr) *windows_out = t.windows;
    return AVR_OK;
  });
}

int avr_renderer_set_tighten(avr_renderer* r, int enabled) {
  return guarded([&]() -> int {
    require(r != nullptr, "null renderer");
    r->change_plan_inputs([&] { r->tighten_exchange = enabled != 0; });  // the next frame plans afresh
    return AVR_OK;
  });
}

int avr_renderer_set_piece_layout(avr_renderer* r, int piece_layout, int band_rows) {
  return guarded([&]() -> int {
    require(r != nullptr, "null renderer");
    require(piece_layout == AVR_PIECES_CONTIGUOUS || piece_layout == AVR_PIECES_ROW_BANDS,
            "unknown piece layout");
    require(piece_layout == AVR_PIECES_CONTIGUOUS ||
                (band_rows >= 1 && (band_rows & (band_rows - 1)) == 0),
            "band_rows must be a power of two");
    r->change_plan_inputs([&] {
      r->piece_layout = piece_layout;
      r->band_rows = (piece_layout == AVR_PIECES_ROW_BANDS) ? band_rows : 1;
    });
    return AVR_OK;
  });
}

int avr_renderer_set_host_backpressure(avr_renderer* r, int mode) {
  return guarded_renderer(r, [&]() -> int {
    require(mode >= -1 && mode <= 1, "mode must be -1, 0 or 1");
    r->drain_all();
    r->host_backpressure = mode;
    return AVR_OK;
  });
}

int avr_renderer_set_overlap(avr_renderer* r, int overlap_classify) {
  return guarded_renderer(r, [&]() -> int {
    r->drain_all();
    r->overlap_classify = overlap_classify;
    ++r->settings_epoch;
    return AVR_OK;
  });
}

int avr_renderer_set_frame_chunks(avr_renderer* r, int chunks) {
  return guarded_renderer(r, [&]() -> int {
    require(chunks == -1 || (chunks >= 1 && chunks <= AVR_MAX_FRAME_CHUNKS),
            "chunks must be -1 or in [1, AVR_MAX_FRAME_CHUNKS]");
    r->frame_chunks = chunks;
    return AVR_OK;
  });
}

int avr_renderer_set_occlusion_culling(avr_renderer* r, int chunks) {
  return guarded_renderer(r, [&]() -> int {
    require(chunks == -1 || chunks == 0 || (chunks >= 2 && chunks <= AVR_MAX_FRAME_CHUNKS),
            "chunks must be -1, 0 or in [2, AVR_MAX_FRAME_CHUNKS]");
    r->drain_all();
    r->occlusion_chunks = chunks;
    return AVR_OK;
  });
}

int avr_renderer_set_visibility_speculation(avr_renderer* r, int mode) {
  return guarded_renderer(r, [&]() -> int {
    require(mode >= -1 && mode <= 1, "mode must be -1, 0 or 1");
    r->drain_all();
    if ((r->speculation != 0) != (mode != 0) && r->spec.state == avr_renderer::Speculating::kActive) {
      r->tuner.restart();  // (the classify pass changes its length)
    }
    r->speculation = mode;
    r->spec.forget();  // (whatever was decided is decided again)
    return AVR_OK;
  });
}

int avr_renderer_debug_set_speculation_threshold(avr_renderer* r, float sampled_fraction) {
  return guarded_renderer(r, [&]() -> int {
    require(sampled_fraction >= 0.0f && sampled_fraction <= 1.0f, "the fraction must be in [0, 1]");
    r->drain_all();
    r->spec_worth_it = sampled_fraction;
    r->spec_min_saving_ms = 0.0;
    r->spec.forget();
    return AVR_OK;
  });
}

int avr_renderer_speculation_state(const avr_renderer* r, int* state, int64_t* speculative_frames,
                                   int64_t* repaired_frames, float* sampled_fraction) {
  if (r == nullptr) return AVR_ERR_INVALID_ARGUMENT;
  const avr_renderer::Speculating& sp = r->spec;
  if (state != nullptr) {
    *state = (r->speculation == 0 || r->n_ranks > 1) ? -1 : static_cast<int>(sp.state);
  }
  if (speculative_frames != nullptr) *speculative_frames = sp.active_frames;
  if (repaired_frames != nullptr) *repaired_frames = sp.repaired_frames;
  if (sampled_fraction != nullptr) *sampled_fraction = sp.sampled_fraction < 0.0f ? -1.0f : sp.sampled_fraction;
  return AVR_OK;
}

int avr_renderer_set_corun_balance(avr_renderer* r, int mode) {
  return guarded_renderer(r, [&]() -> int {
    require(mode >= -1 && mode <= 1, "mode must be -1, 0 or 1");
    r->drain_all();
    r->balance = mode;
    return AVR_OK;
  });
}

int avr_renderer_last_frame_chunks(const avr_renderer* r) { return r != nullptr ? r->last_chunks : -1; }

int avr_renderer_set_deferred_gather(avr_renderer* r, int mode) {
  return guarded_renderer(r, [&]() -> int {
    require(mode >= -1 && mode <= 1, "mode must be -1, 0 or 1");
    require(!r->pending.valid, "a frame's gather is pending: call avr_renderer_synchronize first");
    r->deferred_gather = mode;
    ++r->settings_epoch;
    return AVR_OK;
  });
}

int avr_renderer_set_plan_check(avr_renderer* r, int mode) {
  return guarded_renderer(r, [&]() -> int {
    require(mode == 0 || mode == 1, "mode must be 0 or 1");
    r->plan_check = mode;
    ++r->settings_epoch;
    return AVR_OK;
  });
}

int avr_renderer_set_corun_coordination(avr_renderer* r, int mode) {
  return guarded_renderer(r, [&]() -> int {
    require(mode >= -1 && mode <= 1, "mode must be -1, 0 or 1");
    r->drain_all();
    r->coordinate = mode;
    ++r->settings_epoch;
    return AVR_OK;
  });
}

int avr_renderer_set_corun_history(avr_renderer* r, int frames) {
  return guarded_renderer(r, [&]() -> int {
    require(frames >= 0, "frames must not be negative");
    r->history.clear();
    r->history_capacity = static_cast<size_t>(frames);
    r->history.reserve(r->history_capacity);
    return AVR_OK;
  });
}

int avr_renderer_corun_history(const avr_renderer* r, int16_t* candidates_out, int capacity,
                               int* frames_out) {
  return guarded([&]() -> int {
    require(r != nullptr && frames_out != nullptr && capacity >= 0 &&
                (capacity == 0 || candidates_out != nullptr), "invalid argument");
    *frames_out = static_cast<int>(r->history.size());
    const size_t n = std::min(r->history.size(), static_cast<size_t>(capacity));
    std::copy(r->history.begin(), r->history.begin() + static_cast<std::ptrdiff_t>(n), candidates_out);
    return AVR_OK;
  });
}

const char* avr_renderer_failure(const avr_renderer* r) {
  return (r == nullptr || r->failed.empty()) ? nullptr : r->failed.c_str();
}

int avr_renderer_reference_sample_distance(const avr_renderer* r, float* out) {
  return guarded([&]() -> int {
    require(r != nullptr && out != nullptr, "null argument");
    *out = r->reference_sample_distance;
    return AVR_OK;
  });
}

void* avr_renderer_stream(avr_renderer* r, int which) {
  if (r == nullptr) return nullptr;
  avr_context* ctx = (which == 0) ? r->classify : (which == 1) ? r->march : r->compose;
  try {
    return avr::context_stream(ctx);
  } catch (...) {
    return nullptr;
  }
}

int avr_renderer_synchronize(avr_renderer* r) {
  return guarded_renderer(r, [&]() -> int {
    r->stage = "synchronize";
    hip_ok(hipSetDevice(r->device), "hipSetDevice");
    if (r->pending.valid) {
      // the last frame's RGB8 pieces have not travelled yet: a gather round of their own
      // (collective -- every rank synchronises after the same frame)
      r->stage = "synchronize: the last frame's gather";
      avr_renderer::PendingGather& pending = r->pending;
      uint8_t* gathered = nullptr;
      if (r->rank == 0) {
        const size_t bytes = static_cast<size_t>(pending.pieces.width) * pending.pieces.height * 3 + 1;
        gathered = static_cast<uint8_t*>(r->full_rgb8.reserve(bytes, [&] { r->drain_all(); }));
      }
      avr_gather_op op{};
      op.piece = pending.piece;
      op.bytes_per_pixel = 3;
      op.root = 0;
      op.full = gathered;
      op.begin = pending.begin.data();
      op.end = pending.end.data();
      op.skip_own = pending.own_in_place ? 1 : 0;
      abi_ok(avr_gather_run(r->compose, r->comm, &op));
      if (r->rank == 0 &&
          avr::launch_assemble_rows(pending.pieces, gathered,
                                    static_cast<int64_t>(pending.pieces.width) * 3, /*flip=*/1,
                                    pending.out, r->stream_of(r->compose),
                                    pending.own_in_place ? pending.piece : nullptr,
                                    pending.own_piece) != AVR_OK) {
        throw std::runtime_error(avr_last_error());
      }
      pending.valid = false;
      r->stage = "synchronize";
    }
    r->drain_all();
    r->stage = "idle";
    return AVR_OK;
  });
}

int avr_renderer_outputs_complete(const avr_renderer* r, uint64_t* complete_out, uint64_t* frames_out) {
  return guarded([&]() -> int {
    require(r != nullptr && complete_out != nullptr, "null argument");
    // (a pending gather is the last frame's: its bytes travel with the next round)
    *complete_out = static_cast<uint64_t>(r->frame) - (r->pending.valid ? 1u : 0u);
    if (frames_out != nullptr) *frames_out = r->frame;
    return AVR_OK;
  });
}

int avr_renderer_plan_info(const avr_renderer* r, avr_frame_plan_info* out) {
  return guarded([&]() -> int {
    require(r != nullptr && out != nullptr, "null argument");
    require(r->have_plan, "no frame has been rendered yet");
    return avr_frame_plan_get_info(r->plan, out);
  });
}

int avr_renderer_host_profile(avr_renderer* r, double seconds_out[6], long* frames_out, int reset) {
  return guarded([&]() -> int {
    require(r != nullptr && seconds_out != nullptr && frames_out != nullptr, "null argument");
    std::copy(r->host_seconds, r->host_seconds + 6, seconds_out);
    *frames_out = r->host_frames;
    if (reset) {
      std::fill(r->host_seconds, r->host_seconds + 6, 0.0);
      r->host_frames = 0;
    }
    return AVR_OK;
  });
}

int avr_renderer_set_timing(avr_renderer* r, int enabled) {
  return guarded_renderer(r, [&]() -> int {
    hip_ok(hipSetDevice(r->device), "hipSetDevice");
    r->drain_all();
    r->clear_timing();
    r->timing = enabled != 0;
    if (r->timing) {
      if (r->epoch == nullptr) r->epoch = make_event(true);
      hip_ok(hipEventRecord(r->epoch, r->stream_of(r->march)), "hipEventRecord");
    }
    return AVR_OK;
  });
}

int avr_renderer_timings(avr_renderer* r, double* classify_ms, double* march_ms, double* busy_ms,
                         int* frames) {
  return guarded_renderer(r, [&]() -> int {
    require(classify_ms != nullptr && march_ms != nullptr && busy_ms != nullptr &&
                frames != nullptr, "null argument");
    hip_ok(hipSetDevice(r->device), "hipSetDevice");
    r->drain_all();
    *frames = static_cast<int>(r->timed.size());
    *classify_ms = *march_ms = *busy_ms = 0.0;
    if (r->timed.empty()) return AVR_OK;
    // Each kernel's own duration, and the length of the union of all their execution intervals:
    // the classify pass of frame i+1 runs beside the march of frame i, so the durations overlap.
    std::vector<std::pair<float, float>> spans;
    double classify = 0.0, march = 0.0;
    for (const FrameEvents& e : r->timed) {
      float c = 0.0f, m = 0.0f, c0 = 0.0f, c1 = 0.0f, m0 = 0.0f, m1 = 0.0f;
      hip_ok(hipEventElapsedTime(&c, e.classify_begin, e.classify_end), "hipEventElapsedTime");
      hip_ok(hipEventElapsedTime(&m, e.march_begin, e.march_end), "hipEventElapsedTime");
      hip_ok(hipEventElapsedTime(&c0, r->epoch, e.classify_begin), "hipEventElapsedTime");
      hip_ok(hipEventElapsedTime(&c1, r->epoch, e.classify_end), "hipEventElapsedTime");
      hip_ok(hipEventElapsedTime(&m0, r->epoch, e.march_begin), "hipEventElapsedTime");
      hip_ok(hipEventElapsedTime(&m1, r->epoch, e.march_end), "hipEventElapsedTime");
      classify += c;
      march += m;
      spans.emplace_back(c0, c1);
      spans.emplace_back(m0, m1);
    }
    std::sort(spans.begin(), spans.end());
    double busy = 0.0;
    float cursor = -1e30f;
    for (const auto& span : spans) {
      if (span.second > cursor) {
        busy += span.second - std::max(span.first, cursor);
        cursor = span.second;
      }
    }
    const double n = static_cast<double>(r->timed.size());
    *classify_ms = classify / n;
    *march_ms = march / n;
    *busy_ms = busy / n;
    return AVR_OK;
  });
}

int avr_renderer_prepare(avr_renderer* r, const avr_render_params* render, const avr_camera* camera,
                         const int32_t* group_order) {
  return guarded([&]() -> int {
    require(r != nullptr && render != nullptr && camera != nullptr, "null argument");
    (void)plan_for(r, *render, *camera, group_order, /*use=*/false);
    return AVR_OK;
  });
}

int avr_renderer_render(avr_renderer* r, const avr_render_params* render, const avr_camera* camera,
                        const int32_t* group_order, void* input_stream, uint64_t* samples_out,
                        int want_image, uint8_t* rgb8_out, float* image_out) {
  return guarded_renderer(r, [&]() -> int {
    require(render != nullptr && camera != nullptr, "null argument");
    const int root = validate(*render);
    hip_ok(hipSetDevice(r->device), "hipSetDevice");
    const bool is_root = r->rank == 0;
    require(!is_root || rgb8_out != nullptr, "the root rank needs an rgb8 output buffer");
    require(!is_root || !want_image || image_out != nullptr,
            "want_image needs an image output buffer on the root rank");
    const int width = render->width, height = render->height;

    using Clock = std::chrono::steady_clock;
    auto mark = Clock::now();
    auto lap = [&](int section) {
      const auto now = Clock::now();
      r->host_seconds[section] += std::chrono::duration<double>(now - mark).count();
      mark = now;
    };
    // ---- host plan (re-used while camera and parameters repeat; avr_renderer_prepare may have
    // made it ahead of time on another thread) ---------------------------------------------------
    r->stage = "frame plan";
    const avr_frame_plan* plan = plan_for(r, *render, *camera, group_order, /*use=*/true);
    const avr_frame_plan_info& info = plan->info;
    const int64_t piece_pixels = info.piece_end - info.piece_begin;
    const bool many = r->n_ranks > 1;
    // ---- a NEW plan of a rank of several is agreed on before anything of its first frame is
    // queued: a grouped ncclSend / ncclRecv round whose two sides disagree on a block size never
    // ends (the reference would notice in the metadata message of every transfer,
    // Common/Image.cpp:62-90).  Once per plan (and again after a setting changed), never per frame;
    // on disagreement EVERY rank returns the error here (avr_frame_plan_agree).
    if (many && r->plan_check != 0 && plan->agreed_epoch != r->settings_epoch) {
      r->stage = "plan agreement (control plane)";
      abi_ok(avr_frame_plan_agree(plan, r->comm, r->compose, r->settings_digest()));
      plan->agreed_epoch = r->settings_epoch;
    }

    // Round 1's march (8 workgroups per CU) gained from being capped at 5 beside the classify
    // pass; the present one is admitted 6 per CU by its register budget and runs best uncapped
    // (config-4 frame: uncapped 1.06-1.07 ms, cap 5 1.09-1.11 ms).
    const int cap = (r->march_cap < 0) ? 0 : r->march_cap;

    // Send buffers alternate; the classified volumes rotate through three, so that the classify
    // stream may run a whole frame ahead of the march: with two, classify(f+1) and march(f) both
    // had to wait for the later of classify(f) and march(f-1) and started in lockstep, a launch
    // latency apart from the kernels before them, every frame.
    const int volume = static_cast<int>(r->frame % static_cast<unsigned>(AVR_CLASSIFIED_SLOTS));
    const int slot = volume;  // send buffers rotate with the classified volumes
    auto drain = [&] { r->drain_all(); };

    // ---- everything that can fail for lack of memory happens BEFORE anything is queued and
    // before the frame counter moves: a frame either is not started at all (the renderer stays
    // usable) or has every buffer and event it needs.  (A failure later -- a launch, a collective
    // -- leaves a frame half queued: the caller synchronises and tears the renderer down, on
    // every rank; the peers of a failed rank are otherwise left waiting in the exchange.)
    // 8-bit conversion is per pixel, so without antialiasing it is done on each rank's piece
    // before the gather (3 bytes per pixel on the wire instead of 20); the wireframe of the tight
    // bounds is per pixel too, so each rank overlays its own piece
    r->stage = "frame buffers";
    const bool early_rgb8 = root == 1;
    const bool overlay_piece = early_rgb8 && render->draw_bounds;
    const bool gather_image = want_image != 0;  // the same on every rank: it adds a collective
    const bool bytes_only = early_rgb8 && !overlay_piece && !gather_image;
    // The gathered buffer is piece-major; with contiguous pieces that IS the image, with row
    // bands avr_assemble_rows puts the rows back (for the bytes in the same pass that turns the
    // bottom-up image into the file's top-down rows).
    const bool banded = info.piece_layout == AVR_PIECES_ROW_BANDS;
    const int64_t n_pixels = info.n_pixels;
    auto bytes_of = [](int64_t count, int each) {
      return static_cast<size_t>(std::max<int64_t>(count, 1)) * static_cast<size_t>(each);
    };
    // ---- occlusion culling (one rank): the frame in depth-ordered chunks on ONE stream, every
    // chunk's classify launch leaving out the boxes its predecessors' marches found hidden
    int cull = 0;
    if (!many && !r->cache_classification && info.n_local_runs > 0 && info.n_local_boxes >= 8) {
      cull = std::min(std::max(r->occlusion_chunks, 0), info.n_local_boxes);
    }
    uint8_t* visibility = nullptr;
    if (cull >= 2) {
      visibility = static_cast<uint8_t*>(r->visible_flags[slot].reserve(
          bytes_of(static_cast<int64_t>(cull) * info.n_local_boxes, 1), drain));
    }
    // ---- visibility speculation (one rank): this frame's buffers; what the frame does with them
    // is settled below, when its layout is known
    avr_renderer::Speculating& sp = r->spec;
    const bool spec_considered = !many && r->speculation != 0 && cull < 2 && !r->cache_classification &&
                                 info.n_local_runs > 0 && info.n_local_boxes >= 8 && samples_out == nullptr &&
                                 plan->local_order.size() == static_cast<size_t>(info.n_local_boxes);
    const size_t spec_bytes = (static_cast<size_t>(std::max(info.n_local_boxes, 1)) + 15) / 16 * 16;
    uint8_t* spec_visited = nullptr;
    uint8_t* spec_missed = nullptr;
    uint32_t* spec_count = nullptr;
    uint8_t* spec_dirty = nullptr;
    int64_t spec_workgroups = 0;
    size_t spec_block = 0;
    avr_renderer::Speculating::Observation* spec_observation = nullptr;
    if (spec_considered) {
      ++sp.frame;
      if (sp.last_sampled.size() != static_cast<size_t>(info.n_local_boxes)) {
        sp.last_sampled.assign(static_cast<size_t>(info.n_local_boxes), -1);
        sp.state = avr_renderer::Speculating::kObserving;
      }
      spec_visited = static_cast<uint8_t*>(sp.visited[slot].reserve(spec_bytes, drain));
      // (one block per slot: the missed flags, the miss counter, a byte per march workgroup)
      abi_ok(avr_march_plan_workgroups(plan, &spec_workgroups));
      spec_block = spec_bytes + 16 + (static_cast<size_t>(spec_workgroups) + 15) / 16 * 16;
      spec_missed = static_cast<uint8_t*>(sp.missed[slot].reserve(spec_block, drain));
      spec_count = reinterpret_cast<uint32_t*>(spec_missed + spec_bytes);
      spec_dirty = spec_missed + spec_bytes + 16;
      if (sp.host_miss == nullptr) {
        void* block = nullptr;
        void* mapped = nullptr;
        hip_ok(hipHostMalloc(&block, 64, hipHostMallocMapped), "hipHostMalloc(speculation miss flag)");
        std::memset(block, 0, 64);
        sp.host_miss = static_cast<uint32_t*>(block);
        hip_ok(hipHostGetDevicePointer(&mapped, block, 0), "hipHostGetDevicePointer");
        sp.host_miss_dev = static_cast<uint32_t*>(mapped);
      }
      // the observations that have arrived, oldest first: which boxes those frames' rays sampled
      for (int k = 0; k < avr_renderer::Speculating::kObservations; ++k) {
        avr_renderer::Speculating::Observation* oldest = nullptr;
        for (avr_renderer::Speculating::Observation& o : sp.observations) {
          if (o.pending && (oldest == nullptr || o.frame < oldest->frame)) oldest = &o;
        }
        if (oldest == nullptr) break;
        if (hipEventQuery(oldest->copied) != hipSuccess) {
          (void)hipGetLastError();  // hipErrorNotReady is not an error here
          break;
        }
        oldest->pending = false;
        if (sp.state == avr_renderer::Speculating::kDeciding) sp.state = avr_renderer::Speculating::kObserving;
        if (oldest->stale || oldest->order.size() != sp.last_sampled.size()) {
          oldest->stale = false;
          continue;
        }
        for (size_t position = 0; position < oldest->order.size(); ++position) {
          if (oldest->host[position] != 0) {
            int64_t& last = sp.last_sampled[static_cast<size_t>(oldest->order[position])];
            last = std::max(last, oldest->frame);
          }
        }
        sp.sampled_fraction = -2.0f;  // (to be counted below)
      }
      // a free observation slot for this frame (none: the host is far ahead, this frame is not observed)
      for (avr_renderer::Speculating::Observation& o : sp.observations) {
        if (o.pending) continue;
        if (o.capacity < spec_bytes) {
          if (o.host != nullptr) (void)hipHostFree(o.host);
          o.host = nullptr;
          o.capacity = 0;
          void* block = nullptr;
          void* mapped = nullptr;
          hip_ok(hipHostMalloc(&block, spec_bytes * 2, hipHostMallocMapped), "hipHostMalloc(speculation flags)");
          o.host = static_cast<uint8_t*>(block);
          hip_ok(hipHostGetDevicePointer(&mapped, block, 0), "hipHostGetDevicePointer");
          o.host_dev = static_cast<uint8_t*>(mapped);
          o.capacity = spec_bytes * 2;
        }
        if (o.copied == nullptr) {
          // (WITH the system fence, unlike the ordering events: the host reads what the copy wrote)
          hip_ok(hipEventCreateWithFlags(&o.copied, hipEventDisableTiming), "hipEventCreate");
        }
        spec_observation = &o;
        break;
      }
    }
    float* send = static_cast<float*>(r->send[slot].reserve(bytes_of(info.send_floats, 4), drain));
    float* recv = many ? static_cast<float*>(r->recv.reserve(bytes_of(info.recv_floats, 4), drain))
                       : nullptr;
    float* piece = bytes_only ? nullptr
                              : static_cast<float*>(r->piece.reserve(bytes_of(piece_pixels, 20), drain));
    // (ranks of several: two RGB8 pieces alternate, see PendingGather)
    DeviceBuffer& rgb8_buffer = (many && (r->frame & 1u)) ? r->piece_rgb8_odd : r->piece_rgb8;
    uint8_t* piece_rgb8 =
        early_rgb8 ? static_cast<uint8_t*>(rgb8_buffer.reserve(bytes_of(piece_pixels, 3), drain))
                   : nullptr;
    const bool defer_gather = many && r->deferred_gather != 0 && early_rgb8 && !gather_image;
    uint8_t* gathered_rgb8 = nullptr;
    float* gathered_image = nullptr;
    float* assembled = nullptr;
    float* small = nullptr;
    if (is_root) {
      if (many && (early_rgb8 || r->pending.valid)) {
        int64_t pixels = early_rgb8 ? n_pixels : 0;
        if (r->pending.valid) {
          pixels = std::max<int64_t>(pixels, static_cast<int64_t>(r->pending.pieces.width) *
                                                 r->pending.pieces.height);
        }
        gathered_rgb8 = static_cast<uint8_t*>(r->full_rgb8.reserve(bytes_of(pixels, 3), drain));
      }
      if (many && ((early_rgb8 && gather_image && banded) || !early_rgb8)) {
        gathered_image = static_cast<float*>(r->full_image.reserve(bytes_of(n_pixels, 20), drain));
      }
      if (!early_rgb8 && many && banded) {
        assembled = static_cast<float*>(r->assembled_image.reserve(bytes_of(n_pixels, 20), drain));
      }
      if (!early_rgb8 && !gather_image) {
        small = static_cast<float*>(
            r->small_image.reserve(bytes_of(static_cast<int64_t>(width) * height, 20), drain));
      }
    }
    // ---- back-pressure.  The frame re-uses the classified volume and the send buffer of the
    // frame three before it.  Either the two streams wait for that frame's march and exchange /
    // fold (two wait packets between this frame's kernels and their predecessors, and with the
    // descriptor ring's events the host stays a few frames ahead), or -- for the short frames of
    // a rank of several -- the HOST waits here until they are through and nothing is queued: every
    // packet between two marches is microseconds of a rank's 0.17 ms frame (with the descriptor
    // copies of a repeating camera skipped as well, a rank of eight went from 0.187 to 0.164 ms);
    // at most three frames are then in flight.  One rank's 1 ms frames hide those packets and lose
    // 1-2 % to the shorter queue (measured), so there the streams wait.
    // avr_renderer_set_host_backpressure.  (The waits have the deadline of
    // avr_set_frame_timeout_ms: the exchange of the frame three back involves every peer.)
    const bool host_side = (r->host_backpressure < 0) ? many : (r->host_backpressure != 0);
    if (host_side) {
      if (r->marched_pending[volume]) {
        r->stage = "back-pressure: the march of the frame three before";
        avr::wait_event_deadline(r->marched_event[volume], "the march of the frame three before this one");
      }
      if (r->composed_pending[slot]) {
        r->stage = "back-pressure: the exchange and fold of the frame three before";
        avr::wait_event_deadline(r->composed_event[slot],
                                 "the exchange / fold of the frame three before this one (compositing "
                                 "stream: a peer's blocks have not arrived)");
      }
    }

    // One rank: the classify pass of the next frame runs beside the march of this one (HBM-bound
    // beside issue-bound).  A rank's share of an N-rank frame is two SHORT kernels whose time is
    // their slowest workgroups': side by side each stretched the other (N = 8, slowest rank:
    // 77 us + 151 us alone, 0.26 + 0.27 ms overlapped), so there they run back to back on the
    // march stream and only the exchange / fold / gather of the previous frame overlaps them.
    // Which of the two, and how many classify workgroups a CU admits beside the march, is measured
    // on the running pipeline (CoRunTuner) unless the caller fixed it: avr_renderer_set_overlap,
    // avr_renderer_set_classify_share.  (A cached classification leaves nothing to tune.)
    CoRunTuner& tuner = r->tuner;
    {
      // overlap_classify: -1 everything, 0 back to back, 1 side by side, 2 paired
      int first = CoRunTuner::kBackToBack, last = CoRunTuner::kLastPaired;
      if (r->cache_classification) {  // no classify pass to place
        first = last = (r->overlap_classify == 0) ? CoRunTuner::kBackToBack : 0;
      } else if (r->overlap_classify == 0) {
        last = first;
      } else if (r->overlap_classify == 2) {
        first = CoRunTuner::kPairedBase;
        if (r->share_fixed >= 0) last = first;  // the caller's reserve, kept outside the scale
      } else {
        if (r->overlap_classify > 0) {
          first = 0;
          last = CoRunTuner::kLastCandidate;
        }
        if (r->share_fixed >= 0) {
          // one side-by-side candidate: the caller's reserve (kept outside the candidate scale)
          last = (first == CoRunTuner::kBackToBack) ? 0 : first;
        }
      }
      if (cull >= 2) first = last = CoRunTuner::kBackToBack;  // (one stream: nothing to place)
      tuner.restrict_to(first, last, r->n_ranks == 1);
      // one rank with nothing fixed: the balance of the two kernels is read off their durations
      tuner.set_balance(r->n_ranks == 1 && r->balance != 0);
      // ranks of several search together (avr_corun.h): the same candidate in the same frames,
      // every window's period the maximum over the ranks
      tuner.set_coordinated(many && r->comm != nullptr && r->coordinate != 0);
    }
    if (tuner.tuning() && tuner.report_due()) {
      // The window closed kReportLag frames ago (the back-pressure above has seen its last march
      // through); all ranks are at this frame.  A rank whose window was void (its pipeline had
      // drained: a buffer grew) says so, and then everybody times the candidate again.
      r->stage = "co-run window agreement (control plane)";
      avr::wait_event_deadline(r->window_end, "the march that closes the co-run window");
      // (the message also says which frame of which plan the rank is in: ranks that were driven
      // apart -- same block sizes, another camera -- are found out here at the latest)
      struct Word {
        float period_ms;
        uint32_t frame;
        uint64_t plan_digest;
      } mine{-1.0f, r->frame, plan->agreed_digest};
      static_assert(sizeof(Word) == 16, "control word");
      if (!tuner.window_void) {
        float elapsed_ms = 0.0f;
        hip_ok(hipEventElapsedTime(&elapsed_ms, r->window_begin, r->window_end), "hipEventElapsedTime");
        mine.period_ms = elapsed_ms / static_cast<float>(tuner.window_length);
      }
      std::vector<Word> words(static_cast<size_t>(r->n_ranks));
      abi_ok(avr_comm_control_allgather(r->comm, r->compose, &mine, words.data(), sizeof(Word)));
      float agreed = 0.0f;
      bool any_void = false;
      for (size_t peer = 0; peer < words.size(); ++peer) {
        const Word& word = words[peer];
        if (word.frame != mine.frame || word.plan_digest != mine.plan_digest) {
          throw std::runtime_error("rank " + std::to_string(peer) + " is in frame " +
                                   std::to_string(word.frame) + (word.plan_digest != mine.plan_digest
                                                                     ? " of another frame plan" : "") +
                                   " while rank " + std::to_string(r->rank) + " is in frame " +
                                   std::to_string(mine.frame) + ": the ranks were not driven alike");
        }
        any_void = any_void || !(word.period_ms >= 0.0f);
        agreed = std::max(agreed, word.period_ms);
      }
      if (any_void) {
        tuner.retime();
      } else {
        tuner.report(agreed);
      }
    }
    const bool overlap = tuner.candidate != CoRunTuner::kBackToBack;
    const bool paired = CoRunTuner::is_paired(tuner.candidate);
    const int reserve = !overlap ? 0
                        : (r->share_fixed >= 0)
                            ? r->share_fixed
                            : CoRunTuner::reserve_index(tuner.candidate) * CoRunTuner::kReserveStep;
    // (One stream per kernel kind.  Letting the odd frames take a second march or classify stream
    // -- the frames are independent, so march(f+1) need not queue behind march(f) and its wait /
    // record / copy packets could be worked off early -- was measured in round 3 and is worse by
    // half: with a FOURTH concurrently active queue everything stalls, a 5 us descriptor copy
    // takes 40-50 us, the frame of a rank of eight goes from 0.187 to 0.26-0.33 ms and the one-rank
    // frame from 0.98 to 1.25-1.42 ms.  profiles/r3_experiments/.)
    // Paired layout (avr_corun.h): this frame's classify pass and march back to back on ONE
    // stream, the even frames on stream M, the odd ones on stream B -- with stream X three active
    // queues, the number this GPU runs side by side without penalty.
    avr_context* march_ctx = r->march;
    if (paired && (r->frame & 1u)) {
      if (r->pair_b == nullptr) abi_ok(avr_context_create_with_priority(r->device, 1, &r->pair_b));
      march_ctx = r->pair_b;
    }
    avr_context* classify_ctx = paired ? march_ctx : overlap ? r->classify : r->march;
    abi_ok(avr_context_set_march_occupancy(march_ctx, cap));
    hipStream_t stream_c = r->stream_of(classify_ctx);
    hipStream_t stream_m = r->stream_of(march_ctx);
    hipStream_t stream_x = r->stream_of(r->compose);
    for (avr_context* ctx : {r->classify, r->march, r->compose, r->pair_b}) {
      if (ctx != nullptr) avr::context_set_lean_descriptors(ctx, host_side);
    }
    if (r->history.size() < r->history_capacity) {
      r->history.push_back(static_cast<int16_t>(tuner.candidate));
    }
    struct TimedGuard {  // the frame's four timing events, destroyed unless the frame keeps them
      FrameEvents events;
      bool kept = false;
      ~TimedGuard() {
        if (kept) return;
        for (hipEvent_t ev : {events.classify_begin, events.classify_end, events.march_begin,
                              events.march_end}) {
          if (ev != nullptr) (void)hipEventDestroy(ev);
        }
      }
    } timed_guard;
    FrameEvents& timed = timed_guard.events;
    // kBalance: this frame's kernels are timed through a slot of the probe ring (if one is free)
    avr_renderer::Probe* probe = nullptr;
    if (tuner.balancing() && overlap && !paired && !r->pipeline_idle &&
        r->probe_head - r->probe_tail < avr_renderer::kProbes) {
      probe = &r->probes[r->probe_head % avr_renderer::kProbes];
      for (hipEvent_t* ev : {&probe->events.classify_begin, &probe->events.classify_end,
                             &probe->events.march_begin, &probe->events.march_end}) {
        if (*ev == nullptr) *ev = make_event(true);
      }
      probe->candidate = tuner.candidate;
    }
    if (r->timing) {
      timed.classify_begin = make_event(true);
      timed.classify_end = make_event(true);
      timed.march_begin = make_event(true);
      timed.march_end = make_event(true);
      r->timed.reserve(r->timed.size() + 1);
    }
    ++r->frame;

    // ---- one launch per kernel, or depth-ordered chunks (an idle pipeline: see frame_chunks) ----
    // Chunks need the two kernels on two streams (side by side); a cached classification has no
    // classify pass to cut.
    const bool was_idle = r->pipeline_idle;
    int n_chunks = 1;
    if (overlap && !paired && !r->cache_classification && info.n_local_runs > 0) {
      n_chunks = std::min(std::max(r->frame_chunks, 1), std::max(info.n_local_boxes, 1));
    }
    r->last_chunks = n_chunks;
    // ---- visibility speculation: what this frame does (0 nothing, 1 a plain frame whose march
    // records the boxes it samples, 2 classifies only the set, checks, repairs -- and records)
    int spec_mode = 0;
    {
      using S = avr_renderer::Speculating;
      // (the classify pass changes its length: the co-run balance is found again -- and the frames
      // still in flight, timed under the old length, must not be read as its first steps: they sent
      // the bisection the wrong way, 26 KiB held instead of 43, 0.45 ms instead of 0.42)
      auto restart_corun_search = [&] {
        tuner.restart();
        r->probe_tail = r->probe_head;
      };
      if (spec_considered && n_chunks == 1) {  // (any layout: one stream or two, the protocol is the same)
        const bool missed_lately = *static_cast<volatile uint32_t*>(sp.host_miss) != 0;
        if (missed_lately) {  // a march of an earlier frame missed (its repair redid that frame)
          *static_cast<volatile uint32_t*>(sp.host_miss) = 0;
          ++sp.repaired_frames;
          ++sp.recent_repairs;
          sp.last_repair = sp.frame;
        }
        if (sp.state == S::kActive && ++sp.recent_frames >= 32) {
          // repairs in more than half of the frames: the cells change what is visible faster than
          // the observations follow (a repair redoes the tiles that met an unclassified box)
          if (sp.recent_repairs * 2 > sp.recent_frames) {
            sp.state = S::kBackoff;
            sp.asleep_until = sp.frame + sp.next_backoff;
            sp.next_backoff = std::min(sp.next_backoff * 2, 4096);
            restart_corun_search();  // (the classify pass is the whole pass again)
          } else if (sp.recent_repairs == 0) {
            sp.next_backoff = 64;
          }
          sp.recent_repairs = sp.recent_frames = 0;
        }
        if ((sp.state == S::kRejected || sp.state == S::kBackoff) && sp.frame >= sp.asleep_until) {
          sp.state = S::kObserving;
          std::fill(sp.last_sampled.begin(), sp.last_sampled.end(), int64_t{-1});  // (look afresh)
        }
        // this frame's set: the boxes sampled within the last kSpecMemory frames, in its layer order
        if (sp.state == S::kObserving || sp.state == S::kActive) {
          sp.positions.clear();
          sp.flags.assign(spec_bytes, 0);
          bool any_observation = false;
          for (int position = 0; position < info.n_local_boxes; ++position) {
            const int64_t last = sp.last_sampled[static_cast<size_t>(plan->local_order[static_cast<size_t>(position)])];
            any_observation = any_observation || last >= 0;
            if (last >= 0 && last + avr_renderer::kSpecMemory >= sp.frame) {
              sp.positions.push_back(position);
              sp.flags[static_cast<size_t>(position)] = 1;
            }
          }
          if (any_observation) {
            sp.sampled_fraction = static_cast<float>(sp.positions.size()) /
                                  static_cast<float>(std::max(info.n_local_boxes, 1));
            // Worth it when the part of the classify pass it removes outweighs what it adds (two
            // gated launches and two memsets on the march's stream, ~20 us, and a march that holds a
            // wave per SIMD less): the rank's cells at ~5 TB/s, the unsampled share of that -- at
            // least kSpecMinSavingMs.  (config-4 opaque 0.39 ms saved: frame 0.63 -> 0.43; config-3
            // opaque 0.14, config-2 0.06: 1-2 % SLOWER when tried, their frames are march-bound.)
            double cell_bytes = 0.0;
            for (size_t b = 0; b < r->all_boxes.size(); ++b) {
              if (r->owner[b] != r->rank) continue;
              const avr_box& box = r->all_boxes[b];
              cell_bytes += 8.0 * box.dims[0] * box.dims[1] * box.dims[2];
            }
            const double saving_ms = (1.0 - sp.sampled_fraction) * cell_bytes / 5.0e9;
            const bool worth_it = sp.sampled_fraction <= r->spec_worth_it && !sp.positions.empty() &&
                                  saving_ms >= r->spec_min_saving_ms;
            if (sp.state == S::kObserving && worth_it) {
              sp.state = S::kActive;
              sp.recent_repairs = sp.recent_frames = 0;
              restart_corun_search();  // (a classify pass of a fraction of the boxes: another balance)
            } else if (sp.state == S::kObserving) {
              sp.state = S::kRejected;
              sp.asleep_until = sp.frame + avr_renderer::kSpecProbeEvery;
            } else if (!worth_it) {  // (kActive: the rays reach nearly everything now)
              sp.state = S::kRejected;
              sp.asleep_until = sp.frame + avr_renderer::kSpecProbeEvery;
              restart_corun_search();
            }
          }
        }
        if (sp.state == S::kActive) {
          spec_mode = 2;
        } else if (sp.state == S::kObserving && spec_observation != nullptr) {
          spec_mode = 1;
          sp.state = S::kDeciding;  // (until this observation has arrived)
        }
      }
    }
    void* chunk_events[AVR_MAX_FRAME_CHUNKS] = {};
    for (int k = 0; k < n_chunks && n_chunks > 1; ++k) {
      hipEvent_t& event = r->chunk_event[volume][k];
      if (event == nullptr) event = make_event(false);
      chunk_events[k] = event;
    }
    // (the first frame after a drain classifies alone -- its first chunk, if it is cut: no march
    // to leave room for)
    abi_ok(avr_context_set_classify_lds_reserve(
        classify_ctx, (overlap && (!was_idle || n_chunks > 1)) ? reserve : 0));
    // Side by side the march that reads this frame's bricklets starts a frame later: they are
    // streamed to memory.  Back to back and paired it follows at once: they are stored plainly
    // (a rank of eight 0.143 against 0.149 ms).
    {
      static const char* forced = std::getenv("AVR_CLASSIFY_STREAM");  // A/B only
      // (a chunk's bricklets are marched right away, and so are those of a frame that found the
      // pipeline empty: stored plainly, like back to back)
      const bool stream = forced != nullptr ? std::atoi(forced) != 0
                                            : (overlap && !paired && n_chunks == 1 && !was_idle);
      avr::context_set_classify_stream_stores(classify_ctx, stream);
    }
    r->last_overlap = overlap;
    r->last_paired = paired;
    r->last_reserve = reserve;
    r->pipeline_idle = false;

    lap(0);
    r->stage = "classify";
    // ---- stream C: classify pass of this frame into classified volume `slot` -------------------
    if (input_stream != nullptr) {  // the caller's cell data is produced on that stream
      hipStream_t producer = (input_stream == AVR_DEFAULT_STREAM)
                                 ? nullptr  // recording on stream 0 IS recording on the null stream
                                 : static_cast<hipStream_t>(input_stream);
      hip_ok(hipEventRecord(r->input_event, producer), "hipEventRecord");
      hip_ok(hipStreamWaitEvent(stream_c, r->input_event, 0), "hipStreamWaitEvent");
    }
    // (With host-side back-pressure the re-use of the frame's classified volume and send buffer
    // was settled before anything was queued; otherwise the streams wait -- unless the event has
    // already happened.)
    auto wait_unless_done = [&](hipStream_t stream, hipEvent_t event) {
      if (hipEventQuery(event) == hipSuccess) return;
      (void)hipGetLastError();  // hipErrorNotReady is not an error here
      hip_ok(hipStreamWaitEvent(stream, event, 0), "hipStreamWaitEvent");
    };
    if (!host_side) {
      if (r->marched_pending[volume]) wait_unless_done(stream_c, r->marched_event[volume]);
      if (r->composed_pending[slot]) wait_unless_done(stream_m, r->composed_event[slot]);
    }
    if (paired && r->paired_previous != nullptr) {  // one classify pass at a time
      if (hipEventQuery(r->paired_previous) != hipSuccess) {
        (void)hipGetLastError();
        hip_ok(hipStreamWaitEvent(stream_c, r->paired_previous, 0), "hipStreamWaitEvent");
      }
    }
    if (r->timing) hip_ok(hipEventRecord(timed.classify_begin, stream_c), "hipEventRecord");
    if (probe != nullptr) hip_ok(hipEventRecord(probe->events.classify_begin, stream_c), "hipEventRecord");
    if (cull >= 2) {
      // (classified chunk by chunk between the march launches, below)
    } else if (n_chunks > 1) {
      abi_ok(avr_classify_plan_chunked(classify_ctx, r->scene, plan, volume, n_chunks, chunk_events,
                                       was_idle ? 1 : 0));
    } else if (spec_mode == 2) {
      // only the boxes of the held set: a launch of exactly their tiles
      abi_ok(avr_classify_plan_positions(classify_ctx, r->scene, plan, volume, sp.positions.data(),
                                         static_cast<int>(sp.positions.size())));
    } else {
      abi_ok(avr_classify_plan(classify_ctx, r->scene, plan, volume));
    }
    if (probe != nullptr) hip_ok(hipEventRecord(probe->events.classify_end, stream_c), "hipEventRecord");
    hipEvent_t classified = r->timing ? timed.classify_end : r->classified_event[volume];
    if (overlap || r->timing) hip_ok(hipEventRecord(classified, stream_c), "hipEventRecord");
    // (what the NEXT frame waits on must outlive this frame's timing events, which
    // avr_renderer_set_timing destroys: always the volume's own ordering event)
    if (paired && r->timing) {
      hip_ok(hipEventRecord(r->classified_event[volume], stream_c), "hipEventRecord");
    }
    r->paired_previous = paired ? r->classified_event[volume] : nullptr;

    lap(1);
    r->stage = "march";
    // ---- stream M: march into send buffer `slot` ------------------------------------------------
    // (three batches of descriptors per speculating frame on the march's context: the whole ring)
    avr::context_set_descriptor_lead(march_ctx, spec_mode == 2 ? 9 : 4);
    // (a speculating frame is observed -- a memset, a copy kernel and an event more on the march's
    // stream -- every time while the camera moves or a repair was needed lately: what comes into
    // view is then in the set two or three frames later; every kSpecObserveEvery-th time while the
    // plan stands.  Sparser for a moving camera was tried: a fly-through gains 7 %, sixteen cameras
    // in turn lose 4 % -- the ones that fall between the observations are repaired on every visit.)
    const bool spec_observed =
        spec_mode != 0 && spec_observation != nullptr &&
        (spec_mode == 1 || plan != sp.previous_plan || sp.frame % avr_renderer::kSpecObserveEvery == 0 ||
         sp.frame - sp.last_repair < 2 * avr_renderer::kSpecObserveEvery);
    sp.previous_plan = plan;
    if (spec_observed) {  // (cleared while the classify pass still runs)
      hip_ok(hipMemsetAsync(spec_visited, 0, spec_bytes, stream_m), "hipMemsetAsync(speculation)");
    }
    if (spec_mode == 2) {
      hip_ok(hipMemsetAsync(spec_missed, 0, spec_block, stream_m), "hipMemsetAsync(speculation)");
    }
    if (overlap && !paired && n_chunks == 1) {
      // (paired: the march follows its classify pass on the same stream; chunked: every march
      // launch waits for its own chunk's event)
      hip_ok(hipStreamWaitEvent(stream_m, classified, 0), "hipStreamWaitEvent");
    }
    if (r->timing) hip_ok(hipEventRecord(timed.march_begin, stream_m), "hipEventRecord");
    if (probe != nullptr) hip_ok(hipEventRecord(probe->events.march_begin, stream_m), "hipEventRecord");
    if (cull >= 2) {
      abi_ok(avr_render_plan_culled(march_ctx, r->scene, plan, volume, send, samples_out, cull, visibility));
      r->last_chunks = cull;
    } else if (n_chunks > 1) {
      abi_ok(avr_march_plan_chunked(march_ctx, r->scene, plan, volume, send, samples_out, n_chunks,
                                    chunk_events));
    } else if (spec_mode != 0) {
      avr_speculation first{};
      first.visited = spec_observed ? spec_visited : nullptr;
      if (spec_mode == 2) {
        first.classified_host = sp.flags.data();
        first.missed = spec_missed;
        first.miss_count = spec_count;
        first.host_miss_flag = sp.host_miss_dev;
        first.dirty_workgroups = spec_dirty;
      }
      abi_ok(avr_march_plan_speculative(march_ctx, r->scene, plan, volume, send, &first));
      if (spec_mode == 2) {
        // the repair, queued unconditionally: both launches do nothing unless the march missed
        abi_ok(avr_classify_plan_flagged(march_ctx, r->scene, plan, volume, spec_missed, spec_count));
        avr_speculation again{};
        again.visited = first.visited;
        again.gate = spec_count;
        again.dirty_workgroups = spec_dirty;  // (only the workgroups that met an unclassified box)
        abi_ok(avr_march_plan_speculative(march_ctx, r->scene, plan, volume, send, &again));
        ++sp.active_frames;
      }
      if (spec_observed) {
        // the boxes this frame's rays sampled go to the host (a copy kernel into pinned memory) and
        // are read a few frames on, by box
        abi_ok(avr::launch_upload(spec_visited, spec_observation->host_dev, spec_bytes, stream_m));
        hip_ok(hipEventRecord(spec_observation->copied, stream_m), "hipEventRecord");
        spec_observation->order = plan->local_order;
        spec_observation->frame = sp.frame;
        spec_observation->pending = true;
      }
    } else {
      abi_ok(avr_march_plan(march_ctx, r->scene, plan, volume, send, samples_out));
    }
    if (probe != nullptr) {
      hip_ok(hipEventRecord(probe->events.march_end, stream_m), "hipEventRecord");
      ++r->probe_head;
    }
    // kBalance: the durations of the frames that are through by now, in frame order
    while (r->probe_tail != r->probe_head) {
      avr_renderer::Probe& done = r->probes[r->probe_tail % avr_renderer::kProbes];
      if (hipEventQuery(done.events.march_end) != hipSuccess ||
          hipEventQuery(done.events.classify_end) != hipSuccess) {
        (void)hipGetLastError();  // hipErrorNotReady is not an error here
        break;
      }
      float classify_ms = 0.0f, march_ms = 0.0f;
      hip_ok(hipEventElapsedTime(&classify_ms, done.events.classify_begin, done.events.classify_end),
             "hipEventElapsedTime");
      hip_ok(hipEventElapsedTime(&march_ms, done.events.march_begin, done.events.march_end),
             "hipEventElapsedTime");
      ++r->probe_tail;
      tuner.report_durations(done.candidate, classify_ms, march_ms);
    }
    // the tuner's window: the period of a few frames between two events after the march
    if (tuner.tuning()) {
      if (tuner.closing) {
        if (tuner.coordinated) {
          // (agreed on by all ranks kReportLag frames after the window, above)
        } else if (hipEventQuery(r->window_end) == hipSuccess) {
          float elapsed_ms = 0.0f;
          hip_ok(hipEventElapsedTime(&elapsed_ms, r->window_begin, r->window_end),
                 "hipEventElapsedTime");
          tuner.report(elapsed_ms / static_cast<float>(tuner.window_length));
        } else {
          (void)hipGetLastError();  // hipErrorNotReady is not an error here
        }
      } else {
        switch (tuner.frame()) {
          case CoRunTuner::kOpenWindow:
            if (r->window_begin == nullptr) {
              r->window_begin = make_event(true);
              r->window_end = make_event(true);
            }
            hip_ok(hipEventRecord(r->window_begin, stream_m), "hipEventRecord");
            break;
          case CoRunTuner::kCloseWindow:
            hip_ok(hipEventRecord(r->window_end, stream_m), "hipEventRecord");
            break;
          case CoRunTuner::kNothing:
            break;
        }
      }
    }
    if (r->timing) hip_ok(hipEventRecord(timed.march_end, stream_m), "hipEventRecord");
    hip_ok(hipEventRecord(r->marched_event[volume], stream_m), "hipEventRecord");
    r->marched_pending[volume] = true;
    if (r->timing) {
      r->timed.push_back(timed);
      timed_guard.kept = true;
    }

    lap(2);
    r->stage = "exchange";
    // ---- stream X: exchange, fold, gather, frame tail ------------------------------------------
    hip_ok(hipStreamWaitEvent(stream_x, r->marched_event[volume], 0), "hipStreamWaitEvent");
    // (the rank's block for itself is not copied into the receive buffer: the fold reads it
    // where the march stored it -- the send buffer lives until composed_event -- which is one
    // kernel and one gap less on this stream, the busiest of a rank of eight's frame)
    const float* received = send;
    const float* own = nullptr;
    if (many) {
      // ... and in the same grouped round the RGB8 pieces of the frame before travel to the root
      avr_renderer::PendingGather& pending = r->pending;
      avr_gather_op rider{};
      if (pending.valid) {
        rider.piece = pending.piece;
        rider.bytes_per_pixel = 3;
        rider.root = 0;
        rider.full = gathered_rgb8;
        rider.begin = pending.begin.data();
        rider.end = pending.end.data();
        // (the assemble pass reads the root's own piece where its fold wrote it -- unless the
        // reference's contiguous pieces cut through rows)
        rider.skip_own = pending.own_in_place ? 1 : 0;
      }
      abi_ok(avr_exchange_peers_gather(r->compose, plan, r->comm, send, recv,
                                       pending.valid ? &rider : nullptr));
      if (pending.valid && is_root) {
        if (avr::launch_assemble_rows(pending.pieces, gathered_rgb8,
                                      static_cast<int64_t>(pending.pieces.width) * 3, /*flip=*/1,
                                      pending.out, stream_x,
                                      pending.own_in_place ? pending.piece : nullptr,
                                      pending.own_piece) != AVR_OK) {
          throw std::runtime_error(avr_last_error());
        }
      }
      pending.valid = false;
      received = recv;
      own = send;
    }
    lap(3);
    r->stage = "fold";
    // (one rank without antialiasing or wireframe: the fold writes the output file's rows itself)
    const bool fold_to_image = !many && early_rgb8 && !overlay_piece && is_root;
    avr::context_set_fold_whole_grid(r->compose, was_idle);
    if (fold_to_image) {
      abi_ok(avr_fold_plan_image(r->compose, plan, received, piece, rgb8_out));
    } else {
      abi_ok(avr_fold_plan_own(r->compose, plan, received, own, piece,
                               overlay_piece ? nullptr : piece_rgb8));
    }
    if (overlay_piece && piece_pixels > 0) {
      abi_ok(avr_bbox_overlay_piece(r->compose, plan, r->tight_min, r->tight_max, camera, piece,
                                    piece_rgb8));
    }
    hip_ok(hipEventRecord(r->composed_event[slot], stream_x), "hipEventRecord");
    r->composed_pending[slot] = true;

    lap(4);
    r->stage = "gather and frame tail";
    if (defer_gather) {
      // the bytes travel with the next frame's round (or with avr_renderer_synchronize)
      avr_renderer::PendingGather& pending = r->pending;
      pending.valid = true;
      pending.pieces = plan->pieces;
      pending.begin.resize(static_cast<size_t>(r->n_ranks));
      pending.end.resize(static_cast<size_t>(r->n_ranks));
      abi_ok(avr_frame_plan_piece_ranges(plan, pending.begin.data(), pending.end.data()));
      pending.own_piece = plan->piece_of_rank[static_cast<size_t>(r->rank)];
      pending.own_in_place = banded || (plan->pieces.width > 0 &&
                                        plan->pieces.piece_size % plan->pieces.width == 0);
      pending.piece = piece_rgb8;
      pending.out = rgb8_out;
    } else if (early_rgb8) {
      uint8_t* full = piece_rgb8;
      if (many) {
        full = gathered_rgb8;
        abi_ok(avr_gather(r->compose, plan, r->comm, piece_rgb8, 3, full, 0));
      }
      if (is_root && !fold_to_image) abi_ok(avr_assemble_rows(r->compose, plan, full, 3, 1, rgb8_out));
      if (gather_image) {
        if (many) {
          float* gathered = is_root ? (banded ? gathered_image : image_out) : nullptr;
          abi_ok(avr_gather(r->compose, plan, r->comm, piece, 20, gathered, 0));
          if (is_root && banded) abi_ok(avr_assemble_rows(r->compose, plan, gathered, 20, 0, image_out));
        } else {
          hip_ok(hipMemcpyAsync(image_out, piece, static_cast<size_t>(n_pixels) * 20,
                                hipMemcpyDeviceToDevice, stream_x), "hipMemcpyAsync(image)");
        }
      }
    } else {
      float* full = piece;
      if (many) {
        full = gathered_image;
        abi_ok(avr_gather(r->compose, plan, r->comm, piece, 20, full, 0));
        if (is_root && banded) {
          abi_ok(avr_assemble_rows(r->compose, plan, full, 20, 0, assembled));
          full = assembled;
        }
      }
      if (is_root) {
        float* target = gather_image ? image_out : small;
        abi_ok(avr_downsample_depthsort(r->compose, full, width, height, root, target));
        if (render->draw_bounds) {
          abi_ok(avr_bbox_overlay(r->compose, r->tight_min, r->tight_max, camera, 1, width, height, 0,
                                  static_cast<int64_t>(width) * height, target, nullptr));
        }
        abi_ok(avr_quantize_rgb8(r->compose, target, width, height, 5, rgb8_out));
      }
    }
    lap(5);
    r->stage = "queued";
    ++r->host_frames;
    r->pipeline_idle = false;  // (a buffer that grew drained the streams in between)
    return AVR_OK;
  });
}

}  // extern "C"
