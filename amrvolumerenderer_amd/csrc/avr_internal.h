// Internal types shared by the host prologue (avr_host.cpp), the kernels (avr_kernels.hip)
// and the C ABI (avr_capi.cpp).  Not part of the public boundary (include/avr_hip.h).
#ifndef AVR_INTERNAL_H
#define AVR_INTERNAL_H

#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/avr_hip.h"

namespace avr {

constexpr int kTableSize = 256;  // kColorTableSize, Common/VolumePainter.cpp:35
constexpr int kMaxLdsTables = 16;  // transfer-function tables staged in LDS per launch (4 KiB each)

// Cell index = floor((pos - min) / dx) with an IEEE divide in the reference
// (Common/VolumePainter.cpp:846-852).  The same integer is obtained cheaper when provable:
enum IndexMode : int32_t {
  kPow2Multiply = 0,  // dx is a power of two: (pos - min) * (1/dx) IS the correctly rounded quotient,
                      // and so is fma(pos, 1/dx, -min/dx): scaling by 2^k commutes with rounding
                      // (the host admits the mode only for magnitudes where nothing is subnormal)
  kReciprocal = 1,    // q = (pos - min) * RN(1/dx) is within 2^-22 * q of the rounded quotient, so
                      // floor(q) is the reference's index unless q is within near_tol of an
                      // integer; those samples take the exact divide
  kExactDivide = 2,   // degenerate spacing (dx <= 0 or not finite): always the exact divide
};

// Per-box constants of VolumePainter::paint's host prologue (Common/VolumePainter.cpp:571-692),
// laid out for wave-uniform scalar loads (one 128-byte record per box).
struct alignas(16) BoxDev {
  float minc[3];
  float maxc[3];
  float dx, dy, dz;       // (maxCornerF - minCornerF) / float(n)   (:678-686)
  float mesh_eps;         // 1e-4 * |diag|                          (:688-692)
  float sample_dist;      // max(0.5 * minSpacing, 1e-5)            (:600)
  int32_t nx, ny, nz;
  int32_t lut;            // index of this box's transfer-function table
  int32_t rect[4];        // conservative screen rectangle x0,y0,x1,y1 (inclusive); x1 < x0 = off-screen
  int32_t index_mode;     // how (pos - min) / dx is evaluated, see IndexMode
  const double* cells;    // values(validBox.smallEnd(), component)
  int32_t jstride;        // element strides (Array4); the box spans < 2^28 elements
  int32_t kstride;
  float inv_dx, inv_dy, inv_dz;  // RN(1 / dx)
  float near_tol;         // kReciprocal: |q - rint(q)| <= near_tol sends the sample to the exact divide
  int32_t pad_;
  uint64_t cls_offset;    // byte offset of this box's classified bricklets in the frame's buffer
  float nmin_inv[3];      // kPow2Multiply: -(minc * inv_d), exact (a power-of-two scaling)
  int32_t pad2_;
};
static_assert(sizeof(BoxDev) == 144, "BoxDev is read with scalar loads: keep it a multiple of 16 bytes");

// A speculative frame's bookkeeping on the device (render_runs_kernel; all arrays by position in
// the global layer order).
struct MarchSpecDev {
  const uint8_t* classified;  // != 0: this frame's classify pass covered the box; nullptr: every box
  uint8_t* visited;           // out: set for every box some wave marched; nullptr: not recorded
  uint8_t* missed;            // out: boxes a ray needed that `classified` left out
  uint32_t* miss_count;       // out: one per (wave, box) miss
  uint32_t* host_miss_flag;   // out, host-mapped or nullptr: set to 1 on a miss
  const uint32_t* gate;       // nullptr, or: the launch does nothing unless *gate != 0
  uint8_t* dirty_blocks_out;  // out, or nullptr: per workgroup of the grid, 1 = it met an unclassified box
  const uint8_t* dirty_blocks;  // gated launch: nullptr, or only these workgroups run again
};

// Frame constants (camera basis, scalar mapping); passed to kernels by value (kernarg -> SGPRs).
struct FrameConsts {
  int32_t width, height;
  float aspect, tan_half_fov, inv_width, inv_height;
  float fwd[3], right[3], up[3], eye[3];
  float range_min, inverse_range, clip_start;
  int32_t apply_clip;
  int32_t log_scale, normalize;
  double positive_floor, norm_min, inv_norm_span;
};

// Classified volume: every cell's transfer-function table index (uint8) in 128-byte bricklets
// of 8 x 4 x 4 cells (x fastest inside a bricklet; bricklets z-fastest, then y, then x inside the
// box: see bricklet_offset in avr_kernels.hip).
constexpr int kBrickX = 8, kBrickY = 4, kBrickZ = 4, kBrickBytes = 128;
constexpr int kClassifyChunk = 128;  // cells of one x-row handled by one classify workgroup

#if defined(__HIP__)
#define AVR_HD __host__ __device__
#else
#define AVR_HD
#endif

// How the image's pixels are dealt to the N pieces of the direct-send exchange (one piece per
// rank of the compositing group).
//   kPiecesContiguous  the reference's partition: piece k = pixels [k * floor(P/N), ...), the
//                      last piece takes the remainder (getPieceRange, DirectSendBase.cpp:59-74).
//                      What Compositor::compose promises its caller, so the plugin keeps it.
//   kPiecesRowBands    bands of `band_rows` image rows dealt round-robin: band b belongs to piece
//                      b % N.  The scene covers a compact part of the screen, so contiguous
//                      pieces leave some ranks with nothing to receive and fold and others with
//                      everything (config-4, N = 8: 36.5 MB against 0); dealt bands give every
//                      rank the same share of every region.  Per-pixel results do not depend on
//                      which rank folds a pixel, so the gathered image is the same bit for bit;
//                      only the frame driver (whose caller sees the gathered image, never the
//                      pieces) uses it.
// A piece's rows are numbered 0, 1, ... in image order ("piece rows"); with contiguous pieces a
// piece row IS the image row (blocks count rows from the image's row 0 there).
enum PieceLayout : int32_t { kPiecesContiguous = 0, kPiecesRowBands = 1 };
struct PieceMapDev {
  int32_t layout = kPiecesContiguous;
  int32_t band_rows = 1;
  int32_t n_pieces = 1;
  int32_t width = 0, height = 0;
  int32_t pad_ = 0;
  int64_t piece_size = 0;  // kPiecesContiguous: floor(P / N)
};
AVR_HD inline int piece_of_row(const PieceMapDev& m, int y) { return (y / m.band_rows) % m.n_pieces; }
// position of image row y among the rows of its piece (kPiecesRowBands)
AVR_HD inline int piece_row_of(const PieceMapDev& m, int y) {
  return (y / (m.band_rows * m.n_pieces)) * m.band_rows + y % m.band_rows;
}
// image row of piece row j of piece k (kPiecesRowBands)
AVR_HD inline int image_row_of(const PieceMapDev& m, int k, int j) {
  return ((j / m.band_rows) * m.n_pieces + k) * m.band_rows + j % m.band_rows;
}
// number of rows of piece k (kPiecesRowBands)
AVR_HD inline int piece_row_count(const PieceMapDev& m, int k) {
  const int cycle = m.band_rows * m.n_pieces;
  const int rest = m.height % cycle - k * m.band_rows;  // rows of the last, partial cycle left for k
  return (m.height / cycle) * m.band_rows + (rest <= 0 ? 0 : (rest < m.band_rows ? rest : m.band_rows));
}

// Sparse run layers: a run's layer is stored only inside the run's screen rectangle, cut into
// one block per DirectSend piece (include/avr_hip.h, "frame plan").
struct RunRectDev {
  int32_t x0, y0, x1, y1;  // inclusive; x1 < x0 = empty
};
struct RunBlockDev {
  int64_t offset;     // float offset of the block in the send / recv buffer (unused if empty)
  int32_t first_row;  // piece row (contiguous pieces: image row) stored in the block's first row
  int32_t span_base;  // < 0: every row holds the run rectangle's x0..x1; else the block's rows are
                      // entries span_base + (row - first_row) of the span table
};
// One row of a block of a tightened frame plan (avr_frame_plan_tighten): only the pixels
// x0..x1 of the row are stored, starting `rel` floats after the block's offset; the others are
// known to be empty (conservative per-row extent of the run's boxes on screen).  8 bytes per row:
// the two span tables of a frame travel with its descriptors (hundreds of KB at 16).
struct RunSpanDev {
  uint16_t x0, x1;  // inclusive; x1 < x0 = nothing stored for this row
  uint32_t rel;
};
constexpr int kMaxTightenedWidth = 65535;            // x0 / x1 are 16 bit
constexpr int64_t kMaxTightenedBlockFloats = 0xffffffffLL;  // rel is 32 bit

// One unit of march work: a screen super-tile of one run.  Runs are independent (each has its
// own layer, started from the empty pixel), so a rank's runs are marched by different
// workgroups; a frame's critical path is then one run's rays, not all of a rank's boxes.
struct MarchItemDev {
  uint32_t slot;  // Morton index of the super-tile; kNoMarchItem = padding
  uint32_t run;
};
constexpr uint32_t kNoMarchItem = 0xffffffffu;

// Host results for one frame over a list of boxes.
struct FramePlan {
  FrameConsts consts;
  std::vector<BoxDev> boxes;        // same order as the input boxes
  std::vector<float> tables;        // n_tables * 1024 floats
  int n_tables = 0;
  std::vector<uint32_t> classify_tile_begin;  // n_boxes + 1: prefix sum of classify workgroups
  uint64_t classified_bytes = 0;              // size of the frame's classified buffer
  bool ready = false;                         // set by plan_frame
  // the march's work items (build_march_items) of the frame plan this prologue belongs to: they
  // follow from the plan alone, so a camera that repeats does not sort them again (35-45 us of the
  // host's 0.1 ms per frame at N = 8)
  std::vector<MarchItemDev> march_items;
  bool march_items_ready = false;
};

// Screen tiling of the march kernel: workgroup = 16 x 16 pixels, super-tile = 2 x 2 workgroups
// (Morton order inside); the (super-tile, run) items are dealt round-robin to the 8 XCDs in the
// host's cost order.
constexpr int kTile = 16;
constexpr int kSuperTileSide = 2;
constexpr int kSuperTileTiles = kSuperTileSide * kSuperTileSide;
constexpr int kXcds = 8;
// The march items of a frame: for every run the super-tiles its screen rectangle touches, most
// expensive first (estimated samples), padded with kNoMarchItem to a multiple of kXcds.
void build_march_items(const FramePlan& plan, const int32_t* box_order, const int32_t* run_end,
                       int n_runs, const std::vector<RunRectDev>& run_rects,
                       std::vector<MarchItemDev>* items);

// ---- host prologue (avr_host.cpp) ---------------------------------------------------------
void build_color_table(float alpha_scale, float normalization_factor, const float scalar_range[2],
                       const avr_colormap_point* colormap, int colormap_count, float* out_table);
void box_sampling(const avr_box& box, const avr_paint_params& params, float* sample_distance,
                  float* normalization_factor, float* alpha_scale);
float box_depth_hint(const avr_box& box, const avr_camera& camera);
float reference_sample_distance(const avr_box* boxes, int n_boxes, const double bounds_min[3],
                                const double bounds_max[3]);
int layer_order(const float* hints, const int32_t* owner, const int32_t* local_index, int n_layers,
                int32_t* order_out, int32_t* run_end_out);
// Conservative screen rectangle of a box (x0,y0,x1,y1 inclusive; x1 < x0 = off-screen).
void box_screen_rect(const avr_box& box, const avr_camera& camera, int width, int height,
                     int32_t rect[4]);
// The image rows a caller wants extents for: lo <= y <= hi and, with period > 1, only the rows of
// the bands dealt to one piece ((y / band_rows) % period == phase).
struct RowSet {
  int32_t band_rows = 1, period = 1, phase = 0;
  int32_t lo = 0, hi = 0x7fffffff;
};
// Calls f(a, b) for every maximal interval [a, b] of rows of the set inside [lo, hi].
template <typename F>
inline void for_rows(const RowSet& set, int lo, int hi, F&& f) {
  lo = lo > set.lo ? lo : set.lo;
  hi = hi < set.hi ? hi : set.hi;
  if (hi < lo) return;
  if (set.period <= 1) {
    f(lo, hi);
    return;
  }
  int band = lo / set.band_rows;
  band += ((set.phase - band) % set.period + set.period) % set.period;  // first band of the phase
  for (; band * set.band_rows <= hi; band += set.period) {
    const int a = band * set.band_rows, b = a + set.band_rows - 1;
    f(a > lo ? a : lo, b < hi ? b : hi);
  }
}
// A box on screen: its conservative rectangle and the projected corners whose convex hull bounds
// its projection (`whole`: the box reaches behind the eye, every row keeps the whole rectangle).
struct BoxFootprint {
  int32_t rect[4];  // x0, y0, x1, y1 inclusive; x1 < x0 = off-screen
  bool whole = false;
  double px[8], py[8];
};
void box_footprints(const avr_box* boxes, int n_boxes, const avr_camera& camera, int width,
                    int height, BoxFootprint* out);
// Merges (min / max) the box's conservative extent on every row of `rows` inside its rectangle
// into x0[y - y_base], x1[y - y_base]; an entry with x1 < x0 is empty.
void merge_footprint_rows(const BoxFootprint& footprint, const RowSet& rows, int y_base,
                          int32_t* x0, int32_t* x1);
// Per-row extent inside that rectangle (rows rect[1]..rect[3]; x1 < x0 = nothing on the row).
void box_row_spans(const avr_box& box, const avr_camera& camera, int width, int height,
                   const int32_t rect[4], std::vector<int32_t>* row_x0,
                   std::vector<int32_t>* row_x1);
// Fills plan for the given boxes; throws std::invalid_argument / std::runtime_error.
void plan_frame(const avr_box* boxes, int n_boxes, const avr_scalar_transform& transform,
                const avr_paint_params& params, const avr_camera& camera, FramePlan* plan);

// ---- kernel launchers (avr_kernels.hip); all asynchronous on `stream` (hipStream_t) ---------
struct RenderLaunch {
  FrameConsts consts;
  const BoxDev* boxes_dev;      // scene descriptors for this frame
  const float* tables_dev;      // n_tables * 1024 floats
  int n_tables;
  const int32_t* order_dev;     // box indices in global layer order
  const int32_t* order_rects_dev;  // their conservative screen rectangles (x0, y0, x1, y1), same order
  const MarchSpecDev* spec_dev = nullptr;  // a speculative frame's bookkeeping (staged per launch)
  bool spec_is_repair = false;             // ... of its gated second march (a kernel name of its own)
  const uint32_t* classify_gate = nullptr;  // launch_classify: the gated small-grid kernel
  const int32_t* run_end_dev;   // one-past-last position per run
  int n_order, n_runs, n_pieces;
  const RunRectDev* run_rects_dev;    // n_runs
  const RunBlockDev* run_blocks_dev;  // n_runs x n_pieces
  const RunSpanDev* run_spans_dev;    // rows of the blocks with span_base >= 0 (may be null)
  PieceMapDev pieces;                 // which piece a pixel's row belongs to
  float* out_layers;
  unsigned long long* samples_out;  // may be null
  unsigned long long* counters;     // diagnostics (4 x uint64), only read when samples_out is set
  uint8_t* classified;              // frame's classified buffer (FramePlan::classified_bytes)
  const uint32_t* tile_begin_dev;   // n_boxes + 1 prefix of classify workgroups
  int n_boxes;
  uint32_t n_classify_tiles;
  uint32_t classify_lds_pad;  // bytes of LDS each classify workgroup claims beyond its own
                              // (0 = none): caps its occupancy beside the march
  int classify_stream_stores;  // the classified bricklets are written through to memory as a
                               // stream (their reader is a frame away) instead of stored plainly
  const MarchItemDev* items_dev;  // n_items entries (multiple of kXcds)
  uint32_t n_items;
  int only_mode;                        // the IndexMode shared by every box, or -1
  int workgroups_per_cu;                // resident march workgroups per CU (0 = uncapped)
  // A frame in depth-ordered chunks (avr_classify_plan_chunked / avr_march_plan_chunked): the
  // classify launch takes the n_classify_boxes boxes listed in box_list_dev (tile_begin_dev is
  // then the chunk's prefix), the march launch the positions [pos_begin, pos_end) of the global
  // layer order, its run accumulators resumed from what the launch before stored.
  const int32_t* box_list_dev = nullptr;
  int n_classify_boxes = 0;
  int pos_begin = 0, pos_end = -1;      // pos_end < 0: n_order
  int resume = 0;
  // occlusion culling between the chunks (avr_render_plan_culled): the march launch of chunk k
  // flags the boxes behind it that a ray may still sample (indexed by position in the global layer
  // order), the classify launch of chunk k + 1 leaves out the others (indexed like its box list)
  uint8_t* visible_out = nullptr;
  const uint8_t* visible_in = nullptr;
};
// classify pass (cells -> table indices) and march; the march reads what the classify pass of
// the same frame wrote into `classified`
// Copies `bytes` (rounded up to 16; both blocks are that large) from device-mapped pinned host
// memory to device memory.
int launch_upload(const void* host_mapped, void* dev, size_t bytes, void* stream);
int launch_stall(int milliseconds, void* stream);  // test hook: a bounded busy kernel
int launch_classify(const RenderLaunch& launch, void* stream);
int launch_march(const RenderLaunch& launch, void* stream);
int launch_blend(int kind, const void* top, const void* bottom, void* out, int64_t n, void* stream);
int launch_blend_regions(int kind, const void* top, int64_t tb, int64_t te, const void* bottom,
                         int64_t bb, int64_t be, void* out, void* stream);
int launch_encode_u8(const float* rgba, uint32_t* out, int64_t n, void* stream);
int launch_decode_u8(const uint32_t* in, float* rgba, int64_t n, void* stream);
struct FoldLaunch {
  int width;
  PieceMapDev pieces;
  int piece;                           // this rank's piece
  int64_t piece_begin, piece_end;      // contiguous pieces: the image's pixel range; row bands: [0, n)
  int n_runs;                          // global runs, in order
  const RunRectDev* run_rects_dev;     // n_runs
  const RunBlockDev* run_blocks_dev;   // n_runs: block of this rank's piece in the recv buffer
  const RunSpanDev* run_spans_dev;     // rows of the blocks with span_base >= 0 (may be null)
  const float* recv;
  float* out_piece;
  uint8_t* out_rgb8;                   // may be null
  // Blocks whose offset in the receive layout lies in [own_begin, own_end) -- the rank's own runs
  // -- are read own_delta floats away from there: from the send buffer the march stored them in
  // (avr_fold_plan_own).  own_begin == own_end: everything from recv.
  int64_t own_begin = 0, own_end = 0, own_delta = 0;
  int max_workgroups = 0;              // grid cap (0: the default, 2048)
  int flip_height = 0;                 // > 0: out_rgb8 is the whole image, rows top-down (one rank)
};
int launch_fold_plan(const FoldLaunch& launch, void* stream);
int launch_fold_runs(const float* const* slices_dev, int n_slices, float* out, int64_t n,
                     void* stream);
int launch_downsample(const float* src, int tw, int th, int block, float* dst, void* stream);
int launch_quantize(const float* src, int w, int h, int stride, uint8_t* dst, void* stream);
int launch_flip_rows(const uint8_t* src, int64_t row_bytes, int h, uint8_t* dst, void* stream);
// Gathered, piece-major rows (row bands: piece 0's rows, then piece 1's, ...) -> image order;
// flip != 0 also turns the image upside down (bottom-up image -> top-down file rows).
// own (may be null): piece own_piece is read from there instead of from src.
int launch_assemble_rows(const PieceMapDev& pieces, const uint8_t* src, int64_t row_bytes, int flip,
                         uint8_t* dst, void* stream, const uint8_t* own = nullptr, int own_piece = -1);

constexpr uint32_t kScanWorkgroups = 2048;  // grid of the grid-stride cell scans
// scene statistics (avr_scene_stats.hip).  partial_dev: kScanWorkgroups x 32 bytes of scratch;
// out_dev: 32 bytes {min, max, min positive (double), finite count (int64)}.
int launch_scalar_stats(const BoxDev* boxes_dev, const uint32_t* tile_begin_dev, int n_boxes,
                        uint32_t n_tiles, void* partial_dev, void* out_dev, void* stream);
int launch_histogram(const FrameConsts& consts, const BoxDev* boxes_dev,
                     const uint32_t* tile_begin_dev, int n_boxes, uint32_t n_tiles,
                     float range_min, float range_max, int bin_count, uint64_t* counts_dev,
                     void* stream);
// BoxDev records + classify tiling of a box list without any camera (scene statistics).
void plan_cells(const avr_box* boxes, int n_boxes, const avr_scalar_transform& transform,
                FramePlan* plan);
// The scalar-transform part of BuildSceneGeometry (SceneBuilder.cpp:315-443); throws
// std::runtime_error like the reference.
void scene_transform_from_stats(const double stats[3], int64_t finite_count, bool log_scale,
                                bool normalize_to_data_range, avr_scalar_transform* transform,
                                double processed[2], float processed_range[2],
                                float scalar_range[2]);

// Wireframe overlay (avr_overlay.hip): the 12 edges of the bounds box projected by the host.
struct OverlayEdge {
  float sx, sy, ex, ey;   // projected end points (pixels)
  float dx, dy, len_sq;   // end - start, squared length
  int32_t x_begin, x_end, y_begin, y_end;  // pixel rectangle the reference loops over
  int32_t point;          // degenerate edge: a single full-coverage sample
};
struct OverlayPlan {
  OverlayEdge edges[12];
  int32_t n_edges;
  float pixel_radius;
};
// renderBoundingBoxLayer's host part (VolumeRenderer.cpp:139-232, 265-288).
void plan_overlay(const double bounds_min[3], const double bounds_max[3], const avr_camera& camera,
                  int sqrt_antialiasing, int width, int height, OverlayPlan* plan);
// computeTightBounds (VolumeRenderer.cpp:791-848) over replicated box metadata.
void tight_bounds(const avr_box* boxes, int n_boxes, const double fallback_min[3],
                  const double fallback_max[3], double out_min[3], double out_max[3]);
// image / rgb8 hold the pixels [pixel_begin, pixel_end) of the image, or -- pieces != nullptr with
// row bands -- the rows of piece `piece` in order (pixel_begin = 0, pixel_end = its pixel count).
int launch_overlay(const OverlayPlan& plan, int width, int64_t pixel_begin, int64_t pixel_end,
                   const PieceMapDev* pieces, int piece, float* image, uint8_t* rgb8, void* stream);

// ---- visibility ordering (avr_visibility.cpp) -----------------------------------------------
struct VisBox {
  float lo[3], hi[3];
  int owner;
  float min_depth, max_depth;
};
struct VisPair {
  int i, j, axis;
  int b_below_a;  // 0: a.max == b.min on `axis` (a below b); 1: b.max == a.min
};
void visibility_pairs(const std::vector<VisBox>& boxes, std::vector<VisPair>* pairs);
}  // namespace avr
struct avr_visibility_graph;
namespace avr {
avr_visibility_graph* visibility_graph_create(const avr_box* all_boxes, const int32_t* owner,
                                              int n_boxes, int n_ranks);
void visibility_graph_destroy(avr_visibility_graph* graph);
int visibility_rank_count(const avr_visibility_graph* graph);
// Fills rank_order[n_ranks]; returns false (and the default order) when the graph ordering fails.
bool visibility_order(avr_visibility_graph* graph, const avr_camera& camera, float aspect,
                      const char* dot_prefix, int32_t* rank_order, int* n_splits);

void set_error(const std::string& message);
// Deadline (ms) of every host wait on device work, AVR_FRAME_TIMEOUT_MS (default 30000; 0: none):
// a frame of several ranks contains collectives, and a peer that died or whose calls differ must
// end in an error on this rank, not in a hang (DirectSendBase.cpp:206-220, 277 completes or errors).
int frame_timeout_ms();
void set_frame_timeout_ms(int ms);  // < 0: back to the environment's value
// While one is alive on a thread, that thread's waits take the DEFAULT deadline (30 s) when nobody
// set one: the calls of a renderer of several ranks and of a communicator.  A deadline that was
// asked for (avr_set_frame_timeout_ms / AVR_FRAME_TIMEOUT_MS) holds everywhere; without either, a
// single-rank context waits forever, as a plain HIP synchronise would.
class CollectiveScope {
 public:
  explicit CollectiveScope(bool collective);
  ~CollectiveScope();
  CollectiveScope(const CollectiveScope&) = delete;
  CollectiveScope& operator=(const CollectiveScope&) = delete;

 private:
  bool on_;
};
struct DeadlineExceeded : std::runtime_error {
  using std::runtime_error::runtime_error;
};
// Polling host waits (hipEvent_t / hipStream_t) that throw DeadlineExceeded naming `what`.
void wait_event_deadline(void* hip_event, const char* what);
void wait_stream_deadline(void* hip_stream, const char* what);
// The context's HIP stream (hipStream_t; created on first use) with its device made current.
void* context_stream(avr_context* ctx);
// The context's owner bounds the frames in flight itself: a descriptor batch that repeats then
// leaves no packet at all on the stream (by default its event is still recorded, which is what
// keeps the host a few batches ahead of the GPU at most).
void context_set_lean_descriptors(avr_context* ctx, bool lean);
// One rank's fold normally takes one workgroup per CU (it runs beside the next frame's paint
// kernels); `whole`: the next fold of this context takes the whole grid (nothing else is running).
void context_set_fold_whole_grid(avr_context* ctx, bool whole);
// How many descriptor batches the host may stage ahead of the context's stream (1 .. 9; default 4).
void context_set_descriptor_lead(avr_context* ctx, int batches);
// Whether the context's classify passes stream their bricklets to memory (RenderLaunch).
void context_set_classify_stream_stores(avr_context* ctx, bool stream);

// Flags of the events that only ORDER work between this library's streams on ONE device (a
// frame's classified volume -> its march -> its fold; the descriptor ring's slots): recorded
// without the system-scope fence.  Whoever waits on them is another queue of the same GPU (the
// kernel's own end-of-kernel release makes its writes visible there) or the host asking only
// WHETHER the work is through (buffer re-use), never reading what it wrote -- results reach the
// host through a stream / renderer synchronise, which fences.  Measured (tools/microbench/
// packet_gap.hip): a record costs 3.0 us of stream time with the fence and 1.2 us without; with
// two to three records between a frame's kernels and the next frame's: config-2 0.381 -> 0.376 ms,
// config-4 0.992 -> 0.978 ms at fixed reserves.  (hipEventDisableTiming | hipEventDisableSystemFence)
inline unsigned ordering_event_flags() { return 0x2u | 0x20000000u; }

}  // namespace avr

#endif
