/*
 * avr_hip.h -- C ABI of the MI355X (gfx950) volume-rendering hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types, no exceptions.
 * Every entry point cites the reference interface (file:line under the amrVolumeRenderer tree)
 * it replaces.  All image/cell pointers are DEVICE pointers (HBM) unless a parameter says
 * "host".  Every function returns 0 on success or a negative avr_status; the message of the
 * last failure on the calling thread is available from avr_last_error().
 *
 * There is no CPU fallback behind this ABI: without a HIP device every compute entry point
 * fails with AVR_ERR_NO_DEVICE.
 *
 * Pixel layouts (same as the reference's image buffers):
 *   depth-sort image : 5 floats / pixel  (premultiplied r, g, b, a, depth)   -- ImageRGBAFloatColorDepthSort
 *   float image      : 4 floats / pixel  (premultiplied r, g, b, a)          -- ImageRGBAFloatColorOnly
 *   ubyte image      : 1 uint32 / pixel  (bytes r, g, b, a in memory order)  -- ImageRGBAUByteColorOnly
 * Pixel index p = y * width + x, image origin bottom-left (Common/VolumePainter.cpp:738-739).
 */
#ifndef AVR_HIP_H
#define AVR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  AVR_OK = 0,
  AVR_ERR_INVALID_ARGUMENT = -1, /* std::invalid_argument in the reference API */
  AVR_ERR_RUNTIME = -2,          /* std::runtime_error (HIP failure, bad image type, ...) */
  AVR_ERR_NO_DEVICE = -3,
  AVR_ERR_OUT_OF_MEMORY = -4
} avr_status;

typedef struct avr_context avr_context;
typedef struct avr_scene avr_scene;

/* volume::AmrBox (Common/VolumeTypes.hpp:69-76).  `cells` = address of
 * values(validBox.smallEnd(), component): x fastest, then jstride, kstride (amrex::Array4). */
typedef struct {
  double min_corner[3];
  double max_corner[3];
  int32_t dims[3];
  int32_t level;        /* informational (AMR level); not read by the kernels */
  const double *cells;  /* device pointer */
  int64_t jstride;
  int64_t kstride;
} avr_box;

/* volume::ScalarTransform (Common/VolumeTypes.hpp:21-31): the fields the path reads. */
typedef struct {
  int32_t log_scale_input;
  int32_t normalize_to_unit_range;
  double positive_floor;
  double normalization_min;
  double inverse_normalization_span;
} avr_scalar_transform;

/* volume::CameraParameters (Common/VolumeTypes.hpp:83-90). */
typedef struct {
  double eye[3];
  double look_at[3];
  double up[3];
  float fov_y_degrees;
  float near_plane;
  float far_plane;
} avr_camera;

/* volume::ColorMapControlPoint (Common/VolumeTypes.hpp:92-98). */
typedef struct {
  float value, red, green, blue, alpha;
} avr_colormap_point;

/* The scalar arguments of VolumePainter::paint (Common/VolumePainter.hpp:19-31):
 * image size, scalarRange, boxTransparency, referenceSampleDistance, bounds, colorMap.
 * (rank, numProcs, antialiasing are unused by the reference: VolumePainter.cpp:561-562.) */
typedef struct {
  int32_t width;
  int32_t height;
  float scalar_range[2];
  float box_transparency;
  float reference_sample_distance;
  double bounds_min[3];
  double bounds_max[3];
  const avr_colormap_point *colormap; /* host pointer; NULL/0 = default jet map */
  int32_t colormap_count;
} avr_paint_params;

/* ---- library / context ------------------------------------------------------------------ */

/* Message of the last failure on this thread ("" if none). */
const char *avr_last_error(void);

/* ABI version of this header (bumped on incompatible change).
 *   2 (round 5): for ranks of several the RGB8 bytes of a frame reach rank 0's rgb8_out one frame
 *     late by default (avr_renderer_set_deferred_gather, avr_renderer_outputs_complete) and
 *     avr_renderer_synchronize is collective while such a gather is pending; the test hooks and
 *     diagnostics moved to avr_hip_debug.h; avr_context_set_cu_mask_pattern is gone.
 *     (Added since, compatibly: avr_classify_plan_flagged / _positions, avr_march_plan_speculative,
 *     avr_renderer_set_visibility_speculation, avr_renderer_speculation_state.)
 *   1: rounds 1-4. */
#define AVR_ABI_VERSION 2
int avr_abi_version(void);

/* Deadline, in milliseconds, of every host wait on device work inside this library (0 = wait
 * forever; < 0 = back to the default).  Default: the environment's AVR_FRAME_TIMEOUT_MS; without
 * it 30000 for the calls that involve other ranks (a renderer of several ranks or with a
 * communicator, the avr_comm_* / avr_exchange* / avr_gather* calls) and NONE for single-rank
 * contexts and renderers, which wait as long as a plain HIP synchronise would.
 * The frame of a rank of several contains collectives; the reference's exchange either completes
 * or errors (MPI_Waitany / MPI_Waitall, DirectSend/Base/DirectSendBase.cpp:206-220, 277).  Here a
 * wait that outlasts the deadline -- a peer that died, ranks whose calls differ -- makes the call
 * return AVR_ERR_RUNTIME with avr_last_error() naming what did not finish (stream, frame); the
 * renderer is then failed for good (every later call returns the error at once; destroy it). */
int avr_set_frame_timeout_ms(int milliseconds);

/* Creates a context bound to HIP device `device_id` with its own stream.  Replaces the
 * function-local static VolumePainter / DirectSendBase instances of
 * VolumeRenderer/VolumeRenderer.cpp:909-927 (one context per rank = per GPU). */
int avr_context_create(int device_id, avr_context **out_ctx);
void avr_context_destroy(avr_context *ctx);

/* The same with the context's own stream created in the device's most urgent priority class
 * (high_priority != 0) or in the default class. */
int avr_context_create_with_priority(int device_id, int high_priority, avr_context **out_ctx);

/* The stream the context launches on (hipStream_t as void*), for ordering other work against it. */
void *avr_context_stream(avr_context *ctx);

/* Use an externally owned HIP stream (hipStream_t passed as void*; NULL = the context's own).
 * All launches of the context go to this stream; nothing in this ABI synchronises the stream
 * except avr_context_synchronize and the *_host readbacks. */
int avr_context_set_stream(avr_context *ctx, void *hip_stream);
int avr_context_synchronize(avr_context *ctx);

/* Resident march workgroups per CU for the launches of this context: 0 (default) = as many as
 * fit (7 by the kernel's registers), 1..7 = capped (the launch reserves 160 KiB / n of LDS per
 * workgroup).  Round 1's frame driver capped its 8-per-CU march at 5 to leave room for the
 * classify pass of the next frame; the present driver leaves the march uncapped and holds the
 * classify pass back instead (avr_context_set_classify_lds_reserve; DESIGN.md section 3, "What
 * the two kernels compete for").  Never changes results. */
int avr_context_set_march_occupancy(avr_context *ctx, int workgroups_per_cu);

/* LDS (bytes, 0 = none) each classify workgroup of this context's launches claims beyond the
 * 2 KiB it stages bricklets in: caps how many of them a CU holds at once.  Beside the march the
 * classify pass takes memory-system time from the march in proportion to the bandwidth it
 * reaches, whatever its arithmetic, occupancy or cache policy (profiles/experiments_rounds_1_to_3.md section 3); the
 * renderer's frame driver uses this to make both kernels of a frame take equally long
 * (avr_renderer_set_classify_share).  Never changes results. */
#define AVR_CLASSIFY_LDS_RESERVE_MAX 61440
int avr_context_set_classify_lds_reserve(avr_context *ctx, int bytes);

/* ---- host-side per-frame quantities (no device work) ------------------------------------- */

/* buildColorTable (Common/VolumePainter.cpp:442-516): 256 RGBA entries to host memory. */
int avr_build_color_table(float alpha_scale, float normalization_factor,
                          const float scalar_range[2], const avr_colormap_point *colormap,
                          int colormap_count, float out_table_host[1024]);

/* sampleDistance / normalizationFactor / alphaScale of one box (VolumePainter.cpp:571-613). */
int avr_box_sampling(const avr_box *box, const avr_paint_params *params, float *sample_distance,
                     float *normalization_factor, float *alpha_scale);

/* computeBoxDepthHint (VolumeRenderer/VolumeRenderer.cpp:541-553). */
int avr_box_depth_hint(const avr_box *box, const avr_camera *camera, float *out_hint);

/* The conservative footprint of a box on a width x height image that the frame plans are built
 * from (host only): its screen rectangle rect[4] = {x0, y0, x1, y1} inclusive (x1 < x0: off
 * screen) -- the march only traces the rays of those pixels -- and, for every image row y, the
 * columns row_x0[y] .. row_x1[y] of the tightened layout (avr_frame_plan_tighten; row_x1 < row_x0:
 * nothing on that row; rows outside the rectangle are empty).  row_x0 / row_x1: height entries
 * each, or NULL.  For the parity tests: every pixel the reference's K1 leaves non-empty for the
 * box (VolumePainter.cpp:802-837: slab hit in front of the eye) must lie inside both. */
int avr_box_footprint(const avr_box *box, const avr_camera *camera, int width, int height,
                      int32_t rect[4], int32_t *row_x0, int32_t *row_x1);

/* referenceSampleDistance of renderSingleTrial (VolumeRenderer/VolumeRenderer.cpp:1138-1190)
 * over the boxes given (the caller reduces with MAX over ranks when boxes are distributed:
 * pass the already-reduced coarsest spacing through avr_paint_params instead). */
int avr_reference_sample_distance(const avr_box *boxes, int n_boxes, const double bounds_min[3],
                                  const double bounds_max[3], float *out_distance);

/* Global layer order of DirectSendBase::composeLayered (DirectSend/Base/DirectSendBase.cpp:
 * 363-410): sorts layers by (hint, owner, local_index); order_out[n] = layer ids,
 * run_end_out[r] = one-past-last position of run r (maximal same-owner stretch); *n_runs_out. */
int avr_layer_order(const float *hints, const int32_t *owner, const int32_t *local_index,
                    int n_layers, int32_t *order_out, int32_t *run_end_out, int *n_runs_out);

/* getPieceRange (DirectSend/Base/DirectSendBase.cpp:59-74). */
int avr_piece_range(int64_t image_size, int piece_index, int num_pieces, int64_t *begin,
                    int64_t *end);

/* ---- painter ----------------------------------------------------------------------------- */

/* VolumePainter::paint (Common/VolumePainter.hpp:19-31, .cpp:548-961) for ONE box: writes
 * width*height depth-sort pixels to out_rgbad (device).  If samples_out (device, 1 x uint64)
 * is non-NULL the number of executed cell fetches is ADDED to it. */
int avr_paint_box(avr_context *ctx, const avr_box *box, const avr_scalar_transform *transform,
                  const avr_paint_params *params, const avr_camera *camera, float *out_rgbad,
                  uint64_t *samples_out);

/* A scene = the rank's local boxes (geometry.localBoxes, VolumeRenderer.cpp:1201) with one
 * scalar transform (geometry.scalarTransform).  Descriptors are copied to the device; cell
 * data stays where `cells` points. */
int avr_scene_create(avr_context *ctx, const avr_box *boxes, int n_boxes,
                     const avr_scalar_transform *transform, avr_scene **out_scene);
void avr_scene_destroy(avr_scene *scene);

/* Off by default: every frame's classify pass re-reads the f64 cells, as the reference's
 * per-sample fetch does.  When enabled, a classified volume is kept across frames for as long as
 * what it was derived from is unchanged -- the boxes (cell pointers, strides, dimensions), the
 * scalar transform and the scalar range -- so that a camera moving over a static data set only
 * pays for the march.  The cells are the caller's memory: after changing them in place call
 * avr_scene_invalidate.  Results are identical either way. */
int avr_scene_set_classification_cache(avr_scene *scene, int enabled);
int avr_scene_invalidate(avr_scene *scene);

/* Fused replacement of the per-box loop + owner-side run fold
 * (VolumeRenderer.cpp:1201-1219 + DirectSendBase.cpp:413-426): first a streaming classify pass
 * turns every f64 cell of the scene into its transfer-function table index (the per-sample
 * arithmetic of VolumePainter.cpp:870-883 is a pure function of the cell), then one thread per
 * pixel marches the local boxes box_order[0..n) (indices into the scene, already in global
 * layer order) and folds each run (run_end[r] = one-past-last position of run r in box_order)
 * with the depth-sort blend, in order.  Run r's layer is written for ALL pixels in dense "send
 * layout": the image is cut in n_pieces DirectSend pieces (avr_piece_range) and
 *     out[ piece_offset(k) + (r * piece_len(k) + (p - piece_begin(k))) * 5 .. +5 ]
 * with piece_offset(k) = 5 * n_runs * piece_begin(k); for n_pieces == 1 this is simply
 * out[r][p][5].  The result is bit-identical to painting every box into its own layer and
 * blending (empty pixels are an exact identity of the blend).
 * samples_out as in avr_paint_box. */
int avr_render_runs(avr_context *ctx, const avr_scene *scene, const avr_paint_params *params,
                    const avr_camera *camera, const int32_t *box_order, int n_order,
                    const int32_t *run_end, int n_runs, int n_pieces, float *out_layers,
                    uint64_t *samples_out);

/* ---- frame plan: layered DirectSend with one sparse exchange ------------------------------ */

/* One frame's compositing plan, identical on every rank.  Replaces the per-frame allgather of
 * layer counts / depth hints, the global sort and the run grouping of
 * DirectSendBase::composeLayered (DirectSend/Base/DirectSendBase.cpp:329-410) -- depth hints
 * depend only on box corners and the camera, so every rank derives them from the replicated box
 * metadata -- and adds, per run, the conservative screen rectangle of its boxes: pixels outside
 * it hold the empty layer pixel (0,0,0,0,+inf), an exact identity of the blend, and are neither
 * stored nor sent.
 *
 *   all_boxes[n_boxes]  metadata of every box of the scene (cells may be NULL), level-major;
 *   owner[n_boxes]      owning rank; a rank's local index of a box is its position among the
 *                       boxes it owns (geometry.localBoxes order);
 *   group_order         rank at position k of the ordered MPI group of Compositor::compose
 *                       (Common/Compositor.hpp:19-40); NULL = identity.  Position k holds
 *                       pixel piece k (DirectSendBase.cpp:76-130).  Pixel values do not depend
 *                       on it (SURVEY.md 8c probe 1).
 * Exchange layout (floats; 5 per pixel):
 *   send buffer = for peer rank s = 0..n-1: for local run r: block(piece of s, r)
 *   recv buffer = for source rank s = 0..n-1: for run r of s: block(my piece, r)
 *   block(k, r) = rows [max(rect.y0, first row of piece k), min(rect.y1, last row of piece k)]
 *                 x columns [rect.x0, rect.x1] of run r, row-major.
 * so one all-to-all (split sizes from avr_frame_plan_splits) replaces the N(N-1) messages per
 * run of the reference. */
typedef struct avr_frame_plan avr_frame_plan;

typedef struct {
  int32_t n_ranks, rank;
  int32_t n_runs_total;   /* global runs, all ranks */
  int32_t n_local_runs;   /* runs owned by this rank */
  int32_t n_local_boxes;
  int64_t n_pixels;       /* render width * height */
  int64_t piece_begin;    /* this rank's DirectSend piece [begin, end) */
  int64_t piece_end;
  int64_t send_floats;    /* total size of the send / recv buffers */
  int64_t recv_floats;
  int32_t piece_layout;   /* AVR_PIECES_CONTIGUOUS / AVR_PIECES_ROW_BANDS (see below) */
  int32_t band_rows;
} avr_frame_plan_info;

/* How the image is dealt to the ranks' pieces.  AVR_PIECES_CONTIGUOUS is the reference's
 * partition (DirectSendBase.cpp:59-74) and what Compositor::compose promises its caller.
 * AVR_PIECES_ROW_BANDS deals bands of band_rows image rows round-robin (band b -> piece b % N):
 * every rank then receives and folds the same share of every screen region (contiguous pieces of
 * a scene that covers a third of the image leave some ranks with nothing and others with
 * everything).  Per-pixel results do not depend on which rank folds a pixel, so the gathered
 * image is identical; piece_begin is 0 and piece_end the piece's pixel count there (its rows in
 * image order), the gathered buffer is piece-major, and avr_assemble_rows restores image order.
 * Used by the frame driver (avr_renderer), whose caller only ever sees the gathered image. */
#define AVR_PIECES_CONTIGUOUS 0
#define AVR_PIECES_ROW_BANDS 1

/* One run as seen by the exchange (for inspection and tests). */
typedef struct {
  int32_t owner;          /* owning rank */
  int32_t local_run;      /* index among the owner's runs */
  int32_t first_layer;    /* position of its first layer in the global order */
  int32_t n_layers;
  int32_t rect[4];        /* x0, y0, x1, y1 inclusive; x1 < x0 = empty */
} avr_run_info;

int avr_frame_plan_create(const avr_box *all_boxes, const int32_t *owner, int n_boxes,
                          int n_ranks, int rank, const int32_t *group_order,
                          const avr_paint_params *params, const avr_camera *camera,
                          avr_frame_plan **out_plan);
/* The same with a piece layout (avr_frame_plan_create = AVR_PIECES_CONTIGUOUS). */
int avr_frame_plan_create_pieces(const avr_box *all_boxes, const int32_t *owner, int n_boxes,
                                 int n_ranks, int rank, const int32_t *group_order,
                                 const avr_paint_params *params, const avr_camera *camera,
                                 int piece_layout, int band_rows, avr_frame_plan **out_plan);
/* The same plan for layers that are only known as images with a depth hint -- the generic
 * Compositor::compose(Image*, MPI_Group, MPI_Comm) of a LayeredImageInterface
 * (DirectSend/Base/DirectSendBase.cpp:316-458): hints[l] / owner[l] for ALL layers of all ranks in
 * rank-major order (what the reference's MPI_Allgatherv of the depth hints delivers, :354-361; a
 * layer's local index is its position among its owner's layers).  Every run covers the whole
 * image.  Use with avr_pack_layers, avr_exchange, avr_fold_plan. */
int avr_layered_plan_create(const float *hints, const int32_t *owner, int n_layers, int n_ranks,
                            int rank, const int32_t *group_order, int width, int height,
                            avr_frame_plan **out_plan);
/* Owner-side run fold of already painted layers (DirectSendBase.cpp:413-426) into the send buffer
 * of a layered plan: local_layers[i] = device pointer to this rank's layer i (width*height*5
 * floats), host array. */
int avr_pack_layers(avr_context *ctx, const avr_frame_plan *plan, const float *const *local_layers,
                    int n_local_layers, float *send_buffer);
void avr_frame_plan_destroy(avr_frame_plan *plan);
int avr_frame_plan_get_info(const avr_frame_plan *plan, avr_frame_plan_info *out);
/* all_to_all split sizes in floats, indexed by peer rank (n_ranks entries each). */
int avr_frame_plan_splits(const avr_frame_plan *plan, int64_t *send_splits, int64_t *recv_splits);
/* Global layer order: layer_box[l] = index into all_boxes of the l-th layer; n_boxes entries. */
int avr_frame_plan_layers(const avr_frame_plan *plan, int32_t *layer_box);
/* runs[n_runs_total] in global order. */
int avr_frame_plan_runs(const avr_frame_plan *plan, avr_run_info *runs);
/* Tightens the exchange layout of a plan made by avr_frame_plan_create: instead of its whole
 * screen rectangle, every row of a run stores only the conservative extent of the run's boxes on
 * that row (the convex hull of each box's projected corners, +2 pixels); what lies outside is the
 * cleared layer pixel -- the exact identity of the depth-sort blend (DirectSendBase.cpp:400-446
 * blends it like any other) -- on the sender and the receiver alike, so results do not change
 * while send_floats / recv_floats and the splits shrink (config-4: the rectangles are 48-72 %
 * filled).  Sizes still follow from the replicated metadata: EVERY rank must tighten the plan of
 * a frame or none.  all_boxes as given to avr_frame_plan_create.  The march counts, in the 5th
 * diagnostic counter (avr_context_set_march_counters, avr_hip_debug.h), non-empty pixels it found outside a span:
 * always 0.  avr_frame_plan_send_block / recv_block do not apply to a tightened plan. */
int avr_frame_plan_tighten(avr_frame_plan *plan, const avr_box *all_boxes, int n_boxes);

/* Offset (floats) of block(piece of `peer`, local run r) in the send buffer and of
 * block(my piece, global run g) in the recv buffer; -1 when the block is empty.
 * first_row receives the image row of the block's first row. */
int avr_frame_plan_send_block(const avr_frame_plan *plan, int peer, int local_run,
                              int64_t *offset, int32_t *first_row, int32_t *n_rows);
int avr_frame_plan_recv_block(const avr_frame_plan *plan, int global_run, int64_t *offset,
                              int32_t *first_row, int32_t *n_rows);

/* Classify + march of this rank's runs into the sparse send buffer (send_floats floats).
 * The scene must hold this rank's boxes in local-index order.  samples_out as avr_paint_box. */
int avr_render_plan(avr_context *ctx, const avr_scene *scene, const avr_frame_plan *plan,
                    float *send_buffer, uint64_t *samples_out);

/* The two halves of avr_render_plan as separate calls, for pipelining frames: the classify pass
 * of frame i+1 (any context / stream) may run while frame i is still marched.  `slot`
 * (0 .. AVR_CLASSIFIED_SLOTS-1) selects one of the scene's classified volumes (each allocated
 * when first used); the caller orders march(slot) after classify(slot) of the same frame and
 * classify(slot) after the previous march(slot).  Two slots are enough to overlap; with three
 * the classify stream can run a frame further ahead, so that neither stream waits for a launch
 * on the other (avr_renderer does). */
#define AVR_CLASSIFIED_SLOTS 3
int avr_classify_plan(avr_context *ctx, const avr_scene *scene, const avr_frame_plan *plan,
                      int slot);
int avr_march_plan(avr_context *ctx, const avr_scene *scene, const avr_frame_plan *plan, int slot,
                   float *send_buffer, uint64_t *samples_out);

/* ONE frame in n_chunks (2 .. AVR_MAX_FRAME_CHUNKS) depth-ordered chunks, for the caller who waits
 * for every frame (the reference's Render() returns after one: VolumeRenderer.cpp:1103-1339): the
 * rank's boxes are cut, in global layer order, into chunks of equal classify work; chunk k is
 * classified by launch k of avr_classify_plan_chunked (which then records chunk_events[k],
 * hipEvent_t as void*, on its stream) and marched by launch k of avr_march_plan_chunked (which
 * first makes its stream wait for chunk_events[k]) -- so chunk k is marched while chunk k + 1 is
 * still being classified and the frame's latency is one chunk's classify pass plus the march, not
 * the whole pass plus the march.  The run accumulators are carried from launch to launch through
 * the send buffer: the run's left fold (DirectSendBase.cpp:413-426) is cut, never re-associated,
 * so the results equal avr_march_plan's bit for bit.  chunk_events may be NULL when both calls go
 * to one stream.  first_alone != 0: nothing else runs beside chunk 0's classify launch, so it
 * takes no LDS reserve (avr_context_set_classify_lds_reserve).  Not with a cached classification. */
#define AVR_MAX_FRAME_CHUNKS 16
int avr_classify_plan_chunked(avr_context *ctx, const avr_scene *scene, const avr_frame_plan *plan,
                              int slot, int n_chunks, void *const *chunk_events, int first_alone);
int avr_march_plan_chunked(avr_context *ctx, const avr_scene *scene, const avr_frame_plan *plan,
                           int slot, float *send_buffer, uint64_t *samples_out, int n_chunks,
                           void *const *chunk_events);

/* The same frame with OCCLUSION CULLING between its chunks, one call: classify 0, march 0,
 * classify 1, march 1 ... on the context's stream.  The march skips a box at every pixel whose run
 * accumulator is opaque and in front of the box (the blend would return the accumulator unchanged:
 * the reference's early exit `accumA < 1`, VolumePainter.cpp:837, carried from box to box); the
 * march launch of chunk k evaluates that same test for every box behind it and flags the boxes
 * some ray may still sample (visibility: device, n_chunks * n_local_boxes bytes of scratch), and
 * the classify launch of chunk k + 1 leaves out the others -- their f64 cells are not even read.
 * With the reference's default boxTransparency = 0 most rays saturate in the first boxes, and a
 * frame reads a fraction of its cells.  Results equal avr_render_plan's bit for bit: a box is left
 * out only where the march provably never reads it (proof with the kernel, avr_kernels.hip). */
int avr_render_plan_culled(avr_context *ctx, const avr_scene *scene, const avr_frame_plan *plan,
                           int slot, float *send_buffer, uint64_t *samples_out, int n_chunks,
                           uint8_t *visibility);

/* SPECULATIVE frames: occlusion culling from one frame to the next, exact.  With the reference's
 * default boxTransparency = 0 (VolumeRenderer.hpp:36) most rays saturate in the first boxes they
 * cross (VolumePainter.cpp:837 carried from box to box: the march skips a box wherever the run
 * accumulator in front of it is opaque), so a frame reads the f64 cells of boxes no ray samples.
 * A frame of the same plan as an earlier one may therefore classify only the boxes that earlier
 * frame's march sampled (`visited`, recorded by avr_march_plan_speculative):
 *   avr_classify_plan_flagged(flags = that frame's visited[])      the covered boxes only
 *   avr_march_plan_speculative(classified = the same flags, ...)   checks every box it needs
 * The march that needs a box the flags left out (the cells changed, say) raises missed[position]
 * and miss_count, leaves the box out and goes on; with a less opaque accumulator it can only need
 * MORE boxes than the true frame, so the raised flags cover everything the true frame needs, and
 * the caller queues, unconditionally, the repair behind it on the same stream:
 *   avr_classify_plan_flagged(flags = missed, gate = miss_count)   a small grid, idle unless *gate
 *   avr_march_plan_speculative(classified = NULL, gate = miss_count, ...)   the whole march again
 * Both do nothing when no flag was raised (the usual case: two launches of a few microseconds).
 * Results equal avr_classify_plan + avr_march_plan bit for bit either way.  All arrays are device
 * memory of plan->info.n_local_boxes entries, indexed by the position of a box among this rank's
 * boxes in the plan's global layer order (opaque to the caller: flags one call wrote are what the
 * next one reads); visited / missed / miss_count are cleared by the caller.
 * host_miss_flag: NULL, or device-visible host memory that the march sets to 1 on a miss. */
typedef struct avr_speculation {
  const uint8_t *classified; /* != 0: classified this frame; NULL: every box is */
  uint8_t *visited;          /* out (or NULL): 1 for every box some ray sampled */
  uint8_t *missed;           /* out: boxes needed but not classified (NULL only if classified is) */
  uint32_t *miss_count;      /* out: raised with every miss */
  uint32_t *host_miss_flag;  /* out, or NULL */
  const uint32_t *gate;      /* NULL, or: the march does nothing unless *gate != 0 */
  const uint8_t *classified_host; /* instead of `classified`: the same flags in HOST memory, copied
                                     with the launch's descriptors (a caller who decides per frame) */
  uint8_t *dirty_workgroups; /* NULL, or device memory of avr_march_plan_workgroups(plan) bytes, cleared
                                by the caller: the checking march marks the workgroups that met an
                                unclassified box, the gated march (same pointer) redoes only those --
                                every other layer pixel is final after the first pass */
} avr_speculation;
/* An upper bound of the workgroups of this rank's march launch of the plan: the bytes to give
 * dirty_workgroups. */
int avr_march_plan_workgroups(const avr_frame_plan *plan, int64_t *workgroups);
int avr_classify_plan_flagged(avr_context *ctx, const avr_scene *scene, const avr_frame_plan *plan,
                              int slot, const uint8_t *flags, const uint32_t *gate);
int avr_march_plan_speculative(avr_context *ctx, const avr_scene *scene, const avr_frame_plan *plan,
                               int slot, float *send_buffer, const avr_speculation *speculation);
/* The flagged classify pass for a caller who has the flags on the HOST (copied from `visited` a few
 * frames ago): positions[0 .. n_positions) ascending -- a launch of exactly those boxes' tiles
 * (avr_classify_plan_flagged launches a workgroup per tile of every box and lets the unflagged ones
 * return: a sixth of config-4's flagged pass).  The march is then given the same set as flags. */
int avr_classify_plan_positions(avr_context *ctx, const avr_scene *scene, const avr_frame_plan *plan,
                                int slot, const int32_t *positions, int n_positions);

/* Receiver side of composeLayered (DirectSendBase.cpp:400-446) for this rank's piece: folds
 * the runs in global order from the received buffer (recv_floats floats; with one rank the send
 * buffer itself) into out_piece[(piece_end - piece_begin) * 5]; pixels no run covers become the
 * cleared layer pixel (DirectSendBase.cpp:450-455).  If out_rgb8 is non-NULL the piece is also
 * written as RGB8 (Color::GetComponentAsByte, 3 bytes per pixel, same pixel order); out_piece
 * may then be NULL when only the bytes are wanted (20 of the 23 bytes stored per pixel). */
int avr_fold_plan(avr_context *ctx, const avr_frame_plan *plan, const float *recv_buffer,
                  float *out_piece, uint8_t *out_rgb8);
/* The same fold with the blocks of the rank's OWN runs read where the march stored them -- in
 * own_send_buffer, the rank's send buffer (its block for itself has the same layout there as in
 * the receive buffer) -- instead of from recv_buffer: with avr_exchange_peers a rank's data for
 * itself is never copied (one kernel and one gap less on the compositing stream of every frame,
 * which at N = 8 has five kernels to run in 0.15 ms).  own_send_buffer NULL: avr_fold_plan. */
int avr_fold_plan_own(avr_context *ctx, const avr_frame_plan *plan, const float *recv_buffer,
                      const float *own_send_buffer, float *out_piece, uint8_t *out_rgb8);
/* One rank (the piece is the image): the fold with its bytes written as the output file's rows,
 * top-down (SavePPM.cpp:25) -- avr_fold_plan + avr_assemble_rows(flip) in one pass over the
 * pixels.  out_rgb8_image: width * height * 3 bytes; out_piece as avr_fold_plan (may be NULL). */
int avr_fold_plan_image(avr_context *ctx, const avr_frame_plan *plan, const float *recv_buffer,
                        float *out_piece, uint8_t *out_rgb8_image);

/* ---- image algebra ----------------------------------------------------------------------- */

/* Features::blend over n pixels, out = blend(top, bottom); out may alias top or bottom.
 * ImageRGBAFloatColorDepthSort.hpp:13-27 / ImageRGBAFloatColorOnly.hpp:19-26 /
 * ImageRGBAUByteColorOnly.hpp:19-34 (uint8 wrap-around preserved). */
int avr_blend_depthsort_f32x5(avr_context *ctx, const float *top, const float *bottom,
                              float *out, int64_t n_pixels);
int avr_blend_rgba_f32x4(avr_context *ctx, const float *top, const float *bottom, float *out,
                         int64_t n_pixels);
int avr_blend_rgba_u8x4(avr_context *ctx, const uint32_t *top, const uint32_t *bottom,
                        uint32_t *out, int64_t n_pixels);

/* ImageColorOnly<F>::blend with pixel regions (Common/ImageColorOnly.hpp:119-199): top covers
 * [tb,te), bottom [bb,be), out covers [min(tb,bb), max(te,be)); the regions must touch or
 * overlap.  kind: 0 depth-sort, 1 float, 2 ubyte.  out must not alias the inputs. */
int avr_blend_regions(avr_context *ctx, int kind, const void *top, int64_t tb, int64_t te,
                      const void *bottom, int64_t bb, int64_t be, void *out);

/* Color::GetComponentAsByte / SetComponentFromByte over n RGBA pixels
 * (Common/Color.hpp:66-91, ImageRGBAUByteColorOnly.cpp:16-39). */
int avr_encode_rgba_u8(avr_context *ctx, const float *rgba, uint32_t *out, int64_t n_pixels);
int avr_decode_rgba_u8(avr_context *ctx, const uint32_t *in, float *rgba, int64_t n_pixels);

/* Receiver-side fold of DirectSendBase::composeLayered (DirectSendBase.cpp:400-446) for one
 * piece of n_pixels pixels: slices[r] (device pointers, host array) are the piece's pixels of
 * global run r in global order; out = fold_left(blend, slices).  n_slices == 0 writes the
 * cleared layer (0,0,0,0,+inf) (DirectSendBase.cpp:450-455). */
int avr_fold_runs_depthsort(avr_context *ctx, const float *const *slices_host, int n_slices,
                            float *out, int64_t n_pixels);

/* ---- frame tail --------------------------------------------------------------------------- */

/* downsampleImage (VolumeRenderer/VolumeRenderer.cpp:479-528): block x block box mean of a
 * (target_w*block) x (target_h*block) depth-sort image; depth := +inf. */
int avr_downsample_depthsort(avr_context *ctx, const float *src, int target_w, int target_h,
                             int block, float *dst);

/* SavePPM/SavePNG pixel bytes (Common/SavePPM.cpp:17-36, Common/Color.hpp:86-90): RGB8 from an
 * image of `stride` floats per pixel, rows written top-down (y = h-1 .. 0). dst = w*h*3 bytes. */
int avr_quantize_rgb8(avr_context *ctx, const float *src, int w, int h, int stride,
                      uint8_t *dst);

/* Rows of an h-row byte image in reverse order: the composited image has its origin bottom-left,
 * the output file's rows run top-down (Common/SavePPM.cpp:25, Common/SavePNG.cpp:64-71). */
int avr_flip_rows(avr_context *ctx, const uint8_t *src, int64_t row_bytes, int h, uint8_t *dst);

/* ---- rank communicator: the DirectSend exchange and the gather over RCCL / xGMI ----------- */

/* One communicator per rank (= per GPU, one process per rank).  Replaces the MPI communicator +
 * group handed to Compositor::compose (Common/Compositor.hpp:35-37): the image-fragment exchange
 * of DirectSendBase::PostSends / PostReceives (DirectSend/Base/DirectSendBase.cpp:76-177) and
 * ImageFull::Gather (Common/ImageColorOnly.hpp:220-270) become one grouped ncclSend / ncclRecv
 * round each, on device buffers, launched on the calling context's stream.
 *
 * avr_comm_unique_id is called on ONE rank; the caller distributes the id to the others by its
 * own means (the reference's host already has MPI: MPI_Bcast of 128 bytes; torch.distributed's
 * store in the Python layer); then every rank calls avr_comm_create (collective). */
typedef struct avr_comm avr_comm;
#define AVR_COMM_ID_BYTES 128
int avr_comm_unique_id(char id_out[AVR_COMM_ID_BYTES]);
int avr_comm_create(int device_id, const char id[AVR_COMM_ID_BYTES], int rank, int n_ranks,
                    avr_comm **out_comm);
/* In-process rehearsal of an N-rank frame on ONE GPU: n_ranks connected communicators for n_ranks
 * host threads of this process (host-synchronous device copies).  Not a performance path -- RCCL
 * cannot place two ranks on one device, and this keeps the N-rank frame testable there. */
int avr_comm_create_local(int n_ranks, avr_comm **out_comms /* [n_ranks] */);
/* ONE rank of an N-rank frame played alone (timing studies of a rank's share on a single GPU,
 * tools/rank_share.py): only the block a rank keeps for itself moves, the peers' blocks of the
 * receive buffer keep whatever they held.  Frames rendered this way are not images. */
int avr_comm_create_solo(int rank, int n_ranks, avr_comm **out_comm);
/* The same, with the rank's collectives played through RCCL to ITSELF (a one-rank communicator
 * made here): avr_exchange sends and receives every peer's block -- the smaller of its two sizes
 * -- as one grouped ncclSend / ncclRecv round, avr_gather likewise.  What a one-GPU box can show
 * of the real round: its launch on the host, its kernel beside the paint kernels, its bytes
 * through HBM; not the links.  percent (1..100): the share of every peer's block avr_exchange
 * moves -- on the node the N - 1 blocks travel over N - 1 links at once, here one after the other
 * over one connection, so 100 overstates how long the round's kernel runs.  percent 0 = "one link":
 * only the largest peer block travels, whole, and the root's gather receives one peer's piece --
 * what the busiest of a rank's N - 1 links carries; RCCL works off the operations for ONE peer one
 * after the other (~13 us each, measured), a serialisation the node, with one peer per link, does
 * not have.  Timing only. */
int avr_comm_create_solo_rccl(int device_id, int rank, int n_ranks, int percent,
                              avr_comm **out_comm);
/* Rehearsal of the N-rank frame across PROCESSES that share one GPU (RCCL refuses two ranks on
 * one device): the ranks meet in the POSIX shared-memory segment `name` ("/something"; whoever
 * comes first creates it, rank 0 removes the name when it is destroyed), capacity_bytes of staging
 * per rank -- at least a rank's largest send buffer, piece or image.  Blocks travel device ->
 * the sender's region -> device with the plans, offsets and ordering of the RCCL flavour; every
 * call is host-synchronous and a peer that does not arrive within 120 s fails the call instead of
 * hanging it.  Collective (the call returns when every rank has attached).  Not a performance
 * path: it exists so that the whole multi-process flow -- control plane, frame driver, bench --
 * runs where one GPU is all there is. */
int avr_comm_create_shared(const char *name, int rank, int n_ranks, size_t capacity_bytes,
                           avr_comm **out_comm);
void avr_comm_destroy(avr_comm *comm);
int avr_comm_rank(const avr_comm *comm);
int avr_comm_size(const avr_comm *comm);

/* Control plane of the communicator: a small host-side allgather among the ranks -- what the
 * reference does with MPI_Allgather on MPI_COMM_WORLD for its layer counts and depth hints
 * (DirectSend/Base/DirectSendBase.cpp:329-361).  Used for the agreement check of a new frame plan
 * (avr_frame_plan_agree) and for the decisions of the frame driver's co-run search, which all
 * ranks of a frame take together; never per frame.
 * avr_comm_set_control hands over the caller's own (the reference's host: MPI_Allgather of
 * bytes_per_rank bytes; bench.py: gloo): allgather(user, mine, all, bytes_per_rank) fills
 * all[rank * bytes_per_rank ...] for every rank and returns 0.  Without one, the RCCL flavour runs
 * a grouped round of tiny ncclSend / ncclRecv in band on the given context's stream (host-blocking,
 * with the deadline of avr_set_frame_timeout_ms); the rehearsal flavours meet in their own memory.
 * The caller's allgather runs under the same deadline: it is called on a helper thread over copies
 * of the buffers, and a round that outlasts the deadline returns AVR_ERR_RUNTIME (the renderer
 * that asked is failed for good) while the helper is left behind in the caller's collective -- so
 * a caller's allgather should still carry a timeout of its own (MPI: a communicator error
 * handler; gloo: the process group's timeout) to end that thread. */
#define AVR_CONTROL_MAX_BYTES 2048
typedef int (*avr_control_allgather_fn)(void *user, const void *mine, void *all, int bytes_per_rank);
int avr_comm_set_control(avr_comm *comm, avr_control_allgather_fn allgather, void *user);
/* Collective, host-blocking: bytes (a multiple of 4, <= AVR_CONTROL_MAX_BYTES) of every rank into
 * all[n_ranks * bytes].  ctx: the context whose stream carries the in-band round (RCCL flavour
 * without a caller's control plane); may be NULL otherwise. */
int avr_comm_control_allgather(avr_comm *comm, avr_context *ctx, const void *mine, void *all,
                               int bytes);
long avr_comm_control_rounds(const avr_comm *comm); /* control-plane calls so far (diagnostics) */
/* Do the ranks' plans of one frame describe ONE exchange?  Collective over the control plane: every
 * rank contributes a digest of the plan's replicated part (image, pieces, group order, runs, layer
 * order) mixed with settings_digest (whatever else the caller requires to be equal on all ranks)
 * and its send / receive block sizes; every rank then checks the whole matrix -- rank a's block
 * for rank b must be what b expects from a -- and so every rank reaches the same verdict:
 * AVR_OK, or AVR_ERR_RUNTIME (naming the first pair that disagrees) on ALL of them, before
 * anything has been queued.  A grouped ncclSend / ncclRecv round whose two sides disagree on a
 * size never ends; the reference finds out from the metadata message of every transfer
 * (Common/Image.cpp:62-90).  The frame driver calls it once per new plan (avr_renderer_set_plan_check). */
int avr_frame_plan_agree(const avr_frame_plan *plan, avr_comm *comm, avr_context *ctx,
                         uint64_t settings_digest);

/* The sparse all-to-all of one frame (layout: "frame plan" above): block for peer s of `send` goes
 * to rank s, `recv` receives the blocks of all ranks for this rank's piece.  Collective; on the
 * context's stream. */
int avr_exchange(avr_context *ctx, const avr_frame_plan *plan, avr_comm *comm, const float *send,
                 float *recv);
/* avr_exchange without the copy of the rank's block for itself: recv holds the peers' blocks
 * only, the rank's own stay in send (avr_fold_plan_own reads them there; send must then live until
 * that fold has run). */
int avr_exchange_peers(avr_context *ctx, const avr_frame_plan *plan, avr_comm *comm,
                       const float *send, float *recv);
/* Classic direct send of ONE image per rank -- DirectSendBase::compose(localImage, sendGroup,
 * recvGroup, comm) for a plain (not layered) image, DirectSend/Base/DirectSendBase.cpp:76-177,
 * 257-281: every rank's image (n_pixels * bytes_per_pixel bytes, device) is cut into the n pieces of
 * avr_piece_range; piece k goes to the rank at position k of group_order (NULL = rank order).
 * Received: `slices` = n blocks of this rank's own piece (its pixel count * bytes_per_pixel bytes
 * each), block j cut from the image of the rank at group position j -- the order in which
 * ProcessIncomingImages blends them (:179-255: the lower position on top).  One grouped ncclSend /
 * ncclRecv round on the context's stream; fold the blocks with avr_blend_regions. */
int avr_exchange_pieces(avr_context *ctx, avr_comm *comm, const int32_t *group_order,
                        int64_t n_pixels, int bytes_per_pixel, const void *image, void *slices);
/* ImageFull::Gather: every rank's piece (bytes_per_pixel bytes per pixel: 20 for the depth-sort
 * image, 3 for its RGB8 bytes) lands at its pixel range in `full` on rank `root` (full is ignored
 * elsewhere).  Collective; on the context's stream. */
int avr_gather(avr_context *ctx, const avr_frame_plan *plan, avr_comm *comm, const void *piece,
               int bytes_per_pixel, void *full, int root);
/* The same gather described without a plan: begin / end [n_ranks] = where each rank's piece sits in
 * `full`, in pixels (avr_frame_plan_piece_ranges of the frame the pieces belong to).  skip_own: the
 * root does not copy its own piece into `full` (avr_assemble_rows_own reads it where it is).
 * avr_exchange_peers_gather: avr_exchange_peers with such a gather -- of an EARLIER frame's pieces
 * -- riding in the same grouped ncclSend / ncclRecv round: the frame driver's compositing stream
 * then carries ONE RCCL launch per frame (the reference's Gather is a second collective after the
 * compositing, Common/ImageColorOnly.hpp:220-270; frames are independent, so frame f's bytes may
 * travel to the root while frame f + 1 is exchanged).  rider may be NULL. */
typedef struct {
  const void *piece;       /* this rank's piece (device) */
  int32_t bytes_per_pixel;
  int32_t root;
  void *full;              /* on the root: the gathered, piece-major buffer (device) */
  const int64_t *begin;    /* [n_ranks] */
  const int64_t *end;      /* [n_ranks] */
  int32_t skip_own;
} avr_gather_op;
int avr_frame_plan_piece_ranges(const avr_frame_plan *plan, int64_t *begin_out, int64_t *end_out);
int avr_gather_run(avr_context *ctx, avr_comm *comm, const avr_gather_op *op);
int avr_exchange_peers_gather(avr_context *ctx, const avr_frame_plan *plan, avr_comm *comm,
                              const float *send, float *recv, const avr_gather_op *rider);

/* ---- frame driver -------------------------------------------------------------------------- */

/* VolumeRenderer::RenderParameters (VolumeRenderer/VolumeRenderer.hpp:33-44), the fields the hot
 * path reads.  draw_bounds: the reference always blends the wireframe of the tight bounds over the
 * final image (VolumeRenderer.cpp:1311-1314); 0 leaves it out. */
typedef struct {
  int32_t width, height;
  float box_transparency;
  int32_t antialiasing;
  int32_t use_visibility_graph;
  int32_t draw_bounds;
  int32_t write_visibility_graph;
} avr_render_params;

/* One rank's share of VolumeRenderer::renderSingleTrial (VolumeRenderer.cpp:1103-1339) from the
 * per-box loop to the 8-bit image, pipelined over three HIP streams of its own: frame i+1 is
 * classified while frame i is marched and frame i-1 is exchanged, folded and gathered.
 *   all_boxes / owner   metadata of EVERY box of the scene (replicated on all ranks), level-major;
 *                       the boxes of `rank` must carry their cell pointers (HBM);
 *   comm                NULL for one rank.
 * Replaces the function-local static VolumePainter / DirectSendBase pair of
 * VolumeRenderer.cpp:909-927 together with the loop that drives them. */
typedef struct avr_renderer avr_renderer;
int avr_renderer_create(int device_id, int rank, int n_ranks, avr_comm *comm,
                        const avr_box *all_boxes, const int32_t *owner, int n_boxes,
                        const avr_scalar_transform *transform, const double bounds_min[3],
                        const double bounds_max[3], const float scalar_range[2],
                        const avr_colormap_point *colormap, int colormap_count,
                        avr_renderer **out_renderer);
void avr_renderer_destroy(avr_renderer *renderer);
/* march_workgroups_per_cu: -1 default (uncapped), 0..8 as avr_context_set_march_occupancy.  cache_classification: avr_scene_set_classification_cache. */
int avr_renderer_set_options(avr_renderer *renderer, int march_workgroups_per_cu,
                             int cache_classification);
/* geometry.scalarRange of the scene (VolumeRenderer.hpp:74-89) for the frames that follow. */
int avr_renderer_set_scalar_range(avr_renderer *renderer, const float scalar_range[2]);
/* avr_scene_invalidate for the renderer's scene: call after changing cell data in place while
 * cache_classification is on. */
int avr_renderer_invalidate(avr_renderer *renderer);
/* How a frame's two paint kernels are laid out on the streams:
 *   0  back to back on the march stream;
 *   1  side by side: the classify pass of frame i+1 on its own stream beside the march of frame i;
 *   2  paired: every frame's classify pass and march back to back on ONE stream, the even frames
 *      on stream M, the odd ones on a second stream -- the classify pass of frame i+1 still runs
 *      beside the march of frame i, but a march never queues behind its predecessor, so its tail
 *      runs beside whatever the other stream has next (a rank of eight of config-4: 0.165 ->
 *      0.145 ms; one rank: 0.986 -> 1.015 ms);
 *  -1  (default) whichever the driver measures to be fastest (avr_renderer_set_classify_share).
 * Never changes results. */
int avr_renderer_set_overlap(avr_renderer *renderer, int overlap_classify);
/* A frame re-uses the classified volume and the send buffer of the frame three before it.
 * 0: the frame's streams wait for that frame on the GPU (two wait packets; the host runs a few
 * frames ahead).  1: the HOST waits inside avr_renderer_render until that frame's march and
 * exchange / fold are through and queues no wait -- at most three frames in flight, and a
 * descriptor batch that repeats (same camera and parameters) leaves no packet at all on the
 * streams: for the short frames of a rank of several every packet between two kernels counts
 * (a rank of eight of config-4: 0.187 -> 0.164 ms).  -1 (default): 1 for more than one rank,
 * 0 for one rank (whose 1 ms frames hide the packets and lose 1-2 % to the shorter queue).
 * Never changes results. */
int avr_renderer_set_host_backpressure(avr_renderer *renderer, int mode);
/* How the classify pass and the march of this rank share the GPU is measured by the driver on
 * the running pipeline, unless fixed here and / or by avr_renderer_set_overlap: the candidates --
 * back to back on one stream, or side by side with an LDS reserve of 0, 2, 4 ... KiB per classify
 * workgroup (avr_context_set_classify_lds_reserve) -- are each held for a few frames whose
 * period is timed with HIP events on the march stream, and the best is kept (re-timed now and
 * then; searched again when it drifts).  bytes >= 0 fixes the reserve of the side-by-side mode,
 * -1 (default) leaves it to the driver.  Never changes results.
 * avr_renderer_corun_state: what the last frame used, whether the search has settled, and the
 * number of timed windows so far. */
int avr_renderer_set_classify_share(avr_renderer *renderer, int bytes);
int avr_renderer_corun_state(const avr_renderer *renderer, int *overlap_out,
                             int *reserve_bytes_out, int *settled_out, long *windows_out);
/* Whether the driver tightens the exchange layout (avr_frame_plan_tighten) of every plan it makes
 * (default 1; the 32 most recently used plans are kept, so a camera that repeats pays nothing).
 * The same on every rank.  Never changes results. */
int avr_renderer_set_tighten(avr_renderer *renderer, int enabled);
/* How the image is dealt to the ranks' pieces: AVR_PIECES_ROW_BANDS with bands of 8 rows by
 * default (the caller only sees the gathered image), AVR_PIECES_CONTIGUOUS for the reference's
 * partition.  band_rows: a power of two.  The same on every rank.  Never changes results. */
int avr_renderer_set_piece_layout(avr_renderer *renderer, int piece_layout, int band_rows);
int avr_renderer_reference_sample_distance(const avr_renderer *renderer, float *out);
/* Ranks of several, fail-fast and tuned as ONE system (round 4).
 * avr_renderer_set_plan_check (default 1): the first frame of a NEW plan (and the first after one
 *   of the setters the ranks must agree on) runs avr_frame_plan_agree over the communicator's
 *   control plane before anything is queued; ranks whose plans or settings differ all get
 *   AVR_ERR_RUNTIME there instead of hanging in the exchange.  0: no check (a caller whose camera
 *   never repeats and who vouches for its ranks).
 * avr_renderer_set_corun_coordination (default -1 = on for ranks of several): the co-run search
 *   (avr_renderer_set_classify_share) holds the same candidate on every rank in the same frames
 *   and decides on the MAXIMUM of the ranks' window periods (one 4-byte control-plane allgather
 *   per window, three frames after the window's last); the held candidate is re-timed every
 *   half second.  0: every rank searches on its own (round 3).  A run that wants no search at all
 *   fixes layout and reserve: avr_renderer_set_overlap + avr_renderer_set_classify_share.
 * avr_renderer_failure: NULL, or what did not finish within the deadline of
 *   avr_set_frame_timeout_ms (stream, rank, frame, stage, co-run state); the renderer is then
 *   failed: every call returns AVR_ERR_RUNTIME with that message, avr_renderer_destroy releases
 *   host memory only (device memory and streams of a hung GPU queue are left to process exit). */
/* avr_renderer_set_deferred_gather (default -1 = on for ranks of several; the same on every rank):
 *   the RGB8 pieces of frame f travel to rank 0 inside the grouped round of frame f + 1
 *   (avr_exchange_peers_gather) instead of in a gather round of their own -- ONE RCCL launch per
 *   frame.  Rank 0's rgb8_out of frame f is then complete when the compositing stream has passed
 *   the NEXT frame's round, or after avr_renderer_synchronize -- which, while such a gather is
 *   pending, runs it and is therefore collective (every rank synchronises after the same frame);
 *   the caller's rgb8_out buffer must stay valid until then.  Frames with want_image or
 *   antialiasing gather at once as before.  0: every frame gathers at once. */
int avr_renderer_set_deferred_gather(avr_renderer *renderer, int mode);
/* How a frame's two paint kernels are launched.  -1 / 1 (default): one launch each.  k in
 * [2, AVR_MAX_FRAME_CHUNKS]: every frame in k depth-ordered chunks, chunk i marched while chunk
 * i + 1 is classified (avr_classify_plan_chunked; only where the two kernels run on two streams,
 * never with a cached classification).  Built to shorten the frame of a caller who waits for every
 * frame -- the reference's call shape, one Render() per frame
 * (VolumeRenderer/VolumeRenderer.cpp:1103-1339) -- and measured not to on this GPU
 * (profiles/r5_latency/), hence off by default.  Scheduling only: never changes results. */
int avr_renderer_set_frame_chunks(avr_renderer *renderer, int chunks);
/* Occlusion culling (avr_render_plan_culled; one rank).  k in [2, AVR_MAX_FRAME_CHUNKS]: every
 * frame is classified and marched in k depth-ordered chunks on one stream, each classify launch
 * leaving out the boxes no ray can still sample (the reference's default boxTransparency = 0,
 * VolumeRenderer.hpp:36, saturates rays early).  -1 / 0 (default): never -- exact, but measured
 * not to pay on the configurations of BASELINE.json (profiles/r5_opaque/).  Never changes results. */
int avr_renderer_set_occlusion_culling(avr_renderer *renderer, int chunks);
int avr_renderer_last_frame_chunks(const avr_renderer *renderer); /* what the last frame took */
/* Visibility speculation (avr_classify_plan_positions / avr_march_plan_speculative; one rank).
 * -1 / 1 (default): the driver remembers, per box, the last frame whose rays sampled it (a march
 * that records the boxes it samples, the flags read on the host a few frames later); while at most
 * 85 % of the rank's boxes were sampled in the last 24 frames (and the classify work that saves is
 * worth 0.15 ms: not the short, march-bound frames of config-2 / config-3), a frame classifies only those, its
 * march checks every box it needs, and a repair pass (two gated launches that do nothing as a
 * rule) redoes the frame when the set was wrong -- the cells changed, the camera turned: results
 * never change.  The reference's default boxTransparency = 0: config-4's rays sample 58 of 176
 * boxes, the classify pass takes 0.18 instead of 0.55 ms, the frame 0.43 instead of 0.63.  A
 * translucent frame (every box sampled) is observed once and then left alone but for one observing
 * frame in 512; repairs in more than a quarter of the frames suspend it (64 frames, doubling).
 * 0: never.
 * avr_renderer_speculation_state: state -1 off / 0 observing / 1 waiting for an observation /
 * 2 speculating / 3 not worth it (asleep) / 4 suspended after repairs; frames speculated and
 * repaired so far; the fraction of boxes in the present set (-1: nothing observed yet). */
int avr_renderer_set_visibility_speculation(avr_renderer *renderer, int mode);
int avr_renderer_speculation_state(const avr_renderer *renderer, int *state,
                                   int64_t *speculative_frames, int64_t *repaired_frames,
                                   float *sampled_fraction);
int avr_renderer_set_plan_check(avr_renderer *renderer, int mode);
/* One rank (default -1 = on): instead of timing every candidate of the co-run search, the driver
 * reads off each frame's two kernel durations which of the two is the longer one side by side and
 * bisects the LDS reserve to where they take equally long -- where the frame is shortest, because
 * the two trade one resource linearly (profiles/r5_corun_gap/) -- then holds it: some 100 frames
 * instead of 640.  Frames shorter than 0.35 ms, a held candidate that drifts, and mode 0 take the
 * full search.  Scheduling only. */
int avr_renderer_set_corun_balance(avr_renderer *renderer, int mode);
int avr_renderer_set_corun_coordination(avr_renderer *renderer, int mode);
const char *avr_renderer_failure(const avr_renderer *renderer);
/* One frame, asynchronously.  group_order: rank order of the compositing group, NULL = from the
 * visibility graph (VolumeRenderer.cpp:1235-1241).  input_stream: a HIP stream whose queued work
 * produces the cell data (or zeroes samples_out); the classify pass, and with it the march, is
 * ordered after it.  NULL means "nothing to wait for", NOT the legacy default stream (whose handle
 * is 0 as well): the driver's streams are non-blocking and never order themselves after that one
 * implicitly, so a caller that fills cells on the default stream passes AVR_DEFAULT_STREAM.
 * samples_out (device, may be NULL) as avr_paint_box.  want_image (the SAME on every rank: it
 * adds a gather of the float pieces): also deliver the gathered -- with antialiasing downsampled
 * and overlaid -- depth-sort image.  On rank 0: rgb8_out (device, width*height*3 bytes, rows
 * top-down = the output file's pixel bytes, required) and, with want_image, image_out (device,
 * width*height*5 floats, origin bottom-left).  Other ranks pass NULL for both.  The outputs are
 * complete when the compositing stream (avr_renderer_stream(r, 2)) reaches this point:
 * avr_renderer_synchronize, or order your stream after it -- for ranks of several the RGB8 bytes
 * of a frame without want_image arrive one frame later (avr_renderer_set_deferred_gather). */
#define AVR_DEFAULT_STREAM ((void *)(intptr_t)-1) /* input_stream: the legacy default (null) stream */
int avr_renderer_render(avr_renderer *renderer, const avr_render_params *render,
                        const avr_camera *camera, const int32_t *group_order, void *input_stream,
                        uint64_t *samples_out, int want_image, uint8_t *rgb8_out, float *image_out);
/* Plans a frame ahead of time: makes the frame plan of (render, camera, group_order) -- visibility
 * order (VolumeRenderer.cpp:1235-1241), global layer order and exchange layout
 * (DirectSendBase.cpp:400-446), for N > 1 tightened to the runs' per-row extents -- and keeps it
 * with the renderer's plans, where avr_renderer_render finds it.  Host geometry only (0.1 ms for
 * 176 boxes at N = 8, which is most of what a rank's 0.15 ms frame leaves the host): a caller whose
 * camera never repeats (a scripted fly-through) calls it for frame f + 1 on ANOTHER thread while
 * frame f is being queued.  Safe beside avr_renderer_render / avr_renderer_synchronize of the same
 * renderer; not beside its setters or avr_renderer_destroy.  Calling it is never required and never
 * changes results. */
int avr_renderer_prepare(avr_renderer *renderer, const avr_render_params *render,
                         const avr_camera *camera, const int32_t *group_order);
int avr_renderer_synchronize(avr_renderer *renderer);
/* How many of the frames rendered so far have their outputs (rgb8_out, image_out) written once the
 * compositing stream (avr_renderer_stream(r, 2)) has passed everything queued up to now: all of
 * them, or -- ranks of several with the deferred gather -- all but the last.  A caller that orders
 * its own stream after the compositing stream reads frame `*complete_out - 1` and older; the last
 * frame's bytes follow with the next frame's round or with avr_renderer_synchronize.  frames_out
 * (may be NULL): frames rendered so far. */
int avr_renderer_outputs_complete(const avr_renderer *renderer, uint64_t *complete_out,
                                  uint64_t *frames_out);
/* which: 0 classify, 1 march, 2 exchange / fold / gather / tail (hipStream_t as void*). */
void *avr_renderer_stream(avr_renderer *renderer, int which);
/* The plan of the last frame rendered (runs, piece, exchange volume). */
int avr_renderer_plan_info(const avr_renderer *renderer, avr_frame_plan_info *out);
/* Kernel timing for the benchmark: while enabled every frame records HIP events around its
 * classify pass and its march on the streams they are launched on.  avr_renderer_timings drains
 * the streams and returns the averages per frame since timing was enabled: each kernel's own
 * duration and the length of the union of their execution intervals (they overlap by design). */
/* Host seconds spent inside avr_renderer_render since the last reset, by section: plan,
 * classify call, march call, exchange, fold (+ piece overlay), gather + frame tail; and the number
 * of frames they cover.  (A host that runs ahead of the GPU waits inside these calls for a staging
 * slot: under back-pressure the sum approaches the GPU's frame time.) */
int avr_renderer_host_profile(avr_renderer *renderer, double seconds_out[6], long *frames_out,
                              int reset);
int avr_renderer_set_timing(avr_renderer *renderer, int enabled);
int avr_renderer_timings(avr_renderer *renderer, double *classify_ms, double *march_ms,
                         double *busy_ms, int *frames);

/* ---- visibility ordering ------------------------------------------------ */

/* BuildVisibilityOrderedGroup (Common/VisibilityOrdering.cpp:63-632; called from
 * VolumeRenderer/VolumeRenderer.cpp:1235-1241): the order of the ranks in the MPI group handed to
 * Compositor::compose, from a topological sort of the face-adjacency graph of all boxes oriented
 * by the view direction (cycles are broken by splitting a box).  Host only.
 *
 * The graph object holds the replicated box metadata (what the reference allgathers every
 * frame: corners as floats and owners, rank-major) and the camera-independent list of
 * face-adjacent pairs.  owner[b] = owning rank of all_boxes[b]; the relative order of a rank's
 * boxes is their localBoxes order. */
typedef struct avr_visibility_graph avr_visibility_graph;
int avr_visibility_graph_create(const avr_box *all_boxes, const int32_t *owner, int n_boxes,
                                int n_ranks, avr_visibility_graph **out_graph);
void avr_visibility_graph_destroy(avr_visibility_graph *graph);

/* rank_order_out[n_ranks] = rank at each position of the ordered group.
 * use_visibility_graph == 0 or a failed ordering yields the default order 0..n-1
 * (*succeeded_out = 0 on failure, where the reference prints its warning).  aspect = width /
 * height as float (VolumeRenderer.cpp:1114).  dot_prefix (may be NULL): every graph iteration is
 * also written to "<dot_prefix><counter>.dot" in the reference's format
 * (writeVisibilityGraph, VisibilityOrdering.cpp:318-350; the reference's prefix is
 * "visibility_graph_", written by rank 0).  n_splits_out (may be NULL): boxes split to break
 * cycles. */
int avr_visibility_order(avr_visibility_graph *graph, const avr_camera *camera, float aspect,
                         int use_visibility_graph, const char *dot_prefix,
                         int32_t *rank_order_out, int *succeeded_out, int *n_splits_out);

/* ---- wireframe overlay ------------------------------------------------- */

/* computeTightBounds (VolumeRenderer/VolumeRenderer.cpp:791-848) over the (replicated) box
 * metadata: component-wise min / max of the corners, rounded to float as the reference's
 * MPI_FLOAT reduction does; the fallback bounds when there is no box.  Host only. */
int avr_tight_bounds(const avr_box *all_boxes, int n_boxes, const double fallback_min[3],
                     const double fallback_max[3], double out_min[3], double out_max[3]);

/* renderBoundingBoxLayer (VolumeRenderer/VolumeRenderer.cpp:139-335): white anti-aliased
 * wireframe of [bounds_min, bounds_max] blended, in the reference's edge order, over the pixels
 * [pixel_begin, pixel_end) of a width x height depth-sort image (image = those pixels, 5 floats
 * each, in place).  Pixels are independent, so a rank can overlay its own piece.  If rgb8 is
 * non-NULL the overlaid pixels are also written as RGB8 (3 bytes per pixel, same order). */
int avr_bbox_overlay(avr_context *ctx, const double bounds_min[3], const double bounds_max[3],
                     const avr_camera *camera, int sqrt_antialiasing, int width, int height,
                     int64_t pixel_begin, int64_t pixel_end, float *image, uint8_t *rgb8);
/* The same on this rank's piece of a frame plan (either piece layout): piece = the piece's pixels
 * as avr_fold_plan delivers them, rgb8 (may be NULL) its 8-bit twin. */
int avr_bbox_overlay_piece(avr_context *ctx, const avr_frame_plan *plan, const double bounds_min[3],
                           const double bounds_max[3], const avr_camera *camera, float *piece,
                           uint8_t *rgb8);
/* Gathered pieces (piece-major: piece 0's pixels, then piece 1's, ... as avr_gather delivers them)
 * -> the image in row order; flip != 0: rows top-down (the output file's order; pieces hold the
 * bottom-up image of the renderer).  With contiguous pieces the gathered buffer already is the
 * image, so this is a (flipped) copy -- avr_flip_rows generalised to AVR_PIECES_ROW_BANDS. */
int avr_assemble_rows(avr_context *ctx, const avr_frame_plan *plan, const void *gathered,
                      int bytes_per_pixel, int flip, void *image);
/* The same with this rank's own piece read where its fold wrote it (own_piece, device) instead of
 * from the gathered buffer: the root of a gather with skip_own saves a device copy per frame.  With
 * contiguous pieces only if the pieces are whole rows. */
int avr_assemble_rows_own(avr_context *ctx, const avr_frame_plan *plan, const void *gathered,
                          const void *own_piece, int bytes_per_pixel, int flip, void *image);

/* ---- scene statistics -------------------------------------------------- */

/* reduceLocalScalarStats (VolumeRenderer/SceneBuilder.cpp:53-97) over the scene's boxes: one
 * streaming pass.  stats_host[0..2] = min, max, min positive over the finite cells (+inf, -inf,
 * +inf when there is none), *finite_count_host = their number.  Synchronises the stream. */
int avr_scene_scalar_stats(avr_context *ctx, const avr_scene *scene, double stats_host[3],
                           int64_t *finite_count_host);

/* The scalar-transform part of BuildSceneGeometry (SceneBuilder.cpp:315-443) from statistics
 * already reduced over all ranks (host only): positive floor / processed range for log
 * scaling, normalisation to the data range (SetSceneNormalizationRange, :427-443).
 * processed[2] = processed min/max in double, processed_range / scalar_range as the float
 * pairs of SceneGeometry.  AVR_ERR_RUNTIME where the reference throws std::runtime_error. */
int avr_scene_transform_from_stats(const double stats[3], int64_t finite_count, int log_scale,
                                   int normalize_to_data_range, avr_scalar_transform *transform,
                                   double processed[2], float processed_range[2],
                                   float scalar_range[2]);

/* The binning of ComputeSceneHistogram (SceneBuilder.cpp:445-577): every cell of the scene is
 * transformed, clamped to [range_min, range_max] and counted; counts_dev[bin_count] (device,
 * uint64) is ADDED to (the caller zeroes it and sums over ranks). */
int avr_scene_histogram(avr_context *ctx, const avr_scene *scene,
                        const avr_scalar_transform *transform, float range_min, float range_max,
                        int bin_count, uint64_t *counts_dev);

#ifdef __cplusplus
}
#endif
#endif /* AVR_HIP_H */
