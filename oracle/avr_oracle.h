/*
 * avr_oracle.h -- CPU restatement of the amrVolumeRenderer hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the checker, never the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product path (amrvolumerenderer_amd/) never links or calls it.
 *
 * Every function cites the reference file:line (relative to /root/reference) whose arithmetic it
 * restates.  All "float" arithmetic is IEEE binary32 evaluated in source order with no FMA
 * contraction (build with -ffp-contract=off, no -march=native, no -ffast-math).
 *
 * Pin status (details in oracle/README.md, "Pinning"):
 *   pinned by the reference's own fixtures (Common/Testing/ImageFullTest.cpp): the float and
 *     ubyte over-blends, orc_blend_regions;
 *   pinned by properties measured on the reference: orc_compose_layered,
 *     orc_layer_order, orc_piece_range;
 *   PARITY UNPINNED (the reference holds no test or vector): orc_paint_box,
 *     orc_build_color_table, orc_box_sampling, orc_blend_depthsort, orc_box_depth_hint,
 *     orc_reference_sample_distance, orc_downsample, orc_quantize_rgb8, and the SURVEY 8(f)
 *     functions (orc_scalar_stats, orc_scene_transform, orc_histogram, orc_tight_bounds,
 *     orc_bbox_overlay).
 */
#ifndef AVR_ORACLE_H
#define AVR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* volume::AmrBox (Common/VolumeTypes.hpp:69-76) flattened.  `cells` points at the cell
 * validBox.smallEnd() of the chosen component; x is fastest, then jstride, kstride (Array4). */
typedef struct {
  double min_corner[3];
  double max_corner[3];
  int dims[3];
  const double *cells;
  int64_t jstride;
  int64_t kstride;
} orc_box;

/* volume::ScalarTransform (Common/VolumeTypes.hpp:21-31), fields the path reads. */
typedef struct {
  int log_scale_input;
  int normalize_to_unit_range;
  double positive_floor;
  double normalization_min;
  double inverse_normalization_span;
} orc_transform;

/* volume::CameraParameters (Common/VolumeTypes.hpp:83-90). */
typedef struct {
  double eye[3];
  double look_at[3];
  double up[3];
  float fov_y_degrees;
  float near_plane;
  float far_plane;
} orc_camera;

/* volume::ColorMapControlPoint (Common/VolumeTypes.hpp:92-98). */
typedef struct {
  float value, red, green, blue, alpha;
} orc_colormap_point;

/* Arguments of VolumePainter::paint (Common/VolumePainter.hpp:19-31) that the path reads. */
typedef struct {
  int width;
  int height;
  float scalar_range[2];
  float box_transparency;
  float reference_sample_distance;
  double bounds_min[3]; /* VolumeBounds: only the degenerate-spacing fallback reads it */
  double bounds_max[3];
  const orc_colormap_point *colormap; /* NULL/0 -> default jet */
  int colormap_count;
} orc_paint_params;

/* buildColorTable (Common/VolumePainter.cpp:442-516): 256 x RGBA. */
void orc_build_color_table(float alpha_scale, float normalization_factor,
                           const float scalar_range[2],
                           const orc_colormap_point *colormap, int colormap_count,
                           float out_table[1024]);

/* Host prologue quantities of VolumePainter::paint (Common/VolumePainter.cpp:571-613). */
void orc_box_sampling(const orc_box *box, const orc_paint_params *params,
                      float *sample_distance, float *normalization_factor,
                      float *alpha_scale);

/* VolumePainter::paint (Common/VolumePainter.cpp:548-961): writes width*height*5 floats
 * (r,g,b,a,depth; the ImageRGBAFloatColorDepthSort buffer).  Returns the number of executed
 * cell fetches (iterations reaching VolumePainter.cpp:870).  threads>1 splits rows over
 * OpenMP threads (results are per-pixel independent, so bits do not change). */
/* Search tool only (tools/appb_hash_search.py): variants of what a header shim might have
 * computed differently from AMReX; 0 (the default) is the restatement of the reference. */
void orc_set_shim_variant(int variant);

uint64_t orc_paint_box(const orc_box *box, const orc_transform *transform,
                       const orc_paint_params *params, const orc_camera *camera,
                       float *out_rgbad, int threads);
/* The same for the pixel window [x0, x1) x [y0, y1) only (rays of the full image). */
uint64_t orc_paint_box_window(const orc_box *box, const orc_transform *transform,
                              const orc_paint_params *params, const orc_camera *camera, int x0,
                              int y0, int x1, int y1, float *out_rgbad, int threads);

/* computeBoxDepthHint (VolumeRenderer/VolumeRenderer.cpp:541-553). */
float orc_box_depth_hint(const orc_box *box, const orc_camera *camera);

/* referenceSampleDistance (VolumeRenderer/VolumeRenderer.cpp:1138-1190), single process:
 * the MPI_Allreduce(MAX) is the max over all boxes given here. */
float orc_reference_sample_distance(const orc_box *boxes, int n_boxes,
                                    const double bounds_min[3], const double bounds_max[3]);

/* applyScalarTransform (Common/VolumeTypes.hpp:33-67). */
float orc_apply_scalar_transform(double raw, const orc_transform *transform);

/* Features::blend of the three image types, n pixels, out may not alias inputs.
 * ImageRGBAFloatColorDepthSort.hpp:13-27, ImageRGBAFloatColorOnly.hpp:19-26,
 * ImageRGBAUByteColorOnly.hpp:19-34. */
void orc_blend_depthsort(const float *top, const float *bottom, float *out, int64_t n);
void orc_blend_rgba_f32(const float *top, const float *bottom, float *out, int64_t n);
void orc_blend_rgba_u8(const uint32_t *top, const uint32_t *bottom, uint32_t *out, int64_t n);

/* ImageColorOnly<F>::blend with regions (Common/ImageColorOnly.hpp:119-199): images cover
 * pixel index ranges [tb,te) and [bb,be); out covers [min(tb,bb), max(te,be)).
 * kind: 0 depthsort (5 floats), 1 rgba f32 (4 floats), 2 rgba u8 (1 uint32). */
void orc_blend_regions(int kind, const void *top, int tb, int te, const void *bottom, int bb,
                       int be, void *out);

/* Color::GetComponentAsByte encode / SetComponentFromByte decode (Common/Color.hpp:36-91,
 * ImageRGBAUByteColorOnly.cpp:16-39). */
void orc_encode_rgba_u8(const float *rgba, uint32_t *out, int64_t n);
void orc_decode_rgba_u8(const uint32_t *in, float *rgba, int64_t n);

/* getPieceRange (DirectSend/Base/DirectSendBase.cpp:59-74). */
void orc_piece_range(int image_size, int piece_index, int num_pieces, int *begin, int *end);

/* DirectSendBase::composeLayered + Gather (DirectSend/Base/DirectSendBase.cpp:316-458,
 * Common/ImageColorOnly.hpp:220-270) simulated for n_ranks ranks in one process.
 *   layers[l]      : n_pixels*5 floats, layer l (any rank)
 *   hints[l]       : depth hint, owner[l]: owning rank, local_index[l]: index on that rank
 *   group_order[k] : rank at position k of the ordered MPI group (NULL = identity)
 *   fold_variant   : 0 = blend incoming images left-to-right in group order,
 *                    1 = right-to-left (another legal arrival schedule of
 *                        ProcessIncomingImages, DirectSendBase.cpp:179-255)
 * Writes the gathered image (n_pixels*5) and, if piece_owner != NULL, for each pixel the rank
 * that held it after compose.  Returns the number of runs. */
int orc_compose_layered(const float *const *layers, const float *hints, const int *owner,
                        const int *local_index, int n_layers, int n_ranks, int n_pixels,
                        const int *group_order, int fold_variant, float *out_gathered,
                        int *piece_owner);

/* The global layer order and run grouping alone (DirectSendBase.cpp:363-410):
 * order_out[n_layers] = layer ids sorted by (hint, owner, local_index);
 * run_end_out[r] = one-past-last position of run r in order_out.  Returns #runs. */
int orc_layer_order(const float *hints, const int *owner, const int *local_index, int n_layers,
                    int *order_out, int *run_end_out);

/* downsampleImage (VolumeRenderer/VolumeRenderer.cpp:479-528). src is (w*b)x(h*b)x5. */
void orc_downsample(const float *src, int target_w, int target_h, int block, float *dst);

/* SavePPM pixel bytes (Common/SavePPM.cpp:17-36, Common/Color.hpp:66-91): RGB8, rows
 * written top-down (y = h-1 .. 0); src is w*h pixels with `stride` floats each. */
void orc_quantize_rgb8(const float *src, int w, int h, int stride, uint8_t *dst);

/* ---- scene scalar statistics and histogram ---------------------------------- */

/* reduceLocalScalarStats (VolumeRenderer/SceneBuilder.cpp:53-97) over a list of boxes:
 * stats[0] = min, stats[1] = max, stats[2] = min positive over the finite cells
 * (+inf / -inf / +inf when there is none); returns the number of finite cells. */
int64_t orc_scalar_stats(const orc_box *boxes, int n_boxes, double stats[3]);

/* The scalar-transform part of BuildSceneGeometry (SceneBuilder.cpp:315-443) from the already
 * reduced statistics.  normalize_to_data_range as SceneBuildOptions; processed_range and
 * scalar_range receive the float pairs.  Returns 0, or 1 when log scaling finds no positive
 * value, 2 when the range is not finite (the reference throws std::runtime_error). */
int orc_scene_transform(const double stats[3], int64_t finite_count, int log_scale,
                        int normalize_to_data_range, orc_transform *transform,
                        double *processed_min, double *processed_max, float processed_range[2],
                        float scalar_range[2]);

/* ComputeSceneHistogram's per-cell binning (SceneBuilder.cpp:495-532). */
void orc_histogram(const orc_box *boxes, int n_boxes, const orc_transform *transform,
                   float range_min, float range_max, int bin_count, uint64_t *counts);

/* ---- wireframe overlay of the tight bounds ---------------------------------- */

/* computeTightBounds (VolumeRenderer/VolumeRenderer.cpp:791-848): component-wise min / max of
 * the box corners, reduced in float; `fallback` when there is no box. */
void orc_tight_bounds(const orc_box *boxes, int n_boxes, const double fallback_min[3],
                      const double fallback_max[3], double out_min[3], double out_max[3]);

/* renderBoundingBoxLayer (VolumeRenderer/VolumeRenderer.cpp:139-335): white anti-aliased
 * wireframe of the box [bounds_min, bounds_max] blended over the w x h depth-sort image in
 * place (12 edges in the reference's order, depth := lowest float). */
void orc_bbox_overlay(const double bounds_min[3], const double bounds_max[3],
                      const orc_camera *camera, int sqrt_antialiasing, float *image, int w,
                      int h);

/* 64-bit FNV-1a over a byte buffer (used to compare with hashes recorded in SURVEY.md). */
uint64_t orc_fnv1a64(const void *data, uint64_t n_bytes);

#ifdef __cplusplus
}
#endif
#endif
