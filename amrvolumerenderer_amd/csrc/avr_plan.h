// Frame plan (include/avr_hip.h, "frame plan"): global layer order, runs, per-run screen
// rectangles and the sparse exchange layout.  Host only.
#ifndef AVR_PLAN_H
#define AVR_PLAN_H

#include <vector>

#include "avr_internal.h"

struct avr_frame_plan {
  avr_frame_plan_info info{};
  avr_paint_params params{};
  std::vector<avr_colormap_point> colormap;
  avr_camera camera{};
  std::vector<int32_t> layer_box;        // global order -> index into all_boxes
  std::vector<avr_run_info> runs;        // global runs in order
  std::vector<int32_t> local_order;      // this rank's local box indices in global order
  std::vector<int32_t> local_run_end;
  std::vector<int32_t> group_order;      // rank at group position k
  std::vector<int32_t> piece_of_rank;
  std::vector<int64_t> send_splits, recv_splits;
  // sender tables: [local run][piece]
  std::vector<avr::RunRectDev> local_rects;
  std::vector<avr::RunBlockDev> send_blocks;
  std::vector<int32_t> send_block_rows;  // rows of each block (0 = empty)
  // receiver tables: [global run], this rank's piece
  std::vector<avr::RunRectDev> global_rects;
  std::vector<avr::RunBlockDev> recv_blocks;
  std::vector<int32_t> recv_block_rows;
  // tightened layout (avr_frame_plan_tighten): the rows of the blocks above, sender and receiver;
  // they travel to the device with the descriptors of every call that needs them
  std::vector<avr::RunSpanDev> send_spans, recv_spans;
  bool tightened = false;
  avr::PieceMapDev pieces;  // how the image's pixels are dealt to the ranks' pieces
  bool from_boxes = false;  // built by build_frame_plan (layers = boxes): may be tightened
  std::vector<avr::BoxFootprint> footprints;  // of all boxes on screen (from_boxes)
  // host prologue of this frame's local boxes, filled by the first device call that needs it
  // (avr_classify_plan) and re-used by the next (avr_march_plan)
  avr::FramePlan prologue;
  // the settings epoch of the frame driver under which the ranks agreed on this plan
  // (avr_frame_plan_agree; 0 = not yet)
  mutable uint64_t agreed_epoch = 0;
  mutable uint64_t agreed_digest = 0;  // of the plan's replicated part, as agreed on
};

namespace avr {

// Dense single-buffer layout of avr_render_runs expressed with the same tables: every run's
// rectangle is the full screen.
void dense_run_tables(int width, int height, int n_runs, int n_pieces,
                      std::vector<RunRectDev>* rects, std::vector<RunBlockDev>* blocks);

// A plan for layers known only by (depth hint, owner[, screen rectangle]) -- the generic
// Compositor::compose case; rects == nullptr: every layer covers the whole image.
void build_layer_plan(int n_layers, const float* hints, const int32_t* owner,
                      const int32_t (*rects)[4], int n_ranks, int rank, const int32_t* group_order,
                      int width, int height, int piece_layout, int band_rows, avr_frame_plan* plan);

// Replaces the rectangular blocks of a frame plan by per-row spans (conservative extent of each
// run's boxes on screen) and recomputes the exchange layout; every rank must do the same.
void tighten_frame_plan(const avr_box* all_boxes, int n_boxes, avr_frame_plan* plan);

void build_frame_plan(const avr_box* all_boxes, const int32_t* owner, int n_boxes, int n_ranks,
                      int rank, const int32_t* group_order, const avr_paint_params& params,
                      const avr_camera& camera, int piece_layout, int band_rows,
                      avr_frame_plan* plan);

PieceMapDev make_piece_map(int layout, int band_rows, int n_pieces, int width, int height);
// Where piece k sits in the gathered, piece-major image: pixels [begin, end).  With contiguous
// pieces this IS the image's pixel range (getPieceRange, DirectSendBase.cpp:59-74); with row
// bands the pieces' rows follow each other piece by piece and avr_assemble_rows puts them back.
void piece_pixel_range(const PieceMapDev& map, int k, int64_t* begin, int64_t* end);

}  // namespace avr

#endif
