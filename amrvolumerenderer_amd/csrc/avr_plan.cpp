// Host planning of one frame's layered DirectSend compositing (no device work).
//
// Reference: DirectSend/Base/DirectSendBase.cpp:329-410 (allgather of layer counts and depth
// hints, global sort by (hint, owner, local index), grouping into same-owner runs), :59-74
// (pixel pieces), :76-130 (group position k receives piece k).
#include "avr_plan.h"

#include <algorithm>
#include <array>
#include <stdexcept>

namespace avr {

namespace {

struct PieceRows {
  int64_t begin = 0, end = 0;  // pixel range
  int32_t first_row = 0, last_row = -1;
};

PieceRows piece_rows(int64_t n_pixels, int piece, int n_pieces, int width) {
  PieceRows rows;
  const int64_t size = n_pixels / n_pieces;  // getPieceRange, DirectSendBase.cpp:59-74
  rows.begin = size * piece;
  rows.end = (piece < n_pieces - 1) ? rows.begin + size : n_pixels;
  if (rows.end > rows.begin) {
    rows.first_row = static_cast<int32_t>(rows.begin / width);
    rows.last_row = static_cast<int32_t>((rows.end - 1) / width);
  }
  return rows;
}

// rows of `rect` that fall into the piece: [first, first + count)
void block_rows(const RunRectDev& rect, const PieceRows& piece, int32_t* first, int32_t* count) {
  *first = 0;
  *count = 0;
  if (rect.x1 < rect.x0 || rect.y1 < rect.y0 || piece.last_row < piece.first_row) return;
  const int32_t lo = std::max(rect.y0, piece.first_row);
  const int32_t hi = std::min(rect.y1, piece.last_row);
  if (hi < lo) return;
  *first = lo;
  *count = hi - lo + 1;
}

RunRectDev empty_rect() { return RunRectDev{0, 0, -1, -1}; }

void grow(RunRectDev* into, const int32_t rect[4]) {
  if (rect[2] < rect[0] || rect[3] < rect[1]) return;
  if (into->x1 < into->x0) {
    *into = RunRectDev{rect[0], rect[1], rect[2], rect[3]};
    return;
  }
  into->x0 = std::min(into->x0, rect[0]);
  into->y0 = std::min(into->y0, rect[1]);
  into->x1 = std::max(into->x1, rect[2]);
  into->y1 = std::max(into->y1, rect[3]);
}

}  // namespace

void dense_run_tables(int width, int height, int n_runs, int n_pieces,
                      std::vector<RunRectDev>* rects, std::vector<RunBlockDev>* blocks) {
  const int64_t n_pixels = static_cast<int64_t>(width) * height;
  rects->assign(static_cast<size_t>(n_runs), RunRectDev{0, 0, width - 1, height - 1});
  blocks->assign(static_cast<size_t>(n_runs) * n_pieces, RunBlockDev{0, 0, -1});
  for (int k = 0; k < n_pieces; ++k) {
    const PieceRows piece = piece_rows(n_pixels, k, n_pieces, width);
    const int64_t len = piece.end - piece.begin;
    for (int r = 0; r < n_runs; ++r) {
      // out[5*n_runs*begin + (r*len + (p - begin))*5]  ==  offset + (p - first_row*width)*5
      RunBlockDev& block = (*blocks)[static_cast<size_t>(r) * n_pieces + k];
      block.first_row = piece.first_row;
      block.offset = 5 * static_cast<int64_t>(n_runs) * piece.begin +
                     (static_cast<int64_t>(r) * len - piece.begin +
                      static_cast<int64_t>(piece.first_row) * width) * 5;
    }
  }
}

// The plan of a set of layers described only by (depth hint, owner, screen rectangle): global
// order, runs, exchange layout.  layer l's local index is its position among its owner's layers.
void build_layer_plan(int n_layers, const float* hints, const int32_t* owner,
                      const int32_t (*rects)[4], int n_ranks, int rank, const int32_t* group_order,
                      int width, int height, avr_frame_plan* plan) {
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) throw std::invalid_argument("invalid rank");
  if (n_layers < 0 || (n_layers > 0 && (hints == nullptr || owner == nullptr))) {
    throw std::invalid_argument("invalid layer list");
  }
  if (width <= 0 || height <= 0) {
    throw std::invalid_argument("image width and height must be positive");
  }
  const int64_t n_pixels = static_cast<int64_t>(width) * height;
  if (n_pixels > (int64_t{1} << 31) - 1) throw std::invalid_argument("image too large");
  const int n_boxes = n_layers;

  // group order / pieces
  plan->group_order.resize(static_cast<size_t>(n_ranks));
  plan->piece_of_rank.assign(static_cast<size_t>(n_ranks), -1);
  for (int k = 0; k < n_ranks; ++k) {
    const int32_t member = group_order ? group_order[k] : k;
    if (member < 0 || member >= n_ranks || plan->piece_of_rank[static_cast<size_t>(member)] >= 0) {
      throw std::invalid_argument("group_order must be a permutation of the ranks");
    }
    plan->group_order[static_cast<size_t>(k)] = member;
    plan->piece_of_rank[static_cast<size_t>(member)] = k;
  }

  // what the reference all-gathers: per layer (depth hint, owner, local index)
  std::vector<int32_t> local_index(static_cast<size_t>(n_boxes));
  std::vector<int32_t> boxes_of_rank(static_cast<size_t>(n_ranks), 0);
  for (int b = 0; b < n_boxes; ++b) {
    if (owner[b] < 0 || owner[b] >= n_ranks) throw std::invalid_argument("box owner out of range");
    local_index[static_cast<size_t>(b)] = boxes_of_rank[static_cast<size_t>(owner[b])]++;
  }
  plan->layer_box.assign(static_cast<size_t>(n_boxes), 0);
  std::vector<int32_t> run_end(static_cast<size_t>(std::max(n_boxes, 1)));
  const int n_runs = (n_boxes > 0)
                         ? layer_order(hints, owner, local_index.data(), n_boxes,
                                       plan->layer_box.data(), run_end.data())
                         : 0;

  // runs with their screen rectangles
  plan->runs.assign(static_cast<size_t>(n_runs), avr_run_info{});
  plan->global_rects.assign(static_cast<size_t>(n_runs), empty_rect());
  plan->local_order.clear();
  plan->local_run_end.clear();
  plan->local_rects.clear();
  std::vector<int32_t> runs_of_rank(static_cast<size_t>(n_ranks), 0);
  const int32_t full_rect[4] = {0, 0, width - 1, height - 1};
  int start = 0;
  for (int g = 0; g < n_runs; ++g) {
    const int end = run_end[static_cast<size_t>(g)];
    avr_run_info& run = plan->runs[static_cast<size_t>(g)];
    run.owner = owner[plan->layer_box[static_cast<size_t>(start)]];
    run.local_run = runs_of_rank[static_cast<size_t>(run.owner)]++;
    run.first_layer = start;
    run.n_layers = end - start;
    RunRectDev rect = empty_rect();
    for (int l = start; l < end; ++l) {
      const int32_t box = plan->layer_box[static_cast<size_t>(l)];
      grow(&rect, rects != nullptr ? rects[box] : full_rect);
      if (run.owner == rank) plan->local_order.push_back(local_index[static_cast<size_t>(box)]);
    }
    run.rect[0] = rect.x0;
    run.rect[1] = rect.y0;
    run.rect[2] = rect.x1;
    run.rect[3] = rect.y1;
    plan->global_rects[static_cast<size_t>(g)] = rect;
    if (run.owner == rank) {
      plan->local_run_end.push_back(static_cast<int32_t>(plan->local_order.size()));
      plan->local_rects.push_back(rect);
    }
    start = end;
  }
  const int n_local_runs = static_cast<int>(plan->local_rects.size());

  // sender layout: for peer s: for local run r: block(piece of s, r)
  plan->send_blocks.assign(static_cast<size_t>(n_local_runs) * n_ranks, RunBlockDev{0, 0, -1});
  plan->send_block_rows.assign(static_cast<size_t>(n_local_runs) * n_ranks, 0);
  plan->send_splits.assign(static_cast<size_t>(n_ranks), 0);
  int64_t cursor = 0;
  for (int peer = 0; peer < n_ranks; ++peer) {
    const int piece = plan->piece_of_rank[static_cast<size_t>(peer)];
    const PieceRows rows = piece_rows(n_pixels, piece, n_ranks, width);
    const int64_t before = cursor;
    for (int r = 0; r < n_local_runs; ++r) {
      const RunRectDev& rect = plan->local_rects[static_cast<size_t>(r)];
      int32_t first = 0, count = 0;
      block_rows(rect, rows, &first, &count);
      const size_t at = static_cast<size_t>(r) * n_ranks + static_cast<size_t>(piece);
      plan->send_blocks[at].offset = cursor;
      plan->send_blocks[at].first_row = first;
      plan->send_block_rows[at] = count;
      cursor += static_cast<int64_t>(count) * (rect.x1 - rect.x0 + 1) * 5;
    }
    plan->send_splits[static_cast<size_t>(peer)] = cursor - before;
  }
  const int64_t send_floats = cursor;

  // receiver layout: for source s: for run r of s: block(my piece, r)
  const int my_piece = plan->piece_of_rank[static_cast<size_t>(rank)];
  const PieceRows my_rows = piece_rows(n_pixels, my_piece, n_ranks, width);
  plan->recv_blocks.assign(static_cast<size_t>(n_runs), RunBlockDev{0, 0, -1});
  plan->recv_block_rows.assign(static_cast<size_t>(n_runs), 0);
  plan->recv_splits.assign(static_cast<size_t>(n_ranks), 0);
  cursor = 0;
  for (int source = 0; source < n_ranks; ++source) {
    const int64_t before = cursor;
    for (int g = 0; g < n_runs; ++g) {  // runs of one owner appear in its local run order
      if (plan->runs[static_cast<size_t>(g)].owner != source) continue;
      const RunRectDev& rect = plan->global_rects[static_cast<size_t>(g)];
      int32_t first = 0, count = 0;
      block_rows(rect, my_rows, &first, &count);
      plan->recv_blocks[static_cast<size_t>(g)].offset = cursor;
      plan->recv_blocks[static_cast<size_t>(g)].first_row = first;
      plan->recv_block_rows[static_cast<size_t>(g)] = count;
      cursor += static_cast<int64_t>(count) * (rect.x1 - rect.x0 + 1) * 5;
    }
    plan->recv_splits[static_cast<size_t>(source)] = cursor - before;
  }

  avr_frame_plan_info& info = plan->info;
  info.n_ranks = n_ranks;
  info.rank = rank;
  info.n_runs_total = n_runs;
  info.n_local_runs = n_local_runs;
  info.n_local_boxes = boxes_of_rank[static_cast<size_t>(rank)];
  info.n_pixels = n_pixels;
  info.piece_begin = my_rows.begin;
  info.piece_end = my_rows.end;
  info.send_floats = send_floats;
  info.recv_floats = cursor;
}

// The exchange volume of a frame plan is the area of the runs' screen RECTANGLES; measured on
// config-4 only 48 % (2 ranks) to 72 % (8 ranks) of those pixels carry anything
// (tools/send_occupancy.py).  Per row of a run, only the union of its boxes' conservative
// extents (box_row_spans) is kept: what lies outside is the cleared layer pixel, the exact
// identity of the depth-sort blend, on the sender and on the receiver alike.  Block sizes still
// follow from replicated metadata alone, so no sizes travel.
void tighten_frame_plan(const avr_box* all_boxes, int n_boxes, avr_frame_plan* plan) {
  if (plan->tightened) return;
  if (!plan->from_boxes) throw std::invalid_argument("only a frame plan made from boxes can be tightened");
  const int width = plan->params.width, height = plan->params.height;
  const int n_ranks = plan->info.n_ranks, rank = plan->info.rank;
  const int n_runs = plan->info.n_runs_total;
  const int64_t n_pixels = plan->info.n_pixels;
  // per global run: x-extent of every row of its rectangle
  std::vector<std::vector<int32_t>> run_x0(static_cast<size_t>(n_runs)), run_x1(static_cast<size_t>(n_runs));
  std::vector<int32_t> box_x0, box_x1;
  for (int g = 0; g < n_runs; ++g) {
    const avr_run_info& run = plan->runs[static_cast<size_t>(g)];
    const RunRectDev rect = plan->global_rects[static_cast<size_t>(g)];
    if (rect.x1 < rect.x0 || rect.y1 < rect.y0) continue;
    const int rows = rect.y1 - rect.y0 + 1;
    run_x0[static_cast<size_t>(g)].assign(static_cast<size_t>(rows), 0);
    run_x1[static_cast<size_t>(g)].assign(static_cast<size_t>(rows), -1);
    for (int l = run.first_layer; l < run.first_layer + run.n_layers; ++l) {
      const int32_t b = plan->layer_box[static_cast<size_t>(l)];
      if (b < 0 || b >= n_boxes) throw std::invalid_argument("the boxes do not match the plan");
      int32_t box_rect[4];
      box_screen_rect(all_boxes[b], plan->camera, width, height, box_rect);
      if (box_rect[2] < box_rect[0] || box_rect[3] < box_rect[1]) continue;
      box_row_spans(all_boxes[b], plan->camera, width, height, box_rect, &box_x0, &box_x1);
      for (int y = box_rect[1]; y <= box_rect[3]; ++y) {
        const int32_t x0 = box_x0[static_cast<size_t>(y - box_rect[1])];
        const int32_t x1 = box_x1[static_cast<size_t>(y - box_rect[1])];
        if (x1 < x0 || y < rect.y0 || y > rect.y1) continue;
        int32_t& r0 = run_x0[static_cast<size_t>(g)][static_cast<size_t>(y - rect.y0)];
        int32_t& r1 = run_x1[static_cast<size_t>(g)][static_cast<size_t>(y - rect.y0)];
        if (r1 < r0) {
          r0 = x0;
          r1 = x1;
        } else {
          r0 = std::min(r0, x0);
          r1 = std::max(r1, x1);
        }
      }
    }
  }
  auto append_rows = [&](int g, int32_t first, int32_t count, std::vector<RunSpanDev>* spans,
                         int64_t* cursor) {
    const RunRectDev rect = plan->global_rects[static_cast<size_t>(g)];
    for (int32_t y = first; y < first + count; ++y) {
      RunSpanDev span{0, -1, *cursor};
      const int32_t x0 = run_x0[static_cast<size_t>(g)][static_cast<size_t>(y - rect.y0)];
      const int32_t x1 = run_x1[static_cast<size_t>(g)][static_cast<size_t>(y - rect.y0)];
      if (x1 >= x0) {
        span.x0 = x0;
        span.x1 = x1;
        *cursor += static_cast<int64_t>(x1 - x0 + 1) * 5;
      }
      spans->push_back(span);
    }
  };
  // global index of this rank's local runs
  std::vector<int> global_of_local;
  for (int g = 0; g < n_runs; ++g) {
    if (plan->runs[static_cast<size_t>(g)].owner == rank) global_of_local.push_back(g);
  }
  const int n_local_runs = plan->info.n_local_runs;
  // sender layout, same order as build_layer_plan: for peer s: for local run r: block(piece of s, r)
  plan->send_spans.clear();
  int64_t cursor = 0;
  for (int peer = 0; peer < n_ranks; ++peer) {
    const int piece = plan->piece_of_rank[static_cast<size_t>(peer)];
    const int64_t before = cursor;
    for (int r = 0; r < n_local_runs; ++r) {
      const size_t at = static_cast<size_t>(r) * n_ranks + static_cast<size_t>(piece);
      RunBlockDev& block = plan->send_blocks[at];
      block.offset = cursor;
      block.span_base = static_cast<int32_t>(plan->send_spans.size());
      append_rows(global_of_local[static_cast<size_t>(r)], block.first_row,
                  plan->send_block_rows[at], &plan->send_spans, &cursor);
    }
    plan->send_splits[static_cast<size_t>(peer)] = cursor - before;
  }
  plan->info.send_floats = cursor;
  // receiver layout: for source s: for run g of s: block(my piece, g)
  plan->recv_spans.clear();
  cursor = 0;
  for (int source = 0; source < n_ranks; ++source) {
    const int64_t before = cursor;
    for (int g = 0; g < n_runs; ++g) {
      if (plan->runs[static_cast<size_t>(g)].owner != source) continue;
      RunBlockDev& block = plan->recv_blocks[static_cast<size_t>(g)];
      block.offset = cursor;
      block.span_base = static_cast<int32_t>(plan->recv_spans.size());
      append_rows(g, block.first_row, plan->recv_block_rows[static_cast<size_t>(g)],
                  &plan->recv_spans, &cursor);
    }
    plan->recv_splits[static_cast<size_t>(source)] = cursor - before;
  }
  plan->info.recv_floats = cursor;
  (void)n_pixels;
  (void)height;
  plan->tightened = true;
}

void build_frame_plan(const avr_box* all_boxes, const int32_t* owner, int n_boxes, int n_ranks,
                      int rank, const int32_t* group_order, const avr_paint_params& params,
                      const avr_camera& camera, avr_frame_plan* plan) {
  if (n_boxes < 0 || (n_boxes > 0 && (all_boxes == nullptr || owner == nullptr))) {
    throw std::invalid_argument("invalid box list");
  }
  if (params.width <= 0 || params.height <= 0) {
    throw std::invalid_argument("image width and height must be positive");
  }
  plan->params = params;
  plan->colormap.assign(params.colormap, params.colormap + std::max(params.colormap_count, 0));
  plan->params.colormap = plan->colormap.empty() ? nullptr : plan->colormap.data();
  plan->camera = camera;
  // depth hints depend only on box corners and the camera (VolumeRenderer.cpp:541-553), and so do
  // the conservative screen rectangles: every rank derives them from the replicated metadata
  std::vector<float> hints(static_cast<size_t>(std::max(n_boxes, 1)));
  std::vector<std::array<int32_t, 4>> rects(static_cast<size_t>(std::max(n_boxes, 1)));
  for (int b = 0; b < n_boxes; ++b) {
    hints[static_cast<size_t>(b)] = box_depth_hint(all_boxes[b], camera);
    box_screen_rect(all_boxes[b], camera, params.width, params.height,
                    rects[static_cast<size_t>(b)].data());
  }
  build_layer_plan(n_boxes, hints.data(), owner,
                   reinterpret_cast<const int32_t(*)[4]>(rects.data()), n_ranks, rank, group_order,
                   params.width, params.height, plan);
  plan->from_boxes = true;
}

}  // namespace avr
