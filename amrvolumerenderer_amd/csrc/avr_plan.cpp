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

// rows of piece k among the image rows [0, y)  (kPiecesRowBands)
int rows_of_piece_below(const PieceMapDev& m, int k, int y) {
  const int cycle = m.band_rows * m.n_pieces;
  const int rest = y % cycle - k * m.band_rows;
  return (y / cycle) * m.band_rows + std::clamp(rest, 0, m.band_rows);
}

// The image rows of piece k, as the set the row-extent code iterates over.
RowSet piece_row_set(const PieceMapDev& m, int k) {
  RowSet set;
  if (m.layout == kPiecesRowBands) {
    set.band_rows = m.band_rows;
    set.period = m.n_pieces;
    set.phase = k;
    set.lo = 0;
    set.hi = m.height - 1;
    return set;
  }
  int64_t begin = 0, end = 0;
  piece_pixel_range(m, k, &begin, &end);
  if (end <= begin) {
    set.lo = 0;
    set.hi = -1;
    return set;
  }
  set.lo = static_cast<int32_t>(begin / m.width);
  set.hi = static_cast<int32_t>((end - 1) / m.width);
  return set;
}

// rows of `rect` that fall into piece k, in the piece's row numbering: [first, first + count)
void block_rows(const RunRectDev& rect, const PieceMapDev& m, int k, int32_t* first,
                int32_t* count) {
  *first = 0;
  *count = 0;
  if (rect.x1 < rect.x0 || rect.y1 < rect.y0) return;
  if (m.layout == kPiecesRowBands) {
    const int below = rows_of_piece_below(m, k, rect.y0);
    const int upto = rows_of_piece_below(m, k, rect.y1 + 1);
    if (upto > below) {
      *first = below;
      *count = upto - below;
    }
    return;
  }
  const RowSet rows = piece_row_set(m, k);
  const int32_t lo = std::max(rect.y0, rows.lo);
  const int32_t hi = std::min(rect.y1, rows.hi);
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

PieceMapDev make_piece_map(int layout, int band_rows, int n_pieces, int width, int height) {
  PieceMapDev m;
  m.n_pieces = n_pieces;
  m.width = width;
  m.height = height;
  m.piece_size = (static_cast<int64_t>(width) * height) / n_pieces;  // getPieceRange, :59-74
  m.layout = (layout == kPiecesRowBands && n_pieces > 1) ? kPiecesRowBands : kPiecesContiguous;
  m.band_rows = (m.layout == kPiecesRowBands) ? std::max(band_rows, 1) : 1;
  return m;
}

void piece_pixel_range(const PieceMapDev& m, int k, int64_t* begin, int64_t* end) {
  if (m.layout == kPiecesRowBands) {
    int64_t rows_before = 0;
    for (int j = 0; j < k; ++j) rows_before += piece_row_count(m, j);
    *begin = rows_before * m.width;
    *end = *begin + static_cast<int64_t>(piece_row_count(m, k)) * m.width;
    return;
  }
  const int64_t n_pixels = static_cast<int64_t>(m.width) * m.height;
  *begin = m.piece_size * k;
  *end = (k < m.n_pieces - 1) ? *begin + m.piece_size : n_pixels;
}

void dense_run_tables(int width, int height, int n_runs, int n_pieces,
                      std::vector<RunRectDev>* rects, std::vector<RunBlockDev>* blocks) {
  const PieceMapDev map = make_piece_map(kPiecesContiguous, 1, n_pieces, width, height);
  rects->assign(static_cast<size_t>(n_runs), RunRectDev{0, 0, width - 1, height - 1});
  blocks->assign(static_cast<size_t>(n_runs) * n_pieces, RunBlockDev{0, 0, -1});
  for (int k = 0; k < n_pieces; ++k) {
    int64_t begin = 0, end = 0;
    piece_pixel_range(map, k, &begin, &end);
    const int64_t len = end - begin;
    const int32_t first_row = (end > begin) ? static_cast<int32_t>(begin / width) : 0;
    for (int r = 0; r < n_runs; ++r) {
      // out[5*n_runs*begin + (r*len + (p - begin))*5]  ==  offset + (p - first_row*width)*5
      RunBlockDev& block = (*blocks)[static_cast<size_t>(r) * n_pieces + k];
      block.first_row = first_row;
      block.offset = 5 * static_cast<int64_t>(n_runs) * begin +
                     (static_cast<int64_t>(r) * len - begin + static_cast<int64_t>(first_row) * width) * 5;
    }
  }
}

// The plan of a set of layers described only by (depth hint, owner, screen rectangle): global
// order, runs, exchange layout.  layer l's local index is its position among its owner's layers.
void build_layer_plan(int n_layers, const float* hints, const int32_t* owner,
                      const int32_t (*rects)[4], int n_ranks, int rank, const int32_t* group_order,
                      int width, int height, int piece_layout, int band_rows,
                      avr_frame_plan* plan) {
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) throw std::invalid_argument("invalid rank");
  if (n_layers < 0 || (n_layers > 0 && (hints == nullptr || owner == nullptr))) {
    throw std::invalid_argument("invalid layer list");
  }
  if (width <= 0 || height <= 0) {
    throw std::invalid_argument("image width and height must be positive");
  }
  if (piece_layout != kPiecesContiguous && piece_layout != kPiecesRowBands) {
    throw std::invalid_argument("unknown piece layout");
  }
  if (piece_layout == kPiecesRowBands && (band_rows < 1 || (band_rows & (band_rows - 1)) != 0)) {
    throw std::invalid_argument("band_rows must be a power of two");
  }
  const int64_t n_pixels = static_cast<int64_t>(width) * height;
  if (n_pixels > (int64_t{1} << 31) - 1) throw std::invalid_argument("image too large");
  const int n_boxes = n_layers;
  plan->pieces = make_piece_map(piece_layout, band_rows, n_ranks, width, height);
  const PieceMapDev& map = plan->pieces;

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
    const int64_t before = cursor;
    for (int r = 0; r < n_local_runs; ++r) {
      const RunRectDev& rect = plan->local_rects[static_cast<size_t>(r)];
      int32_t first = 0, count = 0;
      block_rows(rect, map, piece, &first, &count);
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
      block_rows(rect, map, my_piece, &first, &count);
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
  int64_t piece_begin = 0, piece_end = 0;
  piece_pixel_range(map, my_piece, &piece_begin, &piece_end);
  if (map.layout == kPiecesRowBands) {  // not an image range: the piece's rows in order
    piece_end -= piece_begin;
    piece_begin = 0;
  }
  info.piece_begin = piece_begin;
  info.piece_end = piece_end;
  info.send_floats = send_floats;
  info.recv_floats = cursor;
  info.piece_layout = map.layout;
  info.band_rows = map.band_rows;
}

// The exchange volume of a frame plan is the area of the runs' screen RECTANGLES; measured on
// config-4 only 48 % (2 ranks) to 72 % (8 ranks) of those pixels carry anything
// (tools/send_occupancy.py).  Per row of a run, only the union of its boxes' conservative
// extents (merge_box_row_spans) is kept: what lies outside is the cleared layer pixel, the exact
// identity of the depth-sort blend, on the sender and on the receiver alike.  Block sizes still
// follow from replicated metadata alone, so no sizes travel.
//
// A rank needs the extents of its OWN runs on every row (it stores them for every peer) and of
// the other ranks' runs only on the rows of its own piece (it receives nothing else).
void tighten_frame_plan(const avr_box* all_boxes, int n_boxes, avr_frame_plan* plan) {
  if (plan->tightened) return;
  if (!plan->from_boxes || plan->footprints.size() != static_cast<size_t>(n_boxes)) {
    throw std::invalid_argument("only a frame plan made from these boxes can be tightened");
  }
  const int width = plan->params.width, height = plan->params.height;
  // (the same decision on every rank: it depends on the image size alone)
  if (width > kMaxTightenedWidth || plan->info.n_pixels * 5 > kMaxTightenedBlockFloats) return;
  const int n_ranks = plan->info.n_ranks, rank = plan->info.rank;
  const int n_runs = plan->info.n_runs_total;
  const PieceMapDev& map = plan->pieces;
  const int my_piece = plan->piece_of_rank[static_cast<size_t>(rank)];
  const RowSet all_rows{1, 1, 0, 0, height - 1};
  const RowSet my_rows = piece_row_set(map, my_piece);
  // per global run: x-extent of the needed rows of its rectangle (flat, run_at[g] + y - rect.y0)
  thread_local std::vector<int32_t> run_x0, run_x1;
  std::vector<int64_t> run_at(static_cast<size_t>(n_runs) + 1, 0);
  for (int g = 0; g < n_runs; ++g) {
    const RunRectDev rect = plan->global_rects[static_cast<size_t>(g)];
    const int rows = (rect.x1 < rect.x0 || rect.y1 < rect.y0) ? 0 : rect.y1 - rect.y0 + 1;
    run_at[static_cast<size_t>(g) + 1] = run_at[static_cast<size_t>(g)] + rows;
  }
  if (run_x0.size() < static_cast<size_t>(run_at.back())) {
    run_x0.resize(static_cast<size_t>(run_at.back()));
    run_x1.resize(static_cast<size_t>(run_at.back()));
  }
  for (int g = 0; g < n_runs; ++g) {
    const avr_run_info& run = plan->runs[static_cast<size_t>(g)];
    const RunRectDev rect = plan->global_rects[static_cast<size_t>(g)];
    if (rect.x1 < rect.x0 || rect.y1 < rect.y0) continue;
    const RowSet& needed = (run.owner == rank) ? all_rows : my_rows;
    int32_t* x0 = run_x0.data() + run_at[static_cast<size_t>(g)];
    int32_t* x1 = run_x1.data() + run_at[static_cast<size_t>(g)];
    for_rows(needed, rect.y0, rect.y1, [&](int a, int b) {
      for (int y = a; y <= b; ++y) {
        x0[y - rect.y0] = 0;
        x1[y - rect.y0] = -1;
      }
    });
    for (int l = run.first_layer; l < run.first_layer + run.n_layers; ++l) {
      const int32_t b = plan->layer_box[static_cast<size_t>(l)];
      if (b < 0 || b >= n_boxes) throw std::invalid_argument("the boxes do not match the plan");
      // (a run's rectangle is the union of its boxes' rectangles: every box row is a run row)
      merge_footprint_rows(plan->footprints[static_cast<size_t>(b)], needed, rect.y0, x0, x1);
    }
  }
  // the rows of one block: piece rows [first, first + count) of piece k
  auto append_rows = [&](int g, int k, int32_t first, int32_t count, std::vector<RunSpanDev>* spans,
                         int64_t* cursor) {
    if (count <= 0) return;
    const RunRectDev rect = plan->global_rects[static_cast<size_t>(g)];
    const int32_t* x0 = run_x0.data() + run_at[static_cast<size_t>(g)] - rect.y0;
    const int32_t* x1 = run_x1.data() + run_at[static_cast<size_t>(g)] - rect.y0;
    const int64_t base = *cursor;
    int64_t at = base;
    const size_t begin = spans->size();
    spans->resize(begin + static_cast<size_t>(count));
    RunSpanDev* out = spans->data() + begin;
    const bool bands = map.layout == kPiecesRowBands;
    int y = bands ? image_row_of(map, k, first) : first;
    const int skip = bands ? (map.n_pieces - 1) * map.band_rows : 0;  // rows of the other pieces
    for (int32_t j = 0; j < count; ++j) {
      RunSpanDev span{1, 0, static_cast<uint32_t>(at - base)};
      const int32_t a = x0[y], b = x1[y];
      if (b >= a) {
        span.x0 = static_cast<uint16_t>(a);
        span.x1 = static_cast<uint16_t>(b);
        at += static_cast<int64_t>(b - a + 1) * 5;
      }
      out[j] = span;
      ++y;
      if (bands && (y & (map.band_rows - 1)) == 0) y += skip;  // (band_rows is a power of two)
    }
    *cursor = at;
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
      append_rows(global_of_local[static_cast<size_t>(r)], piece, block.first_row,
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
      append_rows(g, my_piece, block.first_row, plan->recv_block_rows[static_cast<size_t>(g)],
                  &plan->recv_spans, &cursor);
    }
    plan->recv_splits[static_cast<size_t>(source)] = cursor - before;
  }
  plan->info.recv_floats = cursor;
  plan->tightened = true;
}

void build_frame_plan(const avr_box* all_boxes, const int32_t* owner, int n_boxes, int n_ranks,
                      int rank, const int32_t* group_order, const avr_paint_params& params,
                      const avr_camera& camera, int piece_layout, int band_rows,
                      avr_frame_plan* plan) {
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
  plan->footprints.resize(static_cast<size_t>(n_boxes));
  box_footprints(all_boxes, n_boxes, camera, params.width, params.height, plan->footprints.data());
  for (int b = 0; b < n_boxes; ++b) {
    hints[static_cast<size_t>(b)] = box_depth_hint(all_boxes[b], camera);
    std::copy(plan->footprints[static_cast<size_t>(b)].rect,
              plan->footprints[static_cast<size_t>(b)].rect + 4, rects[static_cast<size_t>(b)].data());
  }
  build_layer_plan(n_boxes, hints.data(), owner,
                   reinterpret_cast<const int32_t(*)[4]>(rects.data()), n_ranks, rank, group_order,
                   params.width, params.height, piece_layout, band_rows, plan);
  plan->from_boxes = true;
}

}  // namespace avr
