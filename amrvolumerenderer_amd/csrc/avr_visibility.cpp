// Visibility ordering of the ranks: BuildVisibilityOrderedGroup,
// Common/VisibilityOrdering.cpp:63-632.  Host only.
//
// The reference rebuilds, every frame, an O(B^2) graph of face-adjacent boxes, orients every edge
// by the sign of the view direction on the shared face's axis, and topologically sorts it with
// the boxes' projected depth range as tie-break; cycles are broken by splitting a box.  Which
// pairs of boxes share a face does not depend on the camera, so the pair list of the unsplit
// boxes is computed once per box set and kept (same pairs, same (i, j, axis) order, hence the same
// adjacency lists); per frame only the orientation, the depth ranges and the sort are redone.
// After a split the list is recomputed for the modified boxes exactly as the reference does.
//
// amrex::RealVect / amrex::SmallMatrix arithmetic (AMReX 26.04, fetched by the reference's
// CMakeLists.txt:43-52) is restated as published: doubles for RealVect; SmallMatrix<float>
// products accumulate r(i,j) += a(i,k) * b(k,j), k ascending, from zero.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <functional>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "avr_internal.h"

struct avr_visibility_graph {
  std::vector<avr::VisBox> boxes;  // rank-major, as the reference's allgathers lay them out
  std::vector<avr::VisPair> pairs; // face-adjacent pairs of `boxes`
  int n_ranks = 0;
  int dot_counter = 0;             // the reference's static graphFileCounter
};

namespace avr {

namespace {

using Mat4 = std::array<std::array<float, 4>, 4>;  // m[row][col]
using Vec4 = std::array<float, 4>;

struct Vec3d {
  double v[3];
};

Vec3d safe_normalize(const Vec3d& in) {  // Common/CameraUtils.hpp:17-23
  const double length = std::sqrt(in.v[0] * in.v[0] + in.v[1] * in.v[1] + in.v[2] * in.v[2]);
  if (length > 0.0 && std::isfinite(length)) {
    return {{in.v[0] / length, in.v[1] / length, in.v[2] / length}};
  }
  return {{0.0, 0.0, -1.0}};
}
Vec3d cross(const Vec3d& a, const Vec3d& b) {
  return {{a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2],
           a.v[0] * b.v[1] - a.v[1] * b.v[0]}};
}
double dot(const Vec3d& a, const Vec3d& b) {
  return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2];
}

Mat4 identity() {
  Mat4 m{};
  for (int i = 0; i < 4; ++i) m[i][i] = 1.0f;
  return m;
}

Mat4 make_view_matrix(const avr_camera& camera) {  // Common/CameraUtils.hpp:25-63
  const Vec3d eye{{camera.eye[0], camera.eye[1], camera.eye[2]}};
  const Vec3d up{{camera.up[0], camera.up[1], camera.up[2]}};
  const Vec3d forward = safe_normalize({{camera.look_at[0] - camera.eye[0],
                                         camera.look_at[1] - camera.eye[1],
                                         camera.look_at[2] - camera.eye[2]}});
  Vec3d right = cross(forward, up);
  const double right_length = std::sqrt(dot(right, right));
  if (right_length > 0.0 && std::isfinite(right_length)) {
    for (double& c : right.v) c /= right_length;
  } else {
    right = {{1.0, 0.0, 0.0}};
  }
  const Vec3d up_ortho = cross(right, forward);
  Mat4 view = identity();
  for (int c = 0; c < 3; ++c) {
    view[c][0] = static_cast<float>(right.v[c]);
    view[c][1] = static_cast<float>(up_ortho.v[c]);
    view[c][2] = static_cast<float>(-forward.v[c]);
    view[c][3] = 0.0f;
  }
  view[3][0] = static_cast<float>(-dot(right, eye));
  view[3][1] = static_cast<float>(-dot(up_ortho, eye));
  view[3][2] = static_cast<float>(dot(forward, eye));
  view[3][3] = 1.0f;
  return view;
}

Mat4 make_perspective_matrix(float fov_y_degrees, float aspect, float near_plane,
                             float far_plane) {  // VisibilityOrdering.cpp:35-59
  Mat4 m = identity();
  constexpr float kPi = 3.14159265358979323846f;
  const float fov_tangent = std::tan(fov_y_degrees * kPi / 180.0f * 0.5f);
  const float size = near_plane * fov_tangent;
  const float left = -size * aspect;
  const float right = size * aspect;
  const float bottom = -size;
  const float top = size;
  m[0][0] = 2.0f * near_plane / (right - left);
  m[1][1] = 2.0f * near_plane / (top - bottom);
  m[0][2] = (right + left) / (right - left);
  m[1][2] = (top + bottom) / (top - bottom);
  m[2][2] = -(far_plane + near_plane) / (far_plane - near_plane);
  m[3][2] = -1.0f;
  m[2][3] = -(2.0f * far_plane * near_plane) / (far_plane - near_plane);
  m[3][3] = 0.0f;
  return m;
}

Vec4 multiply(const Mat4& m, const Vec4& v) {
  Vec4 out{};
  for (int i = 0; i < 4; ++i) {
    float acc = 0.0f;
    for (int k = 0; k < 4; ++k) acc += m[i][k] * v[k];
    out[i] = acc;
  }
  return out;
}

constexpr float kInf = std::numeric_limits<float>::infinity();

void depth_range(const Mat4& modelview, const Mat4& projection, VisBox* box) {  // :165-192
  float min_depth = kInf, max_depth = -kInf;
  for (int corner = 0; corner < 8; ++corner) {
    const Vec4 homogeneous = {static_cast<float>((corner & 1) ? box->hi[0] : box->lo[0]),
                              static_cast<float>((corner & 2) ? box->hi[1] : box->lo[1]),
                              static_cast<float>((corner & 4) ? box->hi[2] : box->lo[2]), 1.0f};
    const Vec4 clip = multiply(projection, multiply(modelview, homogeneous));
    if (clip[3] != 0.0f) {
      const float normalized = clip[2] / clip[3];
      min_depth = std::min(min_depth, normalized);
      max_depth = std::max(max_depth, normalized);
    }
  }
  if (!std::isfinite(min_depth) || !std::isfinite(max_depth)) {
    min_depth = kInf;
    max_depth = kInf;
  }
  box->min_depth = min_depth;
  box->max_depth = max_depth;
}

bool nearly_equal(float a, float b) {  // :213-216
  const float scale = std::max({1.0f, std::fabs(a), std::fabs(b)});
  return std::fabs(a - b) <= 1e-5f * scale;
}

bool overlaps(float a_min, float a_max, float b_min, float b_max) {  // :218-230
  const float overlap_min = std::max(a_min, b_min);
  const float overlap_max = std::min(a_max, b_max);
  const float scale = std::max({1.0f, std::fabs(a_min), std::fabs(a_max), std::fabs(b_min),
                                std::fabs(b_max), std::fabs(overlap_min), std::fabs(overlap_max)});
  return (overlap_max - overlap_min) > 1e-5f * scale;
}

constexpr float kDirectionTolerance = 1e-6f;

}  // namespace

// The camera-independent half of rebuildAdjacency (:262-316): which (i < j, axis) share a face.
void visibility_pairs(const std::vector<VisBox>& boxes, std::vector<VisPair>* pairs) {
  pairs->clear();
  const int n = static_cast<int>(boxes.size());
  for (int i = 0; i < n; ++i) {
    const VisBox& a = boxes[static_cast<size_t>(i)];
    for (int j = i + 1; j < n; ++j) {
      const VisBox& b = boxes[static_cast<size_t>(j)];
      for (int axis = 0; axis < 3; ++axis) {
        const int axis1 = (axis + 1) % 3, axis2 = (axis + 2) % 3;
        if (!overlaps(a.lo[axis1], a.hi[axis1], b.lo[axis1], b.hi[axis1]) ||
            !overlaps(a.lo[axis2], a.hi[axis2], b.lo[axis2], b.hi[axis2])) {
          continue;
        }
        if (nearly_equal(a.hi[axis], b.lo[axis])) {
          pairs->push_back({i, j, axis, 0});  // a below b on this axis
        } else if (nearly_equal(b.hi[axis], a.lo[axis])) {
          pairs->push_back({i, j, axis, 1});  // b below a
        }
      }
    }
  }
}

namespace {

struct Graph {
  std::vector<std::vector<int>> adjacency;
  std::vector<int> indegree;
};

void orient(const std::vector<VisPair>& pairs, size_t n_boxes, const Vec3d& view_dir,
            Graph* g) {
  g->adjacency.assign(n_boxes, {});
  g->indegree.assign(n_boxes, 0);
  auto add_edge = [&](int from, int to) {
    auto& edges = g->adjacency[static_cast<size_t>(from)];
    if (std::find(edges.begin(), edges.end(), to) == edges.end()) {
      edges.push_back(to);
      ++g->indegree[static_cast<size_t>(to)];
    }
  };
  for (const VisPair& p : pairs) {
    const float dir = static_cast<float>(view_dir.v[p.axis]);
    // lower box first when looking along +axis: the upper one is behind it
    const bool positive = dir > kDirectionTolerance, negative = dir < -kDirectionTolerance;
    if (!positive && !negative) continue;
    const bool j_to_i = (p.b_below_a == 0) ? positive : negative;
    if (j_to_i) {
      add_edge(p.j, p.i);
    } else {
      add_edge(p.i, p.j);
    }
  }
}

void write_dot(const std::string& path, const std::vector<VisBox>& boxes, const Graph& g) {
  std::FILE* file = std::fopen(path.c_str(), "w");
  if (file == nullptr) {
    std::fprintf(stderr, "Failed to write visibility graph to '%s'\n", path.c_str());
    return;
  }
  std::fputs("digraph VisibilityGraph {\n  rankdir=LR;\n", file);
  for (size_t idx = 0; idx < boxes.size(); ++idx) {
    std::fprintf(file, "  box%zu [label=\"box %zu\\nrank %d\\nminDepth %f\\nmaxDepth %f\"];\n", idx,
                 idx, boxes[idx].owner, static_cast<double>(boxes[idx].min_depth),
                 static_cast<double>(boxes[idx].max_depth));
  }
  for (size_t from = 0; from < g.adjacency.size(); ++from) {
    for (int to : g.adjacency[from]) std::fprintf(file, "  box%zu -> box%d;\n", from, to);
  }
  std::fputs("}\n", file);
  std::fclose(file);
}

}  // namespace

bool visibility_order(avr_visibility_graph* graph, const avr_camera& camera, float aspect,
                      const char* dot_prefix, int32_t* rank_order, int* n_splits) {
  const int n_ranks = graph->n_ranks;
  if (n_splits != nullptr) *n_splits = 0;
  for (int r = 0; r < n_ranks; ++r) rank_order[r] = r;
  if (graph->boxes.empty()) return true;  // totalBoxes <= 0 (:99-101)

  const Mat4 modelview = make_view_matrix(camera);
  const Mat4 projection =
      make_perspective_matrix(camera.fov_y_degrees, aspect, camera.near_plane, camera.far_plane);
  const Vec3d view_dir = safe_normalize({{camera.look_at[0] - camera.eye[0],
                                          camera.look_at[1] - camera.eye[1],
                                          camera.look_at[2] - camera.eye[2]}});
  std::vector<VisBox> boxes = graph->boxes;
  for (VisBox& box : boxes) depth_range(modelview, projection, &box);

  auto before = [&](int lhs, int rhs) {  // compareBoxes (:240-258)
    const VisBox& a = boxes[static_cast<size_t>(lhs)];
    const VisBox& b = boxes[static_cast<size_t>(rhs)];
    const bool a_finite = std::isfinite(a.min_depth), b_finite = std::isfinite(b.min_depth);
    if (a_finite != b_finite) return a_finite && !b_finite;
    if (a.min_depth == b.min_depth) {
      if (a.max_depth == b.max_depth) {
        if (a.owner == b.owner) return lhs < rhs;
        return a.owner < b.owner;
      }
      return a.max_depth < b.max_depth;
    }
    return a.min_depth < b.min_depth;
  };

  const int max_iterations = static_cast<int>(std::max<size_t>(graph->boxes.size(), 1)) * 8 + 32;
  std::vector<VisPair> split_pairs;
  const std::vector<VisPair>* pairs = &graph->pairs;
  Graph g;
  for (int iteration = 0; iteration < max_iterations; ++iteration) {
    orient(*pairs, boxes.size(), view_dir, &g);
    if (dot_prefix != nullptr) {
      write_dot(std::string(dot_prefix) + std::to_string(graph->dot_counter++) + ".dot", boxes, g);
    }

    // topoSortBoxes (:358-399): always take the smallest ready box under compareBoxes
    const int n = static_cast<int>(boxes.size());
    std::vector<int> indegree = g.indegree;
    std::vector<int> ready, order;
    for (int i = 0; i < n; ++i) {
      if (indegree[static_cast<size_t>(i)] == 0) ready.push_back(i);
    }
    auto after = [&](int lhs, int rhs) { return before(rhs, lhs); };  // min-heap on `before`
    std::make_heap(ready.begin(), ready.end(), after);
    while (!ready.empty()) {
      std::pop_heap(ready.begin(), ready.end(), after);
      const int current = ready.back();
      ready.pop_back();
      order.push_back(current);
      for (int next : g.adjacency[static_cast<size_t>(current)]) {
        if (--indegree[static_cast<size_t>(next)] == 0) {
          ready.push_back(next);
          std::push_heap(ready.begin(), ready.end(), after);
        }
      }
    }
    if (static_cast<int>(order.size()) == n) {
      std::vector<char> visited(static_cast<size_t>(n_ranks), 0);
      int count = 0;
      for (int index : order) {
        const int owner = boxes[static_cast<size_t>(index)].owner;
        if (owner >= 0 && !visited[static_cast<size_t>(owner)]) {
          visited[static_cast<size_t>(owner)] = 1;
          rank_order[count++] = owner;
        }
      }
      for (int owner = 0; owner < n_ranks; ++owner) {
        if (!visited[static_cast<size_t>(owner)]) rank_order[count++] = owner;
      }
      return true;
    }

    // findCycle (:401-445)
    std::vector<int> state(static_cast<size_t>(n), 0), parent(static_cast<size_t>(n), -1), cycle;
    std::function<bool(int)> dfs = [&](int node) -> bool {
      state[static_cast<size_t>(node)] = 1;
      for (int next : g.adjacency[static_cast<size_t>(node)]) {
        if (state[static_cast<size_t>(next)] == 0) {
          parent[static_cast<size_t>(next)] = node;
          if (dfs(next)) return true;
        } else if (state[static_cast<size_t>(next)] == 1) {
          cycle.clear();
          cycle.push_back(next);
          for (int cur = node; cur != next && cur != -1; cur = parent[static_cast<size_t>(cur)]) {
            cycle.push_back(cur);
          }
          std::reverse(cycle.begin(), cycle.end());
          return true;
        }
      }
      state[static_cast<size_t>(node)] = 2;
      return false;
    };
    for (int node = 0; node < n; ++node) {
      if (indegree[static_cast<size_t>(node)] > 0 && state[static_cast<size_t>(node)] == 0) {
        if (dfs(node)) break;
      }
    }
    if (cycle.empty()) break;

    // breakCycle (:447-569)
    if (cycle.size() < 2) break;
    int chosen_axis = 0;
    float best_alignment = static_cast<float>(std::fabs(view_dir.v[0]));
    for (int axis = 1; axis < 3; ++axis) {
      const float alignment = static_cast<float>(std::fabs(view_dir.v[axis]));
      if (alignment > best_alignment) {
        best_alignment = alignment;
        chosen_axis = axis;
      }
    }
    if (best_alignment <= kDirectionTolerance) {
      float widest = -1.0f;
      for (int axis = 0; axis < 3; ++axis) {
        for (int index : cycle) {
          const VisBox& box = boxes[static_cast<size_t>(index)];
          const float length = box.hi[axis] - box.lo[axis];
          if (length > widest) {
            widest = length;
            chosen_axis = axis;
          }
        }
      }
    }
    const float dir = static_cast<float>(view_dir.v[chosen_axis]);
    if (std::fabs(dir) <= kDirectionTolerance) break;
    const float min_length_tolerance = 1e-6f;
    int target_index = cycle.front();
    float target_length = -1.0f;
    for (int index : cycle) {
      const VisBox& box = boxes[static_cast<size_t>(index)];
      const float length = box.hi[chosen_axis] - box.lo[chosen_axis];
      if (length > target_length && length > min_length_tolerance) {
        target_length = length;
        target_index = index;
      }
    }
    if (target_length <= min_length_tolerance) break;
    const VisBox target = boxes[static_cast<size_t>(target_index)];
    const float min_val = target.lo[chosen_axis], max_val = target.hi[chosen_axis];
    const float length = max_val - min_val;
    const float epsilon = std::max(1e-5f * length, 1e-6f);
    bool have_candidate = false;
    float best_candidate = 0.0f;
    for (int index : cycle) {
      if (index == target_index) continue;
      const VisBox& other = boxes[static_cast<size_t>(index)];
      const float ends[2] = {other.lo[chosen_axis], other.hi[chosen_axis]};
      for (float value : ends) {
        if (value > min_val + epsilon && value < max_val - epsilon) {
          if (!have_candidate) {
            best_candidate = value;
            have_candidate = true;
          } else if (dir > 0.0f) {
            best_candidate = std::max(best_candidate, value);
          } else {
            best_candidate = std::min(best_candidate, value);
          }
        }
      }
    }
    float split = 0.5f * (min_val + max_val);
    if (have_candidate) split = best_candidate;
    if (split <= min_val + epsilon) split = min_val + epsilon;
    if (split >= max_val - epsilon) split = max_val - epsilon;
    if (!(split > min_val && split < max_val)) break;
    VisBox near_box = target, far_box = target;
    if (dir > 0.0f) {
      near_box.hi[chosen_axis] = split;
      far_box.lo[chosen_axis] = split;
    } else {
      near_box.lo[chosen_axis] = split;
      far_box.hi[chosen_axis] = split;
    }
    depth_range(modelview, projection, &near_box);
    depth_range(modelview, projection, &far_box);
    boxes[static_cast<size_t>(target_index)] = near_box;
    boxes.push_back(far_box);
    if (n_splits != nullptr) ++*n_splits;
    visibility_pairs(boxes, &split_pairs);  // the boxes changed: rebuild as the reference does
    pairs = &split_pairs;
  }
  for (int r = 0; r < n_ranks; ++r) rank_order[r] = r;  // fall back to the default order
  return false;
}

avr_visibility_graph* visibility_graph_create(const avr_box* all_boxes, const int32_t* owner,
                                              int n_boxes, int n_ranks) {
  auto* graph = new avr_visibility_graph();
  graph->n_ranks = n_ranks;
  graph->boxes.reserve(static_cast<size_t>(n_boxes));
  for (int rank = 0; rank < n_ranks; ++rank) {  // rank-major, local order kept (:86-152)
    for (int b = 0; b < n_boxes; ++b) {
      if (owner[b] != rank) continue;
      VisBox box;
      for (int c = 0; c < 3; ++c) {  // corners travel as MPI_FLOAT (:107-119)
        box.lo[c] = static_cast<float>(all_boxes[b].min_corner[c]);
        box.hi[c] = static_cast<float>(all_boxes[b].max_corner[c]);
      }
      box.owner = rank;
      box.min_depth = box.max_depth = kInf;
      graph->boxes.push_back(box);
    }
  }
  visibility_pairs(graph->boxes, &graph->pairs);
  return graph;
}

void visibility_graph_destroy(avr_visibility_graph* graph) { delete graph; }
int visibility_rank_count(const avr_visibility_graph* graph) { return graph->n_ranks; }

}  // namespace avr
