// Host prologue of the MI355X volume-rendering path: everything VolumePainter::paint and
// renderSingleTrial compute on the host per frame / per box before the device loop runs.
// The transfer-function table is built here (host libm, 256 entries per sampling level) so
// that pow/tan never run on the GPU and the table bits match the reference's host build
// ("Hard parts").
//
// Reference: Common/VolumePainter.cpp:107-125 (computeScaledAlpha), :127-200 (table nodes),
// :202-320 (CIELAB), :331-440 (colour / opacity mapping), :442-516 (buildColorTable),
// :571-733 (paint prologue); VolumeRenderer/VolumeRenderer.cpp:541-553 (depth hint),
// :1138-1190 (reference sample distance); DirectSend/Base/DirectSendBase.cpp:363-410 (order).
#include "avr_internal.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <stdexcept>

namespace avr {

namespace {

constexpr float kSoftClipTolerance = 1e-5f;
constexpr float kPi = 3.14159265358979323846f;

struct Rgba {
  float r = 0.0f, g = 0.0f, b = 0.0f, a = 0.0f;
};

struct ColorStop {
  float value = 0.0f;
  float r = 0.0f, g = 0.0f, b = 0.0f;
};

struct OpacityStop {
  float value = 0.0f;
  float alpha = 0.0f;
  float midpoint = 0.5f;
  float sharpness = 0.0f;
};

// The reference's ColorTableSpec (VolumePainter.cpp:65-73).
class TransferFunction {
 public:
  bool lab = false;
  bool clamp_ends = true;
  Rgba nan_color{0.5f, 0.0f, 0.0f, 1.0f};
  Rgba below{0.0f, 0.0f, 0.0f, 1.0f};
  Rgba above{0.0f, 0.0f, 0.0f, 1.0f};
  std::vector<ColorStop> colors;
  std::vector<OpacityStop> opacity;

  // insertColorNode / insertOpacityNode: sorted by value, an equal value replaces (:127-151).
  void add(const ColorStop& stop) {
    auto it = std::lower_bound(colors.begin(), colors.end(), stop.value,
                               [](const ColorStop& s, float v) { return s.value < v; });
    if (it != colors.end() && it->value == stop.value) {
      *it = stop;
    } else {
      colors.insert(it, stop);
    }
  }
  void add(const OpacityStop& stop) {
    auto it = std::lower_bound(opacity.begin(), opacity.end(), stop.value,
                               [](const OpacityStop& s, float v) { return s.value < v; });
    if (it != opacity.end() && it->value == stop.value) {
      *it = stop;
    } else {
      opacity.insert(it, stop);
    }
  }

  // rescaleTableToRange (:153-200)
  void rescale(float range_min, float range_max) {
    bool seen = false;
    float lo = 0.0f, hi = 0.0f;
    auto visit = [&](float v) {
      if (!seen) {
        lo = hi = v;
        seen = true;
        return;
      }
      lo = std::min(lo, v);
      hi = std::max(hi, v);
    };
    for (const auto& s : colors) visit(s.value);
    for (const auto& s : opacity) visit(s.value);
    if (!seen) lo = hi = 0.0f;
    const float old_span = hi - lo;
    const float new_span = range_max - range_min;
    if (!(old_span > 0.0f) || !(new_span > 0.0f)) return;
    for (auto& s : colors) {
      const float t = (s.value - lo) / old_span;
      s.value = range_min + new_span * t;
    }
    for (auto& s : opacity) {
      const float t = (s.value - lo) / old_span;
      s.value = range_min + new_span * t;
    }
  }

  Rgba color_at(float value) const;     // mapColorValue (:331-379)
  float opacity_at(float value) const;  // mapOpacityValue (:381-440)
};

float srgb_to_linear(float c) {
  return (c > 0.04045f) ? std::pow((c + 0.055f) / 1.055f, 2.4f) : c / 12.92f;
}
float linear_to_srgb(float c) {
  constexpr float inv_gamma = 1.0f / 2.4f;
  return (c > 0.0031308f) ? 1.055f * std::pow(c, inv_gamma) - 0.055f : 12.92f * c;
}
float lab_f(float t) {
  constexpr float third = 1.0f / 3.0f;
  constexpr float offset = 16.0f / 116.0f;
  return (t > 0.008856f) ? std::pow(t, third) : (7.787f * t) + offset;
}
float lab_f_inverse(float t) {
  constexpr float offset = 16.0f / 116.0f;
  return (std::pow(t, 3.0f) > 0.008856f) ? std::pow(t, 3.0f) : (t - offset) / 7.787f;
}

// rgbToLab (:202-256): sRGB -> XYZ (D65 constants as written in the reference) -> CIELAB.
Rgba to_lab(const Rgba& rgb) {
  const float r = srgb_to_linear(rgb.r);
  const float g = srgb_to_linear(rgb.g);
  const float b = srgb_to_linear(rgb.b);
  const float x = r * 0.4124f + g * 0.3576f + b * 0.1805f;
  const float y = r * 0.2126f + g * 0.7152f + b * 0.0722f;
  const float z = r * 0.0193f + g * 0.1192f + b * 0.9505f;
  const float fx = lab_f(x / 0.9505f);
  const float fy = lab_f(y / 1.0f);
  const float fz = lab_f(z / 1.089f);
  Rgba lab;
  lab.r = (116.0f * fy) - 16.0f;
  lab.g = 500.0f * (fx - fy);
  lab.b = 200.0f * (fy - fz);
  lab.a = rgb.a;
  return lab;
}

// labToRgb (:258-320), including the max-normalisation and the >= 0 clamp.
Rgba from_lab(const Rgba& lab) {
  float y = (lab.r + 16.0f) / 116.0f;
  float x = lab.g / 500.0f + y;
  float z = y - lab.b / 200.0f;
  x = lab_f_inverse(x);
  y = lab_f_inverse(y);
  z = lab_f_inverse(z);
  x *= 0.9505f;
  y *= 1.0f;
  z *= 1.089f;
  float r = x * 3.2406f + y * -1.5372f + z * -0.4986f;
  float g = x * -0.9689f + y * 1.8758f + z * 0.0415f;
  float b = x * 0.0557f + y * -0.2040f + z * 1.0570f;
  r = linear_to_srgb(r);
  g = linear_to_srgb(g);
  b = linear_to_srgb(b);
  const float peak = std::max(r, std::max(g, b));
  if (peak > 1.0f) {
    r /= peak;
    g /= peak;
    b /= peak;
  }
  Rgba rgb;
  rgb.r = std::max(r, 0.0f);
  rgb.g = std::max(g, 0.0f);
  rgb.b = std::max(b, 0.0f);
  rgb.a = lab.a;
  return rgb;
}

Rgba mix(const Rgba& l, const Rgba& r, float t) {  // lerpColor (:322-329)
  return {l.r + (r.r - l.r) * t, l.g + (r.g - l.g) * t, l.b + (r.b - l.b) * t,
          l.a + (r.a - l.a) * t};
}

Rgba TransferFunction::color_at(float value) const {
  if (!std::isfinite(value)) return nan_color;
  if (colors.empty()) return below;
  const ColorStop& first = colors.front();
  const ColorStop& last = colors.back();
  if (value < first.value) return clamp_ends ? Rgba{first.r, first.g, first.b, 1.0f} : below;
  if (value > last.value) return clamp_ends ? Rgba{last.r, last.g, last.b, 1.0f} : above;
  if (value == first.value) return {first.r, first.g, first.b, 1.0f};
  if (value == last.value) return {last.r, last.g, last.b, 1.0f};
  for (std::size_t i = 1; i < colors.size(); ++i) {
    const ColorStop& hi = colors[i];
    if (hi.value >= value) {
      const ColorStop& lo = colors[i - 1];
      const float span = hi.value - lo.value;
      const float t = (span > 0.0f) ? (value - lo.value) / span : 0.0f;
      const Rgba a{lo.r, lo.g, lo.b, 1.0f};
      const Rgba b{hi.r, hi.g, hi.b, 1.0f};
      if (lab) return from_lab(mix(to_lab(a), to_lab(b), t));
      return mix(a, b, t);
    }
  }
  return {last.r, last.g, last.b, 1.0f};
}

float TransferFunction::opacity_at(float value) const {
  if (!std::isfinite(value)) return 1.0f;
  if (opacity.empty()) return 1.0f;
  const OpacityStop& first = opacity.front();
  const OpacityStop& last = opacity.back();
  if (value <= first.value) return first.alpha;
  if (value >= last.value) return last.alpha;
  for (std::size_t i = 1; i < opacity.size(); ++i) {
    const OpacityStop& hi = opacity[i];
    if (!(hi.value >= value)) continue;
    const OpacityStop& lo = opacity[i - 1];
    const float span = hi.value - lo.value;
    float w = (span > 0.0f) ? (value - lo.value) / span : 0.0f;
    if (w < lo.midpoint) {
      w = 0.5f * w / lo.midpoint;
    } else {
      w = 0.5f + 0.5f * (w - lo.midpoint) / (1.0f - lo.midpoint);
    }
    if (lo.sharpness == 1.0f) return (w < 0.5f) ? lo.alpha : hi.alpha;
    if (lo.sharpness == 0.0f) return lo.alpha + (hi.alpha - lo.alpha) * w;
    if (w < 0.5f) {
      w = 0.5f * std::pow(w * 2.0f, 1.0f + 10.0f * lo.sharpness);
    } else if (w > 0.5f) {
      w = 1.0f - 0.5f * std::pow((1.0f - w) * 2.0f, 1.0f + 10.0f * lo.sharpness);
    }
    const float w2 = w * w;
    const float w3 = w2 * w;
    const float h1 = 2.0f * w3 - 3.0f * w2 + 1.0f;
    const float h2 = -2.0f * w3 + 3.0f * w2;
    const float h3 = w3 - 2.0f * w2 + w;
    const float h4 = w3 - w2;
    const float slope = hi.alpha - lo.alpha;
    const float tangent = (1.0f - lo.sharpness) * slope;
    float result = h1 * lo.alpha + h2 * hi.alpha + h3 * tangent + h4 * tangent;
    result = std::max(result, std::min(lo.alpha, hi.alpha));
    result = std::min(result, std::max(lo.alpha, hi.alpha));
    return result;
  }
  return last.alpha;
}

// computeScaledAlpha (:107-125): opacity correction for the box's step length.
float step_corrected_alpha(float base_alpha, float alpha_scale, float normalization_factor) {
  const float scaled = std::clamp(base_alpha * alpha_scale, 0.0f, 1.0f);
  if (normalization_factor <= 0.0f || scaled <= 0.0f) return 0.0f;
  if (scaled >= 1.0f) return 1.0f;
  const double transmittance =
      std::pow(1.0 - static_cast<double>(scaled), static_cast<double>(normalization_factor));
  float alpha = static_cast<float>(1.0 - transmittance);
  if (!std::isfinite(alpha)) alpha = scaled;
  return std::clamp(alpha, 0.0f, 1.0f);
}

struct Vec3d {
  double x = 0.0, y = 0.0, z = 0.0;
};
Vec3d sub(const Vec3d& a, const Vec3d& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
double dot(const Vec3d& a, const Vec3d& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
double norm(const Vec3d& a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
Vec3d cross(const Vec3d& a, const Vec3d& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
Vec3d from(const double v[3]) { return {v[0], v[1], v[2]}; }

// camera::safeNormalize (Common/CameraUtils.hpp:17-23)
Vec3d normalized_or_default(const Vec3d& v) {
  const double len = norm(v);
  if (len > 0.0 && std::isfinite(len)) return {v.x / len, v.y / len, v.z / len};
  return {0.0, 0.0, -1.0};
}

struct CameraBasis {
  Vec3d forward, right, up;
};

// VolumePainter.cpp:631-639
CameraBasis camera_basis(const avr_camera& camera) {
  CameraBasis basis;
  basis.forward = normalized_or_default(sub(from(camera.look_at), from(camera.eye)));
  Vec3d right = cross(basis.forward, from(camera.up));
  const double right_len = norm(right);
  if (right_len > 0.0 && std::isfinite(right_len)) {
    right = {right.x / right_len, right.y / right_len, right.z / right_len};
  } else {
    right = {1.0, 0.0, 0.0};
  }
  basis.right = right;
  basis.up = cross(right, basis.forward);
  return basis;
}

bool is_power_of_two(float v) {
  if (!(v > 0.0f) || !std::isfinite(v)) return false;
  int exponent = 0;
  return std::frexp(v, &exponent) == 0.5f;
}

// kPow2Multiply evaluates the quotient as fma(pos, 1/d, -min/d).  With 1/d = 2^k the exact value
// pos * 2^k - min * 2^k = (pos - min) * 2^k is rounded once, and RN(x * 2^k) = RN(x) * 2^k as
// long as neither side is subnormal or overflows; a difference of two floats that IS subnormal
// is exact, so only the magnitudes have to be bounded: every corner is 0 or in [2^-60, 2^60]
// and so is 1/d.  Anything else (never a real data set) takes the unfused kReciprocal path.
bool fold_is_exact(const BoxDev& dev) {
  auto bounded = [](float v) {
    const float a = std::fabs(v);
    return a == 0.0f || (a >= 0x1p-60f && a <= 0x1p60f);
  };
  for (int axis = 0; axis < 3; ++axis) {
    if (!bounded(dev.minc[axis]) || !bounded(dev.maxc[axis])) return false;
  }
  return bounded(dev.inv_dx) && bounded(dev.inv_dy) && bounded(dev.inv_dz);
}

// Conservative screen rectangle of a box: every pixel whose forward ray can intersect the box
// lies inside.  Pixels outside produce the empty layer pixel (0,0,0,0,+inf) in the reference
// (slab miss, or an intersection interval entirely behind the eye: VolumePainter.cpp:802-837),
// which is an exact identity of the depth-sort blend, so skipping them does not change bits.
// The projected corners are kept with it: the per-row extents are the convex hull of the same
// eight points.
struct Projector {
  CameraBasis basis;
  Vec3d eye;
  double tan_x = 0.0, tan_y = 0.0;
  int width = 0, height = 0;
};

Projector make_projector(const avr_camera& camera, float tan_half_fov, float aspect, int width,
                         int height) {
  Projector pr;
  pr.basis = camera_basis(camera);
  pr.eye = from(camera.eye);
  pr.tan_y = static_cast<double>(tan_half_fov);
  pr.tan_x = pr.tan_y * static_cast<double>(aspect);
  pr.width = width;
  pr.height = height;
  return pr;
}

// (the float tangent and aspect the kernel's ray set-up uses)
Projector make_projector(const avr_camera& camera, int width, int height) {
  return make_projector(camera, std::tan(camera.fov_y_degrees * 0.5f * kPi / 180.0f),
                        static_cast<float>(width) / static_cast<float>(std::max(height, 1)), width,
                        height);
}

void project_box(const avr_box& box, const Projector& pr, BoxFootprint* out) {
  double lo_x = std::numeric_limits<double>::infinity(), hi_x = -lo_x;
  double lo_y = lo_x, hi_y = -lo_x;
  bool full = false;
  // Corners are tested after the float cast the kernel applies (VolumePainter.cpp:658-665).
  for (int corner = 0; corner < 8 && !full; ++corner) {
    const Vec3d p{
        static_cast<double>(static_cast<float>((corner & 1) ? box.max_corner[0] : box.min_corner[0])),
        static_cast<double>(static_cast<float>((corner & 2) ? box.max_corner[1] : box.min_corner[1])),
        static_cast<double>(static_cast<float>((corner & 4) ? box.max_corner[2] : box.min_corner[2]))};
    const Vec3d v = sub(p, pr.eye);
    const double depth = dot(v, pr.basis.forward);
    if (!(depth > 1e-6) || !std::isfinite(depth)) {
      full = true;
      break;
    }
    const double ndc_x = dot(v, pr.basis.right) / (depth * pr.tan_x);
    const double ndc_y = dot(v, pr.basis.up) / (depth * pr.tan_y);
    const double px = (ndc_x + 1.0) * 0.5 * pr.width - 0.5;
    const double py = (ndc_y + 1.0) * 0.5 * pr.height - 0.5;
    if (!std::isfinite(px) || !std::isfinite(py)) {
      full = true;
      break;
    }
    out->px[corner] = px;
    out->py[corner] = py;
    lo_x = std::min(lo_x, px);
    hi_x = std::max(hi_x, px);
    lo_y = std::min(lo_y, py);
    hi_y = std::max(hi_y, py);
  }
  int32_t* rect = out->rect;
  out->whole = full;
  if (full) {  // the box reaches behind the eye: the whole image, on every row
    rect[0] = 0;
    rect[1] = 0;
    rect[2] = pr.width - 1;
    rect[3] = pr.height - 1;
    return;
  }
  constexpr double margin = 2.0;  // pixels; float rounding of the ray setup is << 1 pixel
  const double x0 = std::floor(lo_x - margin), x1 = std::ceil(hi_x + margin);
  const double y0 = std::floor(lo_y - margin), y1 = std::ceil(hi_y + margin);
  if (x1 < 0.0 || y1 < 0.0 || x0 > pr.width - 1.0 || y0 > pr.height - 1.0) {
    rect[0] = 0;
    rect[1] = 0;
    rect[2] = -1;
    rect[3] = -1;
    return;
  }
  rect[0] = static_cast<int32_t>(std::max(x0, 0.0));
  rect[1] = static_cast<int32_t>(std::max(y0, 0.0));
  rect[2] = static_cast<int32_t>(std::min(x1, pr.width - 1.0));
  rect[3] = static_cast<int32_t>(std::min(y1, pr.height - 1.0));
}

void screen_rect(const avr_box& box, const avr_camera& camera, const CameraBasis& basis,
                 const FrameConsts& fc, int32_t rect[4]) {
  Projector pr;
  pr.basis = basis;
  pr.eye = from(camera.eye);
  pr.tan_y = static_cast<double>(fc.tan_half_fov);
  pr.tan_x = pr.tan_y * static_cast<double>(fc.aspect);
  pr.width = fc.width;
  pr.height = fc.height;
  BoxFootprint fp;
  project_box(box, pr, &fp);
  std::copy(fp.rect, fp.rect + 4, rect);
}

}  // namespace

// Conservative per-row extent of a box on screen: for rows of its conservative rectangle
// (box_screen_rect) the pixel columns its projection can touch.  The projection of a box in front
// of the eye is the convex hull of its projected corners; per row the hull's x-extent over the
// band of that row (+- the same 2-pixel margin as the rectangle) is taken from the hull's edges.
// A box that reaches behind the eye keeps its whole rectangle on every row.
//
// Cost matters: for N > 1 this runs for every box of every frame whose camera is new (the
// exchange layout of a frame plan is tightened when the plan is made).  Every hull edge visits
// only the rows whose band it reaches (two chains: ~2 edge visits per row instead of the 28
// corner-to-corner segments' ~14), rows the caller does not need are skipped (RowSet), and the
// interior rows of an edge -- both band boundaries cut it -- are two multiply-adds each.
namespace {

// Andrew's monotone chain over (y, x); collinear points are dropped.  Returns the hull's size
// (counter-clockwise, 1 or 2 for degenerate inputs).
int convex_hull8(const double* px, const double* py, int hull[16]) {
  int idx[8] = {0, 1, 2, 3, 4, 5, 6, 7};
  std::sort(idx, idx + 8, [&](int a, int b) {
    return py[a] < py[b] || (py[a] == py[b] && px[a] < px[b]);
  });
  auto cross = [&](int o, int a, int b) {
    return (px[a] - px[o]) * (py[b] - py[o]) - (py[a] - py[o]) * (px[b] - px[o]);
  };
  int n = 0;
  for (int i = 0; i < 8; ++i) {
    while (n >= 2 && cross(hull[n - 2], hull[n - 1], idx[i]) <= 0.0) --n;
    hull[n++] = idx[i];
  }
  const int lower = n + 1;
  for (int i = 6; i >= 0; --i) {
    while (n >= lower && cross(hull[n - 2], hull[n - 1], idx[i]) <= 0.0) --n;
    hull[n++] = idx[i];
  }
  return std::max(n - 1, 1);  // the last point repeats the first
}

constexpr double kRowMargin = 2.0;  // pixels, as screen_rect

}  // namespace

void merge_footprint_rows(const BoxFootprint& fp, const RowSet& rows, int y_base, int32_t* out_x0,
                          int32_t* out_x1) {
  const int32_t* rect = fp.rect;
  if (rect[2] < rect[0] || rect[3] < rect[1]) return;
  const int y_lo = std::max(rect[1], rows.lo), y_hi = std::min(rect[3], rows.hi);
  if (y_hi < y_lo) return;
  auto merge = [&](int y, int32_t x0, int32_t x1) {
    if (x1 < x0) return;
    int32_t& m0 = out_x0[y - y_base];
    int32_t& m1 = out_x1[y - y_base];
    if (m1 < m0) {
      m0 = x0;
      m1 = x1;
    } else {
      m0 = std::min(m0, x0);
      m1 = std::max(m1, x1);
    }
  };
  if (fp.whole) {  // the whole rectangle on every needed row
    for_rows(rows, y_lo, y_hi, [&](int a, int b) {
      for (int y = a; y <= b; ++y) merge(y, rect[0], rect[2]);
    });
    return;
  }
  const double* px = fp.px;
  const double* py = fp.py;
  // per needed row the extent of the hull's part inside the row's band [y - 0.5 - m, y + 0.5 + m]
  const int n_rows = y_hi - y_lo + 1;
  thread_local std::vector<double> lo_scratch, hi_scratch;
  if (lo_scratch.size() < static_cast<size_t>(n_rows)) {
    lo_scratch.resize(static_cast<size_t>(n_rows));
    hi_scratch.resize(static_cast<size_t>(n_rows));
  }
  double* lo = lo_scratch.data();
  double* hi = hi_scratch.data();
  for_rows(rows, y_lo, y_hi, [&](int a, int b) {
    for (int y = a; y <= b; ++y) {
      lo[y - y_lo] = std::numeric_limits<double>::infinity();
      hi[y - y_lo] = -std::numeric_limits<double>::infinity();
    }
  });
  int hull[16];
  const int n_hull = convex_hull8(px, py, hull);
  const double half = 0.5 + kRowMargin;
  for (int e = 0; e < n_hull; ++e) {
    double ya = py[hull[e]], yb = py[hull[(e + 1) % n_hull]];
    double xa = px[hull[e]], xb = px[hull[(e + 1) % n_hull]];
    if (ya > yb) {
      std::swap(ya, yb);
      std::swap(xa, xb);
    }
    // rows whose band the edge reaches, clamped in double before the cast (a box just in front
    // of the eye can project far outside any integer range)
    const double first = std::max(std::ceil(ya - half), static_cast<double>(y_lo));
    const double last = std::min(std::floor(yb + half), static_cast<double>(y_hi));
    if (!(first <= last)) continue;
    const int r_first = static_cast<int>(first), r_last = static_cast<int>(last);
    const double slope = (yb > ya) ? (xb - xa) / (yb - ya) : 0.0;
    // interior rows: the band's lower boundary lies above ya and its upper boundary below yb, so
    // both ends are interpolated:  ya < y - half  and  y + half < yb.  (Clamped in double before
    // the cast: the corners of a box far off screen are unbounded.)
    const int in_first = static_cast<int>(
        std::clamp(std::floor(ya + half) + 1.0, static_cast<double>(r_first), r_last + 1.0));
    const int in_last = static_cast<int>(
        std::clamp(std::ceil(yb - half) - 1.0, r_first - 1.0, static_cast<double>(r_last)));
    auto generic = [&](int y) {
      const double band_lo = static_cast<double>(y) - half, band_hi = static_cast<double>(y) + half;
      if (yb < band_lo || ya > band_hi) return;
      double x_first = xa, x_last = xb;
      if (yb > ya) {
        if (ya < band_lo) x_first = xa + (band_lo - ya) * slope;
        if (yb > band_hi) x_last = xa + (band_hi - ya) * slope;
      }
      lo[y - y_lo] = std::min(lo[y - y_lo], std::min(x_first, x_last));
      hi[y - y_lo] = std::max(hi[y - y_lo], std::max(x_first, x_last));
    };
    for_rows(rows, r_first, r_last, [&](int a, int b) {
      int y = a;
      for (const int end = std::min(b, in_first - 1); y <= end; ++y) generic(y);
      for (const int end = std::min(b, in_last); y <= end; ++y) {
        // the expressions of `generic` with both ends interpolated
        const double x_first = xa + ((static_cast<double>(y) - half) - ya) * slope;
        const double x_last = xa + ((static_cast<double>(y) + half) - ya) * slope;
        lo[y - y_lo] = std::min(lo[y - y_lo], std::min(x_first, x_last));
        hi[y - y_lo] = std::max(hi[y - y_lo], std::max(x_first, x_last));
      }
      for (; y <= b; ++y) generic(y);
    });
  }
  for_rows(rows, y_lo, y_hi, [&](int a, int b) {
    for (int y = a; y <= b; ++y) {
      const double l = lo[y - y_lo], h = hi[y - y_lo];
      if (!(h >= l)) continue;  // the hull misses the band: nothing on the row
      const double x0 = std::max(std::floor(l - kRowMargin), static_cast<double>(rect[0]));
      const double x1 = std::min(std::ceil(h + kRowMargin), static_cast<double>(rect[2]));
      if (x1 < x0) continue;
      merge(y, static_cast<int32_t>(x0), static_cast<int32_t>(x1));
    }
  });
}

void box_footprints(const avr_box* boxes, int n_boxes, const avr_camera& camera, int width,
                    int height, BoxFootprint* out) {
  const Projector pr = make_projector(camera, width, height);
  for (int b = 0; b < n_boxes; ++b) {
    if (boxes[b].dims[0] <= 0 || boxes[b].dims[1] <= 0 || boxes[b].dims[2] <= 0) {
      out[b].whole = false;
      out[b].rect[0] = out[b].rect[1] = 0;
      out[b].rect[2] = out[b].rect[3] = -1;
      continue;
    }
    project_box(boxes[b], pr, &out[b]);
  }
}

void box_row_spans(const avr_box& box, const avr_camera& camera, int width, int height,
                   const int32_t rect[4], std::vector<int32_t>* row_x0,
                   std::vector<int32_t>* row_x1) {
  row_x0->clear();
  row_x1->clear();
  if (rect[2] < rect[0] || rect[3] < rect[1]) return;
  const int rows = rect[3] - rect[1] + 1;
  row_x0->assign(static_cast<size_t>(rows), 0);
  row_x1->assign(static_cast<size_t>(rows), -1);
  BoxFootprint fp;
  box_footprints(&box, 1, camera, width, height, &fp);
  merge_footprint_rows(fp, RowSet{}, rect[1], row_x0->data(), row_x1->data());
}

void box_screen_rect(const avr_box& box, const avr_camera& camera, int width, int height,
                     int32_t rect[4]) {
  BoxFootprint fp;
  box_footprints(&box, 1, camera, width, height, &fp);
  std::copy(fp.rect, fp.rect + 4, rect);
}

void build_color_table(float alpha_scale, float normalization_factor, const float scalar_range[2],
                       const avr_colormap_point* colormap, int colormap_count, float* out_table) {
  TransferFunction tf;
  tf.clamp_ends = true;
  const float range_min = scalar_range[0];
  const float range_max = scalar_range[1];
  const float range_span = range_max - range_min;

  if (colormap != nullptr && colormap_count > 0) {
    tf.lab = true;
    tf.nan_color = {1.0f, 0.0f, 0.0f, 1.0f};
    for (int i = 0; i < colormap_count; ++i) {
      const avr_colormap_point& pt = colormap[i];
      tf.add(ColorStop{pt.value, std::clamp(pt.red, 0.0f, 1.0f), std::clamp(pt.green, 0.0f, 1.0f),
                       std::clamp(pt.blue, 0.0f, 1.0f)});
      tf.add(OpacityStop{pt.value,
                         step_corrected_alpha(pt.alpha, alpha_scale, normalization_factor), 0.5f,
                         0.0f});
    }
  } else {
    tf.lab = false;
    tf.nan_color = {0.25f, 0.0f, 0.0f, 1.0f};
    // default "jet" colour stops and opacity ramp (:471-487)
    static const std::array<ColorStop, 7> kJet = {{
        {0.0f, 0.0f, 0.0f, 0.5625f},
        {0.111111f, 0.0f, 0.0f, 1.0f},
        {0.3650795f, 0.0f, 1.0f, 1.0f},
        {0.4920635f, 0.5f, 1.0f, 0.5f},
        {0.6190475f, 1.0f, 1.0f, 0.0f},
        {0.873016f, 1.0f, 0.0f, 0.0f},
        {1.0f, 0.5f, 0.0f, 0.0f},
    }};
    for (const ColorStop& stop : kJet) tf.add(stop);
    static const std::array<float, 6> kRampAt = {0.0f, 0.15f, 0.35f, 0.6f, 0.85f, 1.0f};
    static const std::array<float, 6> kRampAlpha = {0.05f, 0.15f, 0.22f, 0.3f, 0.38f, 0.5f};
    for (std::size_t i = 0; i < kRampAt.size(); ++i) {
      tf.add(OpacityStop{kRampAt[i] * range_span + range_min,
                         step_corrected_alpha(kRampAlpha[i], alpha_scale, normalization_factor),
                         0.5f, 0.0f});
    }
    tf.rescale(range_min, range_max);
  }

  for (int i = 0; i < kTableSize; ++i) {
    const float t = static_cast<float>(i) / static_cast<float>(kTableSize - 1);
    const float value = range_min + range_span * t;
    Rgba entry = tf.color_at(value);
    entry.a = tf.opacity_at(value);
    out_table[i * 4 + 0] = entry.r;
    out_table[i * 4 + 1] = entry.g;
    out_table[i * 4 + 2] = entry.b;
    out_table[i * 4 + 3] = entry.a;
  }
}

void box_sampling(const avr_box& box, const avr_paint_params& params, float* sample_distance,
                  float* normalization_factor, float* alpha_scale) {
  double spacing[3] = {0.0, 0.0, 0.0};
  for (int axis = 0; axis < 3; ++axis) {
    if (box.dims[axis] > 0) {
      spacing[axis] =
          (box.max_corner[axis] - box.min_corner[axis]) / static_cast<double>(box.dims[axis]);
    }
  }
  float min_spacing = std::numeric_limits<float>::max();
  for (int axis = 0; axis < 3; ++axis) {
    const float value = static_cast<float>(spacing[axis]);
    if (value > 0.0f && value < min_spacing && std::isfinite(value)) min_spacing = value;
  }
  if (!(min_spacing > 0.0f && std::isfinite(min_spacing))) {
    const float fallback = static_cast<float>(std::min({params.bounds_max[0] - params.bounds_min[0],
                                                        params.bounds_max[1] - params.bounds_min[1],
                                                        params.bounds_max[2] - params.bounds_min[2]}));
    min_spacing = std::max(1e-4f, fallback * 0.01f);
  }
  const float step = std::max(min_spacing * 0.5f, 1e-5f);
  float reference = params.reference_sample_distance;
  if (!(reference > 0.0f && std::isfinite(reference))) reference = step;
  float factor = step / reference;
  if (!std::isfinite(factor)) factor = 1.0f;
  factor = std::max(factor, 0.0f);
  *sample_distance = step;
  *normalization_factor = factor;
  *alpha_scale = std::clamp(1.0f - params.box_transparency, 0.0f, 1.0f);
}

float box_depth_hint(const avr_box& box, const avr_camera& camera) {
  const Vec3d eye = from(camera.eye);
  const Vec3d view = normalized_or_default(sub(from(camera.look_at), eye));
  float nearest = std::numeric_limits<float>::infinity();
  for (int corner = 0; corner < 8; ++corner) {
    const Vec3d p{(corner & 1) ? box.max_corner[0] : box.min_corner[0],
                  (corner & 2) ? box.max_corner[1] : box.min_corner[1],
                  (corner & 4) ? box.max_corner[2] : box.min_corner[2]};
    nearest = std::min(nearest, static_cast<float>(dot(sub(p, eye), view)));
  }
  return nearest;
}

float reference_sample_distance(const avr_box* boxes, int n_boxes, const double bounds_min[3],
                                const double bounds_max[3]) {
  float coarsest = 0.0f;
  for (int b = 0; b < n_boxes; ++b) {
    double spacing[3] = {0.0, 0.0, 0.0};
    for (int axis = 0; axis < 3; ++axis) {
      if (boxes[b].dims[axis] > 0) {
        // the reference divides the double span by float(dim) here (VolumeRenderer.cpp:1143-1151)
        spacing[axis] = (boxes[b].max_corner[axis] - boxes[b].min_corner[axis]) /
                        static_cast<float>(boxes[b].dims[axis]);
      }
    }
    float min_spacing = std::numeric_limits<float>::max();
    for (int axis = 0; axis < 3; ++axis) {
      if (spacing[axis] > 0.0f && spacing[axis] < min_spacing && std::isfinite(spacing[axis])) {
        min_spacing = static_cast<float>(spacing[axis]);
      }
    }
    if (min_spacing > 0.0f && std::isfinite(min_spacing)) coarsest = std::max(coarsest, min_spacing);
  }
  if (!(coarsest > 0.0f && std::isfinite(coarsest))) {
    float fallback = std::numeric_limits<float>::max();
    for (int axis = 0; axis < 3; ++axis) {
      const float length = static_cast<float>(bounds_max[axis] - bounds_min[axis]);
      if (length > 0.0f && std::isfinite(length)) fallback = std::min(fallback, length);
    }
    if (!(fallback > 0.0f && std::isfinite(fallback))) fallback = 1.0f;
    coarsest = std::max(1e-4f, fallback * 0.01f);
  }
  return std::max(coarsest * 0.5f, 1e-5f);
}

namespace {

// Morton index of super-tile (sx, sy): the tile sequence number of its first tile / 4.
uint32_t interleave16(uint32_t v) {
  v &= 0xffffu;
  v = (v | (v << 8)) & 0x00ff00ffu;
  v = (v | (v << 4)) & 0x0f0f0f0fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}

}  // namespace

// Items are ordered by an estimate of their march cost (rays x samples per ray of the run's
// boxes projecting onto the super-tile), most expensive first.  The order only decides WHEN a
// workgroup runs (heavy ones first, so the cheap ones fill the tail), never what it computes.
void build_march_items(const FramePlan& plan, const int32_t* box_order, const int32_t* run_end,
                       int n_runs, const std::vector<RunRectDev>& run_rects,
                       std::vector<MarchItemDev>* items) {
  const int span = kTile * kSuperTileSide;  // pixels per super-tile side
  items->clear();
  std::vector<float> cost;       // per item
  std::vector<float> run_cost;   // dense over the current run's super-tile rectangle
  int position = 0;
  for (int run = 0; run < n_runs; ++run) {
    const int end = run_end[run];
    const RunRectDev& rr = run_rects[static_cast<size_t>(run)];
    if (rr.x1 < rr.x0 || rr.y1 < rr.y0) {
      position = end;
      continue;
    }
    const int sx0 = rr.x0 / span, sx1 = rr.x1 / span, sy0 = rr.y0 / span, sy1 = rr.y1 / span;
    const int nsx = sx1 - sx0 + 1;
    run_cost.assign(static_cast<size_t>(nsx) * (sy1 - sy0 + 1), 0.0f);
    for (; position < end; ++position) {
      const BoxDev& box = plan.boxes[static_cast<size_t>(box_order[position])];
      if (box.rect[2] < box.rect[0] || box.rect[3] < box.rect[1]) continue;
      const float ex = box.maxc[0] - box.minc[0];
      const float ey = box.maxc[1] - box.minc[1];
      const float ez = box.maxc[2] - box.minc[2];
      const float steps = std::sqrt(ex * ex + ey * ey + ez * ez) / std::max(box.sample_dist, 1e-30f);
      if (!std::isfinite(steps)) continue;
      const int bx0 = std::max(box.rect[0], rr.x0), bx1 = std::min(box.rect[2], rr.x1);
      const int by0 = std::max(box.rect[1], rr.y0), by1 = std::min(box.rect[3], rr.y1);
      for (int sy = by0 / span; sy <= by1 / span; ++sy) {
        const int y0 = std::max(by0, sy * span), y1 = std::min(by1, sy * span + span - 1);
        for (int sx = bx0 / span; sx <= bx1 / span; ++sx) {
          const int x0 = std::max(bx0, sx * span), x1 = std::min(bx1, sx * span + span - 1);
          run_cost[static_cast<size_t>(sy - sy0) * nsx + (sx - sx0)] +=
              steps * static_cast<float>((x1 - x0 + 1) * (y1 - y0 + 1));
        }
      }
    }
    for (int sy = sy0; sy <= sy1; ++sy) {
      for (int sx = sx0; sx <= sx1; ++sx) {
        const uint32_t slot = interleave16(static_cast<uint32_t>(sx)) |
                              (interleave16(static_cast<uint32_t>(sy)) << 1);
        items->push_back(MarchItemDev{slot, static_cast<uint32_t>(run)});
        cost.push_back(run_cost[static_cast<size_t>(sy - sy0) * nsx + (sx - sx0)]);
      }
    }
  }
  // sort by cost, most expensive first (ties: generation order = run, then rows of super-tiles)
  std::vector<uint32_t> index(items->size());
  for (uint32_t i = 0; i < index.size(); ++i) index[i] = i;
  std::stable_sort(index.begin(), index.end(),
                   [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
  std::vector<MarchItemDev> sorted;
  sorted.reserve(((items->size() + kXcds - 1) / kXcds) * kXcds);
  for (uint32_t i : index) sorted.push_back((*items)[i]);
  while (sorted.size() % kXcds != 0) sorted.push_back(MarchItemDev{kNoMarchItem, 0});
  items->swap(sorted);
}

void plan_cells(const avr_box* boxes, int n_boxes, const avr_scalar_transform& transform,
                FramePlan* plan) {
  // the statistics kernels only need the cell addressing part of the frame plan; a fixed
  // dummy view keeps plan_frame's validation of strides and sizes
  avr_paint_params params{};
  params.width = 1;
  params.height = 1;
  params.scalar_range[0] = 0.0f;
  params.scalar_range[1] = 1.0f;
  for (int c = 0; c < 3; ++c) {
    params.bounds_min[c] = 0.0;
    params.bounds_max[c] = 1.0;
  }
  avr_camera camera{};
  camera.eye[2] = 1.0;
  camera.up[1] = 1.0;
  camera.fov_y_degrees = 45.0f;
  camera.near_plane = 0.1f;
  camera.far_plane = 10.0f;
  plan_frame(boxes, n_boxes, transform, params, camera, plan);
}

void scene_transform_from_stats(const double stats[3], int64_t finite_count, bool log_scale,
                                bool normalize_to_data_range, avr_scalar_transform* transform,
                                double processed[2], float processed_range[2],
                                float scalar_range[2]) {
  const double inf = std::numeric_limits<double>::infinity();
  const double original_min = (finite_count > 0) ? stats[0] : inf;
  const double original_max = (finite_count > 0) ? stats[1] : -inf;
  double processed_min = original_min, processed_max = original_max;
  std::memset(transform, 0, sizeof(*transform));
  transform->log_scale_input = log_scale ? 1 : 0;
  if (log_scale) {
    const double positive_min = (stats[2] > 0.0 && std::isfinite(stats[2])) ? stats[2] : inf;
    if (!(positive_min < inf) || !(positive_min > 0.0)) {
      throw std::runtime_error("Log scaling requested but no positive scalar values were found.");
    }
    transform->positive_floor = positive_min;
    processed_min = std::log(positive_min);
    processed_max = std::log(std::max(original_max, positive_min));
  }
  if (!std::isfinite(processed_min) || !std::isfinite(processed_max)) {
    throw std::runtime_error("Scalar data does not have a finite range.");
  }
  if (processed_min == processed_max) processed_max = processed_min + 1.0;
  auto to_range = [](double lo, double hi, float out[2]) {  // makeScalarRange (:106-112)
    if (lo == hi) hi = lo + 1.0;
    out[0] = static_cast<float>(lo);
    out[1] = static_cast<float>(hi);
  };
  to_range(processed_min, processed_max, processed_range);
  processed[0] = processed_min;
  processed[1] = processed_max;
  transform->normalization_min = processed_min;
  transform->inverse_normalization_span = 1.0 / (processed_max - processed_min);
  scalar_range[0] = processed_range[0];
  scalar_range[1] = processed_range[1];
  if (normalize_to_data_range) {  // SetSceneNormalizationRange (:427-443)
    const double span = processed_max - processed_min;
    if (!(span > 0.0) || !std::isfinite(span)) {
      throw std::runtime_error("Failed to establish a finite scalar range for color mapping.");
    }
    transform->normalize_to_unit_range = 1;
    transform->normalization_min = processed_min;
    transform->inverse_normalization_span = 1.0 / span;
    scalar_range[0] = 0.0f;
    scalar_range[1] = 1.0f;
  }
}

void tight_bounds(const avr_box* boxes, int n_boxes, const double fallback_min[3],
                  const double fallback_max[3], double out_min[3], double out_max[3]) {
  if (n_boxes <= 0) {
    for (int c = 0; c < 3; ++c) {
      out_min[c] = fallback_min[c];
      out_max[c] = fallback_max[c];
    }
    return;
  }
  for (int c = 0; c < 3; ++c) {
    double lo = std::numeric_limits<float>::max(), hi = -std::numeric_limits<float>::max();
    for (int b = 0; b < n_boxes; ++b) {
      lo = std::min(lo, boxes[b].min_corner[c]);
      hi = std::max(hi, boxes[b].max_corner[c]);
    }
    out_min[c] = static_cast<double>(static_cast<float>(lo));  // reduced as MPI_FLOAT
    out_max[c] = static_cast<double>(static_cast<float>(hi));
  }
}

void plan_overlay(const double bounds_min[3], const double bounds_max[3], const avr_camera& camera,
                  int sqrt_antialiasing, int width, int height, OverlayPlan* plan) {
  std::memset(plan, 0, sizeof(*plan));
  plan->pixel_radius = 0.5f * static_cast<float>(std::max(sqrt_antialiasing, 1));
  if (width <= 0 || height <= 0) return;
  const float aspect = static_cast<float>(width) / static_cast<float>(std::max(height, 1));
  const CameraBasis basis = camera_basis(camera);
  const float tan_half_fov = std::tan(camera.fov_y_degrees * 0.5f * kPi / 180.0f);
  struct Corner {
    float x = 0.0f, y = 0.0f;
    bool valid = false;
  };
  std::array<Corner, 8> corners;
  const float width_scale = (width > 1) ? static_cast<float>(width - 1) : 0.0f;
  const float height_scale = (height > 1) ? static_cast<float>(height - 1) : 0.0f;
  const Vec3d eye = from(camera.eye);
  for (int index = 0; index < 8; ++index) {
    const Vec3d world{(index & 1) ? bounds_max[0] : bounds_min[0],
                      (index & 2) ? bounds_max[1] : bounds_min[1],
                      (index & 4) ? bounds_max[2] : bounds_min[2]};
    const Vec3d relative = sub(world, eye);
    const float depth = static_cast<float>(dot(relative, basis.forward));
    if (!(depth > 0.0f) || !std::isfinite(depth)) continue;
    const float x_cam = static_cast<float>(dot(relative, basis.right));
    const float y_cam = static_cast<float>(dot(relative, basis.up));
    const float ndc_x = x_cam / (depth * tan_half_fov * aspect);
    const float ndc_y = y_cam / (depth * tan_half_fov);
    if (!std::isfinite(ndc_x) || !std::isfinite(ndc_y)) continue;
    corners[static_cast<std::size_t>(index)].x = (ndc_x * 0.5f + 0.5f) * width_scale;
    corners[static_cast<std::size_t>(index)].y = (ndc_y * 0.5f + 0.5f) * height_scale;
    corners[static_cast<std::size_t>(index)].valid = true;
  }
  static constexpr int kEdges[12][2] = {{0, 1}, {1, 3}, {3, 2}, {2, 0}, {4, 5}, {5, 7},
                                        {7, 6}, {6, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
  const float influence_radius = plan->pixel_radius + 0.5f;
  for (const auto& pair : kEdges) {
    const Corner& start = corners[static_cast<std::size_t>(pair[0])];
    const Corner& end = corners[static_cast<std::size_t>(pair[1])];
    if (!start.valid || !end.valid) continue;
    OverlayEdge& edge = plan->edges[plan->n_edges];
    edge.sx = start.x;
    edge.sy = start.y;
    edge.ex = end.x;
    edge.ey = end.y;
    edge.dx = end.x - start.x;
    edge.dy = end.y - start.y;
    edge.len_sq = edge.dx * edge.dx + edge.dy * edge.dy;
    if (!(edge.len_sq > 0.0f)) {
      const int px = static_cast<int>(std::lround(start.x));
      const int py = static_cast<int>(std::lround(start.y));
      if (px < 0 || px >= width || py < 0 || py >= height) continue;
      edge.point = 1;
      edge.x_begin = edge.x_end = px;
      edge.y_begin = edge.y_end = py;
    } else {
      const float min_x = std::min(start.x, end.x) - influence_radius;
      const float max_x = std::max(start.x, end.x) + influence_radius;
      const float min_y = std::min(start.y, end.y) - influence_radius;
      const float max_y = std::max(start.y, end.y) + influence_radius;
      edge.x_begin = std::max(0, static_cast<int>(std::floor(min_x)));
      edge.x_end = std::min(width - 1, static_cast<int>(std::ceil(max_x)));
      edge.y_begin = std::max(0, static_cast<int>(std::floor(min_y)));
      edge.y_end = std::min(height - 1, static_cast<int>(std::ceil(max_y)));
    }
    ++plan->n_edges;
  }
}

int layer_order(const float* hints, const int32_t* owner, const int32_t* local_index, int n_layers,
                int32_t* order_out, int32_t* run_end_out) {
  std::vector<int32_t> ids(static_cast<std::size_t>(std::max(n_layers, 0)));
  for (int i = 0; i < n_layers; ++i) ids[static_cast<std::size_t>(i)] = i;
  // (depth, owningRank, localIndex) is a strict total order over distinct layers
  // (DirectSendBase.cpp:378-388), so the sort result does not depend on the algorithm.
  std::sort(ids.begin(), ids.end(), [&](int32_t a, int32_t b) {
    if (hints[a] == hints[b]) {
      if (owner[a] == owner[b]) return local_index[a] < local_index[b];
      return owner[a] < owner[b];
    }
    return hints[a] < hints[b];
  });
  int runs = 0;
  int i = 0;
  while (i < n_layers) {
    const int32_t run_owner = owner[ids[static_cast<std::size_t>(i)]];
    while (i < n_layers && owner[ids[static_cast<std::size_t>(i)]] == run_owner) ++i;
    run_end_out[runs++] = i;
  }
  for (int k = 0; k < n_layers; ++k) order_out[k] = ids[static_cast<std::size_t>(k)];
  return runs;
}

void plan_frame(const avr_box* boxes, int n_boxes, const avr_scalar_transform& transform,
                const avr_paint_params& params, const avr_camera& camera, FramePlan* plan) {
  if (params.width <= 0 || params.height <= 0) {
    throw std::invalid_argument("image width and height must be positive");
  }
  if (n_boxes < 0 || (n_boxes > 0 && boxes == nullptr)) {
    throw std::invalid_argument("invalid box list");
  }
  if (static_cast<int64_t>(params.width) * params.height > (int64_t{1} << 31) - 1) {
    throw std::invalid_argument("image has more than 2^31-1 pixels");
  }

  FrameConsts& fc = plan->consts;
  std::memset(&fc, 0, sizeof(fc));
  fc.width = params.width;
  fc.height = params.height;
  fc.aspect = static_cast<float>(params.width) / static_cast<float>(std::max(params.height, 1));
  fc.tan_half_fov = std::tan(camera.fov_y_degrees * 0.5f * kPi / 180.0f);
  fc.inv_width = 1.0f / static_cast<float>(params.width);
  fc.inv_height = 1.0f / static_cast<float>(params.height);

  const CameraBasis basis = camera_basis(camera);
  fc.fwd[0] = static_cast<float>(basis.forward.x);
  fc.fwd[1] = static_cast<float>(basis.forward.y);
  fc.fwd[2] = static_cast<float>(basis.forward.z);
  fc.right[0] = static_cast<float>(basis.right.x);
  fc.right[1] = static_cast<float>(basis.right.y);
  fc.right[2] = static_cast<float>(basis.right.z);
  fc.up[0] = static_cast<float>(basis.up.x);
  fc.up[1] = static_cast<float>(basis.up.y);
  fc.up[2] = static_cast<float>(basis.up.z);
  for (int axis = 0; axis < 3; ++axis) fc.eye[axis] = static_cast<float>(camera.eye[axis]);

  // scalar range -> table index mapping and soft clip (VolumePainter.cpp:717-724)
  fc.range_min = params.scalar_range[0];
  const float range_max = params.scalar_range[1];
  fc.inverse_range = 1.0f;
  if (range_max != fc.range_min) fc.inverse_range = 1.0f / (range_max - fc.range_min);
  fc.clip_start = std::clamp(params.scalar_range[1], 0.0f, 1.0f);
  fc.apply_clip = (1.0f > fc.clip_start + kSoftClipTolerance) ? 1 : 0;

  fc.log_scale = transform.log_scale_input ? 1 : 0;
  fc.normalize = transform.normalize_to_unit_range ? 1 : 0;
  fc.positive_floor = transform.positive_floor;
  fc.norm_min = transform.normalization_min;
  fc.inv_norm_span = transform.inverse_normalization_span;

  plan->boxes.assign(static_cast<std::size_t>(n_boxes), BoxDev{});
  plan->tables.clear();
  plan->n_tables = 0;
  plan->classify_tile_begin.assign(static_cast<std::size_t>(n_boxes) + 1, 0u);
  plan->classified_bytes = 0;
  std::map<uint32_t, int> table_of_factor;  // normalization factor bits -> table slot

  for (int b = 0; b < n_boxes; ++b) {
    const avr_box& box = boxes[b];
    BoxDev& dev = plan->boxes[static_cast<std::size_t>(b)];
    std::memset(&dev, 0, sizeof(dev));
    dev.cls_offset = plan->classified_bytes;
    plan->classify_tile_begin[static_cast<std::size_t>(b) + 1] =
        plan->classify_tile_begin[static_cast<std::size_t>(b)];

    float step = 0.0f, factor = 0.0f, alpha_scale = 0.0f;
    box_sampling(box, params, &step, &factor, &alpha_scale);
    uint32_t factor_bits = 0;
    std::memcpy(&factor_bits, &factor, sizeof(factor_bits));
    auto found = table_of_factor.find(factor_bits);
    if (found == table_of_factor.end()) {
      const int slot = plan->n_tables++;
      plan->tables.resize(static_cast<std::size_t>(plan->n_tables) * kTableSize * 4);
      build_color_table(alpha_scale, factor, params.scalar_range, params.colormap,
                        params.colormap_count,
                        plan->tables.data() + static_cast<std::size_t>(slot) * kTableSize * 4);
      found = table_of_factor.emplace(factor_bits, slot).first;
    }
    dev.lut = found->second;

    for (int axis = 0; axis < 3; ++axis) {
      dev.minc[axis] = static_cast<float>(box.min_corner[axis]);
      dev.maxc[axis] = static_cast<float>(box.max_corner[axis]);
    }
    dev.nx = box.dims[0];
    dev.ny = box.dims[1];
    dev.nz = box.dims[2];
    dev.sample_dist = step;
    dev.cells = box.cells;

    if (dev.nx <= 0 || dev.ny <= 0 || dev.nz <= 0) {
      // the reference clears the layer (VolumePainter.cpp:670-673): never hit
      dev.rect[0] = dev.rect[1] = 0;
      dev.rect[2] = dev.rect[3] = -1;
      dev.dx = dev.dy = dev.dz = 1.0f;
      dev.inv_dx = dev.inv_dy = dev.inv_dz = 1.0f;
      dev.index_mode = kExactDivide;
      continue;
    }
    if (box.cells == nullptr) throw std::invalid_argument("box has no cell data");
    // 32-bit element offsets on the device: the addressed span must stay below 2^28 elements
    const int64_t span = static_cast<int64_t>(dev.nx - 1) +
                         static_cast<int64_t>(dev.ny - 1) * box.jstride +
                         static_cast<int64_t>(dev.nz - 1) * box.kstride;
    if (box.jstride < 0 || box.kstride < 0 || span >= (int64_t{1} << 28)) {
      throw std::invalid_argument("box spans more than 2^28 cells (or has negative strides)");
    }
    dev.jstride = static_cast<int32_t>(box.jstride);
    dev.kstride = static_cast<int32_t>(box.kstride);
    {
      const uint64_t bricks_x = static_cast<uint64_t>((dev.nx + kBrickX - 1) / kBrickX);
      const uint64_t bricks_y = static_cast<uint64_t>((dev.ny + kBrickY - 1) / kBrickY);
      const uint64_t bricks_z = static_cast<uint64_t>((dev.nz + kBrickZ - 1) / kBrickZ);
      const uint64_t chunks = static_cast<uint64_t>((dev.nx + kClassifyChunk - 1) / kClassifyChunk);
      const uint64_t tiles = bricks_y * bricks_z * chunks;
      // 24-bit multiplies in the bricklet address arithmetic of the march
      if (bricks_z * bricks_y * kBrickBytes >= (uint64_t{1} << 24)) {
        throw std::invalid_argument("box cross-section too large (ceil(ny/4)*ceil(nz/4) >= 2^17)");
      }
      const uint64_t total_tiles = plan->classify_tile_begin[static_cast<std::size_t>(b)] + tiles;
      if (total_tiles >= (uint64_t{1} << 31)) throw std::invalid_argument("scene has too many cells");
      plan->classify_tile_begin[static_cast<std::size_t>(b) + 1] = static_cast<uint32_t>(total_tiles);
      plan->classified_bytes += bricks_x * bricks_y * bricks_z * kBrickBytes;
    }

    dev.dx = (dev.maxc[0] - dev.minc[0]) / static_cast<float>(dev.nx);
    dev.dy = (dev.maxc[1] - dev.minc[1]) / static_cast<float>(dev.ny);
    dev.dz = (dev.maxc[2] - dev.minc[2]) / static_cast<float>(dev.nz);
    dev.inv_dx = 1.0f / dev.dx;
    dev.inv_dy = 1.0f / dev.dy;
    dev.inv_dz = 1.0f / dev.dz;
    const bool regular = (dev.dx > 0.0f && dev.dy > 0.0f && dev.dz > 0.0f) &&
                         std::isfinite(dev.dx) && std::isfinite(dev.dy) && std::isfinite(dev.dz) &&
                         std::isfinite(dev.inv_dx) && std::isfinite(dev.inv_dy) &&
                         std::isfinite(dev.inv_dz);
    if (!regular) {
      dev.index_mode = kExactDivide;
    } else if (is_power_of_two(dev.dx) && is_power_of_two(dev.dy) && is_power_of_two(dev.dz) &&
               fold_is_exact(dev)) {
      dev.index_mode = kPow2Multiply;
      dev.nmin_inv[0] = -(dev.minc[0] * dev.inv_dx);
      dev.nmin_inv[1] = -(dev.minc[1] * dev.inv_dy);
      dev.nmin_inv[2] = -(dev.minc[2] * dev.inv_dz);
    } else {
      dev.index_mode = kReciprocal;
    }
    // |q - RN(f/dx)| < 2^-22 * q and q <= n * (1 + 2^-22) for an inside sample; 2^-20 * n
    // leaves a 4x margin (derivation in DESIGN.md, "Exact index without the divide")
    const int longest = std::max(dev.nx, std::max(dev.ny, dev.nz));
    dev.near_tol = std::ldexp(static_cast<float>(std::max(longest, 1)), -20);
    const float ex = dev.maxc[0] - dev.minc[0];
    const float ey = dev.maxc[1] - dev.minc[1];
    const float ez = dev.maxc[2] - dev.minc[2];
    dev.mesh_eps = std::sqrt(ex * ex + ey * ey + ez * ez) * 0.0001f;
    screen_rect(box, camera, basis, fc, dev.rect);
  }
  plan->ready = true;
}

}  // namespace avr
