/*
 * avr_oracle.c -- CPU restatement of the amrVolumeRenderer hot path.  TEST INFRASTRUCTURE ONLY
 * (see avr_oracle.h).  Plain C99, serial arithmetic per pixel; OpenMP only splits rows.
 *
 * Build: gcc -O2 -std=gnu99 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 * x86-64 baseline: SSE2 scalar float math, no FMA, IEEE divide/sqrt -- the arithmetic the
 * reference's CPU build performs.
 */
#include "avr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define TABLE_SIZE 256             /* kColorTableSize, VolumePainter.cpp:35 */
static const float kSoftClipTolerance = 1e-5f;                /* VolumePainter.cpp:36 */
static const float kPi = 3.14159265358979323846f;             /* VolumePainter.cpp:37 */

/* 0 = the restatement of the reference (always, except in tools/appb_hash_search.py, which
 * enumerates what an AMReX header shim could have done differently from AMReX 26.04) */
static int orc_shim_variant = 0;
void orc_set_shim_variant(int variant) { orc_shim_variant = variant; }

/* ---- std:: helpers with libstdc++ semantics (NaN handling matters) ---------------------- */
static inline float clampf(float v, float lo, float hi) { /* std::clamp */
  return (v < lo) ? lo : ((hi < v) ? hi : v);
}
static inline float maxf(float a, float b) { return (a < b) ? b : a; } /* std::max */
static inline float minf(float a, float b) { return (b < a) ? b : a; } /* std::min */
static inline double mind(double a, double b) { return (b < a) ? b : a; }

/* ---- amrex::RealVect (double) restated: length, cross, dot in AMReX's term order -------- */
static double v3_len(const double v[3]) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
static double v3_dot(const double a[3], const double b[3]) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
static void v3_cross(const double a[3], const double b[3], double out[3]) {
  out[0] = a[1] * b[2] - a[2] * b[1];
  out[1] = a[2] * b[0] - a[0] * b[2];
  out[2] = a[0] * b[1] - a[1] * b[0];
}
/* camera::safeNormalize, Common/CameraUtils.hpp:17-23 */
static void safe_normalize(const double in[3], double out[3]) {
  const double len = v3_len(in);
  if (len > 0.0 && isfinite(len)) {
    out[0] = in[0] / len;
    out[1] = in[1] / len;
    out[2] = in[2] / len;
  } else {
    out[0] = 0.0;
    out[1] = 0.0;
    out[2] = -1.0;
  }
}

/* Camera basis as VolumePainter::paint builds it (VolumePainter.cpp:631-656). */
static void camera_basis(const orc_camera *cam, float fwd[3], float right[3], float up[3],
                         float eye[3]) {
  double d[3] = {cam->look_at[0] - cam->eye[0], cam->look_at[1] - cam->eye[1],
                 cam->look_at[2] - cam->eye[2]};
  double f[3], r[3], u[3];
  safe_normalize(d, f);
  v3_cross(f, cam->up, r);
  const double rl = v3_len(r);
  if (rl > 0.0 && isfinite(rl)) {
    r[0] /= rl;
    r[1] /= rl;
    r[2] /= rl;
  } else {
    r[0] = 1.0;
    r[1] = 0.0;
    r[2] = 0.0;
  }
  v3_cross(r, f, u);
  for (int c = 0; c < 3; ++c) {
    fwd[c] = (float)f[c];
    right[c] = (float)r[c];
    up[c] = (float)u[c];
    eye[c] = (float)cam->eye[c];
  }
}

/* ========================================================================================= */
/* Transfer-function table (VolumePainter.cpp:107-125, 127-516)                               */
/* ========================================================================================= */

typedef struct { float r, g, b, a; } tf_entry;
typedef struct { float value, r, g, b; } color_node;
typedef struct { float value, alpha, midpoint, sharpness; } opacity_node;

#define MAX_NODES 1024
typedef struct {
  int lab;               /* ColorSpace::kLab */
  int n_colors, n_opacity;
  color_node colors[MAX_NODES];
  opacity_node opacity[MAX_NODES];
  tf_entry nan_color, below, above;
  int use_clamping;
} tf_spec;

/* computeScaledAlpha, VolumePainter.cpp:107-125 */
static float scaled_alpha(float base_alpha, float alpha_scale, float norm_factor) {
  const float scaled_base = clampf(base_alpha * alpha_scale, 0.0f, 1.0f);
  if (norm_factor <= 0.0f || scaled_base <= 0.0f) return 0.0f;
  if (scaled_base >= 1.0f) return 1.0f;
  const double transmittance = pow(1.0 - (double)scaled_base, (double)norm_factor);
  float a = (float)(1.0 - transmittance);
  if (!isfinite(a)) a = scaled_base;
  return clampf(a, 0.0f, 1.0f);
}

/* insertColorNode / insertOpacityNode (VolumePainter.cpp:127-151): sorted insert by value,
 * replace when an equal value already exists (lower_bound semantics). */
static void insert_color(tf_spec *t, color_node n) {
  int pos = 0;
  while (pos < t->n_colors && t->colors[pos].value < n.value) ++pos;
  if (pos < t->n_colors && t->colors[pos].value == n.value) {
    t->colors[pos] = n;
    return;
  }
  if (t->n_colors >= MAX_NODES) return;
  memmove(&t->colors[pos + 1], &t->colors[pos], sizeof(color_node) * (size_t)(t->n_colors - pos));
  t->colors[pos] = n;
  t->n_colors++;
}
static void insert_opacity(tf_spec *t, opacity_node n) {
  int pos = 0;
  while (pos < t->n_opacity && t->opacity[pos].value < n.value) ++pos;
  if (pos < t->n_opacity && t->opacity[pos].value == n.value) {
    t->opacity[pos] = n;
    return;
  }
  if (t->n_opacity >= MAX_NODES) return;
  memmove(&t->opacity[pos + 1], &t->opacity[pos],
          sizeof(opacity_node) * (size_t)(t->n_opacity - pos));
  t->opacity[pos] = n;
  t->n_opacity++;
}

/* getTableRange + rescaleTableToRange, VolumePainter.cpp:153-200 */
static void rescale_table(tf_spec *t, float range_min, float range_max) {
  int has = 0;
  float lo = 0.0f, hi = 0.0f;
  for (int i = 0; i < t->n_colors; ++i) {
    const float v = t->colors[i].value;
    if (!has) { lo = hi = v; has = 1; continue; }
    lo = minf(lo, v);
    hi = maxf(hi, v);
  }
  for (int i = 0; i < t->n_opacity; ++i) {
    const float v = t->opacity[i].value;
    if (!has) { lo = hi = v; has = 1; continue; }
    lo = minf(lo, v);
    hi = maxf(hi, v);
  }
  if (!has) { lo = 0.0f; hi = 0.0f; }
  const float old_span = hi - lo;
  const float new_span = range_max - range_min;
  if (!(old_span > 0.0f) || !(new_span > 0.0f)) return;
  for (int i = 0; i < t->n_colors; ++i) {
    const float u = (t->colors[i].value - lo) / old_span;
    t->colors[i].value = range_min + new_span * u;
  }
  for (int i = 0; i < t->n_opacity; ++i) {
    const float u = (t->opacity[i].value - lo) / old_span;
    t->opacity[i].value = range_min + new_span * u;
  }
}

/* rgbToLab, VolumePainter.cpp:202-256 (float pow = powf) */
static tf_entry rgb_to_lab(tf_entry rgb) {
  float r = rgb.r, g = rgb.g, b = rgb.b;
  r = (r > 0.04045f) ? powf((r + 0.055f) / 1.055f, 2.4f) : r / 12.92f;
  g = (g > 0.04045f) ? powf((g + 0.055f) / 1.055f, 2.4f) : g / 12.92f;
  b = (b > 0.04045f) ? powf((b + 0.055f) / 1.055f, 2.4f) : b / 12.92f;

  const float x = r * 0.4124f + g * 0.3576f + b * 0.1805f;
  const float y = r * 0.2126f + g * 0.7152f + b * 0.0722f;
  const float z = r * 0.0193f + g * 0.1192f + b * 0.9505f;

  const float one_third = 1.0f / 3.0f;
  const float sixteen_over_116 = 16.0f / 116.0f;
  float vx = x / 0.9505f;
  float vy = y / 1.0f;
  float vz = z / 1.089f;
  vx = (vx > 0.008856f) ? powf(vx, one_third) : (7.787f * vx) + sixteen_over_116;
  vy = (vy > 0.008856f) ? powf(vy, one_third) : (7.787f * vy) + sixteen_over_116;
  vz = (vz > 0.008856f) ? powf(vz, one_third) : (7.787f * vz) + sixteen_over_116;

  tf_entry lab;
  lab.r = (116.0f * vy) - 16.0f;
  lab.g = 500.0f * (vx - vy);
  lab.b = 200.0f * (vy - vz);
  lab.a = rgb.a;
  return lab;
}

/* labToRgb, VolumePainter.cpp:258-320 */
static tf_entry lab_to_rgb(tf_entry lab) {
  const float sixteen_over_116 = 16.0f / 116.0f;
  float y = (lab.r + 16.0f) / 116.0f;
  float x = lab.g / 500.0f + y;
  float z = y - lab.b / 200.0f;
  x = (powf(x, 3.0f) > 0.008856f) ? powf(x, 3.0f) : (x - sixteen_over_116) / 7.787f;
  y = (powf(y, 3.0f) > 0.008856f) ? powf(y, 3.0f) : (y - sixteen_over_116) / 7.787f;
  z = (powf(z, 3.0f) > 0.008856f) ? powf(z, 3.0f) : (z - sixteen_over_116) / 7.787f;

  x *= 0.9505f;
  y *= 1.0f;
  z *= 1.089f;

  float r = x * 3.2406f + y * -1.5372f + z * -0.4986f;
  float g = x * -0.9689f + y * 1.8758f + z * 0.0415f;
  float b = x * 0.0557f + y * -0.2040f + z * 1.0570f;

  const float inv_gamma = 1.0f / 2.4f;
  r = (r > 0.0031308f) ? 1.055f * powf(r, inv_gamma) - 0.055f : 12.92f * r;
  g = (g > 0.0031308f) ? 1.055f * powf(g, inv_gamma) - 0.055f : 12.92f * g;
  b = (b > 0.0031308f) ? 1.055f * powf(b, inv_gamma) - 0.055f : 12.92f * b;

  const float max_val = maxf(r, maxf(g, b));
  if (max_val > 1.0f) {
    r /= max_val;
    g /= max_val;
    b /= max_val;
  }
  tf_entry rgb;
  rgb.r = maxf(r, 0.0f);
  rgb.g = maxf(g, 0.0f);
  rgb.b = maxf(b, 0.0f);
  rgb.a = lab.a;
  return rgb;
}

static tf_entry lerp_entry(tf_entry l, tf_entry r, float t) { /* VolumePainter.cpp:322-329 */
  tf_entry o;
  o.r = l.r + (r.r - l.r) * t;
  o.g = l.g + (r.g - l.g) * t;
  o.b = l.b + (r.b - l.b) * t;
  o.a = l.a + (r.a - l.a) * t;
  return o;
}

/* mapColorValue, VolumePainter.cpp:331-379 */
static tf_entry map_color(const tf_spec *t, float value) {
  if (!isfinite(value)) return t->nan_color;
  if (t->n_colors == 0) return t->below;
  const color_node *first = &t->colors[0];
  const color_node *last = &t->colors[t->n_colors - 1];
  tf_entry e;
  if (value < first->value) {
    if (!t->use_clamping) return t->below;
    e.r = first->r; e.g = first->g; e.b = first->b; e.a = 1.0f;
    return e;
  }
  if (value > last->value) {
    if (!t->use_clamping) return t->above;
    e.r = last->r; e.g = last->g; e.b = last->b; e.a = 1.0f;
    return e;
  }
  if (value == first->value) {
    e.r = first->r; e.g = first->g; e.b = first->b; e.a = 1.0f;
    return e;
  }
  if (value == last->value) {
    e.r = last->r; e.g = last->g; e.b = last->b; e.a = 1.0f;
    return e;
  }
  for (int idx = 1; idx < t->n_colors; ++idx) {
    const color_node *right = &t->colors[idx];
    if (right->value >= value) {
      const color_node *left = &t->colors[idx - 1];
      const float span = right->value - left->value;
      const float u = (span > 0.0f) ? (value - left->value) / span : 0.0f;
      tf_entry l = {left->r, left->g, left->b, 1.0f};
      tf_entry r = {right->r, right->g, right->b, 1.0f};
      if (t->lab) {
        l = rgb_to_lab(l);
        r = rgb_to_lab(r);
        return lab_to_rgb(lerp_entry(l, r, u));
      }
      return lerp_entry(l, r, u);
    }
  }
  e.r = last->r; e.g = last->g; e.b = last->b; e.a = 1.0f;
  return e;
}

/* mapOpacityValue, VolumePainter.cpp:381-440 */
static float map_opacity(const tf_spec *t, float value) {
  if (!isfinite(value)) return 1.0f;
  if (t->n_opacity == 0) return 1.0f;
  const opacity_node *first = &t->opacity[0];
  const opacity_node *last = &t->opacity[t->n_opacity - 1];
  if (value <= first->value) return first->alpha;
  if (value >= last->value) return last->alpha;
  for (int idx = 1; idx < t->n_opacity; ++idx) {
    const opacity_node *right = &t->opacity[idx];
    if (right->value >= value) {
      const opacity_node *left = &t->opacity[idx - 1];
      const float span = right->value - left->value;
      float w = (span > 0.0f) ? (value - left->value) / span : 0.0f;
      if (w < left->midpoint) {
        w = 0.5f * w / left->midpoint;
      } else {
        w = 0.5f + 0.5f * (w - left->midpoint) / (1.0f - left->midpoint);
      }
      if (left->sharpness == 1.0f) return (w < 0.5f) ? left->alpha : right->alpha;
      if (left->sharpness == 0.0f) return left->alpha + (right->alpha - left->alpha) * w;
      if (w < 0.5f) {
        w = 0.5f * powf(w * 2.0f, 1.0f + 10.0f * left->sharpness);
      } else if (w > 0.5f) {
        w = 1.0f - 0.5f * powf((1.0f - w) * 2.0f, 1.0f + 10.0f * left->sharpness);
      }
      const float ww = w * w;
      const float www = ww * w;
      const float h1 = 2.0f * www - 3.0f * ww + 1.0f;
      const float h2 = -2.0f * www + 3.0f * ww;
      const float h3 = www - 2.0f * ww + w;
      const float h4 = www - ww;
      const float slope = right->alpha - left->alpha;
      const float tt = (1.0f - left->sharpness) * slope;
      float result = h1 * left->alpha + h2 * right->alpha + h3 * tt + h4 * tt;
      result = maxf(result, minf(left->alpha, right->alpha));
      result = minf(result, maxf(left->alpha, right->alpha));
      return result;
    }
  }
  return last->alpha;
}

/* buildColorTable, VolumePainter.cpp:442-516 */
void orc_build_color_table(float alpha_scale, float normalization_factor,
                           const float scalar_range[2], const orc_colormap_point *colormap,
                           int colormap_count, float out_table[1024]) {
  tf_spec *t = (tf_spec *)calloc(1, sizeof(tf_spec));
  t->use_clamping = 1;
  t->below.a = 1.0f;
  t->above.a = 1.0f;
  if (colormap != NULL && colormap_count > 0) {
    t->lab = 1;
    t->nan_color.r = 1.0f; t->nan_color.g = 0.0f; t->nan_color.b = 0.0f; t->nan_color.a = 1.0f;
    for (int i = 0; i < colormap_count; ++i) {
      color_node n;
      n.value = colormap[i].value;
      n.r = clampf(colormap[i].red, 0.0f, 1.0f);
      n.g = clampf(colormap[i].green, 0.0f, 1.0f);
      n.b = clampf(colormap[i].blue, 0.0f, 1.0f);
      insert_color(t, n);
      opacity_node o;
      o.value = colormap[i].value;
      o.alpha = scaled_alpha(colormap[i].alpha, alpha_scale, normalization_factor);
      o.midpoint = 0.5f;
      o.sharpness = 0.0f;
      insert_opacity(t, o);
    }
  } else {
    t->lab = 0;
    t->nan_color.r = 0.25f; t->nan_color.g = 0.0f; t->nan_color.b = 0.0f; t->nan_color.a = 1.0f;
    static const color_node jet[7] = {
        {0.0f, 0.0f, 0.0f, 0.5625f},      {0.111111f, 0.0f, 0.0f, 1.0f},
        {0.3650795f, 0.0f, 1.0f, 1.0f},   {0.4920635f, 0.5f, 1.0f, 0.5f},
        {0.6190475f, 1.0f, 1.0f, 0.0f},   {0.873016f, 1.0f, 0.0f, 0.0f},
        {1.0f, 0.5f, 0.0f, 0.0f},
    };
    for (int i = 0; i < 7; ++i) insert_color(t, jet[i]);
    static const float positions[6] = {0.0f, 0.15f, 0.35f, 0.6f, 0.85f, 1.0f};
    static const float alphas[6] = {0.05f, 0.15f, 0.22f, 0.3f, 0.38f, 0.5f};
    const float range_min = scalar_range[0];
    const float range_max = scalar_range[1];
    const float range_span = range_max - range_min;
    for (int i = 0; i < 6; ++i) {
      opacity_node o;
      o.value = positions[i] * range_span + range_min;
      o.alpha = scaled_alpha(alphas[i], alpha_scale, normalization_factor);
      o.midpoint = 0.5f;
      o.sharpness = 0.0f;
      insert_opacity(t, o);
    }
    rescale_table(t, scalar_range[0], scalar_range[1]);
  }

  const float range_min = scalar_range[0];
  const float range_max = scalar_range[1];
  const float range_span = range_max - range_min;
  for (int i = 0; i < TABLE_SIZE; ++i) {
    const float u = (float)i / (float)(TABLE_SIZE - 1);
    const float value = range_min + range_span * u;
    tf_entry e = map_color(t, value);
    e.a = map_opacity(t, value);
    out_table[i * 4 + 0] = e.r;
    out_table[i * 4 + 1] = e.g;
    out_table[i * 4 + 2] = e.b;
    out_table[i * 4 + 3] = e.a;
  }
  free(t);
}

/* ========================================================================================= */
/* Scalar transform and soft clip                                                             */
/* ========================================================================================= */

/* applyScalarTransform / toProcessedScalar / sanitizeScalarSample, VolumeTypes.hpp:33-67 */
float orc_apply_scalar_transform(double raw, const orc_transform *tr) {
  double v = isfinite(raw) ? raw : 0.0;
  if (tr->log_scale_input) {
    if (!(v > 0.0)) {
      v = tr->positive_floor;
    } else if (v < tr->positive_floor) {
      v = tr->positive_floor;
    }
    v = log(v);
  }
  if (tr->normalize_to_unit_range) {
    v = (v - tr->normalization_min) * tr->inverse_normalization_span;
    if (v < 0.0) {
      v = 0.0;
    } else if (v > 1.0) {
      v = 1.0;
    }
  }
  return (float)v;
}

/* saturateSoftTail, VolumePainter.cpp:75-105 */
static float saturate_soft_tail(float value, float clip_start, float rolloff_end) {
  const float clamped_end = maxf(clip_start, rolloff_end);
  float cv = value;
  if (cv < 0.0f) {
    cv = 0.0f;
  } else if (cv > clamped_end) {
    cv = clamped_end;
  }
  if (!(clamped_end > clip_start + kSoftClipTolerance)) return cv;
  if (!(cv > clip_start)) return cv;
  if (!(cv < clamped_end)) return clamped_end;
  const float n = (cv - clip_start) / (clamped_end - clip_start);
  const float smooth = n + n * n - n * n * n;
  return clip_start + (clamped_end - clip_start) * smooth;
}

/* ========================================================================================= */
/* VolumePainter::paint                                                                       */
/* ========================================================================================= */

/* VolumePainter.cpp:571-613 */
void orc_box_sampling(const orc_box *box, const orc_paint_params *params, float *sample_distance,
                      float *normalization_factor, float *alpha_scale) {
  double spacing[3] = {0.0, 0.0, 0.0};
  for (int c = 0; c < 3; ++c) {
    const double span = box->max_corner[c] - box->min_corner[c];
    if (box->dims[c] > 0) spacing[c] = span / (double)box->dims[c];
  }
  float min_spacing = 3.402823466e+38f; /* numeric_limits<float>::max() */
  for (int c = 0; c < 3; ++c) {
    const float v = (float)spacing[c];
    if (v > 0.0f && v < min_spacing && isfinite(v)) min_spacing = v;
  }
  if (!(min_spacing > 0.0f && isfinite(min_spacing))) {
    const double fs0 = params->bounds_max[0] - params->bounds_min[0];
    const double fs1 = params->bounds_max[1] - params->bounds_min[1];
    const double fs2 = params->bounds_max[2] - params->bounds_min[2];
    const float fallback_min = (float)mind(mind(fs0, fs1), fs2);
    min_spacing = maxf(1e-4f, fallback_min * 0.01f);
  }
  const float sd = maxf(min_spacing * 0.5f, 1e-5f);
  float ref = params->reference_sample_distance;
  if (!(ref > 0.0f && isfinite(ref))) ref = sd;
  float nf = sd / ref;
  if (!isfinite(nf)) nf = 1.0f;
  nf = maxf(nf, 0.0f);
  *sample_distance = sd;
  *normalization_factor = nf;
  *alpha_scale = clampf(1.0f - params->box_transparency, 0.0f, 1.0f);
}

typedef struct {
  int width, height, nx, ny, nz;
  float aspect, tan_half_fov, inv_width, inv_height;
  float fwd[3], right[3], up[3], eye[3];
  float minc[3], maxc[3];
  float dx, dy, dz;
  float mesh_epsilon, sample_distance;
  float range_min, inverse_range, clip_start;
  int apply_clip;
  const double *cells;
  int64_t jstride, kstride;
  const orc_transform *transform;
  const float *table;
} march_consts;

/* The ParallelFor body, VolumePainter.cpp:737-922, one pixel.  Writes rgba (device-side clamp
 * to <= 1 only) and depth; returns the executed cell fetches of this ray. */
static uint64_t march_pixel(const march_consts *k, int index, float out_color[4],
                            float *out_depth) {
  const int px = index % k->width;
  const int py = index / k->width;

  const float ndc_x = ((float)px + 0.5f) * k->inv_width * 2.0f - 1.0f;
  const float ndc_y = ((float)py + 0.5f) * k->inv_height * 2.0f - 1.0f;
  const float plane_x = ndc_x * k->tan_half_fov * k->aspect;
  const float plane_y = ndc_y * k->tan_half_fov;

  float dir_x = k->fwd[0] + plane_x * k->right[0] + plane_y * k->up[0];
  float dir_y = k->fwd[1] + plane_x * k->right[1] + plane_y * k->up[1];
  float dir_z = k->fwd[2] + plane_x * k->right[2] + plane_y * k->up[2];

  const float len_sq = dir_x * dir_x + dir_y * dir_y + dir_z * dir_z;
  /* host amrex::Math::rsqrt(x) = 1/sqrt(x)   */
  float dir_len = (len_sq > 0.0f) ? (1.0f / (1.0f / sqrtf(len_sq))) : 0.0f;
  if (orc_shim_variant == 1 && len_sq > 0.0f) {
    /* tools/appb_hash_search.py only: a header shim whose rsqrt works in amrex::Real (double) */
    dir_len = (float)(1.0 / (1.0 / sqrt((double)len_sq)));
  } else if (orc_shim_variant == 2 && len_sq > 0.0f) {
    /* ... or one that returns the float result of a double computation */
    dir_len = 1.0f / (float)(1.0 / sqrt((double)len_sq));
  }
  if (dir_len > 0.0f) {
    const float inv = 1.0f / dir_len;
    dir_x *= inv;
    dir_y *= inv;
    dir_z *= inv;
  }

  float tmin = -INFINITY;
  float tmax = INFINITY;
  const float origin[3] = {k->eye[0], k->eye[1], k->eye[2]};
  const float dir[3] = {dir_x, dir_y, dir_z};
  for (int c = 0; c < 3; ++c) { /* updateBounds, :775-800 */
    if (fabsf(dir[c]) < 1e-8f) {
      if (origin[c] < k->minc[c] || origin[c] > k->maxc[c]) {
        tmin = INFINITY;
        tmax = -INFINITY;
      }
      continue;
    }
    const float inv_dir = 1.0f / dir[c];
    float t1 = (k->minc[c] - origin[c]) * inv_dir;
    float t2 = (k->maxc[c] - origin[c]) * inv_dir;
    if (t1 > t2) {
      const float tmp = t1;
      t1 = t2;
      t2 = tmp;
    }
    tmin = (tmin > t1) ? tmin : t1;
    tmax = (tmax < t2) ? tmax : t2;
  }

  if (!(tmax >= tmin)) {
    out_color[0] = out_color[1] = out_color[2] = out_color[3] = 0.0f;
    *out_depth = INFINITY;
    return 0;
  }

  float distance = tmin + k->mesh_epsilon;
  if (distance < 0.0f) distance = k->mesh_epsilon;

  float acc_r = 0.0f, acc_g = 0.0f, acc_b = 0.0f, acc_a = 0.0f;
  uint64_t fetches = 0;

#define INSIDE(x, y, z)                                                                     \
  (!((x) < k->minc[0] || (x) > k->maxc[0] || (y) < k->minc[1] || (y) > k->maxc[1] ||        \
     (z) < k->minc[2] || (z) > k->maxc[2]))

  float pos_x = origin[0] + dir_x * distance;
  float pos_y = origin[1] + dir_y * distance;
  float pos_z = origin[2] + dir_z * distance;
  while (distance < tmax && !INSIDE(pos_x, pos_y, pos_z)) {
    distance += k->sample_distance;
    pos_x = origin[0] + dir_x * distance;
    pos_y = origin[1] + dir_y * distance;
    pos_z = origin[2] + dir_z * distance;
  }

  while (distance < tmax && acc_a < 1.0f) {
    if (!INSIDE(pos_x, pos_y, pos_z)) {
      distance += k->sample_distance;
      pos_x = origin[0] + dir_x * distance;
      pos_y = origin[1] + dir_y * distance;
      pos_z = origin[2] + dir_z * distance;
      continue;
    }
    const float fx = (pos_x - k->minc[0]) / k->dx;
    const float fy = (pos_y - k->minc[1]) / k->dy;
    const float fz = (pos_z - k->minc[2]) / k->dz;
    int i = (int)floorf(fx);
    int j = (int)floorf(fy);
    int kk = (int)floorf(fz);
    if (i < 0) i = 0; else if (i >= k->nx) i = k->nx - 1;
    if (j < 0) j = 0; else if (j >= k->ny) j = k->ny - 1;
    if (kk < 0) kk = 0; else if (kk >= k->nz) kk = k->nz - 1;

    {
      const double raw = k->cells[(int64_t)i + (int64_t)j * k->jstride + (int64_t)kk * k->kstride];
      ++fetches;
      float scalar = orc_apply_scalar_transform(raw, k->transform);
      if (k->apply_clip) scalar = saturate_soft_tail(scalar, k->clip_start, 1.0f);
      float normalized = (scalar - k->range_min) * k->inverse_range;
      normalized = (normalized < 0.0f) ? 0.0f : normalized;
      normalized = (normalized > 1.0f) ? 1.0f : normalized;
      int idx = (int)(normalized * (float)(TABLE_SIZE - 1));
      idx = (idx < 0) ? 0 : idx;
      idx = (idx > TABLE_SIZE - 1) ? (TABLE_SIZE - 1) : idx;
      const float *e = k->table + idx * 4;
      const float alpha = e[3] * (1.0f - acc_a);
      acc_r += e[0] * alpha;
      acc_g += e[1] * alpha;
      acc_b += e[2] * alpha;
      acc_a += alpha;
    }

    distance += k->sample_distance;
    pos_x = origin[0] + dir_x * distance;
    pos_y = origin[1] + dir_y * distance;
    pos_z = origin[2] + dir_z * distance;
  }
#undef INSIDE

  acc_r = (acc_r > 1.0f) ? 1.0f : acc_r;
  acc_g = (acc_g > 1.0f) ? 1.0f : acc_g;
  acc_b = (acc_b > 1.0f) ? 1.0f : acc_b;
  acc_a = (acc_a > 1.0f) ? 1.0f : acc_a;
  out_color[0] = acc_r;
  out_color[1] = acc_g;
  out_color[2] = acc_b;
  out_color[3] = acc_a;

  float depth = INFINITY;
  if (acc_a > 0.0f) {
    const float ex = origin[0] + dir_x * tmin;
    const float ey = origin[1] + dir_y * tmin;
    const float ez = origin[2] + dir_z * tmin;
    depth = (ex - k->eye[0]) * k->fwd[0] + (ey - k->eye[1]) * k->fwd[1] +
            (ez - k->eye[2]) * k->fwd[2];
  }
  *out_depth = depth;
  return fetches;
}

/* VolumePainter::paint restricted to the pixel window [x0, x1) x [y0, y1) of the width x height
 * image: the rays are those of the full image (pixel index y * width + x), only the loop bounds
 * and the output addressing change; out_rgbad holds (y1 - y0) x (x1 - x0) pixels, row-major.
 * Lets the tests check crops of frames whose full per-box layers would not fit in memory
 * (config-5: 1856 boxes x 8192^2 x 20 B). */
uint64_t orc_paint_box_window(const orc_box *box, const orc_transform *transform,
                              const orc_paint_params *params, const orc_camera *camera, int x0,
                              int y0, int x1, int y1, float *out_rgbad, int threads) {
  float sample_distance, norm_factor, alpha_scale;
  orc_box_sampling(box, params, &sample_distance, &norm_factor, &alpha_scale);

  float table[TABLE_SIZE * 4];
  orc_build_color_table(alpha_scale, norm_factor, params->scalar_range, params->colormap,
                        params->colormap_count, table);

  const int width = params->width;
  const int height = params->height;
  if (width <= 0 || height <= 0) return 0;
  if (x0 < 0) x0 = 0;
  if (y0 < 0) y0 = 0;
  if (x1 > width) x1 = width;
  if (y1 > height) y1 = height;
  if (x1 <= x0 || y1 <= y0) return 0;
  const int win_w = x1 - x0;
  const int64_t pixel_count = (int64_t)win_w * (y1 - y0);

  march_consts k;
  memset(&k, 0, sizeof(k));
  k.width = width;
  k.height = height;
  k.aspect = (float)width / (float)((height > 1) ? height : 1);
  camera_basis(camera, k.fwd, k.right, k.up, k.eye);
  for (int c = 0; c < 3; ++c) {
    k.minc[c] = (float)box->min_corner[c];
    k.maxc[c] = (float)box->max_corner[c];
  }
  k.nx = box->dims[0];
  k.ny = box->dims[1];
  k.nz = box->dims[2];
  if (k.nx <= 0 || k.ny <= 0 || k.nz <= 0) { /* image->clear(): (0,0,0,0) + depth +inf */
    for (int64_t p = 0; p < pixel_count; ++p) {
      out_rgbad[p * 5 + 0] = out_rgbad[p * 5 + 1] = out_rgbad[p * 5 + 2] = 0.0f;
      out_rgbad[p * 5 + 3] = 0.0f;
      out_rgbad[p * 5 + 4] = INFINITY;
    }
    return 0;
  }
  k.dx = (k.maxc[0] - k.minc[0]) / (float)k.nx;
  k.dy = (k.maxc[1] - k.minc[1]) / (float)k.ny;
  k.dz = (k.maxc[2] - k.minc[2]) / (float)k.nz;
  const float ex = k.maxc[0] - k.minc[0];
  const float ey = k.maxc[1] - k.minc[1];
  const float ez = k.maxc[2] - k.minc[2];
  const float extent_mag = sqrtf(ex * ex + ey * ey + ez * ez);
  k.mesh_epsilon = extent_mag * 0.0001f;
  k.sample_distance = sample_distance;

  k.range_min = params->scalar_range[0];
  const float range_max = params->scalar_range[1];
  k.inverse_range = 1.0f;
  if (range_max != k.range_min) k.inverse_range = 1.0f / (range_max - k.range_min);
  k.clip_start = clampf(params->scalar_range[1], 0.0f, 1.0f);
  k.apply_clip = 1.0f > k.clip_start + kSoftClipTolerance;

  k.tan_half_fov = tanf(camera->fov_y_degrees * 0.5f * kPi / 180.0f);
  k.inv_width = 1.0f / (float)width;
  k.inv_height = 1.0f / (float)height;

  k.cells = box->cells;
  k.jstride = box->jstride;
  k.kstride = box->kstride;
  k.transform = transform;
  k.table = table;

  uint64_t total = 0;
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : total) num_threads(threads > 0 ? threads : 1)
#endif
  for (int y = y0; y < y1; ++y) {
    for (int x = x0; x < x1; ++x) {
      const int index = y * width + x;
      float color[4];
      float depth;
      total += march_pixel(&k, index, color, &depth);
      /* host epilogue, VolumePainter.cpp:939-955 */
      const float r = clampf(color[0], 0.0f, 1.0f);
      const float g = clampf(color[1], 0.0f, 1.0f);
      const float b = clampf(color[2], 0.0f, 1.0f);
      const float a = clampf(color[3], 0.0f, 1.0f);
      if (!isfinite(depth) || a <= 0.0f) depth = INFINITY;
      float *o = out_rgbad + ((int64_t)(y - y0) * win_w + (x - x0)) * 5;
      o[0] = r;
      o[1] = g;
      o[2] = b;
      o[3] = a;
      o[4] = depth;
    }
  }
  return total;
}

uint64_t orc_paint_box(const orc_box *box, const orc_transform *transform,
                       const orc_paint_params *params, const orc_camera *camera,
                       float *out_rgbad, int threads) {
  return orc_paint_box_window(box, transform, params, camera, 0, 0, params->width, params->height,
                              out_rgbad, threads);
}

/* computeBoxDepthHint, VolumeRenderer.cpp:541-553 */
float orc_box_depth_hint(const orc_box *box, const orc_camera *cam) {
  double d[3] = {cam->look_at[0] - cam->eye[0], cam->look_at[1] - cam->eye[1],
                 cam->look_at[2] - cam->eye[2]};
  double view_dir[3];
  safe_normalize(d, view_dir);
  float min_depth = INFINITY;
  for (int corner = 0; corner < 8; ++corner) {
    double rel[3];
    rel[0] = ((corner & 1) ? box->max_corner[0] : box->min_corner[0]) - cam->eye[0];
    rel[1] = ((corner & 2) ? box->max_corner[1] : box->min_corner[1]) - cam->eye[1];
    rel[2] = ((corner & 4) ? box->max_corner[2] : box->min_corner[2]) - cam->eye[2];
    min_depth = minf(min_depth, (float)v3_dot(rel, view_dir));
  }
  return min_depth;
}

/* VolumeRenderer.cpp:1138-1190 */
float orc_reference_sample_distance(const orc_box *boxes, int n_boxes,
                                    const double bounds_min[3], const double bounds_max[3]) {
  float coarsest = 0.0f;
  for (int b = 0; b < n_boxes; ++b) {
    double spacing[3] = {0.0, 0.0, 0.0};
    for (int c = 0; c < 3; ++c) {
      const double span = boxes[b].max_corner[c] - boxes[b].min_corner[c];
      if (boxes[b].dims[c] > 0) spacing[c] = span / (double)(float)boxes[b].dims[c];
    }
    float min_spacing = 3.402823466e+38f;
    for (int c = 0; c < 3; ++c) {
      /* comparisons happen in double against the float-valued running minimum */
      if (spacing[c] > 0.0 && spacing[c] < (double)min_spacing && isfinite(spacing[c])) {
        min_spacing = (float)spacing[c];
      }
    }
    if (min_spacing > 0.0f && isfinite(min_spacing)) coarsest = maxf(coarsest, min_spacing);
  }
  if (!(coarsest > 0.0f && isfinite(coarsest))) {
    float fallback_min = 3.402823466e+38f;
    for (int c = 0; c < 3; ++c) {
      const float axis = (float)(bounds_max[c] - bounds_min[c]);
      if (axis > 0.0f && isfinite(axis)) fallback_min = minf(fallback_min, axis);
    }
    if (!(fallback_min > 0.0f && isfinite(fallback_min))) fallback_min = 1.0f;
    coarsest = maxf(1e-4f, fallback_min * 0.01f);
  }
  return maxf(coarsest * 0.5f, 1e-5f);
}

/* ========================================================================================= */
/* Image algebra                                                                              */
/* ========================================================================================= */

static inline void blend_px_depthsort(const float *top, const float *bottom, float *out) {
  const float td = top[4], bd = bottom[4];
  const int top_is_front = td <= bd;
  const float *front = top_is_front ? top : bottom;
  const float *back = top_is_front ? bottom : top;
  for (int c = 0; c < 4; ++c) out[c] = front[c] + back[c] * (1.0f - front[3]);
  out[4] = minf(td, bd); /* std::min(topDepth, bottomDepth) */
}
static inline void blend_px_rgba_f32(const float *top, const float *bottom, float *out) {
  for (int c = 0; c < 4; ++c) out[c] = top[c] + bottom[c] * (1.0f - top[3]);
}
static inline void blend_px_rgba_u8(const uint32_t *top_e, const uint32_t *bottom_e,
                                    uint32_t *out_e) {
  const unsigned char *top = (const unsigned char *)top_e;
  const unsigned char *bottom = (const unsigned char *)bottom_e;
  unsigned char *out = (unsigned char *)out_e;
  const float bottom_scale = 1.0f - top[3] / 255.0f;
  for (int c = 0; c < 4; ++c) {
    out[c] = (unsigned char)(top[c] + (unsigned char)(bottom[c] * bottom_scale));
  }
}

void orc_blend_depthsort(const float *top, const float *bottom, float *out, int64_t n) {
  for (int64_t p = 0; p < n; ++p) blend_px_depthsort(top + p * 5, bottom + p * 5, out + p * 5);
}
void orc_blend_rgba_f32(const float *top, const float *bottom, float *out, int64_t n) {
  for (int64_t p = 0; p < n; ++p) blend_px_rgba_f32(top + p * 4, bottom + p * 4, out + p * 4);
}
void orc_blend_rgba_u8(const uint32_t *top, const uint32_t *bottom, uint32_t *out, int64_t n) {
  for (int64_t p = 0; p < n; ++p) blend_px_rgba_u8(top + p, bottom + p, out + p);
}

/* ImageColorOnly<F>::blend region logic, Common/ImageColorOnly.hpp:119-199 */
void orc_blend_regions(int kind, const void *top, int tb, int te, const void *bottom, int bb,
                       int be, void *out) {
  const size_t px_bytes = (kind == 0) ? 20 : (kind == 1) ? 16 : 4;
  const char *t = (const char *)top;
  const char *b = (const char *)bottom;
  char *o = (char *)out;
  int ti = 0, bi = 0, oi = 0;
  const int tn = te - tb, bn = be - bb;
  if (tb < bb) {
    const int n = bb - tb;
    memcpy(o, t, px_bytes * (size_t)n);
    ti += n;
    oi += n;
  } else if (bb < tb) {
    const int n = tb - bb;
    memcpy(o, b, px_bytes * (size_t)n);
    bi += n;
    oi += n;
  }
  while (ti < tn && bi < bn) {
    const void *tp = t + px_bytes * (size_t)ti;
    const void *bp = b + px_bytes * (size_t)bi;
    void *op = o + px_bytes * (size_t)oi;
    if (kind == 0) blend_px_depthsort((const float *)tp, (const float *)bp, (float *)op);
    else if (kind == 1) blend_px_rgba_f32((const float *)tp, (const float *)bp, (float *)op);
    else blend_px_rgba_u8((const uint32_t *)tp, (const uint32_t *)bp, (uint32_t *)op);
    ++ti; ++bi; ++oi;
  }
  if (ti < tn) {
    memcpy(o + px_bytes * (size_t)oi, t + px_bytes * (size_t)ti, px_bytes * (size_t)(tn - ti));
    oi += tn - ti;
  }
  if (bi < bn) {
    memcpy(o + px_bytes * (size_t)oi, b + px_bytes * (size_t)bi, px_bytes * (size_t)(bn - bi));
    oi += bn - bi;
  }
}

static inline unsigned char component_as_byte(float c) { /* Color.hpp:86-90 */
  /* int(c * 256.f) is undefined in C when the product does not fit an int; what the reference gets on
     the CPU it runs on (x86-64, cvttss2si: INT_MIN for NaN and |product| >= 2^31) is spelled out, so
     that this restatement does not depend on the compiler (pinned by tests/golden/ref_blend.npz) */
  const float t = c * 256.f;
  const int tv = (t >= -2147483648.f && t < 2147483648.f) ? (int)t : (-2147483647 - 1);
  return (unsigned char)((tv < 0) ? 0 : (tv > 255) ? 255 : tv);
}

void orc_encode_rgba_u8(const float *rgba, uint32_t *out, int64_t n) {
  for (int64_t p = 0; p < n; ++p) {
    unsigned char *o = (unsigned char *)(out + p);
    for (int c = 0; c < 4; ++c) o[c] = component_as_byte(rgba[p * 4 + c]);
  }
}
void orc_decode_rgba_u8(const uint32_t *in, float *rgba, int64_t n) {
  for (int64_t p = 0; p < n; ++p) {
    const unsigned char *i = (const unsigned char *)(in + p);
    for (int c = 0; c < 4; ++c) {
      float v = (float)i[c] / 255.f; /* Color.hpp:69-77 */
      if (v < 0) v = 0;
      if (v > 1) v = 1;
      rgba[p * 4 + c] = v;
    }
  }
}

/* ========================================================================================= */
/* DirectSend                                                                                 */
/* ========================================================================================= */

void orc_piece_range(int image_size, int piece_index, int num_pieces, int *begin, int *end) {
  const int piece_size = image_size / num_pieces;
  *begin = piece_size * piece_index;
  *end = (piece_index < num_pieces - 1) ? (*begin + piece_size) : image_size;
}

typedef struct { float depth; int owner; int local_index; int id; } layer_entry;
static int layer_cmp(const void *pa, const void *pb) { /* DirectSendBase.cpp:378-388 */
  const layer_entry *a = (const layer_entry *)pa;
  const layer_entry *b = (const layer_entry *)pb;
  if (a->depth == b->depth) {
    if (a->owner == b->owner) return (a->local_index > b->local_index) - (a->local_index < b->local_index);
    return (a->owner > b->owner) - (a->owner < b->owner);
  }
  return (a->depth < b->depth) ? -1 : 1;
}

int orc_layer_order(const float *hints, const int *owner, const int *local_index, int n_layers,
                    int *order_out, int *run_end_out) {
  if (n_layers <= 0) return 0;
  layer_entry *e = (layer_entry *)malloc(sizeof(layer_entry) * (size_t)n_layers);
  /* globalOrder is built rank-major, local index minor (DirectSendBase.cpp:363-376);
   * (depth, owner, local_index) is a total order so the initial arrangement is irrelevant. */
  for (int l = 0; l < n_layers; ++l) {
    e[l].depth = hints[l];
    e[l].owner = owner[l];
    e[l].local_index = local_index[l];
    e[l].id = l;
  }
  qsort(e, (size_t)n_layers, sizeof(layer_entry), layer_cmp);
  int runs = 0;
  int idx = 0;
  while (idx < n_layers) { /* DirectSendBase.cpp:400-410 */
    const int run_owner = e[idx].owner;
    while (idx < n_layers && e[idx].owner == run_owner) ++idx;
    run_end_out[runs++] = idx;
  }
  for (int l = 0; l < n_layers; ++l) order_out[l] = e[l].id;
  free(e);
  return runs;
}

int orc_compose_layered(const float *const *layers, const float *hints, const int *owner,
                        const int *local_index, int n_layers, int n_ranks, int n_pixels,
                        const int *group_order, int fold_variant, float *out_gathered,
                        int *piece_owner) {
  int *order = (int *)malloc(sizeof(int) * (size_t)(n_layers > 0 ? n_layers : 1));
  int *run_end = (int *)malloc(sizeof(int) * (size_t)(n_layers > 0 ? n_layers : 1));
  const int n_runs = orc_layer_order(hints, owner, local_index, n_layers, order, run_end);

  const size_t img = (size_t)n_pixels * 5;
  float *accumulated = (float *)malloc(sizeof(float) * (img ? img : 1));
  float *run_layer = (float *)malloc(sizeof(float) * (img ? img : 1));
  float *tmp = (float *)malloc(sizeof(float) * (img ? img : 1));
  float *incoming = (float *)malloc(sizeof(float) * (img ? img : 1));
  float *folded = (float *)malloc(sizeof(float) * (img ? img : 1));
  int have_accumulated = 0;

  /* position of each rank in the ordered group */
  int *pos_rank = (int *)malloc(sizeof(int) * (size_t)n_ranks);
  for (int k = 0; k < n_ranks; ++k) pos_rank[k] = group_order ? group_order[k] : k;

  int start = 0;
  for (int r = 0; r < n_runs; ++r) {
    const int end = run_end[r];
    const int run_owner = owner[order[start]];
    /* owner-side local fold, DirectSendBase.cpp:413-426 */
    memcpy(run_layer, layers[order[start]], sizeof(float) * img);
    for (int q = start + 1; q < end; ++q) {
      orc_blend_depthsort(run_layer, layers[order[q]], tmp, n_pixels);
      memcpy(run_layer, tmp, sizeof(float) * img);
    }
    /* compose(local, sendGroup=group, recvGroup=group): receiver at position k folds the
     * piece-k windows of every sender in group order (DirectSendBase.cpp:76-255).  Non-owners
     * contribute createEmptyLayer+clear(0,0,0,0) -> (0,0,0,0,+inf) (:427-432). */
    for (int k = 0; k < n_ranks; ++k) {
      int pb, pe;
      orc_piece_range(n_pixels, k, n_ranks, &pb, &pe);
      const int np = pe - pb;
      if (np <= 0) continue;
      float *dst = folded + (size_t)pb * 5;
      if (fold_variant == 0) {
        for (int s = 0; s < n_ranks; ++s) {
          const int sender = pos_rank[s];
          for (int p = 0; p < np; ++p) {
            float *in = incoming + (size_t)p * 5;
            if (sender == run_owner) {
              memcpy(in, run_layer + (size_t)(pb + p) * 5, sizeof(float) * 5);
            } else {
              in[0] = in[1] = in[2] = in[3] = 0.0f;
              in[4] = INFINITY;
            }
          }
          if (s == 0) {
            memcpy(dst, incoming, sizeof(float) * (size_t)np * 5);
          } else {
            orc_blend_depthsort(dst, incoming, tmp, np);
            memcpy(dst, tmp, sizeof(float) * (size_t)np * 5);
          }
        }
      } else {
        for (int s = n_ranks - 1; s >= 0; --s) {
          const int sender = pos_rank[s];
          for (int p = 0; p < np; ++p) {
            float *in = incoming + (size_t)p * 5;
            if (sender == run_owner) {
              memcpy(in, run_layer + (size_t)(pb + p) * 5, sizeof(float) * 5);
            } else {
              in[0] = in[1] = in[2] = in[3] = 0.0f;
              in[4] = INFINITY;
            }
          }
          if (s == n_ranks - 1) {
            memcpy(dst, incoming, sizeof(float) * (size_t)np * 5);
          } else {
            orc_blend_depthsort(incoming, dst, tmp, np); /* earlier member stays on top */
            memcpy(dst, tmp, sizeof(float) * (size_t)np * 5);
          }
        }
      }
    }
    /* accumulate, DirectSendBase.cpp:441-445 */
    if (!have_accumulated) {
      memcpy(accumulated, folded, sizeof(float) * img);
      have_accumulated = 1;
    } else {
      orc_blend_depthsort(accumulated, folded, tmp, n_pixels);
      memcpy(accumulated, tmp, sizeof(float) * img);
    }
    start = end;
  }
  if (!have_accumulated) { /* DirectSendBase.cpp:450-455 */
    for (int p = 0; p < n_pixels; ++p) {
      float *o = accumulated + (size_t)p * 5;
      o[0] = o[1] = o[2] = o[3] = 0.0f;
      o[4] = INFINITY;
    }
  }
  memcpy(out_gathered, accumulated, sizeof(float) * img);
  if (piece_owner) {
    for (int k = 0; k < n_ranks; ++k) {
      int pb, pe;
      orc_piece_range(n_pixels, k, n_ranks, &pb, &pe);
      for (int p = pb; p < pe; ++p) piece_owner[p] = pos_rank[k];
    }
  }
  free(pos_rank);
  free(folded);
  free(incoming);
  free(tmp);
  free(run_layer);
  free(accumulated);
  free(run_end);
  free(order);
  return n_runs;
}

/* ========================================================================================= */
/* Frame tail                                                                                 */
/* ========================================================================================= */

/* downsampleImage, VolumeRenderer.cpp:479-528 */
void orc_downsample(const float *src, int target_w, int target_h, int block, float *dst) {
  const int src_w = target_w * block;
  const float inv_samples = 1.0f / (float)(block * block);
  for (int y = 0; y < target_h; ++y) {
    for (int x = 0; x < target_w; ++x) {
      float sr = 0.0f, sg = 0.0f, sb = 0.0f, sa = 0.0f;
      for (int dy = 0; dy < block; ++dy) {
        const int sy = y * block + dy;
        for (int dx = 0; dx < block; ++dx) {
          const int sx = x * block + dx;
          const float *s = src + ((size_t)sy * src_w + sx) * 5;
          sr += s[0];
          sg += s[1];
          sb += s[2];
          sa += s[3];
        }
      }
      float *o = dst + ((size_t)y * target_w + x) * 5;
      o[0] = sr * inv_samples;
      o[1] = sg * inv_samples;
      o[2] = sb * inv_samples;
      o[3] = sa * inv_samples;
      o[4] = INFINITY;
    }
  }
}

/* SavePPM.cpp:17-36 pixel order + Color::GetComponentAsByte */
void orc_quantize_rgb8(const float *src, int w, int h, int stride, uint8_t *dst) {
  size_t o = 0;
  for (int y = h - 1; y >= 0; --y) {
    for (int x = 0; x < w; ++x) {
      const float *s = src + ((size_t)y * w + x) * (size_t)stride;
      dst[o++] = component_as_byte(s[0]);
      dst[o++] = component_as_byte(s[1]);
      dst[o++] = component_as_byte(s[2]);
    }
  }
}

/* ========================================================================================= */
/* Scene statistics and histogram                                          */
/* ========================================================================================= */

int64_t orc_scalar_stats(const orc_box *boxes, int n_boxes, double stats[3]) {
  double lo = INFINITY, hi = -INFINITY, lo_pos = INFINITY;
  int64_t finite = 0;
  for (int b = 0; b < n_boxes; ++b) {
    const orc_box *box = &boxes[b];
    for (int k = 0; k < box->dims[2]; ++k) {
      for (int j = 0; j < box->dims[1]; ++j) {
        const double *row = box->cells + (int64_t)j * box->jstride + (int64_t)k * box->kstride;
        for (int i = 0; i < box->dims[0]; ++i) {
          const double raw = row[i];
          if (!isfinite(raw)) continue; /* contributes {inf, -inf, inf, 0} */
          if (raw < lo) lo = raw;
          if (raw > hi) hi = raw;
          if (raw > 0.0 && raw < lo_pos) lo_pos = raw;
          ++finite;
        }
      }
    }
  }
  stats[0] = lo;
  stats[1] = hi;
  stats[2] = lo_pos;
  return finite;
}

static void make_scalar_range(double lo, double hi, float out[2]) { /* SceneBuilder.cpp:106-112 */
  if (lo == hi) hi = lo + 1.0;
  out[0] = (float)lo;
  out[1] = (float)hi;
}

int orc_scene_transform(const double stats[3], int64_t finite_count, int log_scale,
                        int normalize_to_data_range, orc_transform *tr, double *processed_min,
                        double *processed_max, float processed_range[2], float scalar_range[2]) {
  const double original_min = (finite_count > 0) ? stats[0] : INFINITY;
  const double original_max = (finite_count > 0) ? stats[1] : -INFINITY;
  double pmin = original_min, pmax = original_max;
  memset(tr, 0, sizeof(*tr));
  tr->log_scale_input = log_scale ? 1 : 0;
  tr->normalize_to_unit_range = 0;
  tr->positive_floor = 0.0;
  if (log_scale) {
    const double positive_min = (stats[2] > 0.0 && isfinite(stats[2])) ? stats[2] : INFINITY;
    if (!(positive_min < INFINITY) || !isfinite(positive_min) || !(positive_min > 0.0)) return 1;
    tr->positive_floor = positive_min;
    pmin = log(positive_min);
    pmax = log((original_max < positive_min) ? positive_min : original_max); /* std::max */
  }
  if (!isfinite(pmin) || !isfinite(pmax)) return 2;
  if (pmin == pmax) pmax = pmin + 1.0;
  make_scalar_range(pmin, pmax, processed_range);
  *processed_min = pmin;
  *processed_max = pmax;
  tr->normalization_min = pmin;
  tr->inverse_normalization_span = 1.0 / (pmax - pmin);
  scalar_range[0] = processed_range[0];
  scalar_range[1] = processed_range[1];
  if (normalize_to_data_range) { /* SetSceneNormalizationRange, SceneBuilder.cpp:427-443 */
    const double span = pmax - pmin;
    if (!(span > 0.0) || !isfinite(span)) return 2;
    tr->normalize_to_unit_range = 1;
    tr->normalization_min = pmin;
    tr->inverse_normalization_span = 1.0 / span;
    scalar_range[0] = 0.0f;
    scalar_range[1] = 1.0f;
  }
  return 0;
}

void orc_histogram(const orc_box *boxes, int n_boxes, const orc_transform *transform,
                   float range_min, float range_max, int bin_count, uint64_t *counts) {
  for (int i = 0; i < bin_count; ++i) counts[i] = 0;
  const float range_width = range_max - range_min;
  if (!(range_width > 0.0f) || !isfinite(range_width)) return;
  const float inverse_width = 1.0f / range_width;
  for (int b = 0; b < n_boxes; ++b) {
    const orc_box *box = &boxes[b];
    if (box->dims[0] <= 0 || box->dims[1] <= 0 || box->dims[2] <= 0) continue;
    for (int k = 0; k < box->dims[2]; ++k) {
      for (int j = 0; j < box->dims[1]; ++j) {
        const double *row = box->cells + (int64_t)j * box->jstride + (int64_t)k * box->kstride;
        for (int i = 0; i < box->dims[0]; ++i) {
          float value = orc_apply_scalar_transform(row[i], transform);
          if (value < range_min) {
            value = range_min;
          } else if (value > range_max) {
            value = range_max;
          }
          float normalized = (value - range_min) * inverse_width;
          if (normalized < 0.0f) {
            normalized = 0.0f;
          } else if (normalized > 1.0f) {
            normalized = 1.0f;
          }
          int index = (int)(normalized * (float)bin_count);
          if (index >= bin_count) {
            index = bin_count - 1;
          } else if (index < 0) {
            index = 0;
          }
          counts[index] += 1;
        }
      }
    }
  }
}

/* ========================================================================================= */
/* Wireframe overlay                                                       */
/* ========================================================================================= */

void orc_tight_bounds(const orc_box *boxes, int n_boxes, const double fallback_min[3],
                      const double fallback_max[3], double out_min[3], double out_max[3]) {
  if (n_boxes <= 0) {
    for (int c = 0; c < 3; ++c) {
      out_min[c] = fallback_min[c];
      out_max[c] = fallback_max[c];
    }
    return;
  }
  for (int c = 0; c < 3; ++c) {
    double lo = 3.402823466e+38, hi = -3.402823466e+38; /* Vec3(numeric_limits<float>::max()) */
    for (int b = 0; b < n_boxes; ++b) {
      if (boxes[b].min_corner[c] < lo) lo = boxes[b].min_corner[c];
      if (boxes[b].max_corner[c] > hi) hi = boxes[b].max_corner[c];
    }
    out_min[c] = (double)(float)lo; /* reduced as MPI_FLOAT */
    out_max[c] = (double)(float)hi;
  }
}

typedef struct {
  double world[3];
  float x, y, depth;
  int valid;
} screen_corner;

void orc_bbox_overlay(const double bounds_min[3], const double bounds_max[3],
                      const orc_camera *cam, int sqrt_antialiasing, float *image, int width,
                      int height) {
  if (width <= 0 || height <= 0) return;
  const float aspect = (float)width / (float)((height > 1) ? height : 1);
  double d[3] = {cam->look_at[0] - cam->eye[0], cam->look_at[1] - cam->eye[1],
                 cam->look_at[2] - cam->eye[2]};
  double forward[3], right[3], up[3];
  safe_normalize(d, forward);
  v3_cross(forward, cam->up, right);
  const double rl = v3_len(right);
  if (rl > 0.0 && isfinite(rl)) {
    right[0] /= rl;
    right[1] /= rl;
    right[2] /= rl;
  } else {
    right[0] = 1.0;
    right[1] = 0.0;
    right[2] = 0.0;
  }
  v3_cross(right, forward, up);
  const float tan_half_fov = tanf(cam->fov_y_degrees * 0.5f * kPi / 180.0f);

  screen_corner corners[8];
  const float width_scale = (width > 1) ? (float)(width - 1) : 0.0f;
  const float height_scale = (height > 1) ? (float)(height - 1) : 0.0f;
  for (int index = 0; index < 8; ++index) {
    screen_corner p;
    memset(&p, 0, sizeof(p));
    p.depth = INFINITY;
    p.world[0] = (index & 1) ? bounds_max[0] : bounds_min[0];
    p.world[1] = (index & 2) ? bounds_max[1] : bounds_min[1];
    p.world[2] = (index & 4) ? bounds_max[2] : bounds_min[2];
    const double rel[3] = {p.world[0] - cam->eye[0], p.world[1] - cam->eye[1],
                           p.world[2] - cam->eye[2]};
    const float depth = (float)v3_dot(rel, forward);
    if (!(depth > 0.0f) || !isfinite(depth)) {
      corners[index] = p;
      continue;
    }
    const float x_cam = (float)v3_dot(rel, right);
    const float y_cam = (float)v3_dot(rel, up);
    const float ndc_x = x_cam / (depth * tan_half_fov * aspect);
    const float ndc_y = y_cam / (depth * tan_half_fov);
    if (!isfinite(ndc_x) || !isfinite(ndc_y)) {
      corners[index] = p;
      continue;
    }
    p.x = (ndc_x * 0.5f + 0.5f) * width_scale;
    p.y = (ndc_y * 0.5f + 0.5f) * height_scale;
    p.depth = depth;
    p.valid = 1;
    corners[index] = p;
  }

  static const int edges[12][2] = {{0, 1}, {1, 3}, {3, 2}, {2, 0}, {4, 5}, {5, 7},
                                   {7, 6}, {6, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
  const float overlay_depth = -3.402823466e+38f; /* numeric_limits<float>::lowest() */
  const float pixel_radius = 0.5f * (float)((sqrt_antialiasing > 1) ? sqrt_antialiasing : 1);
  const float influence_radius = pixel_radius + 0.5f;
  const float coverage_scale = 0.6f;

#define BLEND_SAMPLE(px_, py_, coverage_)                                   \
  do {                                                                      \
    const int bx_ = (px_), by_ = (py_);                                     \
    if (bx_ < 0 || bx_ >= width || by_ < 0 || by_ >= height) break;         \
    const float cc_ = clampf((coverage_), 0.0f, 1.0f);                      \
    if (cc_ <= 0.0f) break;                                                 \
    float *buf_ = image + ((size_t)by_ * width + bx_) * 5;                  \
    const float sa_ = cc_;                                                  \
    const float sr_ = 1.0f * sa_, sg_ = 1.0f * sa_, sb_ = 1.0f * sa_;       \
    buf_[0] = sr_ + buf_[0] * (1.0f - sa_);                                 \
    buf_[1] = sg_ + buf_[1] * (1.0f - sa_);                                 \
    buf_[2] = sb_ + buf_[2] * (1.0f - sa_);                                 \
    buf_[3] = sa_ + buf_[3] * (1.0f - sa_);                                 \
    buf_[4] = overlay_depth;                                                \
  } while (0)

  for (int e = 0; e < 12; ++e) {
    const screen_corner *start = &corners[edges[e][0]];
    const screen_corner *end = &corners[edges[e][1]];
    if (!start->valid || !end->valid) continue;
    const float min_x = minf(start->x, end->x) - influence_radius;
    const float max_x = maxf(start->x, end->x) + influence_radius;
    const float min_y = minf(start->y, end->y) - influence_radius;
    const float max_y = maxf(start->y, end->y) + influence_radius;
    const int fx = (int)floorf(min_x), cx = (int)ceilf(max_x);
    const int fy = (int)floorf(min_y), cy = (int)ceilf(max_y);
    const int x_begin = (fx > 0) ? fx : 0;
    const int x_end = (cx < width - 1) ? cx : width - 1;
    const int y_begin = (fy > 0) ? fy : 0;
    const int y_end = (cy < height - 1) ? cy : height - 1;
    const float edge_dx = end->x - start->x;
    const float edge_dy = end->y - start->y;
    const float len_sq = edge_dx * edge_dx + edge_dy * edge_dy;
    if (!(len_sq > 0.0f)) {
      BLEND_SAMPLE((int)lroundf(start->x), (int)lroundf(start->y), 1.0f);
      continue;
    }
    for (int py = y_begin; py <= y_end; ++py) {
      const float sample_y = (float)py + 0.5f;
      for (int px = x_begin; px <= x_end; ++px) {
        const float sample_x = (float)px + 0.5f;
        const float apx = sample_x - start->x;
        const float apy = sample_y - start->y;
        float t = (apx * edge_dx + apy * edge_dy) / len_sq;
        t = clampf(t, 0.0f, 1.0f);
        const float closest_x = start->x + (end->x - start->x) * t;
        const float closest_y = start->y + (end->y - start->y) * t;
        const float dist_x = sample_x - closest_x;
        const float dist_y = sample_y - closest_y;
        const float distance = sqrtf(dist_x * dist_x + dist_y * dist_y);
        const float coverage =
            clampf((pixel_radius + 0.5f - distance) * coverage_scale, 0.0f, 1.0f);
        if (coverage <= 0.0f) continue;
        BLEND_SAMPLE(px, py, coverage);
      }
    }
  }
#undef BLEND_SAMPLE
}

uint64_t orc_fnv1a64(const void *data, uint64_t n_bytes) {
  const unsigned char *p = (const unsigned char *)data;
  uint64_t h = 0xcbf29ce484222325ULL;
  for (uint64_t i = 0; i < n_bytes; ++i) {
    h ^= p[i];
    h *= 0x100000001b3ULL;
  }
  return h;
}
