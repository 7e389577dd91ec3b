/*
 * avr_hip_debug.h -- test hooks and diagnostics of libavr_hip.so.
 *
 * NOT part of the drop-in boundary (include/avr_hip.h, what INTEGRATION.md binds): nothing a
 * product caller needs, everything the parity tests and the multi-rank rehearsals use to look
 * inside.  None of these changes a result.
 */
#ifndef AVR_HIP_DEBUG_H
#define AVR_HIP_DEBUG_H

#include "avr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: keeps hip_stream busy for `milliseconds` (1..2000; one wave on a bounded timer -- it
 * always ends) so that the deadline above can be exercised on a real stream. */
int avr_debug_stall_stream(void *hip_stream, int milliseconds);

/* Diagnostics for the parity tests: while set (device pointer to 5 x uint64; NULL = off), every
 * march launched with a samples_out counter also ADDS
 *   counters[0]  samples whose cell index took the exact IEEE divide of
 *                Common/VolumePainter.cpp:846-852 because the reciprocal product lay within the
 *                proven error bound of an integer ("Exact index without the divide"),
 *   counters[1..3]  samples of boxes indexed by the exact divide throughout (degenerate spacing) /
 *                by the reciprocal product / by the power-of-two product,
 *   counters[4]  non-empty pixels outside the row span of a tightened plan
 *                (avr_frame_plan_tighten): always 0.
 * Never changes results. */
int avr_context_set_march_counters(avr_context *ctx, uint64_t *counters_dev);

/* The candidate of each of the next `frames` frames of the renderer's co-run search (-1 back to
 * back, k >= 0 side by side with an LDS reserve of k * 2 KiB, 29 + k paired): the tests of the
 * coordinated search compare the ranks' histories frame by frame. */
int avr_renderer_set_corun_history(avr_renderer *renderer, int frames);
int avr_renderer_corun_history(const avr_renderer *renderer, int16_t *candidates_out, int capacity,
                               int *frames_out);

/* The fraction of a rank's boxes a plan's deciding frame may have sampled for the driver to take
 * up visibility speculation (avr_renderer_set_visibility_speculation; default 0.85): the tests set
 * it to 1 so that small scenes, whose rays reach nearly every box, exercise the machinery (the call
 * also drops the other condition: that the classify work saved is worth at least 0.15 ms). */
int avr_renderer_debug_set_speculation_threshold(avr_renderer *renderer, float sampled_fraction);

#ifdef __cplusplus
}
#endif

#endif /* AVR_HIP_DEBUG_H */
