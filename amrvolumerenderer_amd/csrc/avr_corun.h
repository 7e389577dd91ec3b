// The decision logic of the frame driver's co-run search (avr_renderer.cpp), free of any GPU
// call so that it can be exercised on the CPU (tests/cxx/corun_test.cpp).
#ifndef AVR_CORUN_H
#define AVR_CORUN_H

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <ctime>

// How the classify pass and the march of a rank share the GPU is MEASURED, not assumed.  Beside
// the march the classify pass takes memory-system time from it in proportion to the bandwidth it
// reaches -- whatever its arithmetic, occupancy or cache policy (profiles/experiments_rounds_1_to_3.md section 3) -- so the
// frame is shortest where the two take equally long; how many classify workgroups a CU admits
// (an LDS reserve per workgroup, avr_context_set_classify_lds_reserve) moves that balance, and
// for the short kernels of an N-rank share running them back to back can win outright.  A third
// way (round 3): every frame's classify pass and march back to back on ONE stream, the even
// frames on one stream and the odd frames on another ("paired") -- the classify pass of frame f+1
// still runs beside the march of frame f, but a march no longer queues behind its predecessor,
// so its tail (a rank of eight keeps 3.3 of 6 waves per SIMD resident on average) runs beside
// whatever the other stream has next: a rank of eight 0.165 -> 0.145 ms, one rank 0.986 -> 1.015.
// The candidates are tried in turn on the running pipeline -- back to back, side by side with a
// reserve of 0, 4, 8 ... KiB, paired with a reserve of 0, 8, 16 ... KiB -- each for a window of
// frames whose period is timed with two HIP events on the march stream; the driver refines around
// the best, re-times back to back against it, holds the winner, re-times it now and then and
// searches again if it has drifted (two slow windows in a row).  Scheduling only: never changes results.
struct CoRunTuner {
  static constexpr int kBackToBack = -1;       // candidate: both kernels on the march stream
  static constexpr int kReserveStep = 2048;    // candidate k >= 0: side by side, reserve k * step
  static constexpr int kLastCandidate = 28;    // 56 KiB: two classify workgroups per CU
  static constexpr int kPairedBase = 29;       // candidate kPairedBase + k: paired, reserve k * step
  static constexpr int kLastPaired = kPairedBase + kLastCandidate;
  static constexpr int kCoarse = 2;            // the first pass takes every second reserve
  static constexpr int kPairedCoarse = 4;      // ... of the paired layout every fourth
  static bool is_paired(int c) { return c >= kPairedBase; }
  static int reserve_index(int c) { return c >= kPairedBase ? c - kPairedBase : c; }
  static constexpr int kSettleFrames = 3;      // frames ignored after a change of candidate (at least)
  static constexpr int kWindowFrames = 40;     // frames timed per candidate (at most)
  static constexpr float kWindowMs = 8.0f;     // ... as many as fill this time, two at least: the
                                               // 0.2 ms frames of an N = 8 share need ~40 for a
                                               // period good to a per cent, 1 ms frames 8
  static constexpr int kHoldFrames = 360;      // frames between re-timings of the held candidate
  static constexpr float kDrift = 1.05f;       // held candidate this much slower, twice in a row:
                                               // search again (the reserve one step past the best
                                               // one is 6 % slower: a window that flattered it once
                                               // must not keep the driver there)

  // kSearch: (back to back, then) reserves 0, 4, 8 ... KiB; kRefine: the two reserves either
  // side of the best one; kVerify: back to back and the best reserve once more; kHold: the
  // winner.  (No early exit: over the reserve the period is flat, dips and rises again, and for
  // the short kernels of an N-rank share the dip lies at the far end -- a search that stopped
  // on the flat stretch missed it.)
  // kBalance (round 5, one rank's side-by-side frames): decided on the two kernels' own DURATIONS,
  // not on timed windows of the frame's period.  Each kernel is timed by two events on its stream;
  // a duration answers a new reserve within a few frames, where the period only shows it after the
  // classify stream's lead of two frames has run out (what the full search's 40 settling frames
  // per finalist are for); the frame's period is the longer of the two streams' (kernel + what lies
  // between two kernels of the stream).  A bisection, then five finalists (below): ~90 frames
  // instead of 600-1500, then kHold, whose re-timing falls back to the full search (kSearch) if the
  // held reserve's period drifts.
  enum Phase { kSearch, kRefine, kVerify, kHold, kBalance } phase = kSearch;
  int first = kBackToBack, last = kLastCandidate;  // the candidates the caller allows
  int candidate = kBackToBack;
  int best = kBackToBack, best_beside = 0, second_beside = -1;
  float best_ms = 0.0f, best_beside_ms = 0.0f, second_beside_ms = 0.0f;
  // the two best candidates of each layout ([0] side by side, [1] paired) during the search;
  // best_beside / second_beside above are those of the layout the search went on with
  int layout_best[2] = {0, kPairedBase}, layout_second[2] = {-1, -1};
  float layout_best_ms[2] = {0.0f, 0.0f}, layout_second_ms[2] = {0.0f, 0.0f};
  // A window timed while the candidates change reads the side-by-side layout 3-5 % slower than it
  // runs once held (its two streams take tens of frames to find their phase; measured on config-2
  // and config-3: 0.402 in the search, 0.381 held) and the paired layout as it is: paired is only
  // gone on with if it wins by more than that.
  static constexpr float kPairedMargin = 1.02f;
  static constexpr float kPairedMaxPeriodMs = 0.35f;  // one rank: frames longer than this never pair
  int verify[3] = {0, 0, 0}, n_verify = 0, verify_at = 0;  // kVerify: candidates re-timed in turn
  int refined = 0, repeated = 0;
  bool drift_suspected = false;  // kHold: the last window read slow
  long windows = 0;
  // Ranks of several search as ONE system (round 4): every frame is a collective, so a rank that
  // walks through its candidates alone stalls all the others and reads their candidates' periods
  // as its own.  Coordinated, every rank holds the same candidate in the same frames -- the state
  // machine below is a pure function of the frame count and of the reported periods -- and the
  // period reported for a window is the MAXIMUM over the ranks (one small allgather on the
  // communicator's control plane per window, avr_comm_control_allgather), taken by all ranks at
  // the same frame: kReportLag frames after the window's last one, when each rank's own events
  // have long happened (host-side back-pressure keeps at most three frames in flight).  What
  // happens to ONE rank only (its pipeline drained because a buffer grew) must not move its
  // state: it voids the rank's window, and a window void on any rank is timed again by all.
  bool coordinated = false;
  bool window_void = false;
  int frames_since_close = 0;
  static constexpr int kReportLag = 3;
  static constexpr float kHoldMs = 500.0f;  // coordinated: the held candidate is re-timed this often
  // the window in progress
  int frames_at_candidate = 0;  // since the candidate was chosen (or an interruption)
  bool open = false, closing = false;
  int window_length = kWindowFrames;  // of the open / closing window
  float last_period_ms = 0.0f;

  // long frames (config-5: 35 ms) get short windows: the search should take seconds, not minutes
  int frames_per_window() const {
    if (last_period_ms <= 0.0f) return 4;
    const int frames = static_cast<int>(std::ceil(kWindowMs / last_period_ms));
    return std::min(std::max(frames, 2), kWindowFrames);  // 1 ms -> 8, 35 ms -> 2, 0.2 ms -> 40
  }

  // ---- kBalance -------------------------------------------------------------------------------
  // Two stages.  (1) A bisection over the 29 reserves on WHICH stream's period is the longer one:
  // the classify pass's duration rises and the march's falls with the reserve, and away from the
  // balance the sign is robust -- also against the pipeline's memory of the reserve before (after a
  // jump across the scale the classify stream, which runs up to two frames ahead, takes tens of
  // frames to settle: comparing reserves by their timed COST across such jumps was tried and holds
  // the wrong one five times out of six).  (2) What a CU admits is a packing of both kernels'
  // workgroups into its LDS, wave slots and registers, so near the balance the period is a
  // staircase with pockets (config-4: 24 KiB 0.965 / 0.965 ms, 26 KiB 1.02 / 1.09 -- both kernels
  // slower -- 28 KiB 1.03 / 0.98) and the bisection closes a step or two beside the best reserve
  // every third time -- above it, as a rule, and also when the seed, timed right after start-up
  // with both kernels still 10-15 % long, sent it the wrong way.  So the five reserves from two
  // below the closed bracket to one above it are then timed afresh in ascending order (neighbours:
  // small jumps) and the one with the shortest period is held.
  static constexpr int kBalanceSettle = 2;      // reports ignored after a change of reserve (bisection)
  static constexpr int kBalanceStepFrames = 2;  // reports averaged per step of the bisection
  static constexpr int kFinalistSettle = 3;     // ... ignored / averaged per finalist
  static constexpr int kFinalistFrames = 4;
  static constexpr int kFirstFinalistSettle = 6;  // (the first one follows a jump across the bracket)
  static constexpr int kBalanceSeed = 12;    // 24 KiB: where config-4's one-rank frame balances
  static constexpr float kSeedBalanced = 0.06f;  // |classify - march| at the seed, of the longer: no bisection
  static constexpr float kBalanceMinMs = 0.35f;  // shorter frames take the full search (the paired
                                                 // layout and back to back are candidates there)
  static constexpr float kClassifyGapMs = 0.014f, kMarchGapMs = 0.028f;  // between two kernels of a stream
  bool balance_allowed = false;  // set_balance(): one rank, nothing fixed by the caller
  bool balance_failed = false;   // this renderer's frames are too short for it: full search
  int b_lo = 0, b_hi = 0;        // the reserve where both take equally long lies in [b_lo, b_hi]
  bool b_final = false;          // the bracket has closed: the finalists are being timed
  int finalists[5] = {0, 0, 0, 0, 0}, n_finalists = 0, b_final_at = 0;
  float finalist_ms[5] = {0, 0, 0, 0, 0};
  static constexpr float kPlayoffWithin = 1.06f;
  static constexpr float kPlayoffDecided = 0.025f;  // of the smaller sum of readings
  static constexpr float kPlayoffEquivalent = 0.01f;
  static constexpr int kPlayoffReadings = 4;        // per finalist at most (the first one included)
  static constexpr int kPlayoffFrames = 4;
  bool b_playoff = false;        // the two best finalists are being read again
  float playoff_sum_ms[2] = {0, 0};
  int playoff_readings = 0;
  int b_reports = 0;
  float b_classify = 0.0f, b_march = 0.0f;
  void set_balance(bool allowed) {
    if (allowed == balance_allowed) return;
    balance_allowed = allowed;
    restart();
  }
  bool balancing() const { return phase == kBalance; }
  void begin_balance_step(int next) {
    candidate = next;
    b_reports = 0;
    b_classify = b_march = 0.0f;
  }
  // One frame's kernel durations (events on the two streams), in frame order, with the candidate
  // the frame was queued under; frames of an older candidate are still in flight after a change.
  void report_durations(int frame_candidate, float classify_ms, float march_ms) {
    if (phase != kBalance || frame_candidate != candidate) return;
    // (the seed's reading decides whether the bisection runs at all, and the first frames of a
    // pipeline read 10-15 % long: it gets the finalists' settling and four frames)
    const bool seed_step = !b_final && b_lo == 0 && b_hi == kLastCandidate && candidate == kBalanceSeed;
    const int settle = seed_step ? kFirstFinalistSettle
                       : !b_final ? kBalanceSettle
                                  : (b_final_at == 0 ? kFirstFinalistSettle : kFinalistSettle);
    const int frames = b_playoff ? kPlayoffFrames : (b_final || seed_step) ? kFinalistFrames : kBalanceStepFrames;
    if (++b_reports <= settle) return;
    b_classify += classify_ms;
    b_march += march_ms;
    if (b_reports < settle + frames) return;
    // Each stream's period is its kernel plus what lies between two of its kernels: the classify
    // stream runs its passes back to back (14 us), a march also waits for its classify pass's
    // event on the other stream and for its descriptors (28 us) -- measured: config-2 at 51200
    // 0.3643 / 0.3658 ms -> period 0.394, at 53248 0.367 / 0.350 -> 0.381; config-3 at 36864
    // 0.4627 / 0.4616 -> 0.488, at 38912 0.4714 / 0.4486 -> 0.487; config-4 at 24576 0.983 / 0.975
    // -> 1.002.  The frame's period is the longer of the two.
    const float classify = b_classify / static_cast<float>(frames) + kClassifyGapMs;
    const float march = b_march / static_cast<float>(frames) + kMarchGapMs;
    const float longer = std::max(classify, march);
    ++windows;
    last_period_ms = longer;
    static const bool trace = std::getenv("AVR_CORUN_TRACE") != nullptr;
    if (trace) {
      std::fprintf(stderr, "corun: balance candidate %d classify %.4f march %.4f ms (with the streams' gaps) [%d, %d]%s\n",
                   candidate, classify, march, b_lo, b_hi, b_final ? " finalist" : "");
    }
    if (longer < kBalanceMinMs) {  // short frames: the layouts themselves are in question
      balance_failed = true;
      restart();
      return;
    }
    if (b_final) {
      finalist_ms[b_final_at] = longer;
      if (++b_final_at < n_finalists) {
        begin_balance_step(finalists[b_final_at]);
        return;
      }
      int chosen = 0, second = -1;
      for (int i = 1; i < n_finalists; ++i) {
        if (finalist_ms[i] < finalist_ms[chosen]) chosen = i;
      }
      for (int i = 0; i < n_finalists; ++i) {
        if (i != chosen && (second < 0 || finalist_ms[i] < finalist_ms[second])) second = i;
      }
      // A close call (a reading of four frames after three is good to 2-3 %; the pockets beside the
      // best reserve are 4-8 % slower, so one reading each orders the two wrongly one time in
      // eight): within 6 % the best two are read again, in ascending order, until their means
      // differ by 2.5 % (or by less than 1 %: either will do) or each has four readings -- a sequential test: clear cases cost nothing,
      // and only the ambiguous ones the frames.
      if (!b_playoff && second >= 0 && finalist_ms[second] < finalist_ms[chosen] * kPlayoffWithin) {
        b_playoff = true;
        const int lo = std::min(chosen, second), hi = std::max(chosen, second);
        const int pair[2] = {finalists[lo], finalists[hi]};
        const float pair_ms[2] = {finalist_ms[lo], finalist_ms[hi]};
        n_finalists = 2;
        for (int i = 0; i < 2; ++i) {
          finalists[i] = pair[i];
          playoff_sum_ms[i] = pair_ms[i];
        }
        playoff_readings = 1;
        b_final_at = 0;
        begin_balance_step(finalists[0]);
        return;
      }
      if (b_playoff) {
        playoff_sum_ms[0] += finalist_ms[0];
        playoff_sum_ms[1] += finalist_ms[1];
        ++playoff_readings;
        const float gap = std::fabs(playoff_sum_ms[0] - playoff_sum_ms[1]) /
                          std::min(playoff_sum_ms[0], playoff_sum_ms[1]);
        // (means within 1 % after two readings each: the two are as good as each other)
        if (gap < kPlayoffDecided && gap >= kPlayoffEquivalent && playoff_readings < kPlayoffReadings) {
          b_final_at = 0;
          begin_balance_step(finalists[0]);
          return;
        }
        chosen = playoff_sum_ms[0] <= playoff_sum_ms[1] ? 0 : 1;
      }
      best = candidate = finalists[chosen];
      best_beside = best;
      best_ms = 0.0f;  // (a period, not a duration: the first window of the hold sets it)
      phase = kHold;
      interrupt();
      return;
    }
    if (candidate == kBalanceSeed && b_lo == 0 && b_hi == kLastCandidate &&
        std::fabs(classify - march) <= kSeedBalanced * longer) {
      // the seed already balances the two (config-4's 24 KiB): no bisection, the five reserves
      // around it are timed at once
      b_lo = kBalanceSeed;
      b_hi = kBalanceSeed + 1;
    } else if (classify < march) {  // the march's stream is the longer one: hold the classify pass back further
      b_lo = candidate;
    } else {
      b_hi = candidate;
    }
    // (an end of the scale that every step pointed beyond is where the bracket closes)
    const int next = (b_lo + b_hi) / 2;
    if (b_hi - b_lo > 1 || (b_hi - b_lo == 1 && ((b_lo == 0 && candidate != 0) ||
                                                 (b_hi == kLastCandidate && candidate != kLastCandidate)))) {
      begin_balance_step(b_hi - b_lo > 1 ? next : (b_lo == 0 ? 0 : kLastCandidate));
      return;
    }
    b_final = true;
    b_final_at = 0;
    n_finalists = 0;
    for (int c = b_lo - 2; c <= b_hi + 1; ++c) {
      if (c >= 0 && c <= kLastCandidate && n_finalists < 5) finalists[n_finalists++] = c;
    }
    begin_balance_step(finalists[0]);
  }

  // Where the search starts is what a caller keeps who never renders enough frames back to back
  // for a window to complete (bursts of a few frames between synchronisations): side by side
  // without a reserve for one rank (there back to back is 1.29 ms against 1.05), back to back
  // for a rank of several (there the unreserved pair can be the worst choice).  Back to back
  // is always re-timed at the end (kVerify).
  bool start_beside = false;

  void restrict_to(int first_candidate, int last_candidate, bool beside_first) {
    if (first_candidate == first && last_candidate == last && beside_first == start_beside) return;
    first = first_candidate;
    last = last_candidate;
    start_beside = beside_first;
    restart();
  }
  void restart() {
    phase = kSearch;
    candidate = best = (start_beside && last >= 0) ? std::max(first, 0) : first;
    // the whole side-by-side scale is open to one rank that has not been found too short for it
    if (balance_allowed && !balance_failed && !coordinated && start_beside && first <= 0 &&
        last >= kLastCandidate) {
      phase = kBalance;
      b_lo = 0;
      b_hi = kLastCandidate;
      b_final = b_playoff = false;
      begin_balance_step(kBalanceSeed);
      best = candidate;
    }
    best_beside = 0;
    second_beside = -1;
    best_ms = best_beside_ms = second_beside_ms = 0.0f;
    layout_best[0] = 0;
    layout_best[1] = kPairedBase;
    layout_second[0] = layout_second[1] = -1;
    layout_best_ms[0] = layout_best_ms[1] = layout_second_ms[0] = layout_second_ms[1] = 0.0f;
    n_verify = verify_at = 0;
    refined = 0;
    drift_suspected = false;
    shrink = 0;
    interrupt();
  }
  void interrupt() {  // the pipeline drained or the candidate changed: the window is void
    frames_at_candidate = 0;
    open = closing = false;
    window_void = false;
    frames_since_close = 0;
  }
  void drained() {  // a window whose last frame was already queued stays valid
    if (coordinated) {  // (the frame counts are the ranks' common clock: only the window suffers)
      if (open) window_void = true;
      return;
    }
    if (phase == kBalance) {  // (the frames after a drain are not the steady state: settle again)
      begin_balance_step(candidate);
      return;
    }
    if (!closing) {
      if (phase == kVerify && shrink < 2) ++shrink;
      interrupt();
    }
  }
  void set_coordinated(bool on) {
    if (on == coordinated) return;
    coordinated = on;
    restart();
  }
  // Frames let pass after a change of candidate before its window opens.  The finalists get at
  // least kVerifySettle: a candidate one step past the best reserve is classify-bound by a few per
  // cent, and the classify stream's lead of two frames (three classified volumes) takes
  // 2 / 0.06 = ~30 frames to run out -- until then the pipeline still shows the period of the
  // candidate before (the fly-through bench held 26 KiB at 1.08 ms where 24 KiB runs 0.98: its 16
  // settling frames had flattered it in every window).
  static constexpr int kVerifySettle = 40;
  // A caller who drains the pipeline every so many frames voids every window longer than that:
  // each drain that interrupts a finalist halves what the finalists are given (down to the
  // search's own windows), so that the search still ends; reset when a new search starts.
  int shrink = 0;
  int settle_frames() const {
    const int frames = std::max(kSettleFrames, frames_per_window());
    return phase == kVerify ? std::max(frames, kVerifySettle >> shrink) : frames;
  }
  // frames the held candidate runs before it is timed again
  int hold_frames() const {
    if (!coordinated || last_period_ms <= 0.0f) return kHoldFrames;
    const float frames = kHoldMs / last_period_ms;
    return static_cast<int>(std::min(std::max(frames, static_cast<float>(kHoldFrames)), 20000.0f));
  }
  // coordinated: whether the closed window's period is to be agreed on in this frame
  bool report_due() {
    if (!coordinated || !closing) return false;
    return ++frames_since_close >= kReportLag;
  }
  // coordinated: some rank's window was void -- the same candidate is timed again, by all
  void retime() {
    // (all ranks take this step in the same frame, so the shrinking below keeps them in lockstep:
    // a caller who synchronises more often than a finalist's window is long voids every one of
    // them on all ranks -- each void window halves what the finalists are given, down to the
    // search's own windows, so that the search still ends, as the per-rank search does in drained())
    if (phase == kVerify && shrink < 2) ++shrink;
    const int start = (phase == kHold) ? hold_frames() : settle_frames();
    open = closing = false;
    window_void = false;
    frames_since_close = 0;
    frames_at_candidate = std::max(0, start - std::max(kSettleFrames, frames_per_window()));
  }
  // Once per frame, after its march has been queued, while no window is waiting for its end
  // event: what to record on the march stream now.
  enum Action { kNothing, kOpenWindow, kCloseWindow };
  Action frame() {
    if (closing || phase == kBalance) return kNothing;
    ++frames_at_candidate;
    // After a change of candidate the two streams take a while to find their steady phase (side
    // by side: tens of frames; a window timed right after the change read 3-5 % slow, which was
    // harmless while every candidate was side by side and is not beside the paired layout, which
    // settles at once): as many frames are let pass as the window then times.
    const int start = (phase == kHold) ? hold_frames() : settle_frames();
    if (!open && frames_at_candidate >= start) {
      open = true;
      // paired, the window's two events must lie on the same one of the two streams (whose
      // frames end in pairs, not evenly spaced: an odd window read a period 1/L short or long)
      // (the finalists' windows are four times as long: 8 frames of 1 ms are good to 1.5 %, and the
      // reserve next to the best one is often within that)
      // (the held candidate's are four times as long: a rare window, and at 0.1 ms per frame forty
      // frames are 4 ms -- short enough for one hiccup to read 5 % slow; the window that checks a
      // suspected drift is eight times as long)
      window_length = frames_per_window() *
                      (phase == kVerify ? (4 >> shrink) : phase == kHold ? (drift_suspected ? 8 : 4) : 1);
      if (is_paired(candidate)) window_length += window_length & 1;
      return kOpenWindow;
    }
    if (open && frames_at_candidate >= start + window_length) {
      open = false;
      closing = true;
      return kCloseWindow;
    }
    return kNothing;
  }
  bool tuning() const { return first != last; }
  bool settled() const { return !tuning() || phase == kHold; }

  // one timed window of the current candidate
  void report(float period_ms) {
    ++windows;
    last_period_ms = period_ms;
    static const bool trace = std::getenv("AVR_CORUN_TRACE") != nullptr;  // diagnostics
    if (trace) {
      timespec now{};
      clock_gettime(CLOCK_MONOTONIC, &now);
      std::fprintf(stderr, "corun: phase %d candidate %d period %.4f ms at %.4f\n",
                   static_cast<int>(phase), candidate, period_ms,
                   static_cast<double>(now.tv_sec) + 1e-9 * static_cast<double>(now.tv_nsec));
    }
    // diagnostics: AVR_CORUN_REPEAT=n times every candidate n windows on end (how long a layout
    // takes to reach its steady period after a change)
    static const int repeat = std::getenv("AVR_CORUN_REPEAT") ? std::atoi(std::getenv("AVR_CORUN_REPEAT")) : 0;
    if (repeat > 1 && phase != kHold && ++repeated < repeat) {
      open = closing = false;
      frames_at_candidate = std::max(kSettleFrames, frames_per_window()) - 1;
      return;
    }
    repeated = 0;
    if (phase == kHold) {
      if (best_ms == 0.0f) best_ms = period_ms;  // (held after kBalance: the first period seen)
      // (a rank of several shares its period with the other ranks' exchange: twice the margin)
      if (period_ms > best_ms * (start_beside ? kDrift : 2.0f * kDrift - 1.0f)) {
        // one slow window (a hiccup of the exchange, another process on the node) is not a
        // drift: the candidate is timed once more right away, and only a second slow window
        // starts a new search (whose ~30 candidates include much slower ones)
        if (drift_suspected) {
          balance_failed = true;  // (whatever was held, it drifted: the full search this time)
          restart();
        } else {
          drift_suspected = true;
          interrupt();
          frames_at_candidate = hold_frames() - std::max(kSettleFrames, frames_per_window());
        }
      } else {
        drift_suspected = false;
        best_ms = 0.75f * best_ms + 0.25f * period_ms;
        interrupt();
      }
      return;
    }
    if (best_ms == 0.0f || period_ms < best_ms) {
      best_ms = period_ms;
      best = candidate;
    }
    if (phase != kVerify && candidate >= 0) {  // the two best reserves of the candidate's layout
      const int layout = is_paired(candidate) ? 1 : 0;
      if (layout_best_ms[layout] == 0.0f || period_ms < layout_best_ms[layout]) {
        if (layout_best_ms[layout] != 0.0f) {
          layout_second[layout] = layout_best[layout];
          layout_second_ms[layout] = layout_best_ms[layout];
        }
        layout_best_ms[layout] = period_ms;
        layout_best[layout] = candidate;
      } else if (layout_second[layout] < 0 || period_ms < layout_second_ms[layout]) {
        layout_second[layout] = candidate;
        layout_second_ms[layout] = period_ms;
      }
      // the layout to go on with: paired only if it wins by the margin (or is all there is)
      const bool only_paired = first >= kPairedBase;
      const bool go_paired =
          only_paired || (layout_best_ms[1] != 0.0f &&
                          (layout_best_ms[0] == 0.0f ||
                           layout_best_ms[1] * kPairedMargin < layout_best_ms[0]));
      const int chosen = go_paired ? 1 : 0;
      if (layout_best_ms[chosen] != 0.0f) {
        best_beside = layout_best[chosen];
        best_beside_ms = layout_best_ms[chosen];
        second_beside = layout_second[chosen];
        second_beside_ms = layout_second_ms[chosen];
      }
    }
    if (phase == kSearch) {
      // back to back, the side-by-side reserves, then the paired ones
      const int last_beside = std::min(last, kLastCandidate);
      int next = -2;
      if (candidate < 0) {
        next = (last >= 0) ? 0 : -2;
      } else if (!is_paired(candidate)) {
        next = candidate + kCoarse;
        if (next > last_beside) {
          // The paired layout pays for SHORT kernels (a rank of four or eight: 0.15-0.3 ms frames);
          // one rank with frames above 0.35 ms has never held it (config-2 0.414 against 0.377 ms,
          // config-3, config-4 1.015 against 0.986): its eight coarse windows are not spent there.
          const bool worth_pairing = !(start_beside && best_ms > kPairedMaxPeriodMs);
          next = (last >= kPairedBase && (worth_pairing || first >= kPairedBase))
                     ? std::max(first, kPairedBase) : -2;
        }
      } else {
        next = candidate + kPairedCoarse;
        if (next > last) next = -2;
      }
      if (next != -2) {
        candidate = next;
      } else {
        phase = kRefine;
      }
    }
    if (phase == kRefine) {
      // the neighbours of the best one inside its layout (the centre may move), and a second
      // window for ITS neighbours of the coarse pass: one window of 8 frames is good to 1.5 %,
      // and the true best reserve had lost to its neighbour by one unlucky window now and then
      // (24 KiB read 1.010 once, 20 KiB was held: 1.002 ms against 0.993)
      int next = -1;
      while (refined < 4 && next < 0) {
        const bool paired = is_paired(best_beside);
        const int step = paired ? kPairedCoarse / 2 : 1;
        const int lo = paired ? std::max(first, kPairedBase) : std::max(first, 0);
        const int hi = paired ? last : std::min(last, kLastCandidate);
        static constexpr int kOffsets[4] = {-1, 1, -2, 2};
        const int probe = best_beside + kOffsets[refined] * step;
        ++refined;
        if (probe >= lo && probe <= hi && hi > lo) next = probe;
      }
      if (next >= 0) {
        candidate = next;
      } else {
        // The finalists once more, now that the pipeline has run for a while (the very first
        // windows after start-up have read up to 20 % fast) and so that one lucky window does
        // not decide (the reserve just past the dip sits on a cliff): back to back if allowed,
        // the best reserve and the runner-up; the best of THESE windows is held.
        n_verify = verify_at = 0;
        if (first == kBackToBack) verify[n_verify++] = kBackToBack;
        if (last >= 0) verify[n_verify++] = best_beside;
        if (last > 0 && second_beside >= 0) verify[n_verify++] = second_beside;
        if (n_verify >= 2) {
          phase = kVerify;
          candidate = verify[0];
          best_ms = 0.0f;
        } else {
          phase = kHold;
          candidate = best;
        }
      }
    } else if (phase == kVerify) {
      if (++verify_at < n_verify) {
        candidate = verify[verify_at];
      } else {
        phase = kHold;
        candidate = best;  // of the re-timed windows (best_ms was reset before them)
      }
    }
    interrupt();
  }
};

#endif  // AVR_CORUN_H
