#include <cmath>
// pw_model.h -- the planner's timing model: every constant the kernel choice depends on, in ONE table, and the estimates
// built from it.  Pure C++ (no HIP): used by pwlib_api.cpp (batch_plan) and readable by the tests.
//
// The planner has two kinds of kernels for wide pairs: kernels that take ALL pairs of a batch at once, one workgroup per
// pair (K2a, 32-bit / f64 / packed 16-bit body) -- their time is the slowest pair's chain of anti-diagonal steps -- and
// kernels that take the pairs ONE AFTER ANOTHER, each with the whole chip (K2c strips, K2b tiles) -- their time is the sum
// over the pairs.  For a batch of few pairs (latency mode) it estimates both and takes the one-after-another kernel when its
// estimate is below `margin` x the workgroups'.
//
// Where the numbers come from: tests/micro/few_pairs.py, shape_sweep.py, shape_sweep_small.py on MI355X
// (profiles/round2_e_few_pairs.txt, round2_f_shape_sweep.txt, round2_g_shape_sweep_small.txt).  A kernel that gets faster
// silently mis-tunes them: tests/micro/planner_check.py (and tests/test_gpu_planner.py, a bounded version of it) runs the
// planner's pick against the alternatives it rejected and fails when the pick is more than 25 % slower.
#ifndef PW_MODEL_H
#define PW_MODEL_H

#include <math.h>

namespace pw {

struct StepCost { int max_ndiag; double us; };      // microseconds per anti-diagonal step for bands up to max_ndiag diagonals

struct PlanModel {
  // K2c strip pipeline, per pair: launch + finish, per strip of 64 rows (the hop), per column step
  double strip_fixed_ms, strip_per_strip_ms, strip_per_col_ms;
  // K2b tiled kernel, per pair: launches of the first time block + finish, per anti-diagonal
  double tile_fixed_ms, tile_per_step_ms;
  // K2a, one anti-diagonal step of a workgroup's pair (8 wavefronts with 4 / 8 / 16 / 32 diagonals per lane)
  StepCost wg_i32[4];      // 32-bit body
  StepCost wg_f64[5];      // f64 body (the wide lanes spill kilobytes of registers: 8 kb x 8 kb takes 317 ms)
  StepCost wg_p16[3];      // packed 16-bit body (k_fill16_mw)
  double margin;           // one-after-another wins when its estimate < margin x the workgroups' estimate
  // The packed kernels' lane layouts: cost of one slot-step relative to the lane-packed form at 8 diagonals per lane, by
  // diagonals per lane 4, 8, .. 32 (pw_launch.h, kPackedBK), and what one pair per wavefront saves (its descriptor sits in
  // scalar registers).  Fitted to tests/micro/overlap_all_bench.py (bands of 9 .. 111 diagonals under the overlap rule, 20 000 and
  // 50 000 pairs per batch, PWLIB_PACKED_BK = 4s .. 32s) and tests/micro/ab_lane_packing.py (config 2's shape, local rule); both
  // agree within 3 % (profiles/round3_n_lane_width.txt, round3_m_lane_packing.txt).
  double seg_slot_cost[8];
  double one_pair_discount;
};

static const PlanModel kPlanModel = {
  /* strips (after the round-3 hand-over diet: tests/micro/few_pairs.py, profiles/round3_i_few_pairs.txt; config 3 itself comes out 9 % below) */ 0.045, 0.0086, 0.00006,
  /* tiles  */ 0.05, 0.00043,
  /* wg_i32 */ {{2048, 0.45}, {4096, 0.7}, {8192, 1.3}, {0x7fffffff, 5.5}},
  /* wg_f64 */ {{1024, 0.47}, {2048, 0.6}, {4096, 1.0}, {8192, 2.7}, {0x7fffffff, 19.8}},
  /* wg_p16 */ {{4096, 0.38}, {8192, 0.6}, {0x7fffffff, 1.2}},
  /* margin */ 0.9,
  /* seg_slot_cost, 4 .. 32 diagonals per lane */ {1.08, 1.0, 0.98, 1.04, 1.1, 1.1, 1.06, 1.06},
  /* one_pair_discount */ 0.93,
};

// n wavefronts on 1024 SIMDs: the last round is only partly filled.  With one or two rounds that costs in full (1250 wavefronts
// take as long as 2048 would); with many rounds wavefronts of different lengths even most of it out.
inline double last_round_factor(double nwaves) {
  const double r = nwaves < 1024.0 ? 1.0 : nwaves / 1024.0;
  const double c = std::ceil(r) / r;
  return r < 3.0 ? c : 1.0 + 0.3 * (c - 1.0);
}

template <int N> inline double step_us(const StepCost (&t)[N], int ndiag) {
  for (int i = 0; i < N; i++) if (ndiag <= t[i].max_ndiag) return t[i].us;
  return t[N - 1].us;
}

// Estimated times of a batch under each way of running it (ms); built pair by pair.
struct BatchEstimates {
  double strips_ms = 0, wgroups_std_ms = 0;        // standard-mode pairs: strips one after another / 32-bit workgroups at once
  double tiles_ms = 0, wgroups_i32_ms = 0, wgroups_f64_ms = 0, p16_ms = 0;
  // one solvable pair: `steps` anti-diagonals of its (banded) table, `ndiag` diagonals; X, Y only for standard-mode pairs
  void add_pair(const PlanModel& m, double steps, int ndiag) {
    tiles_ms += m.tile_fixed_ms + steps * m.tile_per_step_ms;
    wgroups_i32_ms = fmax(wgroups_i32_ms, steps * step_us(m.wg_i32, ndiag) * 1e-3);
    wgroups_f64_ms = fmax(wgroups_f64_ms, steps * step_us(m.wg_f64, ndiag) * 1e-3);
    p16_ms = fmax(p16_ms, steps * step_us(m.wg_p16, ndiag) * 1e-3);
  }
  void add_std_pair(const PlanModel& m, double X, double Y, int ndiag) {
    strips_ms += m.strip_fixed_ms + m.strip_per_strip_ms * ceil((X + 1) / 64.0) + m.strip_per_col_ms * (Y + 64);
    wgroups_std_ms = fmax(wgroups_std_ms, (X + Y) * step_us(m.wg_i32, ndiag) * 1e-3);
  }
  // the three decisions the model makes
  bool strips_beat_packed_workgroups(const PlanModel& m) const { return strips_ms < m.margin * p16_ms; }
  bool strips_beat_workgroups(const PlanModel& m) const { return strips_ms < m.margin * wgroups_std_ms; }
  bool tiles_beat_workgroups(const PlanModel& m, bool f64) const { return tiles_ms < m.margin * (f64 ? wgroups_f64_ms : wgroups_i32_ms); }
};

}  // namespace pw
#endif
