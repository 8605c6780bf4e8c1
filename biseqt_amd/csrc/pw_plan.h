// pw_plan.h -- host-side planning of one alignment problem: the arithmetic of the reference's
// dptable_init (table dimensions, band clamp, feasibility: _pw_internals.c:8-62) plus the geometry the
// wavefront kernel needs (step range, steady-phase blocks, begin / end rules).  Pure C++, no HIP:
// shared by the product library and by the CPU lane emulator in tests/emu.
#ifndef PW_PLAN_H
#define PW_PLAN_H

#include <stdint.h>
#include <stdlib.h>

#include "pw_types.h"

namespace pw {

struct Plan {
  int rc;              // 0, or -1 exactly where the reference's dptable_init returns -1
  int clamped;         // the band was reduced to the table limits (_pw_internals.c:29-36)
  int dmin, dmax;      // band after the clamp (STD mode: -Y, X)
  int num_rows;        // dptable.num_rows: X + 1 (STD) or 1 + dmax - dmin (banded)
  int64_t cells;       // cells the reference allocates = the GCUPS denominator
  int ndiag;
  int s0, nblocks, steady_b0, steady_b1;
  int brule, endrule;
};

static inline int plan_len(int X, int Y, int d) {       // cells on diagonal d (_pw_internals.c:56)
  return 1 + (d > 0 ? 0 : d) + (X - d > Y ? Y : X - d);
}

// sum over d = dmin .. dmax of plan_len(X, Y, d) in closed form (dmin <= dmax, -Y <= dmin, dmax <= X): three
// arithmetic series -- the number of diagonals, min(d, 0) over the negative ones, min(X - d, Y) split at d = X - Y.
static inline int64_t plan_series(int64_t a, int64_t b) { return a > b ? 0 : (a + b) * (b - a + 1) / 2; }   // a + ... + b
static inline int64_t plan_band_cells(int X, int Y, int dmin, int dmax) {
  const int64_t n = (int64_t)dmax - dmin + 1;
  int64_t cells = n;
  cells += plan_series(dmin, dmax < -1 ? dmax : -1);                      // min(d, 0)
  const int64_t k = (int64_t)X - Y;                                       // X - d > Y  <=>  d < k
  const int64_t lo_end = dmax < k - 1 ? dmax : k - 1;                     // d in [dmin, lo_end]: min = Y
  if (lo_end >= dmin) cells += (lo_end - dmin + 1) * (int64_t)Y;
  const int64_t hi_beg = dmin > k ? dmin : k;                             // d in [hi_beg, dmax]: min = X - d
  if (hi_beg <= dmax) cells += (dmax - hi_beg + 1) * (int64_t)X - plan_series(hi_beg, dmax);
  return cells;
}

// Begin rule (_alnchoice_B, _pw_internals.c:161-209) and end rule (:303-414) of an alignment type.
static inline void plan_rules(int mode, int type, int* brule, int* endrule) {
  if (mode == STD_MODE) {
    *brule = (type == LOCAL || type == END_ANCHORED) ? BRULE_ANY
             : (type == OVERLAP || type == END_ANCHORED_OVERLAP) ? BRULE_EDGE : BRULE_ORIGIN;
    *endrule = (type == GLOBAL || type == END_ANCHORED || type == END_ANCHORED_OVERLAP) ? END_CORNER
               : (type == OVERLAP || type == START_ANCHORED_OVERLAP) ? END_STD_OVERLAP : END_STD_LOCAL;
  } else {
    *brule = type == B_LOCAL ? BRULE_ANY : type == B_OVERLAP ? BRULE_EDGE : BRULE_ORIGIN;
    *endrule = type == B_GLOBAL ? END_CORNER : type == B_OVERLAP ? END_BANDED_OVERLAP : END_BANDED_LOCAL;
  }
}

static inline Plan plan_problem(int mode, int type, int X, int Y, int dmin_in, int dmax_in) {
  Plan p;
  p.rc = 0; p.clamped = 0; p.cells = 0; p.ndiag = 0;
  p.s0 = 0; p.nblocks = 0; p.steady_b0 = 0; p.steady_b1 = 0;
  plan_rules(mode, type, &p.brule, &p.endrule);
  if (mode == STD_MODE) {
    p.dmin = -Y; p.dmax = X; p.num_rows = X + 1;
    p.cells = (int64_t)(X + 1) * (int64_t)(Y + 1);
  } else {
    int dmin = dmin_in, dmax = dmax_in;
    if (dmax > X || dmin < -Y) {                       // :29-36
      dmax = dmax > X ? X : dmax;
      dmin = dmin < -Y ? -Y : dmin;
      p.clamped = 1;
    }
    p.dmin = dmin; p.dmax = dmax;
    const int dend = X - Y;
    if (type == B_GLOBAL && (dend > dmax || dend < dmin || (int64_t)dmax * dmin > 0)) { p.rc = -1; return p; }  // :38-43
    p.num_rows = 1 + dmax - dmin;
    if (p.num_rows < 0) { p.rc = -1; return p; }       // :46-49
    if (p.num_rows == 0) { p.ndiag = 0; return p; }    // an empty table: the reference goes on with zero rows
    p.cells = plan_band_cells(X, Y, dmin, dmax);
  }
  p.ndiag = 1 + p.dmax - p.dmin;
  // first / last anti-diagonal that holds an in-band cell
  const int amin = p.dmin > 0 ? p.dmin : (p.dmax < 0 ? -p.dmax : 0);          // min |d|
  int s_last = 0, tf_max = 0, tl_min = 0x7fffffff;
  p.s0 = amin - (((amin - p.dmin) % 2 + 2) % 2);        // s0 == dmin (mod 2), s0 <= amin
  // |d| is maximal at a band end; tlast(d) = |d| + 2 (len(d) - 1) is piecewise linear in d with
  // breakpoints at 0 and X - Y, so its extrema sit at the band ends or at those breakpoints.
  const int cand[4] = {p.dmin, p.dmax, 0, X - Y};
  for (int k = 0; k < 4; k++) {
    const int d = cand[k];
    if (d < p.dmin || d > p.dmax) continue;
    const int ad = d < 0 ? -d : d;
    const int tl = ad + 2 * (plan_len(X, Y, d) - 1);
    if (tl > s_last) s_last = tl;
    if (ad - p.s0 > tf_max) tf_max = ad - p.s0;
    if (tl - p.s0 < tl_min) tl_min = tl - p.s0;
  }
  const int nsteps = s_last - p.s0 + 1;
  p.nblocks = (nsteps + 15) / 16;
  // block b is steady iff every in-band diagonal holds its FIRST cell strictly before step 16 b (first
  // cells carry the begin rule and have missing predecessors) and none has ended before step 16 b + 15
  p.steady_b0 = tf_max / 16 + 1;
  p.steady_b1 = (tl_min + 1) / 16;
  if (p.steady_b1 > p.nblocks) p.steady_b1 = p.nblocks;
  if (p.steady_b1 < p.steady_b0) p.steady_b1 = p.steady_b0;
  return p;
}

// Smallest supported diagonals-per-lane that covers ndiag diagonals with one wavefront, or 0.
static inline int plan_pick_bk(int ndiag, const int* supported, int n) {
  for (int i = 0; i < n; i++) if ((int64_t)64 * supported[i] >= ndiag) return supported[i];
  return 0;
}

}  // namespace pw
#endif
