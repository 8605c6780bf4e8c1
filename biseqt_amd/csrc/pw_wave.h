// pw_wave.h -- the per-lane program of the banded anti-diagonal wavefront fill ("K1") and of the
// traceback walker ("K4").
//
// What it computes: the reference's dptable_solve (biseqt/pwlib/pw.c:47-114) with its move
// generators (_pw_internals.c:161-299) and end-cell search (:303-414), restated as
// (H, 4-bit ordered tie mask) per cell, and dptable_traceback (pw.c:116-151)
// restated over the tie masks.
//
// How it is laid out on a 64-lane CDNA wavefront (one wavefront per sequence pair):
//   * band coordinates: diagonal d = x - y, anti-diagonal s = x + y.  Lane l owns the BK (even)
//     consecutive diagonals dd = l*BK .. l*BK+BK-1 (dd = d - dmin); their running state
//     (H, the gap offers U/L, mask accumulator, best-so-far) lives in registers.
//   * a step advances one anti-diagonal; slot j holds a cell on the steps t == j (mod 2), so each
//     step a lane updates R = BK/2 independent cells in place: the diagonal predecessor is the slot's
//     own previous value, the up/left predecessors are the neighbouring slots' values from the
//     previous step.  Only the two block-edge values cross lanes, one DPP wave shift per step.
//   * the sequences stream through the lanes systolically: origin letters move towards lower lanes,
//     mutant letters towards higher lanes, one DPP shift per step; the edge lanes are fed from the
//     sequence arena with wave-uniform loads.
//   * every 16 steps each slot has accumulated 8 tie masks = one dword; the wave stores them with
//     16-byte-per-lane, 1 KiB-contiguous stores (layout: pw_types.h, mask_word_index).
//
// The file is written against a tiny platform policy P (lane id and count, shifts by one lane, shuffles,
// broadcast across the wavefronts of a workgroup) so
// that the very same lane program is compiled by hipcc into the gfx950 kernels (pw_kernels.hip) and
// by g++ into a 64-fiber lockstep emulator used only by the CPU tests (tests/emu).  PW_FN is the
// function qualifier each side supplies.
#ifndef PW_WAVE_H
#define PW_WAVE_H

#ifndef PW_FN
#error "PW_FN must be defined by the includer"
#endif

#include "pw_types.h"

namespace pw {

template <typename T> struct ScoreTraits;
template <> struct ScoreTraits<int32_t> {
  static constexpr bool is_int = true;
  // "no such predecessor".  Real scores are kept within +-2^27 by the host planner, so a sentinel
  // (even after a few thousand additions of small scores) never reaches a real score, and the sum of
  // two sentinels still fits an int32.
  PW_FN static int32_t neg() { return -(1 << 28); }
  PW_FN static int32_t zero_blk() { return 0; }
  // (the forms the floating-point path uses; for integers they are the plain expressions)
  PW_FN static int32_t max2(int32_t a, int32_t b) { return a > b ? a : b; }
  PW_FN static int32_t gap_offer(bool kept, int32_t go, int32_t A) { return kept ? A : A + go; }
};
template <> struct ScoreTraits<double> {
  static constexpr bool is_int = false;
  PW_FN static double neg() { return -1.0e300; }
  PW_FN static double zero_blk() { return -0.0; }   // x + (-0.0) == x bit for bit, also for x = +-0
  // The larger of two (finite) scores as ONE instruction (v_max_f64) instead of compare + two selects.  Same value as the
  // reference's "replace when strictly greater" (pw.c:92-103); only the sign of a zero may differ (documented exception).
  PW_FN static double max2(double a, double b) { return __builtin_fmax(a, b); }
  // A gap offer out of a cell whose own score + ge is `A`: A + go unless the same gap op is kept in the cell
  // (_pw_internals.c:268-278), as ONE fused multiply-add on a 0.0 / 1.0 factor whose high dword is the only thing selected:
  // fma(1, go, A) = round(go + A), the reference's (H + ge) + go; fma(0, go, A) = A.  Exact, no double rounding: the
  // product is 0 or go itself.  (Compare + add + two 32-bit selects otherwise.)
  union D2I_ { double d; int32_t i[2]; };
  PW_FN static double gap_offer(bool kept, double go, double A) {
    D2I_ f; f.i[0] = 0; f.i[1] = kept ? 0 : 0x3ff00000;
    return __builtin_fma(f.d, go, A);
  }
};

// ---- cross-lane moves for any score type, built on the platform's 32-bit wave shifts -----------
template <class P> PW_FN int32_t xshr1(int32_t v, int32_t old) { return P::shr1(v, old); }
template <class P> PW_FN int32_t xshl1(int32_t v, int32_t old) { return P::shl1(v, old); }
template <class P> PW_FN uint32_t xshr1(uint32_t v, uint32_t old) { return (uint32_t)P::shr1((int32_t)v, (int32_t)old); }
template <class P> PW_FN uint32_t xshl1(uint32_t v, uint32_t old) { return (uint32_t)P::shl1((int32_t)v, (int32_t)old); }
union D2I { double d; int32_t i[2]; };
template <class P> PW_FN double xshr1(double v, double old) {
  D2I a, o, r; a.d = v; o.d = old;
  r.i[0] = P::shr1(a.i[0], o.i[0]); r.i[1] = P::shr1(a.i[1], o.i[1]);
  return r.d;
}
template <class P> PW_FN double xshl1(double v, double old) {
  D2I a, o, r; a.d = v; o.d = old;
  r.i[0] = P::shl1(a.i[0], o.i[0]); r.i[1] = P::shl1(a.i[1], o.i[1]);
  return r.d;
}
// A score and a 32-bit word moved by one lane in ONE exchange (policies whose shifts cross wavefronts through
// LDS pay one barrier for the group instead of two per value; `phase` alternates their LDS slots).
template <class P, int N> PW_FN void xshrv(int32_t (&v)[N], const int32_t (&old)[N], int phase) {
  if constexpr (P::kBatchedShifts) P::template shrv<N>(v, old, phase);
  else {
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = P::shr1(v[i], old[i]);
  }
}
template <class P, int N> PW_FN void xshlv(int32_t (&v)[N], const int32_t (&old)[N], int phase) {
  if constexpr (P::kBatchedShifts) P::template shlv<N>(v, old, phase);
  else {
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = P::shl1(v[i], old[i]);
  }
}
template <class P, bool LEFT> PW_FN void xshift_pair(int32_t& s, int32_t sold, uint32_t& w, uint32_t wold, int phase) {
  int32_t v[2] = {s, (int32_t)w}; const int32_t o[2] = {sold, (int32_t)wold};
  if (LEFT) xshlv<P, 2>(v, o, phase); else xshrv<P, 2>(v, o, phase);
  s = v[0]; w = (uint32_t)v[1];
}
template <class P, bool LEFT> PW_FN void xshift_pair(double& s, double sold, uint32_t& w, uint32_t wold, int phase) {
  D2I a, b; a.d = s; b.d = sold;
  if constexpr (!P::kBatchedShifts) {
    // Every caller passes "no such predecessor" as the edge lane's value, and such a value only has to be hugely negative:
    // its HIGH dword decides that.  So the low dword moves with a zero-filling shift (no copy of a constant in front of the
    // DPP move), the high dword with the constant.
    a.i[0] = LEFT ? P::shl1z(a.i[0]) : P::shr1z(a.i[0]);
    a.i[1] = LEFT ? P::shl1(a.i[1], b.i[1]) : P::shr1(a.i[1], b.i[1]);
    w = (uint32_t)(LEFT ? P::shl1((int32_t)w, (int32_t)wold) : P::shr1((int32_t)w, (int32_t)wold));
    s = a.d;
    return;
  }
  int32_t v[3] = {a.i[0], a.i[1], (int32_t)w}; const int32_t o[3] = {b.i[0], b.i[1], (int32_t)wold};
  if (LEFT) xshlv<P, 3>(v, o, phase); else xshrv<P, 3>(v, o, phase);
  a.i[0] = v[0]; a.i[1] = v[1]; s = a.d; w = (uint32_t)v[2];
}
template <class P> PW_FN int32_t xshfl_xor(int32_t v, int m) { return P::shfl_xor(v, m); }
template <class P> PW_FN double xshfl_xor(double v, int m) {
  D2I a, r; a.d = v;
  r.i[0] = P::shfl_xor(a.i[0], m); r.i[1] = P::shfl_xor(a.i[1], m);
  return r.d;
}
template <class P> PW_FN uint64_t xshfl_xor(uint64_t v, int m) {
  uint32_t lo = (uint32_t)P::shfl_xor((int32_t)(uint32_t)v, m);
  uint32_t hi = (uint32_t)P::shfl_xor((int32_t)(uint32_t)(v >> 32), m);
  return ((uint64_t)hi << 32) | lo;
}

template <class P> PW_FN int32_t xwave_bcast(int32_t v, int w) { return P::wave_bcast(v, w); }
template <class P> PW_FN double xwave_bcast(double v, int w) {
  D2I a, r; a.d = v;
  r.i[0] = P::wave_bcast(a.i[0], w); r.i[1] = P::wave_bcast(a.i[1], w);
  return r.d;
}
template <class P> PW_FN uint64_t xwave_bcast(uint64_t v, int w) {
  const uint32_t lo = (uint32_t)P::wave_bcast((int32_t)(uint32_t)v, w);
  const uint32_t hi = (uint32_t)P::wave_bcast((int32_t)(uint32_t)(v >> 32), w);
  return ((uint64_t)hi << 32) | lo;
}

PW_FN int32_t pw_clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

// =================================================================================================
// K1: fill one pair.  Template switches:
//   BK       diagonals per lane (even); the wave covers 64*BK diagonals
//   BANY     an alignment may begin anywhere (LOCAL, END_ANCHORED, B_LOCAL)         [fast kernels]
//   TRACK    keep the first best cell of every diagonal (needed by the *LOCAL end rules)
//   GENERIC  everything at run time: substitution matrix lookup, begin rule, sign of the gap-open
//            score, optional dump of the score plane.  (BANY/TRACK are ignored: TRACK is on.)
// =================================================================================================
template <class P, typename T, int BK, bool BANY, bool TRACK_, bool GENERIC>
struct WaveFill {
  static constexpr int R = BK / 2;
  static constexpr bool TRACK = TRACK_ || GENERIC;
  using Tr = ScoreTraits<T>;
  // TAB: the substitution score is READ from a table (LDS on the device) instead of selected by a letter comparison: one
  // add for the address and a read that does not occupy the vector ALU, against compare + select (+ a second select for a
  // 64-bit value).  It also makes ANY substitution matrix -- BLOSUM-style integer matrices over 20 letters, full log-odds
  // matrices -- exactly as cheap as match / mismatch scoring on these kernels (round 2 sent every matrix to the generic
  // kernel).  The letter windows hold byte offsets: the origin letter times the row size, the mutant letter times the
  // element size.  (Round 3; `false` restores the comparison for integer scores.)
  static constexpr bool TAB = true;

  // ---- uniform (per pair) ----
  const FillParams<T>& a;
  const PairDesc& pd;
  const T* sub_tab;           // substitution table (LDS on the device) for GENERIC
  const uint8_t* oseq;
  const uint8_t* mseq;
  int X, Y, ndiag, olast, mlast;

  // ---- per lane ----
  int lane;
  int njl;                    // slot j is inside the band iff j < njl
  int xbase, ybase;           // (x, y) of slot 0 on the even step of the current iteration
  T H[BK], U[BK], L[BK], blkL[BK], best[BK];
  int32_t bestT[BK];
  uint32_t m[BK];
  uint32_t ow[R], mw[R];

  PW_FN WaveFill(const FillParams<T>& a_, const PairDesc& pd_, const T* sub_tab_)
      : a(a_), pd(pd_), sub_tab(sub_tab_) {}

  // (the table sits in LDS -- always for the fast f64 kernels, whose planner admits alphabets of up to kMaxLdsL letters
  //  only; the generic kernels also take larger alphabets, from global memory, and ask at run time)
  uint32_t tab_handle = 0;      // the platform's handle of the LDS copy (P::tab_read)
  bool tab_in_lds = false;
  PW_FN T subst(uint32_t oc, uint32_t mc) const {
    if (TAB) {      // (pre-scaled letters: see TAB)
      if (!GENERIC) return P::template tab_read<T, true>(sub_tab, tab_handle, oc + mc, true);
      return P::template tab_read<T, false>(sub_tab, tab_handle, oc + mc, tab_in_lds);
    }
    return oc == mc ? a.match : a.mismatch;
  }
  PW_FN uint32_t osc() const { return TAB ? (uint32_t)a.L * (uint32_t)sizeof(T) : 1u; }
  PW_FN uint32_t msc() const { return TAB ? (uint32_t)sizeof(T) : 1u; }

  // One cell of slot J on step t (lane-local coordinates x, y).
  template <bool RAMP, int J>
  PW_FN void cell(T up, T left, T sub, int x, int y, int t) {
    const T hD = up, hI = left;
    const T hM = H[J] + sub;
    bool active = true;
    bool ball;
    if (RAMP) {
      active = (J < njl) && ((uint32_t)x <= (uint32_t)X) && ((uint32_t)y <= (uint32_t)Y);
      const int brule = GENERIC ? a.brule : (BANY ? (int)BRULE_ANY : a.brule);
      const bool edge = (x == 0) || (y == 0);
      const bool orig = (x == 0) && (y == 0);
      ball = (brule == BRULE_ANY) || (edge && (brule == BRULE_EDGE || orig));
    } else {
      ball = GENERIC ? (a.brule == BRULE_ANY) : BANY;   // no first cells in the steady phase
    }
    // maximum in the reference's candidate order B, D, I, M, later candidates replace only when
    // strictly greater (pw.c:92-103)
    T Hn;
    if (Tr::is_int) {
      Hn = hD > hI ? hD : hI;
      Hn = hM > Hn ? hM : Hn;
      const T b0 = ball ? T(0) : Tr::neg();
      Hn = b0 > Hn ? b0 : Hn;
    } else {
      // (an alignment that may not begin here: "no such predecessor" as the begin candidate, as the integer path does)
      const T b0 = ball ? T(0) : Tr::neg();
      Hn = Tr::max2(Tr::max2(Tr::max2(b0, hD), hI), hM);
    }
    const bool bB = ball && (Hn == T(0));
    const bool bD = (hD == Hn), bI = (hI == Hn), bM = (hM == Hn);
    // what this cell offers downwards (as a D predecessor) and rightwards (as an I predecessor):
    // _alnchoice_ID scans the kept choices, (H + ge) [+ go when the kept op differs], first strict max
    // (_pw_internals.c:268-278).  All kept choices share H, so only "is the same op kept" matters.
    const T A = Hn + a.ge;
    T Un, Ln;
    if (!GENERIC) {                 // go <= 0 guaranteed by the planner
      if (Tr::is_int) {
        const T gego = a.ge + a.go;
        Un = Hn + (bD ? a.ge : gego);
        Ln = Hn + (bI ? a.ge : gego) + blkL[J];
      } else {
        Un = Tr::gap_offer(bD, a.go, A);
        Ln = Tr::gap_offer(bI, a.go, A) + blkL[J];
      }
    } else {
      const T Bv = A + a.go;
      const T hi = A > Bv ? A : Bv;
      const bool oD = bB || bI || bM, oI = bB || bD || bM;   // some other op kept as well
      Un = bD ? (oD ? hi : A) : Bv;
      Ln = (bI ? (oI ? hi : A) : Bv) + blkL[J];
    }
    // the tie nibble (M, I, D, B from bit 3 down: MM, MI, MD, MB) is shifted into the slot's mask word one flag at a time:
    // on the device each P::shl1_in is ONE add-with-carry whose carry-in is the comparison's lane mask (m + m + flag), four
    // instructions per cell instead of four selects, the ORs and the shift
    if (RAMP) {
      m[J] = P::shl1_in(P::shl1_in(P::shl1_in(P::shl1_in(m[J], bM && active), bI && active), bD && active), bB && active);
      H[J] = active ? Hn : H[J];
      U[J] = active ? Un : U[J];
      L[J] = active ? Ln : L[J];
      if (TRACK) {
        const bool upd = active && (Hn > best[J]);
        best[J] = upd ? Hn : best[J];
        bestT[J] = upd ? t : bestT[J];
      }
    } else {
      m[J] = P::shl1_in(P::shl1_in(P::shl1_in(P::shl1_in(m[J], bM), bI), bD), bB);
      H[J] = Hn; U[J] = Un; L[J] = Ln;
      if (TRACK) {
        const bool upd = Hn > best[J];
        if (Tr::is_int) best[J] = upd ? Hn : best[J];
        else best[J] = Tr::max2(best[J], Hn);           // (one v_max_f64 instead of two selects)
        bestT[J] = upd ? t : bestT[J];
      }
    }
    if (GENERIC) {
      if (a.hdump != nullptr && active && (RAMP || J < njl)) {
        const int aa = x < y ? x : y;
        a.hdump[pd.h_off + (uint64_t)(lane * BK + J) * (uint64_t)pd.h_pitch + (uint64_t)aa] = Hn;
      }
    }
  }

  // compile-time loops over the slots of one parity
  // (AHEAD: the substitution scores of the step were read from the table a step earlier -- sE / sO, see iteration())
  template <bool RAMP, int I> struct EvenLoop {
    PW_FN static void run(WaveFill& w, T uin, int t) {
      w.template cell<RAMP, 2 * I>(I == 0 ? uin : w.U[(2 * I - 1 + BK) % BK], w.L[2 * I + 1],
                                   AHEAD ? w.sE[I] : w.subst(w.ow[I], w.mw[I]), w.xbase + I, w.ybase - I, t);
      EvenLoop<RAMP, I + 1>::run(w, uin, t);
    }
  };
  template <bool RAMP> struct EvenLoop<RAMP, R> { PW_FN static void run(WaveFill&, T, int) {} };
  template <bool RAMP, int I> struct OddLoop {
    PW_FN static void run(WaveFill& w, T lin, int t) {
      w.template cell<RAMP, 2 * I + 1>(w.U[2 * I], I == R - 1 ? lin : w.L[(2 * I + 2) % BK],
                                       AHEAD ? w.sO[I] : w.subst(w.ow[I], w.mw[I]), w.xbase + I + 1, w.ybase - I, t);
      OddLoop<RAMP, I + 1>::run(w, lin, t);
    }
  };
  template <bool RAMP> struct OddLoop<RAMP, R> { PW_FN static void run(WaveFill&, T, int) {} };

  // AHEAD (single-wavefront platforms): the letters do not depend on the arithmetic, so their two window moves run ahead of
  // it and every table read is issued a whole STEP before the cell that needs it -- the odd step's at the start of the even
  // step, the next even step's at the start of the odd step -- instead of a few instructions before (a table read takes
  // ~100 cycles; with two wavefronts per SIMD nothing else covers it).  Platforms whose shifts cross wavefronts through LDS
  // move a letter and a score in ONE exchange and keep the letters in step with the arithmetic.
  static constexpr bool AHEAD = TAB && !P::kBatchedShifts;
  T sE[AHEAD ? R : 1], sO[AHEAD ? R : 1];
  PW_FN static T shift_score_l(T v) { T r = v; uint32_t none = 0; xshift_pair<P, true>(r, Tr::neg(), none, 0u, 0); return r; }
  PW_FN static T shift_score_r(T v) { T r = v; uint32_t none = 0; xshift_pair<P, false>(r, Tr::neg(), none, 0u, 0); return r; }

  // One iteration = the even step 2*it and the odd step 2*it + 1.
  template <bool RAMP>
  PW_FN void iteration(int it, int k) {
    // even step: slot 0 takes its "up" offer from the previous lane's last slot (moved at the end of the
    // previous iteration, together with the mutant window)
    T uin = uin_next;
    if (P::kVirtualLanes) uin = lane == 0 ? Tr::neg() : uin;    // nothing lies below diagonal 0 of the band
    if constexpr (AHEAD) {
      // origin window as of the odd step (lane l takes lane l+1's lowest letter, the last lane is fed from the arena)
      {
        const uint32_t oin = (uint32_t)P::shl1((int32_t)ow[0], (int32_t)(feed_byte(fo_lo, fo_hi, k) * osc()));
#pragma unroll
        for (int i = 0; i + 1 < R; i++) ow[i] = ow[i + 1];
        ow[R - 1] = oin;
      }
#pragma unroll
      for (int i = 0; i < R; i++) sO[i] = subst(ow[i], mw[i]);           // the odd step's scores: a step ahead
      EvenLoop<RAMP, 0>::run(*this, uin, 2 * it);
      const T lin = shift_score_l(L[0]);
      // mutant window as of the next iteration (lane l takes lane l-1's highest letter, lane 0 is fed from the arena)
      {
        const uint32_t min_ = (uint32_t)P::shr1((int32_t)mw[R - 1], (int32_t)(feed_byte(fm_lo, fm_hi, k) * msc()));
#pragma unroll
        for (int i = R - 1; i > 0; i--) mw[i] = mw[i - 1];
        mw[0] = min_;
      }
#pragma unroll
      for (int i = 0; i < R; i++) sE[i] = subst(ow[i], mw[i]);           // the next even step's scores
      OddLoop<RAMP, 0>::run(*this, lin, 2 * it + 1);
      uin_next = shift_score_r(U[BK - 1]);
      xbase++; ybase++;
      return;
    }
    EvenLoop<RAMP, 0>::run(*this, uin, 2 * it);
    // one exchange to the left: the origin window moves on by one letter (lane l takes lane l+1's lowest
    // letter, the last lane is fed from the arena) and the last slot gets its "left" offer for the odd step
    T lin = L[0];
    {
      uint32_t oin = ow[0];
      xshift_pair<P, true>(lin, Tr::neg(), oin, feed_byte(fo_lo, fo_hi, k) * osc(), it & 1);
#pragma unroll
      for (int i = 0; i + 1 < R; i++) ow[i] = ow[i + 1];
      ow[R - 1] = oin;
    }
    OddLoop<RAMP, 0>::run(*this, lin, 2 * it + 1);
    // one exchange to the right: the mutant window moves on (lane l takes lane l-1's highest letter, lane 0
    // is fed from the arena) and the next even step's "up" offer travels with it
    {
      uint32_t min_ = mw[R - 1];
      uin_next = U[BK - 1];
      xshift_pair<P, false>(uin_next, Tr::neg(), min_, feed_byte(fm_lo, fm_hi, k) * msc(), it & 1);
#pragma unroll
      for (int i = R - 1; i > 0; i--) mw[i] = mw[i - 1];
      mw[0] = min_;
    }
    xbase++; ybase++;
  }

  // ---- edge-lane feeders (wave-uniform) ----------------------------------------------------------
  // The last lane needs origin letter o[xfeed_o + it] and lane 0 mutant letter m[yfeed_m + it] on
  // iteration `it`.  The 8 letters a block consumes per sequence are fetched as three aligned dwords ONE
  // BLOCK AHEAD (feed_issue) and funnel-shifted into two registers when the block starts (feed_commit),
  // so no iteration ever waits on memory.  Word indices are clamped: letters outside the sequence feed
  // only cells outside the table.
  T uin_next;                 // slot 0's "up" offer for the next even step
  int xfeed_o, yfeed_m;       // arena index the edge lanes are fed from at iteration 0
  int owlast, mwlast;         // last dword of each frame
  uint32_t fo_lo, fo_hi, fm_lo, fm_hi;                 // the current block's 8 + 8 letters
  uint32_t fo_n0, fo_n1, fo_n2, fm_n0, fm_n1, fm_n2;   // next block's raw dwords

  PW_FN static uint32_t feed_byte(uint32_t lo, uint32_t hi, int k) {
    return ((k < 4 ? lo : hi) >> (8 * (k & 3))) & 0xffu;
  }
  PW_FN static uint32_t funnel(uint32_t hi, uint32_t lo, int r) {
    return (uint32_t)(((((uint64_t)hi) << 32) | (uint64_t)lo) >> (8 * r));
  }
  PW_FN void feed_issue(int b) {
    // (wave-uniform indices into the read-only letter arena: scalar loads, which leave vmcnt to the mask stores)
    const int wo = (xfeed_o + 8 * b) >> 2, wm = (yfeed_m + 8 * b) >> 2;
    fo_n0 = P::const_dword(oseq, pw_clampi(wo, 0, owlast)); fo_n1 = P::const_dword(oseq, pw_clampi(wo + 1, 0, owlast));
    fo_n2 = P::const_dword(oseq, pw_clampi(wo + 2, 0, owlast));
    fm_n0 = P::const_dword(mseq, pw_clampi(wm, 0, mwlast)); fm_n1 = P::const_dword(mseq, pw_clampi(wm + 1, 0, mwlast));
    fm_n2 = P::const_dword(mseq, pw_clampi(wm + 2, 0, mwlast));
  }
  PW_FN void feed_commit(int b) {
    const int ro = (xfeed_o + 8 * b) & 3, rm = (yfeed_m + 8 * b) & 3;
    fo_lo = funnel(fo_n1, fo_n0, ro); fo_hi = funnel(fo_n2, fo_n1, ro);
    fm_lo = funnel(fm_n1, fm_n0, rm); fm_hi = funnel(fm_n2, fm_n1, rm);
  }

  // A block = 16 steps = 8 iterations = one mask dword per slot.  The steady body is fully unrolled (the
  // letter-window shifts become register renames); the predicated ramp body runs only at the two ends
  // of a pair and is kept rolled to bound code size and scalar-register pressure.
  template <bool RAMP>
  PW_FN void block(int b) {
    if (RAMP) {
#pragma unroll 1
      for (int k = 0; k < 8; k++) iteration<true>(8 * b + k, k);
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) iteration<false>(8 * b + k, k);
    }
  }

  PW_FN void store_masks(int b) {
    if (lane >= 0 && lane * BK < ndiag && P::central()) {
      uint32_t* dst = a.masks + pd.mask_off;
#pragma unroll
      for (int j = 0; j < BK; j++) dst[mask_word_index(BK, pd.nl, b, lane, j)] = m[j];
    }
  }

  PW_FN void init(int it0 = 0) {
    lane = P::lane();
    X = pd.X; Y = pd.Y; ndiag = pd.ndiag;
    oseq = a.arena + pd.o_off; mseq = a.arena + pd.m_off;
    olast = X > 0 ? X - 1 : 0; mlast = Y > 0 ? Y - 1 : 0;
    owlast = olast >> 2; mwlast = mlast >> 2;
    njl = lane < 0 ? 0 : ndiag - lane * BK;      // (tiles may carry virtual lanes below diagonal 0)
    uin_next = Tr::neg();
    // s0 == dmin (mod 2): e, f are exact
    const int e = (pd.s0 + pd.dmin) >> 1;       // x of diagonal dd = 0 on step t = 0
    const int f = (pd.s0 - pd.dmin) >> 1;       // y of diagonal dd = 0 on step t = 0
    xbase = e + it0 + lane * R;
    ybase = f + it0 - lane * R;
    xfeed_o = e + P::nlanes() * R - 1;          // letter o[xbase + R - 1] of a virtual lane beyond the last one
    yfeed_m = f - P::lane0() * R;               // letter m[ybase] of the first lane
#pragma unroll
    for (int j = 0; j < BK; j++) {
      H[j] = Tr::neg(); U[j] = Tr::neg(); L[j] = Tr::neg();
      best[j] = Tr::neg(); bestT[j] = 0; m[j] = 0;
      // the first diagonal above the band must never offer an insertion into the band
      blkL[j] = (lane * BK + j == ndiag) ? Tr::neg() : Tr::zero_blk();
    }
#pragma unroll
    for (int i = 0; i < R; i++) {
      ow[i] = (uint32_t)oseq[pw_clampi(xbase + i - 1, 0, olast)] * osc();
      mw[i] = (uint32_t)mseq[pw_clampi(ybase - i - 1, 0, mlast)] * msc();
    }
    if (AHEAD) {
#pragma unroll
      for (int i = 0; i < (AHEAD ? R : 1); i++) sE[i] = subst(ow[i], mw[i]);
    }
  }

  PW_FN void run() {
    init();
    feed_issue(0);
    for (int b = 0; b < pd.nblocks; b++) {
      feed_commit(b);
      if (b + 1 < pd.nblocks) feed_issue(b + 1);
      if (b >= pd.steady_b0 && b < pd.steady_b1) block<false>(b);
      else block<true>(b);
      store_masks(b);
    }
    finish();
  }

  // ---- K2b: one tile of a time-blocked, ghost-zone tiled fill (tables too wide for one workgroup) -------
  // The workgroup owns P::nlanes() - P::lane0() lanes of the band; the outer `ghost` lanes on each side are
  // recomputed copies of the neighbouring tiles' lanes.  Dependencies travel one diagonal per step, so after
  // tile_nb * 16 <= ghost * BK steps the centre lanes (P::central()) are still exact; only they store masks
  // and hand their state to the next time block.  State: [5][st_pitch] = H, U, L, best, bestT per diagonal.
  PW_FN void run_tile() {
    const int bb0 = a.tile_b0, bb1 = a.tile_b0 + a.tile_nb, pitch = a.st_pitch;
    lane = P::lane();
    ndiag = pd.ndiag;
    // steps on which this tile's in-band diagonals (ghost lanes included) hold their first / last cell
    int w4[4] = {-0x7fffffff, -0x7fffffff, -0x7fffffff, -0x7fffffff};   // max of: -tf, tf, -tl, tl
#pragma unroll
    for (int j = 0; j < BK; j++) {
      const int dd = lane * BK + j;
      if (lane >= 0 && dd < ndiag) {
        const int d = pd.dmin + dd;
        const int tf = (d < 0 ? -d : d) - pd.s0;
        const int tl = tf + 2 * ((d > 0 ? 0 : d) + (pd.X - d > pd.Y ? pd.Y : pd.X - d));
        w4[0] = -tf > w4[0] ? -tf : w4[0]; w4[1] = tf > w4[1] ? tf : w4[1];
        w4[2] = -tl > w4[2] ? -tl : w4[2]; w4[3] = tl > w4[3] ? tl : w4[3];
      }
    }
    P::wg_max4(w4);
    const int tfmin = -w4[0], tfmax = w4[1], tlmin = -w4[2], tlmax = w4[3];
    // the blocks of this time block in which any of them is live
    int bA = tfmin >> 4, bB = tlmax >> 4;
    if (w4[1] == -0x7fffffff) { bA = 1; bB = 0; }          // no diagonal of the band in this tile
    bA = bA < bb0 ? bb0 : bA; bB = bB > bb1 - 1 ? bb1 - 1 : bB;
    if (bA > bB) {
      // not started yet or already finished: the state passes through unchanged
      if (P::central()) {
#pragma unroll
        for (int j = 0; j < BK; j++) {
          const int dd = lane * BK + j;
          if (lane >= 0 && dd < ndiag) {
#pragma unroll
            for (int r = 0; r < 5; r++)
              a.st_out[r * pitch + dd] = bb0 > 0 ? a.st_in[r * pitch + dd] : (r == 4 ? T(0) : Tr::neg());
          }
        }
      }
      return;
    }
    init(8 * bA);
    if (bb0 > 0) {
#pragma unroll
      for (int j = 0; j < BK; j++) {
        const int dd = lane * BK + j;
        if (lane >= 0 && dd < ndiag) {
          H[j] = a.st_in[dd]; U[j] = a.st_in[pitch + dd]; L[j] = a.st_in[2 * pitch + dd];
          best[j] = a.st_in[3 * pitch + dd]; bestT[j] = (int32_t)a.st_in[4 * pitch + dd];
        }
      }
      uint32_t none = 0;
      uin_next = U[BK - 1];
      xshift_pair<P, false>(uin_next, Tr::neg(), none, 0u, 1);
    }
    feed_issue(bA);
    for (int b = bA; b <= bB; b++) {
      feed_commit(b);
      if (b < bB) feed_issue(b + 1);
      // steady: every diagonal of the tile started strictly before the block and none ends inside it
      if (16 * b > tfmax && 16 * b + 15 <= tlmin) block<false>(b);
      else block<true>(b);
      store_masks(b);
    }
    if (P::central()) {
#pragma unroll
      for (int j = 0; j < BK; j++) {
        const int dd = lane * BK + j;
        if (lane >= 0 && dd < ndiag) {
          a.st_out[dd] = H[j]; a.st_out[pitch + dd] = U[j]; a.st_out[2 * pitch + dd] = L[j];
          a.st_out[3 * pitch + dd] = best[j]; a.st_out[4 * pitch + dd] = (T)bestT[j];
        }
      }
    }
  }

  // ---- end-cell search: reduce (score desc, scan rank asc) over the in-band diagonals ----------
  PW_FN void finish() {
    const int endrule = a.endrule;
    T cs = Tr::neg(); uint64_t ck = ~(uint64_t)0; int cx = -1, cy = -1; bool have = false;
#pragma unroll
    for (int j = 0; j < BK; j++) {
      const int dd = lane * BK + j;
      const int d = pd.dmin + dd;
      const bool inband = dd < ndiag;
      // last cell of diagonal d
      const bool ends_right = d < X - Y;           // ends on the right column y = Y, else bottom row x = X
      const int lx = ends_right ? d + Y : X, ly = ends_right ? Y : X - d;
      T s; uint64_t k; int x, y; bool ok = inband;
      if (endrule == END_CORNER) {
        ok = ok && (d == X - Y); s = H[j]; k = 0; x = X; y = Y;
      } else if (endrule == END_STD_OVERLAP) {
        // row-major over last column (x < X) then last row: rank x for (x, Y), X + y for (X, y)
        s = H[j]; x = lx; y = ly; k = ends_right ? (uint64_t)(uint32_t)lx : (uint64_t)(uint32_t)(X + ly);
      } else if (endrule == END_BANDED_OVERLAP) {
        s = H[j]; x = lx; y = ly; k = (uint64_t)(uint32_t)dd;
      } else {
        // first best cell of the diagonal: step bestT -> index along the diagonal
        const int tfirst = (d < 0 ? -d : d) - pd.s0;
        const int aa = (bestT[j] - tfirst) >> 1;
        s = best[j]; x = aa + (d > 0 ? d : 0); y = aa - (d < 0 ? d : 0);
        if (endrule == END_STD_LOCAL) k = (uint64_t)(uint32_t)x * (uint64_t)(uint32_t)(Y + 1) + (uint64_t)(uint32_t)y;
        else k = ((uint64_t)(uint32_t)dd << 32) | (uint64_t)(uint32_t)aa;
      }
      const bool better = ok && (!have || s > cs || (s == cs && k < ck));
      if (better) { cs = s; ck = k; cx = x; cy = y; have = true; }
    }
    // butterfly over the 64 lanes
    int hv = have ? 1 : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const T os = xshfl_xor<P>(cs, off);
      const uint64_t ok_ = xshfl_xor<P>(ck, off);
      const int ox = P::shfl_xor(cx, off), oy = P::shfl_xor(cy, off), oh = P::shfl_xor(hv, off);
      const bool take = oh && (!hv || os > cs || (os == cs && ok_ < ck));
      if (take) { cs = os; ck = ok_; cx = ox; cy = oy; hv = 1; }
    }
    // several wavefronts per pair (wide bands): reduce the per-wave winners the same way
    if (P::nwaves() > 1) {
      T bs = cs; uint64_t bk_ = ck; int bx = cx, by = cy, bh = 0;
      for (int wv = 0; wv < P::nwaves(); wv++) {
        const T os = xwave_bcast<P>(cs, wv);
        const uint64_t ok_ = xwave_bcast<P>(ck, wv);
        const int ox = P::wave_bcast(cx, wv), oy = P::wave_bcast(cy, wv), oh = P::wave_bcast(hv, wv);
        const bool take = oh && (!bh || os > bs || (os == bs && ok_ < bk_));
        if (take) { bs = os; bk_ = ok_; bx = ox; by = oy; bh = 1; }
      }
      cs = bs; ck = bk_; cx = bx; cy = by; hv = bh;
    }
    if (lane == 0) {
      Result r;
      r.score = (double)cs * a.score_mul;
      // table coordinates as dptable_solve returns them (_cellpos_from_xy, _pw_internals.c:87-98)
      r.opt_i = a.banded ? cx - cy - pd.dmin : cx;
      r.opt_j = a.banded ? (cx < cy ? cx : cy) : cy;
      r.origin_idx = 0; r.mutant_idx = 0; r.tx_len = 0; r.status = 0;
      // LOCAL / START_ANCHORED start from the score of cell (0,0), i.e. 0 (_pw_internals.c:342)
      if (!hv || (endrule == END_STD_LOCAL && !(cs > T(0)))) { r.opt_i = -1; r.opt_j = -1; r.score = 0.0; }
      a.results[pair_slot] = r;
    }
  }
  int pair_slot;
};

// =================================================================================================
// K1, packed 16-bit steady phase.
//
// Measured on gfx950 (profiles/round1_valu_issue_table.txt): only v_add/v_sub/v_xor/v_mov issue at the
// 2-cycle wave64 rate; v_max/v_min/v_cmp/v_cndmask, shifts, every VOP3 op, DPP and the packed VOP3P ops
// all take ~4 cycles.  The 32-bit cell update above is almost entirely 4-cycle ops, one cell each.  The
// packed form below does the same recurrence on TWO cells per 4-cycle v_pk_* op, with the tie bits
// obtained arithmetically (n = min(H - candidate, 1) is 0 iff the candidate is kept) instead of through
// compare -> select, so a steady-phase cell costs roughly half the issue cycles.
//
// Register layout: a lane's BK diagonals are split into a low and a high half-block of BK/2; slot j of the
// low half-block and slot j + BK/2 of the high one share a register (lo / hi 16 bits).  With that pairing
// the up / left neighbours of a register pair are again whole registers of the other parity; only the
// two block edges need one DPP wave shift + one v_alignbit per step.
//
// Blocks in which diagonals start or end run a slightly longer packed body (cellpair<EDGE = true>) that
// needs no predication at all -- see there.  Eligibility (host planner): LOCAL and B_LOCAL (begin anywhere,
// end anywhere: the end-cell search uses the tracked bests only), match/mismatch scoring, go <= 0, every
// score within +-100, min(X,Y) * max(match, mismatch, 0) <= 8000 (below the 8192-deep sentinel that blocks the first diagonal
// above the band), X + Y + 2 < 32000 (steps as signed 16-bit).
// =================================================================================================
// 16-byte group of the mask plane; an 8-byte store at any byte address
struct __attribute__((aligned(8))) U4 { uint32_t x, y, z, w; };
struct __attribute__((packed)) PackedU64 { uint64_t v; };   // an 8-byte store at any byte address
struct __attribute__((packed)) PackedU32 { uint32_t v; };   // a 4-byte load at any byte address

namespace pk {
#if defined(__HIP_DEVICE_COMPILE__)
// clang vector types: the backend selects v_pk_add_u16 / v_pk_sub_i16 / v_pk_max_i16 / v_pk_min_u16 /
// v_pk_mad_u16 itself and knows their latencies (inline asm would be padded with s_nop)
typedef short s2_t __attribute__((ext_vector_type(2)));
typedef unsigned short u2_t __attribute__((ext_vector_type(2)));
PW_FN s2_t as_s2(uint32_t v) { return __builtin_bit_cast(s2_t, v); }
PW_FN u2_t as_u2(uint32_t v) { return __builtin_bit_cast(u2_t, v); }
PW_FN uint32_t add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (u2_t)(as_u2(a) + as_u2(b))); }
PW_FN uint32_t sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (u2_t)(as_u2(a) - as_u2(b))); }
// unsigned saturating subtract per half: max(a - b, 0) (v_pk_sub_u16 ... clamp)
PW_FN uint32_t subsat(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(as_u2(a), as_u2(b))); }
PW_FN uint32_t max(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(as_s2(a), as_s2(b))); }
// NB min(x, 1) and (x >> 15) & c with LITERAL constants would be rewritten by the optimizer into compare +
// select per half (the very pattern this kernel avoids), and inline asm gets padded with s_nop by the
// hazard recogniser.  Callers therefore pass the constants in registers made opaque once (pk::opaque).
PW_FN uint32_t minu(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(as_u2(a), as_u2(b))); }
PW_FN uint32_t maxu(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(as_u2(a), as_u2(b))); }
// logical shift right per half by sh = (n, n) held in an opaque register
PW_FN uint32_t shru(uint32_t a, uint32_t sh) { return __builtin_bit_cast(uint32_t, (u2_t)(as_u2(a) >> as_u2(sh))); }
PW_FN uint32_t mad(uint32_t a, uint32_t b, uint32_t c) { return __builtin_bit_cast(uint32_t, (u2_t)(as_u2(a) * as_u2(b) + as_u2(c))); }
PW_FN uint32_t mins(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(as_s2(a), as_s2(b))); }
PW_FN uint32_t align16(uint32_t hi, uint32_t lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }
// sign mask per half (0xffff where negative): arithmetic shift by sh15 = (15, 15) held in an opaque register
PW_FN uint32_t sign(uint32_t a, uint32_t sh15) { return __builtin_bit_cast(uint32_t, (s2_t)(as_s2(a) >> as_s2(sh15))); }
PW_FN uint32_t opaque(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
// v_perm_b32: byte i of the result is byte sel[i] of the 8 bytes {a (4 .. 7), b (0 .. 3)}; selector 8 .. 11 = the sign of
// byte 1 / 3 / 5 / 7 spread over the byte, 12 = 0x00, 13 and up = 0xff
PW_FN uint32_t perm(uint32_t a, uint32_t b, uint32_t sel) { return __builtin_amdgcn_perm(a, b, sel); }
#else
PW_FN uint32_t mk(uint32_t lo, uint32_t hi) { return (lo & 0xffffu) | (hi << 16); }
PW_FN int32_t sl(uint32_t v) { return (int16_t)(v & 0xffffu); }
PW_FN int32_t sh(uint32_t v) { return (int16_t)(v >> 16); }
PW_FN uint32_t add(uint32_t a, uint32_t b) { return mk((a & 0xffffu) + (b & 0xffffu), (a >> 16) + (b >> 16)); }
PW_FN uint32_t sub(uint32_t a, uint32_t b) { return mk((a & 0xffffu) - (b & 0xffffu), (a >> 16) - (b >> 16)); }
PW_FN uint32_t subsat(uint32_t a, uint32_t b) {
  const uint32_t al = a & 0xffffu, bl = b & 0xffffu, ah = a >> 16, bh = b >> 16;
  return mk(al > bl ? al - bl : 0u, ah > bh ? ah - bh : 0u);
}
PW_FN uint32_t max(uint32_t a, uint32_t b) { return mk((uint32_t)(sl(a) > sl(b) ? sl(a) : sl(b)), (uint32_t)(sh(a) > sh(b) ? sh(a) : sh(b))); }
PW_FN uint32_t minu(uint32_t a, uint32_t b) {
  const uint32_t al = a & 0xffffu, bl = b & 0xffffu, ah = a >> 16, bh = b >> 16;
  return mk(al < bl ? al : bl, ah < bh ? ah : bh);
}
PW_FN uint32_t mad(uint32_t a, uint32_t b, uint32_t c) {
  return mk((uint32_t)(sl(a) * sl(b) + sl(c)), (uint32_t)(sh(a) * sh(b) + sh(c)));
}
PW_FN uint32_t maxu(uint32_t a, uint32_t b) {
  const uint32_t al = a & 0xffffu, bl = b & 0xffffu, ah = a >> 16, bh = b >> 16;
  return mk(al > bl ? al : bl, ah > bh ? ah : bh);
}
PW_FN uint32_t shru(uint32_t a, uint32_t sh) { return mk((a & 0xffffu) >> (sh & 0xfu), (a >> 16) >> ((sh >> 16) & 0xfu)); }
PW_FN uint32_t mins(uint32_t a, uint32_t b) { return mk((uint32_t)(sl(a) < sl(b) ? sl(a) : sl(b)), (uint32_t)(sh(a) < sh(b) ? sh(a) : sh(b))); }
PW_FN uint32_t align16(uint32_t hi, uint32_t lo) { return (lo >> 16) | (hi << 16); }
PW_FN uint32_t sign(uint32_t a, uint32_t) { return mk(sl(a) < 0 ? 0xffffu : 0u, sh(a) < 0 ? 0xffffu : 0u); }
PW_FN uint32_t opaque(uint32_t v) { return v; }
PW_FN uint32_t perm(uint32_t a, uint32_t b, uint32_t sel) {
  const uint64_t in = ((uint64_t)a << 32) | b;
  uint32_t r = 0;
  for (int i = 0; i < 4; i++) {
    const uint32_t c = (sel >> (8 * i)) & 0xffu;
    uint32_t v;
    if (c >= 13) v = 0xffu;
    else if (c == 12) v = 0u;
    else if (c >= 8) v = ((in >> (8 * (2 * (c - 8) + 1) + 7)) & 1u) ? 0xffu : 0u;
    else v = (uint32_t)(in >> (8 * c)) & 0xffu;
    r |= v << (8 * i);
  }
  return r;
}
#endif
PW_FN uint32_t both(int32_t v) { return ((uint32_t)v & 0xffffu) | ((uint32_t)v << 16); }
PW_FN uint32_t pack(int32_t lo, int32_t hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
PW_FN int32_t lo_s(uint32_t v) { return (int32_t)(int16_t)(v & 0xffffu); }
PW_FN int32_t hi_s(uint32_t v) { return (int32_t)(int16_t)(v >> 16); }
}  // namespace pk

// RULE selects the begin / end rules the kernel serves:
//   0  LOCAL / B_LOCAL            begin anywhere (B = 0 in every started cell), end = first best cell (tracked)
//   1  B_OVERLAP                  begin on the table edge (B = 0 only in the FIRST cell of a diagonal), end = the best
//                                 LAST cell of a diagonal
//   2  B_GLOBAL                   begin at (0, 0) only, end = cell (X, Y)
// Rules 1 and 2 have negative scores, so their sentinel is deeper (-24000; real scores stay within [-23000, 30000],
// see the host planner), the first diagonal above the band is silenced by clamping its offer (adding a second sentinel to a
// real score could wrap), every cell takes max(.., sentinel) -- which also pins cells before a diagonal's start and
// beyond its end -- and instead of tracking a best the value of each diagonal's last cell is captured.
//   4  END_ANCHORED (standard)   begin anywhere like rule 0 (scores >= 0, shallow sentinel), end = cell (X, Y): the last
//                                 cell of diagonal X - Y is captured, nothing is tracked
//   5  START_ANCHORED (standard) begin at (0, 0) only like rule 2 (negative scores, deep sentinel), end = the first best
//                                 cell anywhere, which must beat 0 (_pw_internals.c:342): the running best is tracked as a
//                                 key over max(H, 0), so real scores must stay within [-23000, 8000]
// CONTRACT of the plain (match / mismatch) form: letters outside a sequence never match, i.e. cells that have not started and
// cells beyond a diagonal's end take the MISMATCH score on their diagonal move -- which must be <= 0, or they creep up from
// the sentinel (the host planner admits the plain form only then; the matrix form scores such letters with the matrix minimum).
// MAT: the diagonal candidate takes its score from an integer substitution matrix of up to 4 x 4 letters instead of
// match / mismatch (_alnchoice_M, _pw_internals.c:217-245: subst_scores[o][m]).  The origin window then carries, per
// cell, the origin letter's ROW of the matrix -- four bytes subst[o][.] - min(subst) (times 4 under rule 3), at most 127
// each -- and the mutant window carries byte SELECTORS (0x0c00 | letter, + 4 in the high half of a register): one
// v_perm_b32 looks up both cells of a pair.  Letters outside a sequence are the all-zero row / a selector that reads as
// 0x00: the matrix minimum, which the planner requires to be <= 0 ("matches nothing").
template <class P, int BK, bool SEG, int RULE = 0, bool MAT = false>
struct WaveFill16 {
  // RULE 3 = rule 0 with every score held times 4 (admitted when the scores stay below 2048): a kept-or-not difference
  // of two running values is then 0 or at least 4, so min(x, 2) and min(x, 4) deliver the D and I tie bits already
  // weighted and the tie nibble is one three-operand add instead of two multiply-adds.
  static constexpr bool SC4 = RULE == 3;
  static constexpr int RL = SC4 ? 0 : RULE;
  static constexpr int SCL = SC4 ? 4 : 1;
  // what the rules are made of
  static constexpr bool ANYB = RL == 0 || RL == 4;   // an alignment may begin in every cell: scores >= 0, sentinel -8192
  static constexpr bool TRK = RL == 0 || RL == 5;    // the first best cell of every diagonal is tracked (as a key)
  static constexpr bool CAP = !TRK;                  // the value of every diagonal's last cell is captured instead
  static_assert(BK % 4 == 0, "packed layout needs an even number of cells per step");
  static constexpr int R = BK / 2;      // cells per lane and step
  static constexpr int RH = R / 2;      // packed registers per parity
#ifdef PW_FILL16_UNR
  static constexpr int UNR = PW_FILL16_UNR;                      // (A/B builds: build.py, PW_FILL16_UNR_OVERRIDE)
#else
  static constexpr int UNR = BK <= 8 ? 4 : (BK <= 16 ? 2 : 1);   // iterations unrolled per loop trip
#endif
  static constexpr int32_t NEG16 = ANYB ? -8192 : -24000;
  static constexpr uint32_t SENT_O = 0xfffeu, SENT_M = 0xffffu;   // letters outside a sequence: match nothing
  // MAT: selector codes of a mutant letter in the low / high half of a register; outside the sequence 8 / 12, which both
  // read as 0x00 (8 = the sign of a row byte, and row bytes stay below 128)
  static constexpr uint32_t MSEL = 0x0c00u, MSENT_LO = 0x0c08u, MSENT_HI = 0x0c0cu;
  using Base = WaveFill<P, int32_t, BK, true, true, false>;       // only its static feeder helpers are used

  // SEG = true, lane packing: a wavefront holds `count` pairs side by side, `nl` lanes each (WaveDesc).
  // Everything that is per pair is then per LANE (pd is a per-lane copy); the DPP wave shifts still move
  // whole-wave, so the values that cross a pair boundary are replaced on the first / last lane of every
  // pair (sentinel offers, or the pair's own sequence feeder).
  // SEG = false: one pair per wavefront (wd.nl == 64); the descriptor is wave-uniform (scalar registers and
  // scalar loads) and the DPP `old` operand alone handles the two wave edges.
  const FillParams<int32_t>& a;
  const WaveDesc wd;
  PairDesc pd;
  int li;                               // lane within the pair
  bool valid, segfirst, seglast;
  int pair_slot;
  const uint8_t* oseq;
  const uint8_t* mseq;
  int X, Y, ndiag, owlast, mwlast, xfeed_o, yfeed_m;
  uint32_t fo_lo, fo_hi, fm_lo, fm_hi, fo_n0, fo_n1, fo_n2, fm_n0, fm_n1, fm_n2;

  // packed state: index p <-> even slots (2p, 2p + R) [E*] / odd slots (2p + 1, 2p + 1 + R) [O*]
  uint32_t HE[RH], UE[RH], LE[RH], HO[RH], UO[RH], LO[RH];
  uint32_t bestE[RH], bestO[RH], btE[RH], btO[RH];     // running best and the step it was first reached
  uint32_t gebE[RH], gebO[RH];                          // ge (+ the band-top block) per half
  uint32_t clE[RH], clO[RH];                            // RULE != 0: clamp of the "left" offer (sentinel for the slot above the band)
  uint32_t tfE[RH], tfO[RH], tlE[RH], tlO[RH];          // first / last step of each diagonal
  uint32_t accE[RH], accO[RH], acc2E[RH], acc2O[RH];    // inverted tie nibbles: cells 0-3 / 4-7 of a block
  uint32_t OW[RH], MW[RH];
  uint32_t ROW[MAT ? R : 1];                            // MAT: the matrix row of every cell's origin letter
  uint32_t BIASV, MADJ;                                 // MAT: -min(subst) in both halves; the selector fix-up of the mutant window
  uint32_t MR0, MR1, MR2, MR3;                          // MAT: the four rows of the matrix
  uint32_t ONE, SH15, C2, C4, C16, NDELTA, MATCHV, GOV, GOVI, NEGV, LIMV;
  // RULE 0, steady blocks: the running best of a slot as a key 8 H + (7 - cell within the block) -- one multiply-add and
  // one unsigned maximum per cell pair instead of maximum, compare, subtract and multiply-add (H <= 8191: the planner
  // admits scores up to 8000); turned back into (best, step) once per block
  uint32_t kbE[RH], kbO[RH], C8, SH3, SEVEN, NEG2;
  uint32_t up_prev;             // the lane below's last odd slot, as of the end of the previous iteration
  int phase_l, phase_r;         // LDS slot of the next exchange to the left / right (multi-wavefront platforms)

  PW_FN WaveFill16(const FillParams<int32_t>& a_, const WaveDesc& wd_) : a(a_), wd(wd_) {}

  // Two cells at once.  `acc` collects the INVERTED tie bits (1 = candidate not kept), 4 bits per cell:
  // bit 0 B, bit 1 D, bit 2 I; bit 3 (M) stays 0 -- with go <= 0 the walker never needs it: the first
  // kept op is M exactly when none of B, D, I is kept (pw_first_op).
  //
  // EDGE = false: steady phase, every in-band diagonal holds an in-table cell.
  // EDGE = true : blocks in which diagonals start or end.  No state is frozen; instead
  //   * a diagonal that has not started yet (cells with a negative coordinate) may not begin an alignment:
  //     its B candidate is the sentinel instead of 0.  Such cells have only sentinels as predecessors and
  //     letters outside the sequences match nothing, so they stay at the sentinel and in-table cells
  //     never pick them;
  //   * a diagonal that has ended keeps computing cells beyond the table; nothing in the table reads them
  //     (predecessors have smaller coordinates), their tie nibbles land in mask slots the walker never
  //     visits, and they are kept out of the running best by lowering them by 32767 first.
  // For RULE != 0: `bests` holds the captured value of the diagonal's LAST cell, `bts` is unused, `tf` is the step of
  // the one cell that may begin an alignment (32767 = none), `geb` is ge in both halves and `clampL` silences the
  // first diagonal above the band.
  // EK: 0 steady; bit 0: some diagonal of the wavefront has not started yet in this block; bit 1: some has ended (rules
  // 1 / 2 know steady and "edge" = 3 only).
  template <int EK>
  PW_FN void cellpair(uint32_t& Hs, uint32_t& Us, uint32_t& Ls, uint32_t& bests, uint32_t& bts, uint32_t geb,
                      uint32_t tf, uint32_t tl, uint32_t& acc, uint32_t up, uint32_t left, uint32_t oc,
                      uint32_t mc, uint32_t tv, uint32_t clampL = 0, uint32_t och = 0) {
    constexpr bool EDGE = EK != 0;
    uint32_t hM;
    // (MAT, begin-anywhere rules outside the blocks where diagonals start: a started cell's score is never negative, so the
    //  bias comes off with an unsigned SATURATING subtract -- the diagonal candidate is max(H + subst, 0) already and the
    //  maximum with the begin candidate 0 below is not needed: one op less per cell pair.  Cells that have not started hold
    //  the sentinel, a large unsigned value that the subtract leaves alone.)
    constexpr bool FLOOR0 = MAT && ANYB && (EK & 1) == 0;
    if (MAT) {
      // oc / och: the matrix rows of the low / high cell's origin letter, mc: the two selectors
      const uint32_t hb = pk::add(Hs, pk::perm(och, oc, mc));
      hM = FLOOR0 ? pk::subsat(hb, BIASV) : pk::sub(hb, BIASV);
    } else {
      const uint32_t ne = pk::minu(oc ^ mc, ONE);               // 0 where the letters match
      hM = pk::add(Hs, pk::mad(ne, NDELTA, MATCHV));
    }
    uint32_t Hn = pk::max(pk::max(up, left), hM);
    uint32_t nB;
    if (ANYB) {
      if (EK & 1) Hn = pk::max(Hn, pk::sign(pk::sub(tv, tf), SH15) & NEGV);   // B = 0 once started, sentinel before
      else if (!FLOOR0) Hn = pk::max(Hn, 0u);                    // B: an alignment may begin anywhere, score 0
      nB = pk::minu(Hn, ONE);
    } else {
      // B = 0 in the one cell that may begin (step tf), the sentinel everywhere else
      const uint32_t Bc = EDGE ? pk::mad(pk::minu(tv ^ tf, ONE), NEGV, 0u) : NEGV;
      Hn = pk::max(Hn, Bc);
      nB = pk::minu(Hn ^ Bc, ONE);
    }
    // "is the candidate kept" only asks whether H == candidate: xor (a 2-cycle op) instead of a packed subtract
    const uint32_t nD = pk::minu(Hn ^ up, SC4 ? C2 : ONE);       // SC4: 0 or 2
    const uint32_t nI = pk::minu(Hn ^ left, SC4 ? C4 : ONE);     // SC4: 0 or 4
    const uint32_t hg = pk::add(Hn, geb);
    Us = pk::mad(nD, GOV, hg);                                   // (H + ge) + go unless a D choice is kept
    Ls = pk::mad(nI, GOVI, hg);
    if (!ANYB) Ls = pk::mins(Ls, clampL);
    // nibble = nB + 2 nD + 4 nI, appended to the accumulator: three packed multiply-adds
    if (SC4) acc = pk::mad(acc, C16, nB + nD + nI);             // (halves below 8: a plain 32-bit three-operand add)
    else acc = pk::mad(acc, C16, pk::mad(nI, C4, pk::mad(nD, C2, nB)));
    if (TRK) {
      // (the running best is kept as a key by iteration16)
    } else if (EK & 2) {
      const uint32_t e = pk::minu(tv ^ tl, ONE);                 // 0 in the diagonal's last cell
      bests = pk::mad(e, pk::sub(bests, Hn), Hn);                // e ? bests : Hn
    }
    Hs = Hn;
  }

  PW_FN void key_to_best(uint32_t& bests, uint32_t& bts, uint32_t kb, uint32_t kb0, uint32_t base) {
    const uint32_t ch = pk::minu(kb ^ kb0, ONE);                 // 1 where the best strictly improved in this block
    const uint32_t st = pk::mad(kb & SEVEN, NEG2, base);         // its step: base - 2 (7 - cell)
    bts = pk::mad(ch, pk::sub(st, bts), bts);
    // (unchanged where the key is; SC4: key = 2 (4 H) + cell)
    bests = SC4 ? (pk::shru(kb, ONE) & 0xfffcfffcu) : pk::shru(kb, SH3);
  }

  // HALF selects the accumulator set: iterations 0-3 of a block (cells 0-3 of every slot) or 4-7.
  // Rule 0: the slot's running best as a key (see kbE): cells of a diagonal that has not started hold the sentinel and
  // count as 0, cells beyond a diagonal's end are masked out -- neither can beat 8 best + 7.
  template <int EK>
  PW_FN void track_key(uint32_t& kb, uint32_t H, uint32_t tl, uint32_t tv, int k) {
    const uint32_t hk = (!ANYB || (EK & 1)) ? pk::max(H, 0u) : H;   // (rule 5: real scores go negative and count as 0)
    uint32_t key = pk::mad(hk, SC4 ? C2 : C8, pk::both(7 - k));
    if (EK & 2) key &= ~pk::sign(pk::sub(tl, tv), SH15);
    kb = pk::maxu(kb, key);
  }

  template <int EK, int HALF>
  PW_FN void iteration16(int it, int k) {
    constexpr bool EDGE = EK != 0;
    const uint32_t tv0 = pk::both(2 * it), tv1 = pk::both(2 * it + 1);
    // even step: slot 0 <- previous lane's last slot, slot R <- own slot R - 1
    {
      // (the lane below's last odd slot: moved at the end of the previous iteration, together with the mutant window)
      uint32_t prev = up_prev;
      if (SEG) prev = segfirst ? NEGV : prev;
      const uint32_t up0 = pk::align16(UO[RH - 1], prev);         // (prev.hi, own.lo)
#pragma unroll
      for (int p = 0; p < RH; p++)
      {
        cellpair<EK>(HE[p], UE[p], LE[p], bestE[p], btE[p], gebE[p], tfE[p], tlE[p], HALF == 0 ? accE[p] : acc2E[p],
                     p == 0 ? up0 : UO[p == 0 ? 0 : p - 1], LO[p], MAT ? ROW[MAT ? p : 0] : OW[p], MW[p], tv0, clE[p],
                     MAT ? ROW[MAT ? p + RH : 0] : 0u);
        if (TRK) track_key<EK>(kbE[p], HE[p], tlE[p], tv0, k);
      }
    }
    // origin window moves on: last register <- (own first.hi, next lane's first.lo | the pair's feeder).
    // Only cells outside the table read letters outside a sequence; in steady blocks those are out-of-band
    // slots whose values nobody reads, so the range check is needed in EDGE blocks only.
    // The letter and the left offer of the odd step travel the same way: ONE exchange (platforms whose shifts cross
    // wavefronts through LDS pay one barrier for the pair).
    uint32_t nxt_left;
    {
      const int oi = xfeed_o + it;
      const uint32_t fb = Base::feed_byte(fo_lo, fo_hi, k);
      uint32_t feed = (!EDGE || (uint32_t)oi < (uint32_t)X) ? fb : SENT_O;
      if (MAT) feed = (!EDGE || (uint32_t)oi < (uint32_t)X) ? row_of(fb) : 0u;
      int32_t mv[2] = {(int32_t)(MAT ? ROW[0] : OW[0]), (int32_t)LE[0]};
      const int32_t mo[2] = {(int32_t)feed, (int32_t)NEGV};
      xshlv<P, 2>(mv, mo, phase_l); phase_l ^= 1;
      uint32_t nxt = (uint32_t)mv[0];
      nxt_left = (uint32_t)mv[1];
      if (SEG) nxt = seglast ? feed : nxt;
      if (MAT) {
        // whole registers: the window of rows moves on by renaming
#pragma unroll
        for (int i = 0; i + 1 < (MAT ? R : 1); i++) ROW[i] = ROW[i + 1];
        ROW[MAT ? R - 1 : 0] = nxt;
      } else {
        const uint32_t last = pk::align16(nxt, OW[0]);
#pragma unroll
        for (int p = 0; p + 1 < RH; p++) OW[p] = OW[p + 1];
        OW[RH - 1] = last;
      }
    }
    // odd step: slot BK - 1 <- next lane's slot 0, slot R - 1 <- own slot R
    {
      uint32_t nxt = nxt_left;
      if (SEG) nxt = seglast ? NEGV : nxt;
      const uint32_t leftl = pk::align16(nxt, LE[0]);              // (own.hi, next.lo)
#pragma unroll
      for (int p = 0; p < RH; p++)
      {
        cellpair<EK>(HO[p], UO[p], LO[p], bestO[p], btO[p], gebO[p], tfO[p], tlO[p], HALF == 0 ? accO[p] : acc2O[p],
                     UE[p], p == RH - 1 ? leftl : LE[p == RH - 1 ? p : p + 1], MAT ? ROW[MAT ? p : 0] : OW[p], MW[p], tv1, clO[p],
                     MAT ? ROW[MAT ? p + RH : 0] : 0u);
        if (TRK) track_key<EK>(kbO[p], HO[p], tlO[p], tv1, k);
      }
    }
    // mutant window moves on: first register <- (previous lane's last.hi | the pair's feeder, own last.lo)
    {
      const int mi = yfeed_m + it;
      const uint32_t fbm = Base::feed_byte(fm_lo, fm_hi, k);
      // (MAT: the letter arrives in a high half and is moved to the low half of the first register below: high-half code)
      const uint32_t feed = (MAT ? ((!EDGE || (uint32_t)mi < (uint32_t)Y) ? (MSEL | (fbm + 4u)) : MSENT_HI)
                                 : ((!EDGE || (uint32_t)mi < (uint32_t)Y) ? fbm : SENT_M)) << 16;
      // ... together with the up offer of the next iteration's even step (one exchange)
      int32_t mv[2] = {(int32_t)MW[RH - 1], (int32_t)UO[RH - 1]};
      const int32_t mo[2] = {(int32_t)feed, (int32_t)NEGV};
      xshrv<P, 2>(mv, mo, phase_r); phase_r ^= 1;
      uint32_t prv = (uint32_t)mv[0];
      up_prev = (uint32_t)mv[1];
      if (SEG) prv = segfirst ? feed : prv;
      uint32_t first = pk::align16(MW[RH - 1], prv);
      // MAT: what moved from a high half into the low one sheds its + 4, what moved from a low half into the high one gains
      // it (one 32-bit add: no half borrows, codes are 4 .. 12 in the low half here)
      if (MAT) first += MADJ;
#pragma unroll
      for (int p = RH - 1; p > 0; p--) MW[p] = MW[p - 1];
      MW[0] = first;
    }
  }

  template <int EK>
  PW_FN void block16(int b) {
#pragma unroll
    for (int p = 0; p < RH; p++) { accE[p] = 0; accO[p] = 0; acc2E[p] = 0; acc2O[p] = 0; }
    uint32_t kb0E[RH], kb0O[RH];
    if (TRK) {
#pragma unroll
      for (int p = 0; p < RH; p++) {        // a later cell with the same score loses against 8 best + 7
        kb0E[p] = kbE[p] = pk::mad(bestE[p], SC4 ? C2 : C8, SEVEN); kb0O[p] = kbO[p] = pk::mad(bestO[p], SC4 ? C2 : C8, SEVEN);
      }
    }
    // unroll depth: full for narrow lanes (the letter-window shifts become register renames), shallower for
    // wide ones, where the live state already fills the register file
#pragma clang loop unroll_count(UNR)
    for (int k = 0; k < 4; k++) iteration16<EK, 0>(8 * b + k, k);
#pragma clang loop unroll_count(UNR)
    for (int k = 4; k < 8; k++) iteration16<EK, 1>(8 * b + k, k);
    if (TRK) {
      // cell c of this block (iteration 8 b + c) is step 16 b + 2 c of an even slot, 16 b + 2 c + 1 of an odd one
      const uint32_t baseE = pk::both(16 * b + 14), baseO = pk::both(16 * b + 15);
#pragma unroll
      for (int p = 0; p < RH; p++) {
        key_to_best(bestE[p], btE[p], kbE[p], kb0E[p], baseE);
        key_to_best(bestO[p], btO[p], kbO[p], kb0O[p], baseO);
      }
    }
    // 8 cells per slot -> one dword, first cell in the top nibble; un-invert: kept = 7 - (not kept).
    // Slots are gathered into their natural order so that every group of 4 goes out as one 16-byte store.
    // Lanes with nothing to store (padding lanes, lanes beyond the plane's rows, blocks past the pair's last)
    // write into the plane's spare row (row `nblocks`, slot 0) instead of branching: a branch here splits the
    // unrolled block and serialises the packed ops (dependent VOP3P ops need a wait state between them).
    {
      const bool st = valid && li < pd.nl && b < pd.nblocks;
      const int bb = st ? b : pd.nblocks, ll = st ? li : 0;
      uint32_t mwd[BK];
#pragma unroll
      for (int p = 0; p < RH; p++) {
        mwd[2 * p] = 0x77777777u - (((accE[p] & 0xffffu) << 16) | (acc2E[p] & 0xffffu));
        mwd[2 * p + R] = 0x77777777u - ((accE[p] & 0xffff0000u) | (acc2E[p] >> 16));
        mwd[2 * p + 1] = 0x77777777u - (((accO[p] & 0xffffu) << 16) | (acc2O[p] & 0xffffu));
        mwd[2 * p + 1 + R] = 0x77777777u - ((accO[p] & 0xffff0000u) | (acc2O[p] >> 16));
      }
      uint32_t* dst = a.masks + pd.mask_off;
#pragma unroll
      for (int g = 0; g < BK / 4; g++) {
        U4 v; v.x = mwd[4 * g]; v.y = mwd[4 * g + 1]; v.z = mwd[4 * g + 2]; v.w = mwd[4 * g + 3];
        *(U4*)(dst + mask_word_index(BK, pd.nl, bb, ll, 4 * g)) = v;
      }
    }
  }

  PW_FN void feed_issue(int b) {
    const uint32_t* o32 = (const uint32_t*)oseq;
    const uint32_t* m32 = (const uint32_t*)mseq;
    const int wo = (xfeed_o + 8 * b) >> 2, wm = (yfeed_m + 8 * b) >> 2;
    if (!SEG) {
      // one pair per wavefront (or workgroup): the indices are wave-uniform -- scalar loads, which leave vmcnt to the mask stores
      fo_n0 = P::const_dword(oseq, pw_clampi(wo, 0, owlast)); fo_n1 = P::const_dword(oseq, pw_clampi(wo + 1, 0, owlast));
      fo_n2 = P::const_dword(oseq, pw_clampi(wo + 2, 0, owlast));
      fm_n0 = P::const_dword(mseq, pw_clampi(wm, 0, mwlast)); fm_n1 = P::const_dword(mseq, pw_clampi(wm + 1, 0, mwlast));
      fm_n2 = P::const_dword(mseq, pw_clampi(wm + 2, 0, mwlast));
      return;
    }
    fo_n0 = o32[pw_clampi(wo, 0, owlast)]; fo_n1 = o32[pw_clampi(wo + 1, 0, owlast)]; fo_n2 = o32[pw_clampi(wo + 2, 0, owlast)];
    fm_n0 = m32[pw_clampi(wm, 0, mwlast)]; fm_n1 = m32[pw_clampi(wm + 1, 0, mwlast)]; fm_n2 = m32[pw_clampi(wm + 2, 0, mwlast)];
  }
  PW_FN void feed_commit(int b) {
    const int ro = (xfeed_o + 8 * b) & 3, rm = (yfeed_m + 8 * b) & 3;
    fo_lo = Base::funnel(fo_n1, fo_n0, ro); fo_hi = Base::funnel(fo_n2, fo_n1, ro);
    fm_lo = Base::funnel(fm_n1, fm_n0, rm); fm_hi = Base::funnel(fm_n2, fm_n1, rm);
  }

  PW_FN int tfirst_of(int j) const {      // first step of slot j's diagonal; never for a diagonal outside the band
    const int dd = li * BK + j, d = pd.dmin + dd;
    return (valid && dd < ndiag) ? (d < 0 ? -d : d) - pd.s0 : 32767;
  }
  PW_FN int tlast_of(int j) const {
    const int dd = li * BK + j, d = pd.dmin + dd;
    if (!valid || dd >= ndiag) return -1;
    const int len = 1 + (d > 0 ? 0 : d) + (X - d > Y ? Y : X - d);
    return (d < 0 ? -d : d) - pd.s0 + 2 * (len - 1);
  }
  PW_FN int tbegin_of(int j) const {      // RULE != 0: the step of the diagonal's cell that may begin an alignment
    const int dd = li * BK + j, d = pd.dmin + dd;
    if (!valid || dd >= ndiag) return 32767;
    if ((RL == 2 || RL == 5 || a.brule == BRULE_ORIGIN) && d != 0) return 32767;   // begin at cell (0, 0) only (B_GLOBAL, GLOBAL, START_ANCHORED, START_ANCHORED_OVERLAP)
    return (d < 0 ? -d : d) - pd.s0;                       // the first cell of the diagonal lies on the table edge
  }
  // MAT: the row of origin letter l (uniform rows, selected by compares: scalar code where l is wave-uniform)
  // (the rows are copied out of the kernel arguments once -- MR0 .. MR3 -- so that the selection is a chain of selects on
  //  values already in registers; read in place, each arm became a load behind a branch, and a branch in the unrolled block
  //  serialises the packed ops)
  PW_FN uint32_t row_of(uint32_t l) const {
    uint32_t r = l == 3u ? MR3 : 0u;
    r = l == 2u ? MR2 : r;
    r = l == 1u ? MR1 : r;
    return l == 0u ? MR0 : r;
  }
  PW_FN uint32_t row_at(int i) const { return (uint32_t)i < (uint32_t)X ? row_of((uint32_t)oseq[i]) : 0u; }
  PW_FN uint32_t msel_lo(int i) const { return (uint32_t)i < (uint32_t)Y ? (MSEL | (uint32_t)mseq[i]) : MSENT_LO; }
  PW_FN uint32_t msel_hi(int i) const { return (uint32_t)i < (uint32_t)Y ? (MSEL | ((uint32_t)mseq[i] + 4u)) : MSENT_HI; }
  PW_FN uint32_t letter_o(int i) const { return (uint32_t)i < (uint32_t)X ? (uint32_t)oseq[i] : SENT_O; }
  PW_FN uint32_t letter_m(int i) const { return (uint32_t)i < (uint32_t)Y ? (uint32_t)mseq[i] : SENT_M; }
  PW_FN int blocked(int j) const { return li * BK + j == ndiag ? NEG16 : 0; }   // first diagonal above the band

  PW_FN void run() {
    const int lane = P::lane();
    const int nl = wd.nl;
    const int seg = lane / nl;
    li = lane - seg * nl;
    valid = seg < wd.count;
    segfirst = li == 0; seglast = li == nl - 1;
    const int slot = wd.first + ((SEG && valid) ? seg : 0);   // lanes beyond the last pair shadow the first one
    pair_slot = a.order ? a.order[slot] : slot;
    if (!SEG) pair_slot = P::uniform(pair_slot);             // one pair: keep its descriptor in scalar registers
    pd = a.pairs[pair_slot];
    X = pd.X; Y = pd.Y; ndiag = pd.ndiag;
    oseq = a.arena + pd.o_off; mseq = a.arena + pd.m_off;
    owlast = (X > 0 ? X - 1 : 0) >> 2; mwlast = (Y > 0 ? Y - 1 : 0) >> 2;
    const int e = (pd.s0 + pd.dmin) >> 1;       // x of diagonal dd = 0 on step t = 0 (s0 == dmin mod 2)
    const int f = (pd.s0 - pd.dmin) >> 1;       // y likewise
    const int xbase = e + li * R, ybase = f - li * R;
    xfeed_o = e + nl * R - 1;                   // the letter a virtual lane `nl` would hand down
    yfeed_m = f;
    ONE = pk::opaque(0x00010001u); SH15 = pk::opaque(0x000f000fu);
    C2 = pk::opaque(0x00020002u); C4 = pk::opaque(0x00040004u); C16 = pk::opaque(0x00100010u);
    C8 = pk::opaque(0x00080008u); SH3 = pk::opaque(0x00030003u); SEVEN = pk::opaque(0x00070007u); NEG2 = pk::opaque(0xfffefffeu);
    NEGV = pk::both(NEG16); LIMV = pk::both(-32767);
    NDELTA = pk::both(SCL * (a.mismatch - a.match)); MATCHV = pk::both(SCL * a.match);
    // the multipliers of the "not kept" values: 0 / 1 each, or (SC4) 0 / 2 for D and 0 / 4 for I
    GOV = pk::both(SC4 ? 2 * a.go : a.go); GOVI = pk::both(a.go);
    BIASV = pk::both(a.mat_bias); MADJ = 0x0003fffcu;
    MR0 = a.mat_rows[0]; MR1 = a.mat_rows[1]; MR2 = a.mat_rows[2]; MR3 = a.mat_rows[3];
    if (MAT) {
#pragma unroll
      for (int i = 0; i < (MAT ? R : 1); i++) ROW[i] = row_at(xbase + i - 1);
    }
#pragma unroll
    for (int p = 0; p < RH; p++) {
      const int e0 = 2 * p, e1 = 2 * p + R, o0 = 2 * p + 1, o1 = 2 * p + 1 + R;
      if (ANYB) {
        gebE[p] = pk::pack(SCL * a.ge + blocked(e0), SCL * a.ge + blocked(e1));
        gebO[p] = pk::pack(SCL * a.ge + blocked(o0), SCL * a.ge + blocked(o1));
        tfE[p] = pk::pack(tfirst_of(e0), tfirst_of(e1)); tfO[p] = pk::pack(tfirst_of(o0), tfirst_of(o1));
        clE[p] = clO[p] = 0;
      } else {
        gebE[p] = gebO[p] = pk::both(a.ge);
        clE[p] = pk::pack(blocked(e0) ? NEG16 : 32767, blocked(e1) ? NEG16 : 32767);
        clO[p] = pk::pack(blocked(o0) ? NEG16 : 32767, blocked(o1) ? NEG16 : 32767);
        tfE[p] = pk::pack(tbegin_of(e0), tbegin_of(e1)); tfO[p] = pk::pack(tbegin_of(o0), tbegin_of(o1));
      }
      tlE[p] = pk::pack(tlast_of(e0), tlast_of(e1)); tlO[p] = pk::pack(tlast_of(o0), tlast_of(o1));
      HE[p] = UE[p] = LE[p] = HO[p] = UO[p] = LO[p] = NEGV;
      up_prev = NEGV; phase_l = 0; phase_r = 0;
      // rule 0: scores never go below 0, and a diagonal whose best stays 0 reports its first cell (score 0 on the table edge)
      // (rule 5: a best of 0 never wins -- the end cell must beat 0 -- so where it "was reached" does not matter)
      bestE[p] = bestO[p] = TRK ? 0u : NEGV;
      btE[p] = (TRK && ANYB) ? tfE[p] : 0u; btO[p] = (TRK && ANYB) ? tfO[p] : 0u;
      OW[p] = pk::pack((int32_t)letter_o(xbase + p - 1), (int32_t)letter_o(xbase + p + RH - 1));
      if (MAT) MW[p] = pk::pack((int32_t)msel_lo(ybase - p - 1), (int32_t)msel_hi(ybase - p - RH - 1));
      else MW[p] = pk::pack((int32_t)letter_m(ybase - p - 1), (int32_t)letter_m(ybase - p - RH - 1));
    }
    // The planner's steady range allows a diagonal's LAST cell to be the last step of a steady block (every cell of
    // the block is still valid); rules 1 / 2 capture that cell, which only the edge body does: give up that block.
    const int sb1 = CAP ? wd.steady_b1 - 1 : wd.steady_b1;
    feed_issue(0);
    for (int b = 0; b < wd.nblocks; b++) {
      feed_commit(b);
      if (b + 1 < wd.nblocks) feed_issue(b + 1);
      if (b >= wd.steady_b0 && b < sb1) block16<0>(b);
      else if (!ANYB) block16<3>(b);
      else if (b >= wd.steady_b0) block16<2>(b);                   // every diagonal has started, some may have ended
      else if (sb1 > wd.steady_b0) block16<1>(b);                  // some have not started; none has ended before the steady range
      else block16<3>(b);                                          // (no steady range: the planner's bounds cannot tell)
    }
    finish();
  }

  // End-cell search for the two rules this kernel serves (END_STD_LOCAL, END_BANDED_LOCAL): the first best
  // cell of every in-band diagonal, reduced on (score desc, scan rank asc) -- once per pair of the wave.
  PW_FN void finish() {
    const int endrule = a.endrule;
    int32_t cs = -32768; uint64_t ck = ~(uint64_t)0; int cx = -1, cy = -1; bool have = false;
    // Runs once per pair: keep it out of the register budget of the fill loop.  The packed bests are parked
    // in a small private array and scanned by a rolled loop.
    uint32_t parked[4 * RH];
#pragma unroll
    for (int p = 0; p < RH; p++) {
      parked[4 * p] = bestE[p]; parked[4 * p + 1] = btE[p]; parked[4 * p + 2] = bestO[p]; parked[4 * p + 3] = btO[p];
    }
#pragma unroll 1
    for (int q = 0; q < 2 * RH; q++) {
      const int p = q >> 1, odd = q & 1;
      const uint32_t bq = parked[4 * p + 2 * odd], tq = parked[4 * p + 2 * odd + 1];
#pragma unroll 1
      for (int h = 0; h < 2; h++) {
        const int j = 2 * p + odd + h * R;
        const int dd = li * BK + j, d = pd.dmin + dd;
        const int32_t s = (h ? pk::hi_s(bq) : pk::lo_s(bq)) / SCL;   // (exact: every running value is a multiple of SCL)
        const int bt = (int)(h ? (tq >> 16) : (tq & 0xffffu));
        const int tfirst = (d < 0 ? -d : d) - pd.s0;
        const int aa = (bt - tfirst) >> 1;
        int x = aa + (d > 0 ? d : 0), y = aa - (d < 0 ? d : 0);
        uint64_t k;
        bool ok = true;
        if (CAP) {
          // `s` is the captured value of the diagonal's last cell (_banded_find_optimal, _pw_internals.c:364-414)
          const bool ends_right = d < X - Y;
          x = ends_right ? d + Y : X; y = ends_right ? Y : X - d;
          // ties: banded overlap takes the diagonals in ascending order, standard overlap the last column top-down and then
          // the last row left to right (_std_find_optimal: on the full band every diagonal ends on one of the two)
          k = endrule == END_STD_OVERLAP ? (uint64_t)(uint32_t)(x < X ? x : X + y) : (uint64_t)(uint32_t)dd;
          if (RL == 2 || RL == 4 || endrule == END_CORNER) ok = d == X - Y;      // end at cell (X, Y) only
        } else if (endrule == END_STD_LOCAL) k = (uint64_t)(uint32_t)x * (uint64_t)(uint32_t)(Y + 1) + (uint64_t)(uint32_t)y;
        else k = ((uint64_t)(uint32_t)dd << 32) | (uint64_t)(uint32_t)aa;
        const bool better = ok && valid && dd < ndiag && (!have || s > cs || (s == cs && k < ck));
        if (better) { cs = s; ck = k; cx = x; cy = y; have = true; }
      }
    }
    const int lane = P::lane();
    const int seg = lane / wd.nl;
    for (int sidx = 0; sidx < wd.count; sidx++) {
      int hv = (have && seg == sidx) ? 1 : 0;
      int32_t rs = cs; uint64_t rk = ck; int rx = cx, ry = cy;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const int32_t os = P::shfl_xor(rs, off);
        const uint64_t ok_ = xshfl_xor<P>(rk, off);
        const int ox = P::shfl_xor(rx, off), oy = P::shfl_xor(ry, off), oh = P::shfl_xor(hv, off);
        const bool take = oh && (!hv || os > rs || (os == rs && ok_ < rk));
        if (take) { rs = os; rk = ok_; rx = ox; ry = oy; hv = 1; }
      }
      // several wavefronts per pair (K2a with this body; one pair, SEG = false): the per-wave winners the same way
      if (P::nwaves() > 1) {
        int32_t bs = rs; uint64_t bk_ = rk; int bx = rx, by = ry, bh = 0;
        for (int wv = 0; wv < P::nwaves(); wv++) {
          const int32_t os = P::wave_bcast(rs, wv);
          const uint64_t ok_ = xwave_bcast<P>(rk, wv);
          const int ox = P::wave_bcast(rx, wv), oy = P::wave_bcast(ry, wv), oh = P::wave_bcast(hv, wv);
          const bool take = oh && (!bh || os > bs || (os == bs && ok_ < bk_));
          if (take) { bs = os; bk_ = ok_; bx = ox; by = oy; bh = 1; }
        }
        rs = bs; rk = bk_; rx = bx; ry = by; hv = bh;
      }
      if (seg == sidx && li == 0) {
        Result r;
        r.score = (double)rs * a.score_mul;
        r.opt_i = a.banded ? rx - ry - pd.dmin : rx;
        r.opt_j = a.banded ? (rx < ry ? rx : ry) : ry;
        r.origin_idx = 0; r.mutant_idx = 0; r.tx_len = 0; r.status = 0;
        // LOCAL starts from the score of cell (0,0), i.e. 0 (_pw_internals.c:342)
        if (!hv || (endrule == END_STD_LOCAL && !(rs > 0))) { r.opt_i = -1; r.opt_j = -1; r.score = 0.0; }
        a.results[pair_slot] = r;
      }
    }
  }
};

// =================================================================================================
// K4: traceback of one pair by one lane, over the tie-mask plane.
// Predecessor rule after a gap op g (pinned by oracle `maskrule_ok`): go < 0 -> g if g is
// kept in the predecessor else its first kept op; go == 0 -> first kept; go > 0 -> first kept op
// other than g, else g.  After M/S: first kept (_pw_internals.c:235).
// =================================================================================================
PW_FN int pw_first_op(uint32_t mask) {   // index of the lowest set bit: 0 B, 1 D, 2 I, 3 M
  return (mask & 1u) ? 0 : (mask & 2u) ? 1 : (mask & 4u) ? 2 : 3;
}


// ---- K4a: the walk --------------------------------------------------------------------------------
// One lane walks one pair; a wavefront holds 64 walkers.  The walk is a chain of dependent steps, and
// 64 divergent walkers share one program counter, so ANY lane touching memory stalls all 64.  The loop
// is therefore split: an outer iteration in which every unfinished lane refills a private register
// cache with ONE batched access -- the 16-byte mask groups (4 adjacent diagonals) of its current block
// and of the block before it -- and an inner loop that only steps while the next cell is inside that
// cache: no loads at all, just byte stores of the ops.  A lane that leaves its cached window idles
// until the others do; all refill together.  Sequence letters are not read here: diagonal moves are
// written as 'X' and turned into 'M'/'S' by the fix-up pass (K4b), which is fully parallel.
// Window of the mask plane a walker keeps on chip: WIN_B consecutive blocks x 2 adjacent diagonal groups
// (4 diagonals each), one dword per (block, group, diagonal): 64 steps along the path x 8 diagonals.
enum { WIN_B = 4, WIN_WORDS = WIN_B * 2 * 4 };

PW_FN void trace_walk(const TraceParams& p, int pair, uint32_t* win /* WIN_WORDS dwords private to this walker (LDS) */) {
  // everything the loops need is copied into locals first: the byte stores of the ops go through a
  // uint8_t*, which the compiler must assume aliases the descriptors
  const PairDesc pdv = p.pairs[pair];
  if (!pdv.solvable || pdv.layout != 0) return;      // (strip-layout pairs have their own walker: pw_strip.h)
  Result r = p.results[pair];
  const int ei = p.ends ? p.ends[2 * pair] : r.opt_i;
  const int ej = p.ends ? p.ends[2 * pair + 1] : r.opt_j;
  if (ei < 0 || ej < 0) { r.tx_len = 0; r.status = 0; p.results[pair] = r; return; }
  const int dmin = pdv.dmin, s0 = pdv.s0, ndiag = pdv.ndiag, bk = pdv.bk, nl = pdv.nl, gosign = p.gosign;
  int x = ei, y = ej;
  if (p.banded) {                 // _xy_from_cellpos (_pw_internals.c:100-114)
    const int d = ei + dmin;
    x = ej + (d > 0 ? d : 0); y = ej - (d > 0 ? 0 : d);
  }
  const uint32_t* __restrict__ plane = p.masks + pdv.mask_off;
  uint8_t* __restrict__ tx = p.transcripts + pdv.tx_off;
  const int lgG = bk < 4 ? 1 : 2;               // log2 of the dwords per lane group (pw_types.h: mask_word_index)
  const int G = 1 << lgG;
  const int njg = bk >> lgG;                    // groups per lane
  const int ngroups = (ndiag + G - 1) >> lgG;   // groups that hold in-band diagonals
  int pos = pdv.tx_cap;           // ops are written backwards, ending right-aligned in the slot
  int nms = 0, bad = 0;
  int prev = 3;                   // op that led to the current cell; "M" makes the end cell use its first choice (pw.c:123)
  bool done = false;
  while (!done) {
    // ---- refill: blocks cb .. cb - 3 of the current cell's diagonal group and of the neighbouring group on
    //      the side the cell sits on (paths drift by single diagonals); all loads issued back to back ----
    const int dd0 = x - y - dmin, t0 = x + y - s0;
    // (never expected -- see below; x > X or y > Y would index the mask plane behind its last block)
    if (x < 0 || y < 0 || x > pdv.X || y > pdv.Y || dd0 < 0 || dd0 >= ndiag) { bad = 1; break; }
    const int cb = t0 >> 4;
    const int ga = dd0 >> lgG;
    int gb = ((dd0 & (G - 1)) >= (G >> 1)) ? ga + 1 : ga - 1;
    gb = gb < 0 ? ga + 1 : (gb >= ngroups ? ga - 1 : gb);
    if (gb < 0) gb = ga;          // a band of a single group
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) {
      const int gg = s2 ? gb : ga;
      const int ln = gg / njg, jg = gg - ln * njg;
#pragma unroll
      for (int k = 0; k < WIN_B; k++) {
        const int bb = cb - k > 0 ? cb - k : 0;
        // (for bk = 2 a group is 8 bytes: the upper half of the 16-byte load is the neighbouring lane's group,
        //  in bounds thanks to the slack behind the mask workspace, and never selected below)
        const U4 v = *(const U4*)(plane + ((uint64_t)((uint64_t)bb * njg + jg) * nl + ln) * G);
        uint32_t* wslot = win + (k * 2 + s2) * 4;
        wslot[0] = v.x; wslot[1] = v.y; wslot[2] = v.z; wslot[3] = v.w;
      }
    }
    // ---- step while the cell is inside the window: no global loads, only byte stores of the ops ----
    while (true) {
      const int dd = x - y - dmin, t = x + y - s0;
      // a well-formed mask plane never leads outside the table; if it ever did (a kernel bug), stop
      if (x < 0 || y < 0 || x > pdv.X || y > pdv.Y || dd < 0 || dd >= ndiag) { bad = 1; done = true; break; }
      const int gg = dd >> lgG, b = t >> 4, kb = cb - b;
      if ((gg != ga && gg != gb) || kb < 0 || kb >= WIN_B) break;              // miss: refill
      const uint32_t w = win[(kb * 2 + (gg == ga ? 0 : 1)) * 4 + (dd & (G - 1))];
      const uint32_t pm = (w >> (4 * (7 - ((t & 15) >> 1)))) & 15u;
      // which kept choice of this cell is the base of the step that led here
      int op;
      if (prev == 3 || gosign == 0) op = pw_first_op(pm);
      else if (gosign < 0) op = (pm & (1u << prev)) ? prev : pw_first_op(pm);
      else { const uint32_t others = pm & ~(1u << prev); op = others ? pw_first_op(others) : prev; }
      if (op == 0 || pos <= 0) { done = true; break; }
      tx[--pos] = op == 3 ? 'X' : (op == 1 ? 'D' : 'I');
      nms += (op == 3);
      x -= (op != 2); y -= (op != 1);
      prev = op;
      if (op == 3) {
        // bulk step: a run of diagonal moves stays on this diagonal and, while the cell index n inside the
        // block stays >= 0, inside this very dword.  After M the next op is the cell's first kept choice,
        // which is M again exactly when none of B, D, I is kept: (nibble & 7) == 0.  Count those cells
        // with one ctz and write eight 'X' at once: ops are written backwards, so the bytes below the
        // run are overwritten by the ops that follow (the slot's first 8 bytes are left to the slow path).
        const int n = ((t & 15) >> 1) - 1;           // cell index of the new (x, y) within the dword
        if (n >= 0 && pos >= 16) {
          const uint64_t v = (uint64_t)((w & 0x77777777u) >> (4 * (7 - n))) | ((uint64_t)1 << (4 * (n + 1)));
          int k = (int)(__builtin_ctzll(v) >> 2);     // pure-M cells ahead, at most n + 1
          const int lim = x < y ? x : y;               // a diagonal move needs x >= 1 and y >= 1
          k = k < lim ? k : lim;
          if (k > 0) {
            PackedU64 xs; xs.v = 0x5858585858585858ull;
            *(PackedU64*)(tx + pos - 8) = xs;
            pos -= k; nms += k; x -= k; y -= k;
          }
        }
      }
    }
  }
  r.origin_idx = x; r.mutant_idx = y;
  r.tx_len = pdv.tx_cap - pos;
  r.status = ST_TRACED | (r.tx_len == 0 ? ST_EMPTY : 0) | ((x + y + nms <= 0) ? ST_PANICK : 0) | (bad ? ST_BADPATH : 0);
  p.results[pair] = r;
}

// ---- K4b: turn the diagonal moves 'X' into 'M' / 'S' (_pw_internals.c:232) ------------------------
// Serial form (one pair, one thread): what the wave-parallel device version in pw_trace.hip computes
// with ballots + popcounts; the CPU lane emulator uses this one.
PW_FN void trace_fixup_serial(const TraceParams& p, int pair) {
  const PairDesc& pd = p.pairs[pair];
  if (!pd.solvable) return;
  const Result r = p.results[pair];
  if (!(r.status & ST_TRACED) || r.tx_len <= 0) return;
  const uint8_t* oseq = p.arena + pd.o_off;
  const uint8_t* mseq = p.arena + pd.m_off;
  uint8_t* tx = p.transcripts + pd.tx_off + pd.tx_cap - r.tx_len;
  int x = r.origin_idx, y = r.mutant_idx;
  for (int k = 0; k < r.tx_len; k++) {
    const uint8_t ch = tx[k];
    if (ch == 'X') { tx[k] = (oseq[x] == mseq[y]) ? 'M' : 'S'; x++; y++; }
    else if (ch == 'D') x++;
    else y++;
  }
}

}  // namespace pw
#endif
