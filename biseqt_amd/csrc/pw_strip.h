// pw_strip.h -- K2c: ONE pair whose table is wider than a workgroup can hold (standard-mode tables with more than
// 16 384 diagonals; BASELINE config 3, 100 kb x 100 kb), computed as a PIPELINE OF ROW STRIPS.
//
// What it computes: the same cell update as pw_wave.h (reference dptable_solve, biseqt/pwlib/pw.c:47-114, move
// generators _pw_internals.c:161-299, end-cell search :303-360), cell = (H, 4-bit ordered tie mask).
//
// Why another decomposition.  With lanes = DIAGONALS (pw_wave.h) a cell needs its two neighbouring diagonals, so
// wavefronts that share a band exchange values in BOTH directions every step: inside a workgroup that is an LDS
// round trip + barrier per step (K2a), across workgroups it forces ghost zones and a kernel boundary every few dozen
// steps (K2b: 0.37 us per anti-diagonal, 75 ms for config 3).  With lanes = ROWS the dependencies point one way
// only: cell (x, y) needs (x-1, y), (x-1, y-1) from the row above and (x, y-1) from its own row.  A wavefront owns a
// strip of 64 consecutive rows and sweeps the columns; lane i works on column y = k - i at step k, takes lane i-1's
// values of the previous step by ONE DPP wave shift (wave_shr:1) and nothing ever flows back.  The strip below only
// needs the last row of the strip above, a step behind: a FIFO in memory, written by lane 63 of the producer (one
// 8-byte granule {2 H + "the D choice is kept", tag} per column) and read by lane 0 of the consumer in chunks of 32
// columns.  Strips are handed to the resident wavefronts in index order from a work queue, so a strip only ever waits
// for one that was started before it: no grid barrier, no kernel boundary, no recomputed ghost cells -- the critical
// path is X + Y steps of ONE cell per lane plus one FIFO hop per strip.
//
// Tie masks: lane i packs the 8 cells of 8 consecutive steps into a dword (first cell in the top nibble) and stores
// 16 bytes per 32 steps; plane layout [strip][step / 32][lane][4] dwords = 0.5 B per cell of the 64-row-aligned table
// (5.0 GB for config 3, against 10 GB for the bounding parallelogram of the diagonal layout).  strip_walk() is the
// traceback over this layout: one wavefront per pair, the walk itself runs on wave-uniform (scalar) values and reads
// the mask words out of a 64-row x 64-step register window with v_readlane.
//
// Written against a platform policy P like pw_wave.h, so the CPU lane emulator (tests/emu) runs the same code.
#ifndef PW_STRIP_H
#define PW_STRIP_H

#ifndef PW_FN
#error "PW_FN must be defined by the includer"
#endif

#include "pw_types.h"
#include "pw_wave.h"      // U4, PackedU64, pw_first_op

namespace pw {

constexpr int kStripBlock = 32;      // steps per block: one 16-byte mask store per lane, one FIFO chunk

struct StripBest { int32_t score, x, y, have; };

// Uniform parameters of one strip-pipeline launch (one pair).
struct StripParams {
  const uint8_t* arena;
  uint64_t o_off, m_off;
  uint64_t* fifo;            // [nstrips][fifo_pitch] granules: low dword 2 H + (D kept in that cell) of the strip's last
                             //   row, high dword the epoch; strip w reads row w - 1
  uint32_t* masks;           // this pair's plane: [nstrips][nkq][64][4] dwords
  StripBest* sbest;          // [nstrips] per-strip end-cell candidates
  uint32_t* ctl;             // [0 .. 7] work-queue heads, one per XCD queue; [kStripAbort] abort flag (a wait ran out of patience);
                             // [kStripRows + o] BROW kernels: row o of the substitution table, byte m = subst[o][m] (match / mismatch
                             // scoring, or any matrix over at most 4 letters whose entries fit a signed byte: _alnchoice_M,
                             // _pw_internals.c:217-245).  In memory, not in the kernel arguments: the kernel is short of scalar
                             // registers as it is (83 spilled), and a lane reads its one row once per strip.
  Result* result;            // the pair's record
  int32_t X, Y;
  int32_t nstrips, nkq;      // strips of 64 rows; blocks of 32 steps per strip (steps 0 .. Y + 63)
  int32_t fifo_pitch;        // granules per FIFO row (strip_fifo_pitch(Y))
  uint32_t epoch;            // tag of this solve: granules of earlier solves never match (the buffer is never cleared)
  int32_t brule, endrule;
  int32_t match, mismatch, go, ge;
  int32_t spin_limit;        // polls of one FIFO chunk before giving up
  double score_mul;          // reported score = the kernel's integer value times this (dyadic scaling, pw_types.h)
  // Placement (speed and store flavour only): strips are dealt in RUNS of run_len consecutive strips; run r is worked
  // on by the wavefronts of ONE XCD (queue r mod nq), so that a strip and the strip above it normally share an L2 and the
  // FIFO between them never leaves it (plain stores, L2-served loads).  Only the last strip of a run hands over to another
  // XCD and writes its row through to memory.  xcc_queue maps the hardware's XCC id to a queue (-1: no such XCD here).
  int32_t run_len, nq;
  int32_t xcc_queue[8];
  uint64_t* stamps;          // tuning aid (PWLIB_STRIP_TRACE): [nstrips][16]: clock stamps [0..6], placement [7], shader-clock counts [8 + i], or null
};
constexpr int kStripAbort = 8;     // index of the abort flag in StripParams::ctl
constexpr int kStripRows = 12;     // ... of the four byte rows

struct StripTraceParams {
  const uint32_t* masks;
  Result* result;
  uint8_t* tx;               // the pair's transcript slot
  const int32_t* ends;       // optional explicit end cell (i, j)
  int32_t X, Y, nkq, tx_cap, gosign;
};

// Granules per FIFO row: the columns 0 .. Y and, behind them, the virtual columns a strip's lane 63 still computes in
// its last blocks (up to 32 nkq - 64 <= Y + 31: they are written like any other, so the end blocks flush without a range
// check) -- rounded up to whole 64-granule lines.
PW_FN int strip_fifo_pitch(int Y) { return (Y + 1 + 32 + 63) / 64 * 64; }

PW_FN uint64_t strip_mask_index(int nkq, int w, int q, int lane) {      // in dwords
  return ((uint64_t)((uint64_t)w * nkq + q) * 64 + lane) * 4;
}

// BROW (byte rows): alphabets of at most 4 letters and scores -- match / mismatch or a whole substitution matrix -- that fit
// a signed byte: the lane holds its row of the substitution table as 4 bytes (rowreg: byte m = score of the row's letter
// against mutant letter m), ONE v_perm_b32 turns
// the 4 mutant letters of a group of 4 steps into their 4 scores, and a step adds its byte, sign-extended, to the diagonal
// predecessor in one SDWA add -- instead of a byte compare, a select and an add per step.
template <class P, bool TRACK, bool BROW = false>
struct StripFill {
  static constexpr int32_t NEG = -(1 << 28);       // "no such predecessor" (pw_wave.h, ScoreTraits<int32_t>)
#ifndef PW_STRIP_ROLL_START
#define PW_STRIP_ROLL_START 0     /* 1: steps 0 .. 63 as a rolled loop (smaller code) */
#endif
#ifndef PW_STRIP_AHEAD
#define PW_STRIP_AHEAD 0    /* steps between a sub-chunk's entry into the feeders and lane 0's first use of it: 0 (it enters lanes
                               0 .. 15 on the step that needs it) | 16 (round 2: lanes 16 .. 31, a sub-chunk early -- and every
                               strip trailing the one above by 16 columns more) */
#endif
#ifndef PW_STRIP_SUB
#define PW_STRIP_SUB 16    /* measured: 16 -> 33.9 ms, 8 -> 35.8 ms on config 3 (one wait for memory per hand-off) */
#endif
  static constexpr int SUB = PW_STRIP_SUB;         // FIFO granules per hand-off (one store / one load of SUB lanes)
  static constexpr int NSB = kStripBlock / SUB;    // hand-offs per block
  static constexpr int AH = PW_STRIP_AHEAD;        // feeder lanes AH .. AH + 15 take a sub-chunk
  static_assert(AH == 0 || AH == SUB, "feeder look-ahead");
  static_assert(SUB == 16, "hand-off size");
  const StripParams& a;
  int lane, w, x;
  // per lane: the cell computed last (H, what it offers downwards / rightwards), the diagonal predecessor of the
  // next cell, the row's letter and the mutant letter that travels with the wavefront
  int32_t Hout, Uout, Lo, Hdiag, best, bestY, hlast, b0, bfirst;
  int32_t kbest;              // steady blocks: the row's running best as a key, 32 * H + (31 - step within the block)
  int32_t ksnap, kY;          // MODE 3: the key as it was on the step of the row's last cell; that step (Y + lane)
  int32_t bqv;                // MODE 4: the begin candidate of this lane's current cell (moves down one lane per step)
  uint32_t oc;
  uint32_t rowreg;            // BROW: the scores of the row's letter against mutant letters 0 .. 3, one byte each
  // the mutant letters this lane meets: lane i needs m[k - 1 - i] at step k, so over the 4 steps of a group (k = 4 g ..
  // 4 g + 3) the 4 bytes that begin at byte 3 - (i mod 4) of the dword pair m32[g - 1 - i / 4], m32[g - i / 4] -- the
  // byte offset is fixed per lane, and the dword a lane needs next is the one the lane 4 below it needed a group ago
  uint32_t w0, w1, wpend, wshift, wfrom;
  // lane-0 feeders: lane j holds what lane 0 needs j steps from now (moved down one lane per step)
  int32_t cH, cU;
  // what lane 63 produced during the last steps, newest in lane 63 (moved down one lane per step) ...
  int32_t gP;                 // ... as 2 H + "the D choice is kept" (|H| < 2^27)
  int32_t vmatch, vmis, vge, vgego;   // the scores, held in vector registers (scalar registers are scarce in the loop)
  // the mutant letters of a block, m[32 q .. 32 q + 31], as 8 wave-uniform dwords (scalar loads: they neither occupy the
  // vector memory counter nor a lane), this block's and the next one's; lane 0 is fed m[k - 1] at step k
  uint32_t mwin[8], mnext[8];
  // the tie masks of the block just finished; they are stored behind the NEXT hand-over, so that a wait that has to drain
  // this wavefront's stores (a poll, the uncounted waits) finds them a whole sub-block old
  uint32_t mprev[4];
  int mprev_q;
  const uint8_t* mseq;
  uint64_t* fout;             // this strip's FIFO row (written from lane 63's values) or null
  const uint64_t* fin;        // the FIFO row of the strip above or null
  bool cross_out;             // the strip below runs on another XCD: write the row through to memory
  bool cross_in;              // the strip above ran on another XCD: read its row from memory, not from this XCD's L2
  // per-lane constants of the fast hand-over (FAST blocks): lanes AH .. AH + 15 expect the tag (epoch, column), every other lane
  // the "nothing loaded" tag 0 -- so ONE comparison over all 64 lanes says whether the sub-chunk has arrived
  uint32_t wtag, wm255;       // lanes AH .. AH + 15: epoch << 8 and 255; else 0 and 0
  int32_t we0;                // lane - AH: this lane's column within a sub-chunk that sits in lanes AH .. AH + 15
  int32_t npolls, nspins;     // tuning aid (PWLIB_STRIP_TRACE): hand-overs that found their sub-chunk not there yet, polls they took

  PW_FN explicit StripFill(const StripParams& a_) : a(a_) {}
  PW_FN void stamp(int i) {
    if (a.stamps != nullptr && lane == 0) {
      a.stamps[(uint64_t)w * 16 + i] = P::clock();            // 100 MHz
      a.stamps[(uint64_t)w * 16 + 8 + i] = P::cycles();       // the shader clock: what frequency the strip ran at
    }
  }

  // MODE 0  steady: every lane holds an in-table cell that is neither the first nor the last of its row
  //      1  start of a strip (steps 0 .. 63) when its rows are all in the table and the table has more than 64 columns:
  //         lane i starts at step i; nothing else can happen (this phase is on the critical path of every hop)
  //      2  anything (rows beyond the table, tiny tables): every update predicated
  //      3  end of a strip whose rows are all in the table (blocks that hold last columns and the steps after them): the
  //         unpredicated update again -- cells right of the last column are virtual; only the row's running best and
  //         the capture of its last cell look at the column.  (The slowest ~100-step stretch of a strip's life sets
  //         the pace of the whole pipeline: each strip trails the one above by that many columns.)
  template <int MODE>
  PW_FN void step(int k, int j /* step within the block */, uint32_t& macc, uint32_t x4 /* the group's 4 letters | scores */, int s4) {
    constexpr bool RAMP = MODE == 2;
    // lane 0 takes the feeders' lane-0 values; the feeders then move down a lane (what enters at lane 63 is never used:
    // zero fill, so the shifted copy does not depend on the old one and the old register can take the shift below)
    const int32_t fH = cH, fU = cU;
    cH = P::shl1z(fH); cU = P::shl1z(fU);
    const int32_t Hin = P::shr1(Hout, fH);
    const int32_t Uin = P::shr1(Uout, fU);
    const int y = k - lane;
    const int32_t hD = Uin, hI = Lo;
    const int32_t hM = Hdiag + (BROW ? P::sbyte_of(x4, s4) : (oc == P::byte_of(x4, s4) ? vmatch : vmis));
    bool active = true;
    int32_t bq = b0;
    if (RAMP) {
      active = y >= 0 && y <= a.Y && x <= a.X;
      const bool edge = x == 0 || y == 0, orig = x == 0 && y == 0;
      const bool ball = a.brule == BRULE_ANY || (edge && (a.brule == BRULE_EDGE || orig));
      bq = ball ? 0 : NEG;
    } else if (MODE == 4) {
      // MODE 1 for strips below the first, where the begin candidates are the same in every row: NEG left of column 0,
      // `bfirst` in column 0, `b0` after it -- a sequence that simply moves down one lane per step
      bq = bqv = P::shr1(bqv, k == 0 ? bfirst : b0);
    } else if (MODE == 1) {
      // Lanes that have not reached column 0 yet run the same unpredicated update on "virtual" cells: with no begin
      // candidate (and nothing but the initial "no predecessor" values around them) their scores stay below -2^27, which
      // every real cell treats as "no such predecessor" -- exactly what the cells left of column 0 are.  So the first
      // 64 steps of a strip, which sit on the critical path of every hop, cost the same as steady ones.  The first real
      // cell of a row (y == 0) may begin an alignment on the table edge (b0 already covers "anywhere" and row 0).
      bq = y < 0 ? NEG : (y == 0 ? bfirst : b0);
    }
    // maximum in the reference's candidate order B, D, I, M (pw.c:92-103); every kept choice shares the score
    int32_t Hn = hD > hI ? hD : hI;
    Hn = hM > Hn ? hM : Hn;
    Hn = bq > Hn ? bq : Hn;
    const bool bB = Hn == bq, bD = hD == Hn, bI = hI == Hn;
    // offers to the cell below / to the right (_alnchoice_ID, _pw_internals.c:268-278; go <= 0 here)
    const int32_t Un = Hn + (bD ? vge : vgego), Ln = Hn + (bI ? vge : vgego);
    // the M bit stays 0: with go <= 0 the walker never looks at it (the first kept op is M exactly when none of
    // B, D, I is kept: pw_first_op)
    // (the nibble is shifted into the mask word flag by flag: a plain shift for the M bit, then one add-with-carry per flag
    //  -- P::shl1_in, m + m + flag with the comparison's lane mask as carry-in -- instead of three selects, two ORs and a shift)
    Hdiag = Hin;
    if (RAMP) {
      macc = P::shl1_in(P::shl1_in(P::shl1_in(macc << 1, bI && active), bD && active), bB && active);
      Hout = active ? Hn : Hout; Uout = active ? Un : Uout; Lo = active ? Ln : Lo;
      if (active && y == a.Y) hlast = Hn;
      if (TRACK) {
        const bool upd = active && Hn > best;
        best = upd ? Hn : best; bestY = upd ? y : bestY;
      }
    } else if (MODE == 3) {
      macc = P::shl1_in(P::shl1_in(P::shl1_in(macc << 1, bI), bD), bB);
      Hout = Hn; Uout = Un; Lo = Ln;
      // the row's last cell is computed on step kY = Y + lane; cells after it are virtual.  The running key takes them too,
      // so its value at that step is set aside (block<3> then picks, per lane, the snapshot, the key or nothing)
      const bool last = k == kY;
      hlast = last ? Hn : hlast;
      if (TRACK) {
        const int32_t key = (int32_t)(((uint32_t)Hn << 5) | (uint32_t)(31 - j));
        kbest = key > kbest ? key : kbest;
        ksnap = last ? kbest : ksnap;
      }
    } else {
      // (MODE 1: the masks of virtual cells are never visited by the walker, and their scores lose against any real
      //  cell's in the row's running best)
      macc = P::shl1_in(P::shl1_in(P::shl1_in(macc << 1, bI), bD), bB);
      Hout = Hn; Uout = Un; Lo = Ln;
      if (TRACK && (MODE == 0 || MODE == 4)) {
        // every cell is a real one here (|H| < 2^25): one shift-or and one max instead of compare, two selects and the
        // column; block<0> turns the key back into (best, bestY).  (MODE 4: virtual cells, far below, are clamped.)
        const int32_t floor25 = -(1 << 25);
        const int32_t Hk = MODE == 4 ? (Hn > floor25 ? Hn : floor25) : Hn;
        const int32_t key = (int32_t)(((uint32_t)Hk << 5) | (uint32_t)(31 - j));
        kbest = key > kbest ? key : kbest;
      } else if (TRACK) {
        const bool upd = Hn > best;
        best = upd ? Hn : best; bestY = upd ? y : bestY;
      }
    }
    // what the strip below will read: lane 63's cells, collected across the lanes (lane 63 has no source and keeps
    // its own new value; a cell outside the table is dropped when the granules are written)
    gP = P::shl1(gP, P::twice_plus(Hn, bD));
  }

  // After the SUB steps k0 .. k0 + SUB - 1: lane 64 - SUB + j holds lane 63's cell of step k0 + j, column y = k0 + j - 63.
  PW_FN bool flush_out(int k0) {             // true if a store was issued (wave-uniform)
    if (fout == nullptr || k0 + SUB - 64 < 0 || k0 - 63 > a.Y) return false;
    const int y = k0 + (lane - (64 - SUB)) - 63;
    if (lane >= 64 - SUB && y >= 0 && y <= a.Y) {
      const uint64_t g = ((uint64_t)tag_of(y) << 32) | (uint64_t)(uint32_t)gP;
      if (cross_out) P::fifo_store(fout + y, g);
      else P::fifo_store_local(fout + y, g);
    }
    return true;
  }

  // A granule's tag: the solve's epoch (24 bits) and the low bits of its column.  Whatever a register or a FIFO slot held
  // before -- a granule of an earlier solve, of another column, nothing -- fails the comparison, so a load that has not
  // arrived yet (or was waited for with the wrong count) is noticed and fetched again; it can never pass for data.
  PW_FN uint32_t tag_of(int y) const { return (a.epoch << 8) | ((uint32_t)y & 0xffu); }

  // FIFO sub-chunk S = columns SUB S .. SUB S + SUB - 1 of the row above, one granule per lane `first` .. `first` + SUB - 1,
  // loaded into hand-over SLOT (0 / 1: even / odd sub-chunk numbers).  The load is NOT tracked by the compiler and does
  // not land in a register the compiler manages (P::fifo_load_async: on the device the slot is a pair of accumulation
  // registers) -- it is waited for by hand (P::wait_vm) in sub_block.
  template <int SLOT>
  PW_FN void load_sub(int S, int first) const {
    const int e = SUB * S + lane - first;
    P::template slot_zero<SLOT>();
    if (lane >= first && lane < first + SUB && e <= a.Y) P::template fifo_load_async<SLOT>(fin + e, cross_in);
  }
#ifndef PW_STRIP_FAST
#define PW_STRIP_FAST 1     /* 0: every block hands over through the general code (A/B) */
#endif
#ifndef PW_STRIP_ALWAYS_SC1
#define PW_STRIP_ALWAYS_SC1 1     /* FAST blocks read and write the FIFO with agent-scope operations only: one flavour, no branch per
                                     hand-over (a taken branch costs a lone wavefront ~25 cycles; A/B: steady step 74.8 -> 73.6 ns, hops unchanged) */
#endif
#ifndef PW_STRIP_LEAD
#define PW_STRIP_LEAD 8     /* FAST blocks: steps between the issue of a hand-over load and the hand-over: 4 | 8 | 12 | 16 */
#endif
  // FAST blocks (steady ones and the first two of a strip below the first; run() admits them only when every column the
  // hand-overs of the block touch exists): the same hand-over without the range checks.
  //   load   columns c0 .. c0 + 15 into lanes AH .. AH + 15 of SLOT
  //          (LIM, end blocks: only the columns up to Y are waited for -- the strip above writes its virtual columns as well,
  //           but the last of them in its very last flush, and nothing real depends on them: the lanes that would hold them
  //           load nothing and expect nothing; what they put into the feeders only ever reaches virtual cells)
  template <int SLOT, bool LIM>
  PW_FN void load_fast(int c0) const {
    P::template slot_zero<SLOT>();
    const bool have = (unsigned)we0 < (unsigned)SUB && (!LIM || c0 + we0 <= a.Y);
    if (have) P::template fifo_load_async<SLOT>(fin + (c0 + we0), PW_STRIP_ALWAYS_SC1 ? true : cross_in);
  }
  //   merge  the sub-chunk whose first column is c0 (sub-chunk S1) from SLOT into the feeders; VM = vector memory operations
  //          issued after its load.  Tags are compared in all lanes at once (wtag / wm255); anything but "all there" goes
  //          through the polling loop.  Lanes AH .. AH + 15 of the feeders are written by a DPP move with a row mask.
  template <int SLOT, int VM, bool LIM>
  PW_FN void merge_fast(int c0, int S1) {
    uint64_t t = P::template wait_vm<SLOT, VM>();
    uint32_t want = ((uint32_t)(c0 + we0) & wm255) | wtag;
    if (LIM) want = c0 + we0 <= a.Y ? want : 0u;
    // (the one join of the two paths is in front of the unpacking, so the fast path runs straight through)
    if (__builtin_expect(!P::all((uint32_t)(t >> 32) == want), 0)) t = poll_sub<LIM>(t, S1);
    const int32_t pk = (int32_t)(uint32_t)t;
    const int32_t h = pk >> 1;
    const int32_t u = h + ((pk & 1) ? vge : vgego);
    cH = P::template rowmov<AH / SUB>(cH, h); cU = P::template rowmov<AH / SUB>(cU, u);
  }
  // the granules of sub-chunk S1 for lanes AH .. AH + 15, polled until all of them carry their tags
  template <bool LIM>
  PW_FN uint64_t poll_sub(uint64_t t, int S1) {
    const int e = SUB * S1 + we0;
    const bool need = (unsigned)we0 < (unsigned)SUB && (!LIM || e <= a.Y);
    const uint32_t want = tag_of(e);
    int spins = 0;
    while (!P::all(!need || (uint32_t)(t >> 32) == want)) {
      if (++spins > a.spin_limit || ((spins & 63) == 0 && P::uniform((int32_t)P::flag_poll(a.ctl + kStripAbort)) != 0)) give_up();
      P::sleep();
      if (need) t = (cross_in || (spins & 3) == 0) ? P::fifo_poll(fin + e) : P::fifo_poll_local(fin + e);
    }
    npolls++; nspins += spins;
    return need ? t : 0;
  }
  //   flush  lane 63's cells of steps k0 .. k0 + 15 (lanes 48 .. 63 of gP): columns k0 - 63 .. k0 - 48 (virtual ones behind
  //          column Y included: the row has room for them)
  PW_FN void flush_fast(int k0) {
    const int y = k0 + lane - (64 - SUB) - 63;
    if (lane >= 64 - SUB && y >= 0) {
      const uint64_t g = ((uint64_t)tag_of(y) << 32) | (uint64_t)(uint32_t)gP;
      if (PW_STRIP_ALWAYS_SC1 || cross_out) P::fifo_store(fout + y, g);
      else P::fifo_store_local(fout + y, g);
    }
  }
  // Checks that the granules of sub-chunk S (in `t`, lanes first ..) carry their tags -- polling for those that do not --
  // and puts them into the feeders of those lanes.  A wait that is abandoned raises the abort flag and ENDS THE WAVEFRONT
  // on the spot (P::exit_wave: every other wavefront leaves at its next look at the flag, the host sees the flag): nothing
  // above this function carries a "gave up" result, which keeps the hand-over code of the steady loop free of it.
  PW_FN void give_up() {
    P::flag_set(a.ctl + kStripAbort);
    P::exit_wave();
  }
  PW_FN void merge_value(uint64_t t, int S, int first, int count = SUB) {
    const int e = SUB * S + lane - first;
    const bool mine = lane >= first && lane < first + count;
    const bool need = mine && e <= a.Y;
    const uint32_t want = tag_of(e);
    int spins = 0;
    while (!P::all(!need || (uint32_t)(t >> 32) == want)) {
      if (++spins > a.spin_limit || ((spins & 63) == 0 && P::uniform((int32_t)P::flag_poll(a.ctl + kStripAbort)) != 0)) give_up();
      P::sleep();
      if (need) t = (cross_in || (spins & 3) == 0) ? P::fifo_poll(fin + e) : P::fifo_poll_local(fin + e);
    }
    const int32_t pk = (int32_t)(uint32_t)t;
    const int32_t h = need ? (pk >> 1) : NEG;
    const int32_t u = need ? h + ((pk & 1) ? a.ge : a.ge + a.go) : NEG;
    cH = mine ? h : cH; cU = mine ? u : cU;
  }
  PW_FN void store_masks() {
    if (mprev_q < 0) return;
    U4 v; v.x = mprev[0]; v.y = mprev[1]; v.z = mprev[2]; v.w = mprev[3];
    *(U4*)(a.masks + strip_mask_index(a.nkq, w, mprev_q, lane)) = v;
    mprev_q = -1;
  }
  PW_FN void load_letters(int q, uint32_t (&win)[8]) const {        // m[32 q .. 32 q + 31], dwords clamped to the frame
    const int last = a.Y > 0 ? (a.Y - 1) >> 2 : 0;
    if (__builtin_expect(8 * q + 7 <= last, 1)) { P::letters_x8(mseq, 8 * q, win); return; }      // one 32-byte scalar load
#pragma unroll
    for (int d = 0; d < 8; d++) {
      const int idx = 8 * q + d;
      win[d] = P::letters_dword(mseq, idx > last ? last : idx);
    }
  }

  // Start of the 4-step group whose first step is k = 4 g; `fresh` = m32[g] (wave-uniform).  Returns the lane's 4 letters.
  PW_FN uint32_t letters_group(uint32_t fresh) {
    const uint32_t nw = lane < 4 ? fresh : wpend;
    w0 = w1; w1 = nw;
    wpend = P::lane_from(wfrom, w1);          // for the next group: what the lane 4 below holds now
    P::issue_here();                          // (issued now, so that its latency is over by then)
    return P::alignbyte(w1, w0, wshift);
  }

  // The SUB steps of sub-chunk J of block q.  In front of them what lane 63 produced during the previous SUB steps goes
  // out.  In their MIDDLE the hand-over: the granules lane 0 will need from step SUB (S + 1) on (sub-chunk S + 1) go from
  // their hand-over slot into the feeders (lanes SUB/2 ..), and the load of sub-chunk S + 2 is issued -- late enough to
  // find the granules written when this strip trails the one above by the ~108 columns it starts with (so the start of a
  // strip, which sets the pace of the whole pipeline, does not have to poll), and SUB steps ahead of its own hand-over.
  // Right behind the load goes the previous block's mask store.  The wait for a slot counts out what was issued after its
  // load -- that mask store and the FIFO store in front of this sub-chunk (vmcnt(0 .. 2)); everything older has had SUB
  // steps to complete.  (On gfx9-family targets loads and stores share one counter, so a wait for "everything" drains the
  // stores: ~1.2 us, once per hand-over, was the cost of not counting.)
  template <int MODE, int J, bool NOIN>
  PW_FN void sub_block(int q, uint32_t (&mw)[4]) {
    const int S = NSB * q + J;
    const int SM = S + AH / SUB;                          // the sub-chunk that enters the feeders here
    const int k0 = kStripBlock * q + SUB * J;
    constexpr int SLOT = (J & 1) ? 0 : 1;                // its hand-over slot
    constexpr bool FAST = (MODE == 0 || MODE == 3 || MODE == 4) && PW_STRIP_FAST;
    constexpr bool LIM = MODE == 3;
    // (FAST blocks know at compile time whether the strip has a row above it -- NOIN: the first strip -- and every strip has
    //  a row to write: no pointer is looked at per hand-over)
    if (FAST ? NOIN : fin == nullptr) {
      // no row above: "no predecessor" keeps entering the feeders where the granules would (the shift fills with zeros)
      if (FAST) { cH = P::template rowmov<AH / SUB>(cH, NEG); cU = P::template rowmov<AH / SUB>(cU, NEG); }
      else {
        const bool mine = lane >= AH && lane < AH + SUB;
        cH = mine ? NEG : cH; cU = mine ? NEG : cU;
      }
    }
    else if (FAST) {
      // what was issued behind the load that this hand-over waits for: the mask store of block q - 1 in front of J = 1
      // (steady blocks: q >= 2, there always is one) and, when the load went out at the very start of the previous
      // sub-block (PW_STRIP_LEAD 16), the FIFO store of that sub-block
      const bool mstore = PW_STRIP_LEAD != 4 && (J & 1) && (MODE != 4 || q > 0);    // (lead 4: the load goes out behind it)
      const bool fstore = PW_STRIP_LEAD == 16 && MODE != 4;
      if (mstore && fstore) merge_fast<SLOT, 2, LIM>(SUB * SM, SM);
      else if (mstore || fstore) merge_fast<SLOT, 1, LIM>(SUB * SM, SM);
      else merge_fast<SLOT, 0, LIM>(SUB * SM, SM);
      if (PW_STRIP_LEAD == 16) load_fast<1 - SLOT, LIM>(SUB * (SM + 1));
    }
    else {
      // J odd: the mask store of block q - 1 was issued behind the load (there is none in front of block 0)
      const uint64_t t = ((J & 1) && q > 0) ? P::template wait_vm<SLOT, 1>() : P::template wait_vm<SLOT, 0>();
      merge_value(t, SM, AH);
    }
    if (AH == 0 && MODE != 0 && MODE != 3 && J == 0 && q == 0) stamp(2);       // the first granules are in
    bool flushed = false;
    if (FAST) { if (MODE != 4) flush_fast(k0 - SUB); }   // (MODE 4: steps 0 .. 47 hold no cell of lane 63)
    else flushed = flush_out(k0 - SUB);
    (void)flushed;
#pragma unroll
    for (int h = 0; h < SUB / 8; h++) {                  // one mask dword per 8 steps
      const int hb = J * (SUB / 8) + h;                  // 8-step group within the block
      if (h == SUB / 16) {
        if (FAST) { if (!NOIN && PW_STRIP_LEAD == 8) load_fast<1 - SLOT, LIM>(SUB * (SM + 1)); }
        else if (fin != nullptr) load_sub<1 - SLOT>(SM + 1, AH);
        if (J == 0) store_masks();
      }
      uint32_t m = 0;
#pragma unroll
      for (int g2 = 0; g2 < 2; g2++) {
        if (FAST && !NOIN && PW_STRIP_LEAD == 12 && h == 0 && g2 == 1) load_fast<1 - SLOT, LIM>(SUB * (SM + 1));
        if (FAST && !NOIN && PW_STRIP_LEAD == 4 && h == 1 && g2 == 1) load_fast<1 - SLOT, LIM>(SUB * (SM + 1));
        const uint32_t l4 = letters_group(mwin[2 * hb + g2]);
        const uint32_t x4 = BROW ? P::perm_bytes(rowreg, l4) : l4;
        // the next block's letters: scalar loads share lgkmcnt with the lane exchange above, so they are issued right
        // behind one group's wait and have the 4 steps to the next one to arrive
        if (J == NSB - 1 && h == 0 && g2 == 0) load_letters(q + 1, mnext);
        const int j0 = 8 * hb + 4 * g2;                    // step within the block
        if (MODE == 2 || ((MODE == 1 || MODE == 3 || MODE == 4) && PW_STRIP_ROLL_START)) {
#pragma unroll 1
          for (int s = 0; s < 4; s++) step<MODE>(kStripBlock * q + j0 + s, j0 + s, m, x4, s);
        } else {
#pragma unroll
          for (int s = 0; s < 4; s++) step<MODE>(kStripBlock * q + j0 + s, j0 + s, m, x4, s);
        }
      }
      mw[hb] = m;
    }
  }
  template <int MODE, bool NOIN = false>
  PW_FN void block(int q) {
#pragma unroll
    for (int d = 0; d < 8; d++) mwin[d] = mnext[d];
    uint32_t mw[4];
    int32_t kb0 = 0;
    if (TRACK && (MODE == 0 || MODE == 3 || MODE == 4)) {
      const int32_t floor25 = -(1 << 25);
      kb0 = (int32_t)(((uint32_t)(best > floor25 ? best : floor25) << 5) | 31u);   // an equal score later in the row loses
      kbest = kb0; ksnap = kb0;
    }
    sub_block<MODE, 0, NOIN>(q, mw);
    sub_block<MODE, 1, NOIN>(q, mw);
    if (NSB == 4) {
      sub_block<MODE, 2, NOIN>(q, mw);
      sub_block<MODE, 3, NOIN>(q, mw);
    }
#pragma unroll
    for (int d = 0; d < 4; d++) mprev[d] = mw[d];
    mprev_q = q;
    if (TRACK && (MODE == 0 || MODE == 3 || MODE == 4)) {
      int32_t kb = kbest;
      if (MODE == 3) {
        // rows whose last cell lies in this block take the snapshot, rows that ended before it nothing
        const int k0 = kStripBlock * q;
        kb = kY < k0 ? kb0 : (kY < k0 + kStripBlock ? ksnap : kbest);
      }
      const bool ch = kb != kb0;
      best = ch ? (kb >> 5) : best;
      bestY = ch ? kStripBlock * q + (31 - (kb & 31)) - lane : bestY;
    }
  }

  PW_FN void run(int w_, bool cross_in_, bool cross_out_) {
    lane = P::lane();
    w = w_;
    cross_in = cross_in_; cross_out = cross_out_;
    x = 64 * w + lane;
    const uint8_t* oseq = a.arena + a.o_off;
    mseq = a.arena + a.m_off;
    const int oi = x - 1 < 0 ? 0 : (x - 1 > a.X - 1 ? (a.X > 0 ? a.X - 1 : 0) : x - 1);
    oc = (uint32_t)P::in_vgpr((int32_t)oseq[oi]);      // opaque: a compare known to be 8 bits wide is not folded into a byte select
    rowreg = 0;
    // (made opaque at once: a loaded value whose first use sits in the steady loop would put the compiler's wait for it --
    //  s_waitcnt vmcnt(0), a drain of this wavefront's stores -- INTO the loop, once per block: 17.3 -> 18.9 ms)
    if (BROW) rowreg = (uint32_t)P::in_vgpr((int32_t)P::flag_load(a.ctl + kStripRows + (oc & 3u)));
    w0 = 0; w1 = 0; wpend = 0; wshift = 3u - ((uint32_t)lane & 3u); wfrom = (uint32_t)((lane - 4) & 63);
    kbest = 0; mprev_q = -1; bqv = NEG; ksnap = 0; kY = a.Y + lane;
    Hout = NEG; Uout = NEG; Lo = NEG; Hdiag = NEG; best = NEG; bestY = 0; hlast = NEG;
    gP = 0; cH = NEG; cU = NEG;
    we0 = lane - AH;
    npolls = 0; nspins = 0;
    wtag = (unsigned)we0 < (unsigned)SUB ? (a.epoch << 8) : 0u;
    wm255 = (unsigned)we0 < (unsigned)SUB ? 255u : 0u;
    vmatch = P::in_vgpr(a.match); vmis = P::in_vgpr(a.mismatch); vge = P::in_vgpr(a.ge); vgego = P::in_vgpr(a.ge + a.go);
    // steady blocks hold no first-row / first-column cell except row 0 itself
    b0 = (a.brule == BRULE_ANY || (a.brule == BRULE_EDGE && x == 0)) ? 0 : NEG;
    // ... and the first cell of a row (y == 0): the table edge, or the origin for row 0
    bfirst = (a.brule == BRULE_ANY || a.brule == BRULE_EDGE || x == 0) ? 0 : NEG;
    fin = w > 0 ? a.fifo + (uint64_t)(w - 1) * (uint64_t)a.fifo_pitch : nullptr;
    // (the last strip writes its row too -- the FIFO has a row per strip -- and nobody reads it)
    fout = a.fifo + (uint64_t)w * (uint64_t)a.fifo_pitch;
    load_letters(0, mnext);
    stamp(1);
    if (fin != nullptr) {
      if (AH == 0) {
        // the first hand-over (block 0) waits for sub-chunk 0 like any other
        load_sub<1>(0, 0);
      } else {
        // sub-chunks 0 and 1 in ONE poll (granule = lane): once both are there they go straight into the feeders
        const bool need = lane < 2 * SUB && lane <= a.Y;
        uint64_t t = need ? P::fifo_load(fin + lane) : 0;
        int spins = 0;
        while (!P::all(!need || (uint32_t)(t >> 32) == tag_of(lane))) {
          if (++spins > a.spin_limit || ((spins & 63) == 0 && P::uniform((int32_t)P::flag_poll(a.ctl + kStripAbort)) != 0)) give_up();
          P::sleep();
          if (need) t = (cross_in || (spins & 3) == 0) ? P::fifo_poll(fin + lane) : P::fifo_poll_local(fin + lane);
        }
        merge_value(t, 0, 0);
        load_sub<1>(1, SUB);
      }
    }
    if (AH != 0) stamp(2);
    // steady: every lane holds an in-table cell on every step of the block and none its first or last one: blocks
    // 2 .. q_end - 1.  They get a loop of their own, so that nothing another kind of block needs is carried or updated in it.
    // Their hand-overs are not range-checked: every column they touch (up to k0 + 47) must be one the strip above writes --
    // it writes the virtual columns up to 32 nkq - 64 too (strip_fifo_pitch) -- and writes before its last flush: q <= nkq - 4.
    const int q_last_cell = a.Y >= kStripBlock ? (a.Y - kStripBlock) / kStripBlock + 1 : 0;     // first q with k0 + 31 >= Y
    // (... and only in tables of more than 128 columns, where the strip above -- all of whose rows are in the table -- writes
    //  its virtual columns too; narrower tables run on the general code throughout)
    const int q_end = a.Y > 128 ? (q_last_cell < a.nkq - 3 ? q_last_cell : a.nkq - 3) : 0;
    // every row of the strip is in the table (and the first two blocks' hand-overs stay below column 96)
    const bool whole = a.Y > 128 && 64 * w + 63 <= a.X;
    for (int q = 0; q < a.nkq; q++) {
      const int k0 = kStripBlock * q;
      if (q == 2) stamp(3);
      if (q == 3) stamp(4);
      if (q >= 2 && q < q_end) {
        if (w == 0) {
          for (; q < q_end; q++) block<0, true>(q);
        } else {
          for (; q < q_end; q++) {
            if (__builtin_expect(a.stamps != nullptr, 0) && q == 3) stamp(4);
            block<0, false>(q);
          }
        }
        q--;
        stamp(6);                                                       // first block past the steady ones
        continue;
      }
      const bool starting = k0 < 64 && whole;
      const bool ending = k0 >= 64 && whole;
      if (starting && w > 0) block<4>(q);
      else if (starting) block<1>(q);
      else if (ending && w > 0) block<3>(q);
      else if (ending) block<3, true>(q);
      else block<2>(q);
    }
    // the last SUB steps' cells: a whole strip writes the virtual columns too (the end blocks of the strip below, whole as
    // well, do not range-check), any other one the columns up to Y
    if (whole && PW_STRIP_FAST) flush_fast(kStripBlock * a.nkq - SUB);
    else flush_out(kStripBlock * a.nkq - SUB);
    store_masks();
    stamp(5);
    if (a.stamps != nullptr && lane == 0) a.stamps[(uint64_t)w * 16 + 15] = ((uint64_t)(uint32_t)nspins << 32) | (uint32_t)npolls;
    finish();
  }

  // The strip's candidate for the end cell: (score desc, scan rank asc) over its rows (_std_find_optimal,
  // _pw_internals.c:303-360).
  PW_FN static uint64_t rank_of(int endrule, int X, int Y, int cx, int cy) {
    if (endrule == END_CORNER) return 0;
    if (endrule == END_STD_OVERLAP) return cx < X ? (uint64_t)(uint32_t)cx : (uint64_t)(uint32_t)(X + cy);
    return (uint64_t)(uint32_t)cx * (uint64_t)(uint32_t)(Y + 1) + (uint64_t)(uint32_t)cy;
  }
  PW_FN void finish() {
    const int endrule = a.endrule;
    int32_t cs = NEG; int cx = x, cy = a.Y; bool have = false;
    if (x <= a.X) {
      if (endrule == END_CORNER) { have = x == a.X; cs = hlast; }
      else if (endrule == END_STD_OVERLAP) {
        // last column (x < X): cell (x, Y); last row: its first best cell
        if (x < a.X) { have = true; cs = hlast; }
        else { have = true; cs = best; cy = bestY; }
      } else { have = true; cs = best; cy = bestY; }
    }
    uint64_t ck = rank_of(endrule, a.X, a.Y, cx, cy);
    int hv = have ? 1 : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const int32_t os = P::shfl_xor(cs, off);
      const uint32_t klo = (uint32_t)P::shfl_xor((int32_t)(uint32_t)ck, off), khi = (uint32_t)P::shfl_xor((int32_t)(uint32_t)(ck >> 32), off);
      const uint64_t ok_ = ((uint64_t)khi << 32) | klo;
      const int ox = P::shfl_xor(cx, off), oy = P::shfl_xor(cy, off), oh = P::shfl_xor(hv, off);
      const bool take = oh && (!hv || os > cs || (os == cs && ok_ < ck));
      if (take) { cs = os; ck = ok_; cx = ox; cy = oy; hv = 1; }
    }
    if (lane == 0) {
      StripBest sb; sb.score = cs; sb.x = cx; sb.y = cy; sb.have = hv;
      a.sbest[w] = sb;
    }
  }
};

// After every strip has finished: the pair's end cell from the per-strip candidates (one wavefront).
template <class P>
PW_FN void strip_reduce(const StripParams& a) {
  const int lane = P::lane();
  int32_t cs = -(1 << 28); uint64_t ck = ~(uint64_t)0; int cx = -1, cy = -1, hv = 0;
  for (int w = lane; w < a.nstrips; w += 64) {
    const StripBest sb = a.sbest[w];
    if (!sb.have) continue;
    const uint64_t k = StripFill<P, true>::rank_of(a.endrule, a.X, a.Y, sb.x, sb.y);
    if (!hv || sb.score > cs || (sb.score == cs && k < ck)) { cs = sb.score; ck = k; cx = sb.x; cy = sb.y; hv = 1; }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const int32_t os = P::shfl_xor(cs, off);
    const uint32_t klo = (uint32_t)P::shfl_xor((int32_t)(uint32_t)ck, off), khi = (uint32_t)P::shfl_xor((int32_t)(uint32_t)(ck >> 32), off);
    const uint64_t ok_ = ((uint64_t)khi << 32) | klo;
    const int ox = P::shfl_xor(cx, off), oy = P::shfl_xor(cy, off), oh = P::shfl_xor(hv, off);
    const bool take = oh && (!hv || os > cs || (os == cs && ok_ < ck));
    if (take) { cs = os; ck = ok_; cx = ox; cy = oy; hv = 1; }
  }
  if (lane == 0) {
    Result r;
    r.score = (double)cs * a.score_mul; r.opt_i = cx; r.opt_j = cy;          // standard mode: table coordinates are (x, y)
    r.origin_idx = 0; r.mutant_idx = 0; r.tx_len = 0; r.status = 0;
    // LOCAL / START_ANCHORED start from the score of cell (0,0), i.e. 0 (_pw_internals.c:342)
    if (!hv || (a.endrule == END_STD_LOCAL && !(cs > 0))) { r.opt_i = -1; r.opt_j = -1; r.score = 0.0; }
    if (P::flag_load(a.ctl + kStripAbort) != 0u) { r.opt_i = -1; r.opt_j = -1; r.score = 0.0; r.status = ST_BADPATH; }
    *a.result = r;
  }
}

// ---- traceback over the strip layout: one wavefront per pair ------------------------------------------------------
// A walk is a chain of dependent steps, but in an alignment of similar sequences nine steps out of ten continue a
// run of diagonal moves, and a run can be taken in ONE step: the wave keeps a window of the mask plane on chip (LDS:
// all 64 rows of kWalkStrips strips x 160 steps each = 20 KB), lane j looks at the cell j moves up the diagonal from the
// current one, a ballot finds how many of them in a row are "pure M" (none of B, D, I kept: the first kept op is M
// whatever led here), and the whole run is written with one store instruction.  Only the cells that break a run go
// through the scalar predecessor rule (pw_wave.h, trace_walk).  The next window (the strips above, along the same
// diagonal) is loaded while the current one is being walked.
constexpr int kWalkGroups = 5;                        // 32-step groups per strip of the window
#ifndef PW_WALK_STRIPS
#define PW_WALK_STRIPS 4
#endif
constexpr int kWalkStrips = PW_WALK_STRIPS;           // strips per window
constexpr int kWalkWinWords = kWalkStrips * kWalkGroups * 64 * 4;   // dwords of LDS

template <class P>
PW_FN void strip_walk(const StripTraceParams& p, uint32_t* win) {
  const int lane = P::lane();
  Result r = *p.result;
  if (r.status & ST_BADPATH) return;
  const int ei = P::uniform(p.ends ? p.ends[0] : r.opt_i), ej = P::uniform(p.ends ? p.ends[1] : r.opt_j);
  if (ei < 0 || ej < 0) {
    if (lane == 0) { r.tx_len = 0; r.status = 0; *p.result = r; }
    return;
  }
  const int gosign = p.gosign, nkq = p.nkq;
  int x = ei, y = ej;
  int pos = p.tx_cap, nms = 0, bad = 0, prev = 3;
  uint8_t* tx = p.tx;
  // The window: kWalkStrips strips at once -- sub-window s is strip cw - s with its groups cgs[s] - (kWalkGroups - 1) .. cgs[s],
  // chosen along the diagonal through the cell the window was loaded for (a strip is crossed in at most 64 diagonal moves =
  // 128 steps; five groups hold that from any entry point).  Round 3: with ONE strip per window the latency of a window's
  // loads was exposed at every strip crossing -- config 3's 1563 crossings x 2.1 us were its whole 3.35 ms -- although the
  // next window was already being loaded: a window lasted one or two rounds of the walk.  Now a miss comes every kWalkStrips
  // strips, the runs of diagonal moves the lanes look along cross strip boundaries, and the NEXT kWalkStrips strips are in
  // flight meanwhile.  cgs[s] < 0: that strip is not there (above the table, or the chain left it).
  constexpr int NS = kWalkStrips;
  int cw = -1, cgs[NS];
  int pw = -1, pgs[NS];                 // the window being loaded ahead: strips pw - s
  U4 pre[NS][kWalkGroups];
#pragma unroll
  for (int s = 0; s < NS; s++) {
    cgs[s] = -1; pgs[s] = -1;
#pragma unroll
    for (int q = 0; q < kWalkGroups; q++) { pre[s][q].x = pre[s][q].y = pre[s][q].z = pre[s][q].w = 0; }
  }
  // the top groups of the NS strips the diagonal through (ex, ey) crosses, starting with the strip of (ex, ey) itself (top group
  // g0 given) or -- FROM_NEXT -- with the strip above it; returns the cell with which the diagonal leaves the last of them
  auto chain = [&](int ex, int ey, int w0, bool from_next, int g0, int (&out)[NS], int& lx, int& ly) {
    int cx_ = ex, cy_ = ey, cwv = w0;
#pragma unroll
    for (int s = 0; s < NS; s++) {
      if (s > 0 || from_next) {
        const int mi = (cx_ & 63) + 1;                       // diagonal moves to the strip above
        const bool ok = cwv > 0 && cy_ - mi >= 0 && cx_ >= 0;
        cx_ = ok ? cx_ - mi : -1; cy_ = ok ? cy_ - mi : -1; cwv = ok ? cwv - 1 : -1;
        out[s] = ok ? ((cy_ + 63) >> 5) : -1;               // the entry cell sits in row 63 of that strip: k = y + 63
      } else {
        out[s] = g0;
      }
    }
    lx = cx_; ly = cy_;
  };
  int nx_ = -1, ny_ = -1;               // the cell with which the diagonal leaves the window loaded ahead
  while (true) {
    if (x < 0 || y < 0 || x > p.X || y > p.Y) { bad = 1; break; }
    const int w = x >> 6, i = x & 63, k = y + i, g = k >> 5;
    const int s0 = cw - w;
    int cg0 = -1;
#pragma unroll
    for (int s = 0; s < NS; s++) cg0 = s0 == s ? cgs[s] : cg0;
    if (cw < 0 || s0 < 0 || s0 >= NS || cg0 < 0 || g > cg0 || g < cg0 - (kWalkGroups - 1)) {
      // ---- new window.  Use the one loaded ahead if its first strip holds the cell with at least two groups below it.
      const bool hit = w == pw && pgs[0] >= 0 && g <= pgs[0] && g >= pgs[0] - (kWalkGroups - 3);
      int lx = -1, ly = -1;
      cw = w;
      if (hit) {
#pragma unroll
        for (int s = 0; s < NS; s++) cgs[s] = pgs[s];
        lx = nx_; ly = ny_;
      } else {
        chain(x, y, w, false, g, cgs, lx, ly);
      }
      P::wave_sync();
#pragma unroll
      for (int s = 0; s < NS; s++) {
#pragma unroll
        for (int q = 0; q < kWalkGroups; q++) {
          U4 v = pre[s][q];
          if (!hit) {
            v.x = v.y = v.z = v.w = 0;
            if (cgs[s] >= 0 && cgs[s] - q >= 0) v = *(const U4*)(p.masks + strip_mask_index(nkq, cw - s, cgs[s] - q, lane));
          }
          uint32_t* dst = win + ((s * kWalkGroups + q) * 64 + lane) * 4;
          dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
      }
      P::wave_sync();
      // ---- load ahead: the NS strips above this window, along the same diagonal
      pw = cw - NS;
      int dummy = 0;
      if (pw >= 0 && lx >= 0 && ly >= 0 && cgs[NS - 1] >= 0) chain(lx, ly, cw - (NS - 1), true, dummy, pgs, nx_, ny_);
      else {
        pw = -1;
#pragma unroll
        for (int s = 0; s < NS; s++) pgs[s] = -1;
      }
#pragma unroll
      for (int s = 0; s < NS; s++) {
#pragma unroll
        for (int q = 0; q < kWalkGroups; q++) {
          pre[s][q].x = pre[s][q].y = pre[s][q].z = pre[s][q].w = 0;
          if (pw >= 0 && pgs[s] >= 0 && pgs[s] - q >= 0) pre[s][q] = *(const U4*)(p.masks + strip_mask_index(nkq, pw - s, pgs[s] - q, lane));
        }
      }
      continue;                          // (look the cell up again in the new window)
    }
    // ---- lane j looks at the cell j diagonal moves up from (x, y), in whichever strip of the window that is
    const int xj = x - lane, yj = y - lane;
    const int sj = cw - (xj >> 6), ij = xj & 63, kj = yj + ij;
    int cgj = -1;
#pragma unroll
    for (int s = 0; s < NS; s++) cgj = sj == s ? cgs[s] : cgj;
    const bool inwin = xj >= 0 && yj >= 0 && sj >= 0 && sj < NS && cgj >= 0 && (kj >> 5) <= cgj && (kj >> 5) >= cgj - (kWalkGroups - 1);
    uint32_t nibj = 0xfu;
    if (inwin) {
      const uint32_t word = win[((sj * kWalkGroups + (cgj - (kj >> 5))) * 64 + ij) * 4 + ((kj >> 3) & 3)];
      nibj = (word >> (4 * (7 - (kj & 7)))) & 15u;
    }
    const uint64_t pure = P::ballot(inwin && (nibj & 7u) == 0u);
    const uint64_t seen = P::ballot(inwin);
    const int run0 = (int)__builtin_ctzll(~pure | ((uint64_t)1 << 63));   // pure-M cells in a row, starting with this one
    const int lim = x < y ? x : y;                                  // a diagonal move needs x >= 1 and y >= 1
    int run = run0 < lim ? run0 : lim;
    run = run < pos ? run : pos;
    if (run > 0) {
      if (lane < run) tx[pos - 1 - lane] = (uint8_t)'X';
      pos -= run; nms += run; x -= run; y -= run; prev = 3;
      // the cell that ends the run is lane `run`'s: if this window shows it, take its op in the same round (after an M the
      // rule is "first kept op", and a cell that is not pure M keeps B, D or I)
      if (run == run0 && run0 < 63 && ((seen >> run) & 1u) && pos > 0) {      // (63 is the ballot's sentinel, not a break)
        const int op = pw_first_op(P::readlane(nibj, run));
        if (op == 0) break;
        if (lane == 0) tx[pos - 1] = (uint8_t)(op == 1 ? 'D' : 'I');
        pos -= 1;
        x -= (op != 2); y -= (op != 1);
        prev = op;
      }
      continue;
    }
    // ---- the current cell breaks the run: the predecessor rule on its mask
    const uint32_t pm = P::readlane(nibj, 0);
    int op;
    if (prev == 3 || gosign == 0) op = pw_first_op(pm);
    else if (gosign < 0) op = (pm & (1u << prev)) ? prev : pw_first_op(pm);
    else { const uint32_t others = pm & ~(1u << prev); op = others ? pw_first_op(others) : prev; }
    if (op == 0 || pos <= 0) break;
    if (lane == 0) tx[pos - 1] = (uint8_t)(op == 3 ? 'X' : (op == 1 ? 'D' : 'I'));
    pos -= 1;
    nms += (op == 3);
    x -= (op != 2); y -= (op != 1);
    prev = op;
  }
  if (lane == 0) {
    r.origin_idx = x; r.mutant_idx = y;
    r.tx_len = p.tx_cap - pos;
    r.status = ST_TRACED | (r.tx_len == 0 ? ST_EMPTY : 0) | ((x + y + nms <= 0) ? ST_PANICK : 0) | (bad ? ST_BADPATH : 0);
    *p.result = r;
  }
}

}  // namespace pw
#endif
