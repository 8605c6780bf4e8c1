// pw_types.h -- plain-data structures shared by the host planner, the HIP kernels and the
// lane-level CPU emulator used by the tests.  No HIP or torch types here.
#ifndef PW_TYPES_H
#define PW_TYPES_H

#include <stdint.h>

namespace pw {

// Alignment modes / types: numeric values are the reference's enums (pwlib.h:30-33, 39-54, 60-65).
enum { STD_MODE = 0, BANDED_MODE = 1 };
enum { GLOBAL = 0, LOCAL, START_ANCHORED, END_ANCHORED, OVERLAP, START_ANCHORED_OVERLAP,
       END_ANCHORED_OVERLAP };
enum { B_GLOBAL = 0, B_LOCAL, B_OVERLAP };

// Tie-mask bits in the reference's candidate order B, D, I, M (pw.c:77-80): "first kept choice" is
// the lowest set bit.
enum { MB = 1, MD = 2, MI = 4, MM = 8 };

// Where may an alignment begin (_alnchoice_B, _pw_internals.c:161-209).
enum { BRULE_ORIGIN = 0,   // only cell (0,0)
       BRULE_EDGE = 1,     // x == 0 or y == 0
       BRULE_ANY = 2 };    // anywhere

// Which cells may end an alignment and in which order the reference scans them
// (_std_find_optimal :303-360, _banded_find_optimal :364-414).  "first strict maximum in scan
// order" == max score, ties broken by the smallest scan rank.
enum { END_CORNER = 0,        // cell (X,Y)
       END_STD_OVERLAP = 1,   // last row U last column, row-major, init -INT_MAX
       END_BANDED_OVERLAP = 2,// last cell of each diagonal, diagonals ascending, init -INT_MAX
       END_STD_LOCAL = 3,     // all cells, row-major, must beat the score of cell (0,0) (= 0)
       END_BANDED_LOCAL = 4 };// all cells, diagonal-major then along the diagonal, init -INT_MAX

// One alignment problem as the fill kernel sees it.  Everything is expressed in "band coordinates":
// a standard-mode table is the band dmin = -Y, dmax = X.  Diagonal index dd = d - dmin, d = x - y;
// anti-diagonal s = x + y; step t = s - s0 with s0 == dmin (mod 2) so that diagonal slot dd holds a
// cell exactly on the steps t == dd (mod 2).
struct PairDesc {
  uint64_t o_off;      // byte offset of the origin frame in the sequence arena (1 B / base)
  uint64_t m_off;      // same for the mutant frame
  uint64_t mask_off;   // dword offset of this pair's tie-mask plane in the mask workspace
  uint64_t h_off;      // element offset of this pair's score plane (only when scores are dumped)
  uint64_t tx_off;     // byte offset of this pair's transcript slot
  int32_t X, Y;        // frame lengths
  int32_t dmin;        // lowest diagonal of the (clamped) band
  int32_t ndiag;       // number of diagonals, 1 + dmax - dmin
  int32_t s0;          // base anti-diagonal
  int32_t nblocks;     // number of 16-step blocks to run
  int32_t steady_b0;   // blocks [steady_b0, steady_b1) are "steady": every in-band diagonal holds an
  int32_t steady_b1;   //   in-table cell on each of its steps, so the unpredicated body may run
  int32_t h_pitch;     // row pitch (elements) of the score plane
  int32_t tx_cap;      // bytes in the transcript slot (>= X + Y + 1)
  int32_t bk;          // diagonals per lane of the fill kernel that owns this pair (mask plane layout)
  int32_t solvable;    // 0: dptable_init fails for this pair (or its table is empty); kernels skip it
  int32_t nl;          // lanes that hold this pair = row length of its mask plane (64 unless lane-packed)
  int32_t layout;      // mask plane layout: 0 diagonal-major (mask_word_index), 1 row strips (pw_strip.h)
};                     // 96 bytes

// One wavefront of the lane-packed fill kernel: `count` pairs side by side, `nl` lanes each.
struct WaveDesc {
  int32_t first;       // index into the launch order of the wave's first pair
  int32_t count;       // pairs in this wave (<= 64 / nl)
  int32_t nblocks;     // max over its pairs
  int32_t steady_b0;   // blocks [steady_b0, steady_b1) are steady for ALL of its pairs
  int32_t steady_b1;
  int32_t nl;
  int32_t pad_[2];
};                     // 32 bytes

// Per-pair result record (device and host; identical to the public pw_result of include/pw_batch.h).
// 32 bytes, the unit the multi-GPU gather moves.
struct Result {
  double score;        // cells[opt].choices[0].score
  int32_t opt_i, opt_j;// optimal end cell in the reference's TABLE coordinates -- (x, y) in standard mode,
                       // (d - dmin, a) in banded mode -- exactly what dptable_solve returns; -1,-1: none
  int32_t origin_idx;  // alignment start relative to the frame start (traceback output)
  int32_t mutant_idx;
  int32_t tx_len;      // transcript length; the ops sit right-aligned in the pair's transcript slot
  int32_t status;      // bit 0: traceback ran; bit 1: empty transcript (reference returns NULL,
                       // pw.c:135-138); bit 2: the reference would exit(1) here (pw.c:132-134)
};

enum { ST_TRACED = 1, ST_EMPTY = 2, ST_PANICK = 4, ST_BADPATH = 8 };

// Uniform parameters of one fill launch.  T is the score type (int32_t or double).
template <typename T>
struct FillParams {
  const PairDesc* pairs;
  const int32_t* order;       // launch order (longest first) or null
  const WaveDesc* waves;      // lane-packed kernels: one descriptor per wavefront
  // tiled single-pair kernel (K2b): per-diagonal state handed from one time block to the next, laid out as
  // [5][st_pitch]: H, U, L, best, bestT (all as T); the launch covers blocks [tile_b0, tile_b0 + tile_nb)
  const T* st_in;
  T* st_out;
  int32_t st_pitch;
  int32_t tile_b0, tile_nb;
  const uint8_t* arena;
  uint32_t* masks;
  T* hdump;                   // score plane or null
  Result* results;
  const T* subst;             // L x L row-major (generic kernels only)
  int32_t npairs;
  int32_t L;
  int32_t brule;
  int32_t endrule;
  int32_t banded;             // results are reported in (d - dmin, a) table coordinates
  T match, mismatch;          // simple scoring (fast kernels)
  T go, ge;
  // Dyadic scaling (host planner): scores that are multiples of 2^-k are held times 2^k by the integer kernels; the
  // reported score is the kernel's value times score_mul = 2^-k (exact).  1.0 otherwise.
  double score_mul;
  // Packed kernels with a substitution matrix (WaveFill16<.., MAT = true>; alphabets of up to 4 letters): row o holds the
  // four bytes scale * (subst[o][m] - min(subst)), m = 0 .. 3 from the low byte up; mat_bias = scale * -min(subst)
  uint32_t mat_rows[4];
  int32_t mat_bias;
};

struct TraceParams {
  const PairDesc* pairs;
  const uint8_t* arena;
  const uint32_t* masks;
  Result* results;
  uint8_t* transcripts;
  int32_t npairs;
  int32_t gosign;             // sign of the gap-open score: -1, 0, +1
  int32_t banded;             // table coordinates are (d - dmin, a)
  const int32_t* ends;        // optional explicit end cells (table i,j pairs) overriding results[].opt_*
  int32_t fix_segments;       // wavefronts per transcript in the fix-up pass (0 / 1: one)
};

// ---- mask plane addressing (shared by fill, traceback and tests) -------------------------------
// Per pair the plane is [nblocks][BK / G][nl lanes][G] dwords, G = min(4, BK): one 16-byte
// (or 8-byte for BK = 2) store per lane and group, nl * 16 B contiguous per group and store instruction.
// A dword holds the 8 cells one diagonal slot visits in a 16-step block, first cell in the top nibble.
static inline int mask_group(int bk) { return bk < 4 ? bk : 4; }
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline uint64_t mask_word_index(int bk, int nl, int b, int lane, int j) {
  const int G = bk < 4 ? bk : 4;
  return ((uint64_t)((uint64_t)b * (bk / G) + (j / G)) * nl + lane) * G + (j % G);
}

}  // namespace pw
#endif
