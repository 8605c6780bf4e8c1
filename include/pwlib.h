// pwlib.h -- C ABI of the MI355X-native pairwise alignment library (libpwlib / pwlib.so).
//
// DROP-IN BOUNDARY.  This header declares exactly the surface the reference's only FFI binds
// (reference biseqt/pw.py:56-66 dlopens biseqt/pwlib/pwlib.so and cdef()s biseqt/pwlib/pwlib.h minus
// the lines starting with "#define"): the same type names, field names, field order, enum values and
// the same four functions, so `ffi.cdef(this file)` + `ffi.dlopen(pwlib.so)` in the reference's pw.py
// works unchanged.  The file is cdef-clean on purpose: no #include, no #ifdef, no multi-line macros.
//
//   type / function            replaces (reference biseqt/pwlib/pwlib.h)
//   intpair                    :15-18
//   alnmode                    :30-33      STD_MODE = 0, BANDED_MODE = 1
//   std_alntype                :39-54      GLOBAL .. END_ANCHORED_OVERLAP = 0 .. 6
//   banded_alntype             :60-65      B_GLOBAL = 0, B_LOCAL = 1, B_OVERLAP = 2
//   alnscores                  :71-78
//   alnframe                   :84-89
//   std_alnparams              :95-97
//   banded_alnparams           :103-107
//   alnprob                    :113-123
//   alnchoice                  :141-152
//   dpcell                     :158-163
//   dptable                    :168-173
//   alignment                  :178-187
//   dptable_init               :204   (implementation pw.c:10-26)
//   dptable_free               :211   (pw.c:28-45)
//   dptable_traceback          :227   (pw.c:116-151)
//   dptable_solve              :242   (pw.c:47-114)
//
// What is different behind the boundary: the DP table is filled, searched and traced back on the GPU
// (hand-written HIP kernels for gfx950); host memory holds only what the reference's callers read
// directly -- see "materialisation" below.  The batch entry points the GPU path is really built for
// live in pw_batch.h and are exported by the same shared object.

typedef struct {
  int i; /* The "first" element. */
  int j; /* The "second" element. */
} intpair;

typedef enum {
  STD_MODE,    /* Standard alignment: the whole (X+1) x (Y+1) table. */
  BANDED_MODE, /* Banded alignment: only diagonals dmin..dmax. */
} alnmode;

typedef enum {
  GLOBAL,                 /* Needleman-Wunsch. */
  LOCAL,                  /* Smith-Waterman. */
  START_ANCHORED,         /* local, must start at the start of both frames. */
  END_ANCHORED,           /* local, must end at the end of both frames. */
  OVERLAP,                /* suffix-prefix alignments in either direction. */
  START_ANCHORED_OVERLAP, /* overlap starting at the start of both frames. */
  END_ANCHORED_OVERLAP,   /* overlap ending at the end of both frames. */
} std_alntype;

typedef enum {
  B_GLOBAL,  /* banded global. */
  B_LOCAL,   /* banded local. */
  B_OVERLAP, /* banded suffix-prefix. */
} banded_alntype;

typedef struct {
  double **subst_scores;   /* substitution score matrix, rows and columns in alphabet order. */
  double gap_open_score;   /* gap open score (0 for a linear gap model). */
  double gap_extend_score; /* gap extension score. */
} alnscores;

typedef struct {
  int* origin;          /* the "from" sequence as an array of letter indices. */
  int* mutant;          /* the "to" sequence as an array of letter indices. */
  intpair origin_range; /* vertical span of the frame, [min, max). */
  intpair mutant_range; /* horizontal span of the frame, [min, max). */
} alnframe;

typedef struct {
  std_alntype type;
} std_alnparams;

typedef struct {
  banded_alntype type;
  int dmin; /* the upper diagonal bounding the band; clamped IN PLACE by dptable_init like the reference. */
  int dmax; /* the lower diagonal bounding the band; clamped IN PLACE by dptable_init like the reference. */
} banded_alnparams;

typedef struct {
  alnframe *frame;
  alnscores *scores;
  int max_new_mins; /* only values <= 0 are accepted (the reference reads uninitialised memory otherwise). */
  alnmode mode;
  union {
    std_alnparams* std_params;
    banded_alnparams* banded_params;
  };
} alnprob;

typedef struct alnchoice {
  char op;                /* 'B', 'M', 'S', 'I' or 'D'. */
  double score;           /* score of the alignment ending with this choice. */
  struct alnchoice *base; /* always NULL here: the predecessor chain lives on the GPU as tie masks. */
  int mins_cd;
  int cur_min;
} alnchoice;

typedef struct {
  int num_choices;           /* 0, or 1 for every materialised cell. */
  struct alnchoice *choices; /* choices[0].score is what callers read. */
} dpcell;

typedef struct {
  dpcell** cells; /* the table in (i,j) (standard) or (d,a) (banded) coordinates; see materialisation. */
  int num_rows;
  int* row_lens;
  alnprob* prob;
} dptable;

typedef struct alignment {
  int origin_idx;   /* start of the alignment on the origin sequence (NOT relative to the frame). */
  int mutant_idx;   /* start of the alignment on the mutant sequence (NOT relative to the frame). */
  double score;
  char* transcript; /* NUL-terminated edit transcript over M, S, I, D. */
} alignment;

// dptable_init: computes table dimensions exactly as the reference does (clamps the band in place and
// prints the same messages to stdout, rejects infeasible banded-global problems), allocates
// num_rows / row_lens / the host cell rows (all cells empty).  Returns 0, or -1 on error -- also for
// problems this library does not support (max_new_mins > 0, letters >= 256, a band or table wider than
// 2^21 diagonals): it fails loudly on stderr, it never falls back to a CPU path.
int dptable_init(dptable* T);

// dptable_free: releases everything dptable_init / dptable_solve / dptable_traceback allocated for T,
// including the alignments returned by dptable_traceback (the reference leaks those) and row_lens in
// both modes (the reference leaks it in standard mode).  NULL-safe.
void dptable_free(dptable* T);

// dptable_traceback: walks back from table cell `end` on the GPU.  Returns a malloc'd alignment owned
// by T, or NULL if the transcript is empty.  Where the reference calls exit(1) (path starting at cell
// (0,0) without any M/S, pw.c:132-134) this prints the same message and returns NULL instead.
alignment* dptable_traceback(dptable* T, intpair end);

// dptable_solve: fills the table on the GPU and returns the optimal end cell in table coordinates, or
// {-1,-1}.  Materialisation: afterwards cells[opt.i][opt.j].choices[0] holds the optimal score
// (what pw.py:272 reads); in standard mode every cell's choices[0].score is filled in (what
// Aligner.table_scores, pw.py:278-285, reads) unless the environment variable PWLIB_NO_TABLE=1 is set.
intpair dptable_solve(dptable* T);
