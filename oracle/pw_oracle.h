/* oracle/pw_oracle.h -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * CPU restatement of the reference pwlib dynamic-programming path
 * (reference: biseqt/pwlib/pw.c, biseqt/pwlib/_pw_internals.c, biseqt/pwlib/pwlib.h).
 *
 * Parity status: PINNED.  oracle/tests compare this restatement, field by field, against
 *   (1) the reference library itself compiled from /root/reference (oracle/_ref/pwlib_ref.so,
 *       recipe oracle/Makefile) on random problems over all 7 standard + 3 banded alignment types,
 *   (2) the golden vectors under tests/golden/ (generated from that compiled reference by
 *       tests/golden/make_golden.py), which include every known-answer case of the reference's own
 *       tests/test_pw.py:33-103 and the pw.py:11-21 docstring example.
 *
 * State per cell is (H, ordered tie set) exactly as SURVEY.md section 8a derives it: the reference keeps
 * every candidate whose score == the cell maximum in the fixed order B, D, I, M (pw.c:77-80,92-108);
 * all kept choices share one score, so a 4-bit mask + one double is the whole cell.
 */
#ifndef PW_ORACLE_H
#define PW_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* mask bits, low to high = the reference's candidate order B, D, I, M (pw.c:77-80) */
#define PWO_B 1
#define PWO_D 2
#define PWO_I 4
#define PWO_M 8

typedef struct {
  int mode;            /* 0 STD_MODE, 1 BANDED_MODE            (pwlib.h:30-33)   */
  int type;            /* std_alntype 0..6 / banded_alntype 0..2 (pwlib.h:39-65)  */
  int X, Y;            /* frame lengths: origin_range.j-.i, mutant_range.j-.i      */
  const int *origin;   /* already offset to the frame start (origin + origin_range.i) */
  const int *mutant;   /* likewise */
  int L;               /* alphabet size */
  const double *subst; /* row-major L x L: subst[o*L + m] = subst_scores[o][m]      */
  double go, ge;       /* gap open / extend scores */
  int dmin, dmax;      /* banded only: diag_range as the caller gave it (unclamped)  */
  int max_new_mins;    /* must be <= 0 (the reference reads uninitialised memory otherwise) */
} pwo_problem;

typedef struct {
  int init_rc;         /* what dptable_init returns: 0 or -1  (pw.c:10-26)            */
  int dmin_c, dmax_c;  /* band after the clamp of _pw_internals.c:29-36                */
  int num_rows;        /* dptable.num_rows                                              */
  long long cells;     /* sum of row_lens = cells the reference allocates               */
  int opt_i, opt_j;    /* dptable_solve's return value, table coordinates, or -1,-1     */
  double score;        /* cells[opt].choices[0].score                                   */
  int would_panick;    /* 1 iff the reference's dptable_traceback would exit(1) (pw.c:132-134) */
  int tb_null;         /* 1 iff dptable_traceback returns NULL (empty transcript, pw.c:135-138) */
  int origin_idx, mutant_idx; /* alignment start RELATIVE TO THE FRAME START (caller adds range.i) */
  int tx_len;
  char *transcript;    /* malloc'd, NUL-terminated; caller frees with pwo_free_result           */
  int maskrule_ok;     /* 1 iff the mask-only predecessor rule (see pw_oracle.c) reproduces the explicit
                          base chain of this traceback -- pins the rule the device traceback relies on */
} pwo_result;

/* Solve + end-cell search + traceback from the optimal cell.
 * Hout / maskout, if non-NULL, receive the table in the reference's own row layout
 * (STD: row x, column y, pitch Y+1; BANDED: row d-dmin_c, column a, rows packed back to back in
 * row order) -- `cells` entries each.  Empty cells get mask 0 and H = NaN.
 * Returns 0, or -2 for an unsupported problem (max_new_mins > 0). */
int pwo_solve(const pwo_problem *p, pwo_result *r, double *Hout, unsigned char *maskout);

/* Fill only (no end-cell search/traceback); for CPU-baseline timing of the recurrence alone. */
long long pwo_cells(const pwo_problem *p);

void pwo_free_result(pwo_result *r);

#ifdef __cplusplus
}
#endif
#endif
