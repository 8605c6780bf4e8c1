/* oracle/pw_oracle.c -- TEST INFRASTRUCTURE ONLY.  See pw_oracle.h for scope and parity status.
 *
 * A from-the-spec restatement of what the reference computes, NOT a copy of its
 * text: no per-cell heap objects, no pointer-linked choices.  A cell is (H, 4-bit ordered tie mask)
 * plus, for the oracle's own faithful traceback, the op of the predecessor choice each kept gap
 * choice is based on.  Every rule cites the reference line it restates.
 */
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "pw_oracle.h"

enum { STD_MODE = 0, BANDED_MODE = 1 };
enum { GLOBAL = 0, LOCAL, START_ANCHORED, END_ANCHORED, OVERLAP, START_ANCHORED_OVERLAP,
       END_ANCHORED_OVERLAP };
enum { B_GLOBAL = 0, B_LOCAL, B_OVERLAP };

typedef struct {
  const pwo_problem *p;
  int X, Y, dmin, dmax, num_rows;
  long long *row_off;   /* num_rows + 1 prefix offsets into the flat cell arrays */
  int *row_len;
  double *H;
  unsigned char *mask;  /* 0 = empty cell (num_choices == 0) */
  unsigned char *base;  /* bits 0-1: op index (0..3 = B,D,I,M) the kept D choice is based on;
                           bits 2-3: same for the kept I choice (_pw_internals.c:274-281) */
} tbl;

static int imax(int a, int b) { return a > b ? a : b; }

/* _std_table_init_dims (_pw_internals.c:8-20) and _banded_table_init_dims (:22-62) */
static int table_dims(tbl *t) {
  const pwo_problem *p = t->p;
  int i;
  t->X = p->X; t->Y = p->Y;
  if (p->mode == STD_MODE) {
    t->num_rows = t->X + 1;
    t->dmin = t->dmax = 0;
  } else {
    int dmin = p->dmin, dmax = p->dmax, dend = t->X - t->Y;
    if (dmax > t->X || dmin < -t->Y) {              /* clamp, :29-36 */
      dmax = dmax > t->X ? t->X : dmax;
      dmin = dmin < -t->Y ? -t->Y : dmin;
    }
    t->dmin = dmin; t->dmax = dmax;
    /* global alignments need both end points inside the band, :38-43 */
    if (p->type == B_GLOBAL && (dend > dmax || dend < dmin || (long long)dmax * dmin > 0)) return -1;
    t->num_rows = 1 + dmax - dmin;
    if (t->num_rows < 0) return -1;                  /* :46-49 */
  }
  t->row_len = (int *)malloc(sizeof(int) * (size_t)imax(t->num_rows, 1));
  t->row_off = (long long *)malloc(sizeof(long long) * (size_t)(t->num_rows + 1));
  t->row_off[0] = 0;
  for (i = 0; i < t->num_rows; i++) {
    if (p->mode == STD_MODE) t->row_len[i] = t->Y + 1;            /* :16-18 */
    else {
      int d = t->dmin + i;                                         /* :53-56 */
      t->row_len[i] = 1 + (d > 0 ? 0 : d) + (t->X - d > t->Y ? t->Y : t->X - d);
      if (t->row_len[i] <= 0) return -1;  /* the reference PANICKs here (:57-59); unreachable after the clamp */
    }
    t->row_off[i + 1] = t->row_off[i] + t->row_len[i];
  }
  return 0;
}

/* (x,y) -> flat cell index or -1 if outside the ragged table: _cellpos_from_xy (:87-98) followed by
 * _cellpos_valid (:76-85), which is what enforces the band. */
static long long cell_of(const tbl *t, int x, int y) {
  int i, j;
  if (t->p->mode == STD_MODE) { i = x; j = y; }
  else { i = x - y - t->dmin; j = x < y ? x : y; }
  if (i < 0 || j < 0 || i >= t->num_rows || j >= t->row_len[i]) return -1;
  return t->row_off[i] + j;
}

/* _alnchoice_B (:161-209): may an alignment begin at (x,y)? */
static int b_allowed(const pwo_problem *p, int x, int y) {
  if (p->mode == STD_MODE) {
    if (x == 0 && y == 0) return 1;
    if (p->type == LOCAL || p->type == END_ANCHORED) return 1;
    if ((p->type == OVERLAP || p->type == END_ANCHORED_OVERLAP) && (x == 0 || y == 0)) return 1;
    return 0;
  }
  if (p->type == B_GLOBAL && x == 0 && y == 0) return 1;
  if (p->type == B_OVERLAP && (x == 0 || y == 0)) return 1;
  if (p->type == B_LOCAL) return 1;
  return 0;
}

static const char OPCH[4] = {'B', 'D', 'I', 'M'};

/* _alnchoice_ID (:247-291): gap candidate `op` (1 = D, 2 = I) out of predecessor cell c.
 * Walk the predecessor's kept choices in their stored order; each offers
 * (score + ge) [+ go if its op differs]; the FIRST strict maximum wins (:274), starting from
 * -INT_MAX (:267).  Returns the candidate score, *basek = op index of the chosen predecessor choice. */
static double gap_candidate(const tbl *t, long long c, int op, int *basek) {
  double max_score = -INT_MAX, score;
  int k, first = 1;
  *basek = -1;
  for (k = 0; k < 4; k++) {
    if (!(t->mask[c] & (1 << k))) continue;
    if (first) { *basek = k; first = 0; }          /* base_idx = 0 until something beats -INT_MAX */
    score = t->H[c] + t->p->ge;
    if (k != op) score += t->p->go;
    if (score > max_score) { max_score = score; *basek = k; }
  }
  return max_score;
}

/* dptable_solve's fill loop (pw.c:59-110) in the reference's own (x outer, y inner) order with
 * the limits of _xlim/_ylim (_pw_internals.c:116-153). */
static void fill(tbl *t) {
  const pwo_problem *p = t->p;
  int x, y, x0, x1, y0, y1;
  if (p->mode == STD_MODE) { x0 = 0; x1 = t->X + 1; }
  else { x0 = t->dmin > 0 ? t->dmin : 0; x1 = 1 + (t->X > t->Y + t->dmax ? t->Y + t->dmax : t->X); }
  for (x = x0; x < x1; x++) {
    if (p->mode == STD_MODE) { y0 = 0; y1 = t->Y + 1; }
    else { y0 = x - t->dmax > 0 ? x - t->dmax : 0; y1 = 1 + (t->Y > x - t->dmin ? x - t->dmin : t->Y); }
    for (y = y0; y < y1; y++) {
      double cand[4]; int have[4] = {0, 0, 0, 0}; int bk[4] = {0, 0, 0, 0};
      long long c = cell_of(t, x, y), pc;
      int k, any = 0; double best = 0; unsigned char m = 0;
      if (c < 0) continue;   /* cannot happen: the limits are exactly the in-table cells */
      /* candidates in the order B, D, I, M (pw.c:77-80) */
      if (b_allowed(p, x, y)) { cand[0] = 0.0; have[0] = 1; }
      pc = cell_of(t, x - 1, y);                                   /* D: from (x-1, y) */
      if (pc >= 0 && t->mask[pc]) { cand[1] = gap_candidate(t, pc, 1, &bk[1]); have[1] = 1; }
      pc = cell_of(t, x, y - 1);                                   /* I: from (x, y-1) */
      if (pc >= 0 && t->mask[pc]) { cand[2] = gap_candidate(t, pc, 2, &bk[2]); have[2] = 1; }
      pc = cell_of(t, x - 1, y - 1);                               /* M/S: _alnchoice_M (:217-245) */
      if (pc >= 0 && t->mask[pc]) {
        cand[3] = t->H[pc] + p->subst[p->origin[x - 1] * p->L + p->mutant[y - 1]];
        have[3] = 1;
      }
      /* keep every candidate equal to the maximum, order preserved (pw.c:92-108) */
      for (k = 0; k < 4; k++) {
        if (!have[k]) continue;
        if (!any) { best = cand[k]; m = (unsigned char)(1 << k); any = 1; }
        else if (cand[k] == best) m |= (unsigned char)(1 << k);
        else if (cand[k] > best) { best = cand[k]; m = (unsigned char)(1 << k); }
      }
      t->mask[c] = m;                       /* 0 <=> num_choices == 0 (pw.c:84-87) */
      t->H[c] = any ? best : NAN;
      t->base[c] = (unsigned char)((bk[1] & 3) | ((bk[2] & 3) << 2));
    }
  }
}

/* _std_find_optimal (:303-360) / _banded_find_optimal (:364-414) */
static void find_optimal(const tbl *t, int *oi, int *oj) {
  const pwo_problem *p = t->p;
  int i, j; double max; long long c;
  *oi = -1; *oj = -1;
  if (p->mode == STD_MODE) {
    int type = p->type;
    if (type == GLOBAL || type == END_ANCHORED || type == END_ANCHORED_OVERLAP) {
      c = t->row_off[t->X] + t->Y;
      if (t->mask[c]) { *oi = t->X; *oj = t->Y; }
    } else if (type == OVERLAP || type == START_ANCHORED_OVERLAP) {
      max = -INT_MAX;                                    /* row-major over last row U last column */
      for (i = 0; i <= t->X; i++) for (j = 0; j <= t->Y; j++) {
        if (i != t->X && j != t->Y) continue;
        c = t->row_off[i] + j;
        if (!t->mask[c]) continue;
        if (t->H[c] > max) { *oi = i; *oj = j; max = t->H[c]; }
      }
    } else { /* LOCAL, START_ANCHORED: init = score of cell (0,0) (:342) */
      max = t->H[0];
      for (i = 0; i <= t->X; i++) for (j = 0; j <= t->Y; j++) {
        c = t->row_off[i] + j;
        if (!t->mask[c]) continue;
        if (t->H[c] > max) { *oi = i; *oj = j; max = t->H[c]; }
      }
    }
    if (*oi == -1 || *oj == -1) { *oi = -1; *oj = -1; }
    return;
  }
  if (p->type == B_GLOBAL) {
    i = t->X - t->Y - t->dmin; j = t->X < t->Y ? t->X : t->Y;
    if (t->mask[t->row_off[i] + j]) { *oi = i; *oj = j; }
  } else if (p->type == B_OVERLAP) {                      /* last cell of each diagonal, rows ascending */
    max = -INT_MAX;
    for (i = 0; i < t->num_rows; i++) {
      j = t->row_len[i] - 1; c = t->row_off[i] + j;
      if (t->mask[c] && t->H[c] > max) { max = t->H[c]; *oi = i; *oj = j; }
    }
  } else if (p->type == B_LOCAL) {                        /* diagonal-major, init -INT_MAX (:397) */
    max = -INT_MAX;
    for (i = 0; i < t->num_rows; i++) for (j = 0; j < t->row_len[i]; j++) {
      c = t->row_off[i] + j;
      if (t->mask[c] && t->H[c] > max) { max = t->H[c]; *oi = i; *oj = j; }
    }
  }
}

static int first_op(unsigned char m) { int k; for (k = 0; k < 4; k++) if (m & (1 << k)) return k; return -1; }

/* dptable_traceback (pw.c:116-151) from table cell (ei, ej). */
static void traceback(const tbl *t, int ei, int ej, pwo_result *r) {
  const pwo_problem *p = t->p;
  int x, y, op, n = 0, nms = 0, cap;
  char *rev;
  long long c;
  if (p->mode == STD_MODE) { x = ei; y = ej; }
  else { int d = ei + t->dmin; x = ej + (d > 0 ? d : 0); y = ej - (d > 0 ? 0 : d); }  /* _xy_from_cellpos :100-114 */
  cap = x + y + 1;
  rev = (char *)malloc((size_t)cap + 1);
  c = cell_of(t, x, y);
  op = first_op(t->mask[c]);                       /* choices[0] of the end cell (pw.c:123) */
  while (op != 0) {                                /* base == NULL only for 'B' */
    int nx = x - (op == 2 ? 0 : 1), ny = y - (op == 1 ? 0 : 1), nop;
    long long pc = cell_of(t, nx, ny);
    if (op == 3) {
      rev[n++] = (p->origin[x - 1] == p->mutant[y - 1]) ? 'M' : 'S';   /* :232 */
      nms++;
      nop = first_op(t->mask[pc]);                 /* base = &prev.choices[0]  (:235) */
    } else {
      rev[n++] = OPCH[op];
      nop = (op == 1) ? (t->base[c] & 3) : ((t->base[c] >> 2) & 3);       /* :281 */
    }
    x = nx; y = ny; c = pc; op = nop;
  }
  r->would_panick = (x + y + nms <= 0);            /* pos <= 0, checked first (pw.c:132-134) */
  r->tb_null = (!r->would_panick && n == 0);       /* pos == len-1 (pw.c:135-138) */
  r->origin_idx = x; r->mutant_idx = y;
  r->tx_len = n;
  r->transcript = (char *)malloc((size_t)n + 1);
  { int k; for (k = 0; k < n; k++) r->transcript[k] = rev[n - 1 - k]; r->transcript[n] = 0; }
  free(rev);
}

/* The 4-bit-mask-only predecessor rule the device traceback uses ("validated recurrence"):
 * after a gap op g the predecessor's active choice is, for go<0: g if g is kept else the first kept;
 * go==0: the first kept; go>0: the first kept op different from g, else g.  This follows from
 * _alnchoice_ID's first-strict-max scan (:268-278) because all kept choices share one score.
 * Returns 1 iff walking with this rule reproduces the explicit base chain everywhere on the path. */
static int maskrule_agrees(const tbl *t, int ei, int ej) {
  const pwo_problem *p = t->p;
  int x, y, op;
  long long c;
  if (p->mode == STD_MODE) { x = ei; y = ej; }
  else { int d = ei + t->dmin; x = ej + (d > 0 ? d : 0); y = ej - (d > 0 ? 0 : d); }
  c = cell_of(t, x, y);
  op = first_op(t->mask[c]);
  while (op != 0) {
    int nx = x - (op == 2 ? 0 : 1), ny = y - (op == 1 ? 0 : 1), nop, rule;
    long long pc = cell_of(t, nx, ny);
    unsigned char pm = t->mask[pc];
    if (op == 3) { nop = first_op(pm); rule = nop; }
    else {
      nop = (op == 1) ? (t->base[c] & 3) : ((t->base[c] >> 2) & 3);
      if (p->go < 0) rule = (pm & (1 << op)) ? op : first_op(pm);
      else if (p->go == 0) rule = first_op(pm);
      else { unsigned char others = (unsigned char)(pm & ~(1 << op)); rule = others ? first_op(others) : op; }
    }
    if (rule != nop) return 0;
    x = nx; y = ny; c = pc; op = nop;
  }
  return 1;
}

static void tbl_free(tbl *t) {
  free(t->row_off); free(t->row_len); free(t->H); free(t->mask); free(t->base);
}

long long pwo_cells(const pwo_problem *p) {
  tbl t; long long n = -1;
  memset(&t, 0, sizeof t); t.p = p;
  if (table_dims(&t) == 0) n = t.row_off[t.num_rows];
  tbl_free(&t);
  return n;
}

int pwo_solve(const pwo_problem *p, pwo_result *r, double *Hout, unsigned char *maskout) {
  tbl t; long long n;
  memset(&t, 0, sizeof t); memset(r, 0, sizeof *r);
  r->opt_i = r->opt_j = -1;
  if (p->max_new_mins > 0) return -2;
  t.p = p;
  r->init_rc = table_dims(&t);
  r->dmin_c = t.dmin; r->dmax_c = t.dmax; r->num_rows = t.num_rows;
  if (r->init_rc != 0) { tbl_free(&t); return 0; }
  n = t.row_off[t.num_rows];
  r->cells = n;
  t.H = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  t.mask = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);   /* _table_init_cells: all empty (:64-74) */
  t.base = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
  { long long k; for (k = 0; k < n; k++) t.H[k] = NAN; }
  fill(&t);
  find_optimal(&t, &r->opt_i, &r->opt_j);
  if (r->opt_i != -1) {
    r->score = t.H[t.row_off[r->opt_i] + r->opt_j];
    traceback(&t, r->opt_i, r->opt_j, r);
    r->maskrule_ok = maskrule_agrees(&t, r->opt_i, r->opt_j);
  }
  if (Hout) memcpy(Hout, t.H, sizeof(double) * (size_t)n);
  if (maskout) memcpy(maskout, t.mask, (size_t)n);
  tbl_free(&t);
  return 0;
}

void pwo_free_result(pwo_result *r) { free(r->transcript); r->transcript = NULL; }
