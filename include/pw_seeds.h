/* pw_seeds.h -- exact-match k-mer seeds of a pair of sequences in diagonal coordinates, on one MI355X.
 *
 * C ABI of the seeding stage that feeds the banded aligner.  It replaces, for one pair of
 * sequences (S, T), what the reference builds out of SQLite tables and Python generators:
 *
 *   biseqt/kmers.py:164-241   kmer_as_int / as_kmer_seq   k-mers as integers in base |alphabet|, masked k-mers
 *   biseqt/kmers.py:437-509   KmerIndex.index_kmers / hits / kmers     (kmer, seqid, pos) table + SQL index
 *   biseqt/seeds.py:117-162   SeedIndex._index_seeds      the seeds table: one row (d, a) = (i - j, i + j) per
 *                                                         pair of equal k-mers, S at i, T at j
 *   biseqt/seeds.py:164-237   SeedIndex.seeds / seed_count             enumeration and band counts
 *   biseqt/blot.py:607-625    the in-memory variant of the same enumeration (WordBlotLocalRef.seeds)
 *
 * Row order is the reference's rowid order: k-mers ascending, then i ascending, then j ascending; for a self
 * comparison (S and T equal by content, seeds.py:33) per k-mer all pairs i < j in that order, then the trivial
 * pairs (i, i) -- exactly what combinations(hits, 2) + [(x, x)] inserts.
 *
 * Plain pointers and sizes only; letters are one byte each (index into the alphabet).  Everything fails loudly
 * (NULL / negative return + pw_seeds_last_error()); there is no CPU fallback.
 */
#ifndef PW_SEEDS_H
#define PW_SEEDS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pw_seed_index pw_seed_index;

/* Plan an index for (S, T): copies both sequences to the device `device`; nothing is computed yet.
 *   alphabet_len L <= 36 and wordlen k with L^k < 2^62 (kmers.py:266-271 bounds them the same way);
 *   mask_sets / n_masks: k-mers whose SET of letters equals one of these sets are dropped (as_kmer_seq's
 *     `mask`, kmers.py:232-236); each set is a bit mask over letter indices (bit c = letter c);
 *   self_comp: 1 = self comparison (T is ignored), 0 = two different sequences, -1 = decide by content
 *     (S == T), which is what SeedIndex.__init__ does. */
pw_seed_index* pw_seeds_create(int device, const uint8_t* S, int64_t nS, const uint8_t* T, int64_t nT,
                               int alphabet_len, int wordlen, const uint64_t* mask_sets, int n_masks,
                               int self_comp);

/* Build the seeds table on the device (k-mer encoding, two sorts, join, expansion); asynchronous on `stream`
 * (a hipStream_t) except for one 8-byte read of the row count.  May be called again (re-builds).
 * Returns 0, or -1 (e.g. the table would not fit: more than max_rows rows; max_rows <= 0 means 2^31 - 1). */
int pw_seeds_build(pw_seed_index* idx, int64_t max_rows, void* stream);

int64_t pw_seeds_num_rows(const pw_seed_index* idx);          /* rows of the seeds table (seed_count()) */
int pw_seeds_is_self(const pw_seed_index* idx);

/* Rows in table order as interleaved (d, a) int32 pairs: on the device (valid until the next build / destroy)
 * or copied to the host (cap = capacity in rows). */
const int32_t* pw_seeds_rows_device(const pw_seed_index* idx);
int pw_seeds_rows(const pw_seed_index* idx, int32_t* da, int64_t cap);

/* COUNT(*) of rows with dmin <= d <= dmax (if have_d) and amin <= a <= amax (if have_a): seeds.py:199-237. */
int64_t pw_seeds_count(const pw_seed_index* idx, int have_d, int32_t dmin, int32_t dmax,
                       int have_a, int32_t amin, int32_t amax);

/* The k-mer of every position of S (which = 0) or T (which = 1) as as_kmer_seq returns them
 * (masked positions: -1); n - k + 1 entries.  Device work, synchronous. */
int64_t pw_seeds_kmers(const pw_seed_index* idx, int which, int64_t* out, int64_t cap);

/* Band selection support (biseqt/blot.py:497-531, WordBlotOverlap.score_seeds): for every row, in table order,
 * the number of OTHER rows whose scaled diagonal d' / radius(d') is within 1 of its own d / radius(d) -- what
 * cKDTree.query_ball_tree(r = 1, p = inf) over the points (d / r(d),) returns, minus the point itself.
 * radius[d + nT] is the band radius of diagonal d for d = -nT .. nS (n_radius = nS + nT + 1 doubles: the
 * reference divides by the float np.ceil(...) returns).  Not defined for self comparisons (-1). */
int pw_seeds_band_neighbours(const pw_seed_index* idx, const double* radius, int64_t n_radius, int32_t* counts,
                             int64_t cap);

/* Local-similarity support (biseqt/blot.py:343-490, WordBlot.find_all_neighbors / score_seeds /
 * similar_segments).  pw_seeds_graph_build links every pair of rows with
 *     max(|d * d_coeff - d' * d_coeff|, |a - a'|) <= radius
 * -- cKDTree.query_ball_tree(radius, p = inf) over the points (d * d_coeff, a), each row's own entry removed --
 * and keeps the adjacency in HBM as CSR.  Returns the number of directed edges (every pair counts twice), or -1.
 * The POINTS of the graph are the rows -- for a self comparison what SeedIndex.seeds(exclude_trivial=True) yields
 * (seeds.py:186-197): every non-trivial row followed by its mirror image (-d, a), trivial rows dropped. */
int64_t pw_seeds_graph_build(pw_seed_index* idx, double d_coeff, double radius);
int64_t pw_seeds_graph_num_points(const pw_seed_index* idx);
int pw_seeds_graph_points(const pw_seed_index* idx, int32_t* da, int64_t cap);      /* (d, a) of every point, in order */
int pw_seeds_graph_counts(const pw_seed_index* idx, int32_t* counts, int64_t cap);          /* neighbours per point */
/* offsets: num_points + 1 entries; neighbours: pw_seeds_graph_build's return value entries (row indices, the order
 * inside a row's list is unspecified -- it is in the reference as well). */
int pw_seeds_graph_fetch(const pw_seed_index* idx, int64_t* offsets, int32_t* neighbours);
/* Connected components of the graph restricted to the rows with avail[row] != 0 (the depth-first growth of
 * similar_segments, blot.py:452-468, finds exactly these): labels[row] = smallest row index of its component,
 * -1 for rows that are not available. */
int pw_seeds_graph_components(const pw_seed_index* idx, const uint8_t* avail, int32_t* labels);

double pw_seeds_build_ms(const pw_seed_index* idx);           /* device time of the last build (HIP events) */
int64_t pw_seeds_algorithmic_bytes(const pw_seed_index* idx); /* see DESIGN.md: bytes the build must move */
void pw_seeds_destroy(pw_seed_index* idx);
const char* pw_seeds_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
