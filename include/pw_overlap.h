/* pw_overlap.h -- overlap band selection for MANY read pairs in one pass, on one MI355X.
 *
 * The reference discovers overlaps by scoring every pair of reads separately
 * (experiments/blot_overlaps.py:262-272: WordBlotOverlapRef(reads[i]).highest_scoring_overlap_band(reads[j])),
 * i.e. per pair biseqt/blot.py:497-579: all exact k-mer seeds, for every seed the number of seeds within 1 of it on
 * the scaled diagonal axis d / r(d), the estimated match probability of its diagonal band, the best band.
 * Everything in that computation depends on a seed only through its DIAGONAL, so for a batch of pairs the device
 * builds one histogram of seeds per (pair, diagonal) directly from a segmented sort-merge join of the k-mers
 * -- rows are never materialised -- and evaluates every occupied diagonal of every pair:
 *
 *   L(d)  = ceil( (2 / (2 - g)) * (min(|S| - d, |T|) + min(d, 0)) )          blot.py:78-112
 *   r(d)  = max(1, ceil( C * sqrt(L(d)) )),  C = erfcinv(1 - sensitivity) sqrt(2 g)   blot.py:116-139
 *   n(d)  = #{seeds with |d / r(d) - d' / r(d')| <= 1} - 1                   blot.py:521-527
 *   w(d)  = (n(d) + 1 - 2 r(d) L(d) p0^k) / L(d)                             blot.py:533-540  (p̂ = w^(1/k), monotone)
 *
 * in IEEE double with the reference's operation order.  The caller turns w into p̂ = min(exp(log(w) / k), 1)
 * (0 where w <= 0) and the z-score with the reference's closed forms; `tie` tells it when more than one diagonal
 * could reach the same p̂, in which case the reference's answer depends on its row order and the single-pair path
 * (pw_seeds.h) must be used for that pair.
 */
#ifndef PW_OVERLAP_H
#define PW_OVERLAP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  uint64_t s_off, t_off;   /* offsets of the two reads in the letter arena (one byte per letter) */
  int32_t s_len, t_len;
} pw_read_pair;

typedef struct {
  int64_t n_seeds;         /* rows the pair's seeds table would hold */
  double w_best;           /* largest w(d) over the occupied diagonals (undefined when n_seeds == 0) */
  int32_t d_best;          /* its diagonal; among equal w the smallest d */
  int32_t n_best;          /* n(d_best): neighbours of a seed on that diagonal */
  int32_t r_best, len_best;        /* r(d_best), L(d_best) */
  int32_t band_best;       /* seeds with d in [d_best - r_best, d_best + r_best] (seed_count of the band) */
  int32_t tie;             /* occupied diagonals whose w is within 1e-9 relative of w_best, or >= 1 when
                              w_best >= 1, or all of them when w_best <= 0 */
  /* the first row of the reference's table order (k-mer asc, i asc, j asc): it wins when every p̂ is 0 */
  int32_t d_first, n_first, r_first, len_first, band_first;
  int32_t pad_;
} pw_overlap_band;         /* 64 bytes */

/* Scores all pairs.  radius_coeff = C above, len_coeff = 2 / (2 - g), word_p_null = (1 / alphabet_len)^wordlen --
 * computed by the caller exactly as the reference computes them (scipy erfcinv, numpy sqrt).  Synchronous.
 * Returns 0, or -1 (pw_overlap_last_error()). */
int pw_overlap_bands(int device, const uint8_t* arena, uint64_t arena_bytes, const pw_read_pair* pairs,
                     int64_t n_pairs, int alphabet_len, int wordlen, double len_coeff, double radius_coeff,
                     double word_p_null, pw_overlap_band* out);

/* ALL pairs of a set of reads through ONE k-mer index (every read is encoded and sorted once): the pairs (a < b)
 * that share at least one seed are found by a self join of the index; every other pair has no seeds and the
 * reference returns None for it.  read a plays S and read b plays T, as in the reference's double loop
 * (experiments/blot_overlaps.py:267-271).  On return *n_out pairs are listed in pair_a / pair_b (ascending
 * (a, b)) with their records in out; capacity max_pairs each.  Multi-GPU: rank shard_rank of shard_world scores
 * only the pairs with a mod shard_world == shard_rank (no data-path collective; (0, 1) = everything).
 * Returns 0 or -1. */
int pw_overlap_all_pairs(int device, const uint8_t* arena, uint64_t arena_bytes, const uint64_t* read_off,
                         const int32_t* read_len, int64_t n_reads, int alphabet_len, int wordlen, double len_coeff,
                         double radius_coeff, double word_p_null, int shard_rank, int shard_world,
                         int64_t max_pairs, int32_t* pair_a, int32_t* pair_b, pw_overlap_band* out,
                         int64_t* n_out);

double pw_overlap_last_ms(void);           /* device time of the last call (HIP events) */
const char* pw_overlap_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
