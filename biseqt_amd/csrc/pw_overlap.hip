// pw_overlap.hip -- overlap band selection for many read pairs at once (C ABI: include/pw_overlap.h).
//
//   K8a k_enc_batch   k-mer of every position of every pair's S (or T) read, key = (pair << kbits) | k-mer
//       sort          both sides by key (rocPRIM radix sort, stable: positions ascending inside a k-mer)
//   K8b k_join_hist   one thread per sorted S element: its run of equal keys on the T side; every (i, j) of the run
//                     is one seed: atomicAdd into the pair's per-diagonal histogram (rows are never written), the
//                     pair's row count, and atomicMin of the element index (-> the first row in table order)
//   K8c k_band_select one workgroup per pair: prefix sums of the histogram, then for every occupied diagonal the
//                     window of diagonals within 1 on the axis d / r(d) (binary searches on the reference's own
//                     floating-point predicate; d / r(d) is monotone in d), n(d), w(d); block reduction to the
//                     best diagonal, the tie count and the seed count of its band
// All floating point is IEEE double in the reference's operation order (compiled with -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>
#include <rocprim/rocprim.hpp>

#include "../../include/pw_overlap.h"

namespace {

thread_local std::string g_err;
thread_local double g_ms = 0.0;
void set_err(const std::string& s) { g_err = s; }
#define OV_CHECK(call)                                                                       \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      set_err(std::string(#call) + ": " + hipGetErrorString(e_));                            \
      return -1;                                                                             \
    }                                                                                        \
  } while (0)

struct DPair { uint64_t s_off, t_off; int32_t s_len, t_len; uint64_t hbase; };   // hbase: start of its histogram

__device__ __forceinline__ int64_t ub_u64(const uint64_t* __restrict__ a, int64_t n, uint64_t key) {   // first > key
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] <= key) lo = mid + 1; else hi = mid; }
  return lo;
}
__device__ __forceinline__ int64_t lb_u64(const uint64_t* __restrict__ a, int64_t n, uint64_t key) {   // first >= key
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}

// side 0: S reads, side 1: T reads.  start[p] = index of pair p's first k-mer on this side (start[n] = total).
__global__ __launch_bounds__(256) void k_enc_batch(const uint8_t* __restrict__ arena, const DPair* __restrict__ pairs,
                                                   const uint64_t* __restrict__ start, int64_t npairs, int64_t total, int side,
                                                   int k, int L, int kbits, uint64_t* __restrict__ keys, uint32_t* __restrict__ pos) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= total) return;
  const int64_t p = ub_u64(start, npairs + 1, (uint64_t)g) - 1;
  const DPair pr = pairs[p];
  const uint32_t q = (uint32_t)(g - (int64_t)start[p]);
  const uint8_t* __restrict__ s = arena + (side ? pr.t_off : pr.s_off) + q;
  uint64_t v = 0;
  for (int t = 0; t < k; t++) v = v * (uint64_t)L + s[t];
  keys[g] = ((uint64_t)p << kbits) | v;
  pos[g] = q;
}

__global__ __launch_bounds__(256) void k_join_hist(const uint64_t* __restrict__ ks, const uint32_t* __restrict__ ps, int64_t ns,
                                                   const uint64_t* __restrict__ kt, const uint32_t* __restrict__ pt, int64_t nt,
                                                   const DPair* __restrict__ pairs, int kbits, uint32_t* __restrict__ hist,
                                                   unsigned long long* __restrict__ nrows, uint32_t* __restrict__ first_e,
                                                   int32_t* __restrict__ dlist /* [npairs][64]: the diagonals of a pair's first 64 seeds */) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= ns) return;
  const uint64_t key = ks[e];
  const int64_t lo = lb_u64(kt, nt, key);
  if (lo >= nt || kt[lo] != key) return;
  const int64_t hi = ub_u64(kt, nt, key);
  const int64_t p = (int64_t)(key >> kbits);
  const DPair pr = pairs[p];
  const int i = (int)ps[e];
  uint32_t* __restrict__ h = hist + pr.hbase + pr.t_len;      // h[d], d = -t_len .. s_len
  const unsigned long long slot0 = atomicAdd(&nrows[p], (unsigned long long)(hi - lo));
  for (int64_t t = lo; t < hi; t++) {
    const int d = i - (int)pt[t];
    atomicAdd(&h[d], 1u);
    const unsigned long long slot = slot0 + (unsigned long long)(t - lo);
    if (slot < 64ull) dlist[p * 64 + (int64_t)slot] = d;      // pairs with <= 64 seeds are scored from this list (K8d)
  }
  atomicMin(&first_e[p], (uint32_t)e);                        // (fewer than 2^32 k-mers per chunk, checked by the host)
}

struct BandConst { double q, C, p0; };
__device__ __forceinline__ int ov_len(int d, int ls, int lt, double q) {
  const int wall = (ls - d < lt ? ls - d : lt) + (d < 0 ? d : 0);
  return (int)ceil(q * (double)wall);
}
__device__ __forceinline__ int ov_rad(int L, double C) {
  const int r = (int)ceil(C * sqrt((double)L));
  return r < 1 ? 1 : r;
}
__device__ __forceinline__ double ov_x(int d, int ls, int lt, BandConst c) {
  return (double)d / (double)ov_rad(ov_len(d, ls, lt, c.q), c.C);
}

// one workgroup per pair
__global__ __launch_bounds__(256) void k_band_select(const DPair* __restrict__ pairs, uint32_t* hist,
                                                     const unsigned long long* __restrict__ nrows, const int32_t* __restrict__ d_first,
                                                     uint64_t hist_first, int small_max, BandConst c, pw_overlap_band* __restrict__ out) {
  __shared__ uint32_t s_sum[256];
  __shared__ double s_w[256];
  __shared__ int s_d[256], s_cnt[256];
  // the occupied diagonals of the pair, in no particular order: the seeds of a true overlap sit on a few dozen neighbouring
  // diagonals, i.e. in the chunks of one or two threads -- evaluating them where they are found left 254 threads idle
  // (0.5 ms per pair); listed first, they are dealt round-robin
  constexpr int kMaxOcc = 8192;
  __shared__ int s_occ[kMaxOcc];
  __shared__ int s_nocc;
  const int p = (int)blockIdx.x, tid = (int)threadIdx.x;
  const DPair pr = pairs[p];
  const int ls = pr.s_len, lt = pr.t_len, nd = ls + lt + 1;
  uint32_t* h = hist + (pr.hbase - hist_first);        // `hist` starts at counter hist_first; index dd = d + lt
  pw_overlap_band o;
  memset(&o, 0, sizeof o);
  o.n_seeds = (int64_t)nrows[p];
  if (o.n_seeds == 0) { if (tid == 0) out[p] = o; return; }
  if (o.n_seeds <= small_max) return;                  // the sparse kernel (k_band_small) scores these
  // ---- inclusive prefix sums in place: every thread owns a contiguous chunk ----
  const int chunk = (nd + 255) / 256;
  const int b = tid * chunk, e = b + chunk < nd ? b + chunk : nd;
  uint32_t sum = 0;
  for (int dd = b; dd < e; dd++) sum += h[dd];
  s_sum[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const uint32_t v = tid >= off ? s_sum[tid - off] : 0u;
    __syncthreads();
    s_sum[tid] += v;
    __syncthreads();
  }
  if (tid == 0) s_nocc = 0;
  __syncthreads();
  uint32_t run = tid ? s_sum[tid - 1] : 0u;
  for (int dd = b; dd < e; dd++) {
    const uint32_t v = h[dd];
    run += v; h[dd] = run;
    if (v) { const int q = atomicAdd(&s_nocc, 1); if (q < kMaxOcc) s_occ[q] = dd; }
  }
  __threadfence_block();
  __syncthreads();
  const int nocc = s_nocc;
  const bool listed = nocc <= kMaxOcc;                 // else: every thread walks its own chunk (as before)
  auto pre = [&](int dd) -> uint32_t { return dd < 0 ? 0u : h[dd < nd ? dd : nd - 1]; };
  // neighbours of a seed on diagonal d and the score of its band
  auto eval = [&](int d, int& n, int& L, int& r) -> double {
    L = ov_len(d, ls, lt, c.q); r = ov_rad(L, c.C);
    const double x = (double)d / (double)r;
    // The window of diagonals within 1 of d on the axis d / r(d).  Both predicates are monotone in d' (what the binary
    // searches of the first version relied on), so the same two boundaries are found by walking from a guess -- d -+ r,
    // where they are unless r(d') changes in between: a handful of evaluations of x(d') instead of 2 x 14.
    int first = d - r < -lt ? -lt : d - r;             // smallest d' in [-lt, d] with x - x(d') <= 1
    if (x - ov_x(first, ls, lt, c) <= 1.0) { while (first > -lt && x - ov_x(first - 1, ls, lt, c) <= 1.0) first--; }
    else { do first++; while (first < d && !(x - ov_x(first, ls, lt, c) <= 1.0)); }
    int lo = d + r > ls ? ls : d + r;                  // largest d' in [d, ls] with x(d') - x <= 1
    if (ov_x(lo, ls, lt, c) - x <= 1.0) { while (lo < ls && ov_x(lo + 1, ls, lt, c) - x <= 1.0) lo++; }
    else { do lo--; while (lo > d && !(ov_x(lo, ls, lt, c) - x <= 1.0)); }
    n = (int)(pre(lo + lt) - pre(first + lt - 1)) - 1;
    const long long area = 2ll * r * L;
    return ((double)(n + 1) - (double)area * c.p0) / (double)L;
  };
  // ---- pass 1: the best occupied diagonal (largest w, then smallest d) ----
  double bw = 0.0; int bd = 0x7fffffff;
  constexpr int kKeep = 4;                             // scores of this thread's first diagonals, kept for pass 2
  double wkeep[kKeep];
  if (listed) {
    int it = 0;
    for (int q = tid; q < nocc; q += 256, it++) {
      const int d = s_occ[q] - lt;
      int n, L, r;
      const double w = eval(d, n, L, r);
      if (it < kKeep) wkeep[it] = w;
      if (bd == 0x7fffffff || w > bw || (w == bw && d < bd)) { bw = w; bd = d; }
    }
  } else {
    for (int dd = b; dd < e; dd++) {
      if (pre(dd) == pre(dd - 1)) continue;
      int n, L, r;
      const double w = eval(dd - lt, n, L, r);
      if (bd == 0x7fffffff || w > bw) { bw = w; bd = dd - lt; }
    }
  }
  s_w[tid] = bw; s_d[tid] = bd;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (tid < off) {
      const double ow = s_w[tid + off]; const int od = s_d[tid + off];
      const bool have = s_d[tid] != 0x7fffffff, ohave = od != 0x7fffffff;
      if (ohave && (!have || ow > s_w[tid] || (ow == s_w[tid] && od < s_d[tid]))) { s_w[tid] = ow; s_d[tid] = od; }
    }
    __syncthreads();
  }
  const double wbest = s_w[0]; const int dbest = s_d[0];
  __syncthreads();
  // ---- pass 2: how many occupied diagonals could reach the same p ----
  int ties = 0;
  const double wcap = wbest >= 1.0 ? 1.0 : wbest;           // p = min(w^(1/k), 1): everything at or above 1 ties
  const double thr = wcap - fabs(wcap) * 1e-9;
  if (listed) {
    int it = 0;
    for (int q = tid; q < nocc; q += 256, it++) {
      int n, L, r;
      const double w = it < kKeep ? wkeep[it] : eval(s_occ[q] - lt, n, L, r);
      ties += (wbest <= 0.0 || w >= thr) ? 1 : 0;
    }
  } else {
    for (int dd = b; dd < e; dd++) {
      if (pre(dd) == pre(dd - 1)) continue;
      int n, L, r;
      const double w = eval(dd - lt, n, L, r);
      ties += (wbest <= 0.0 || w >= thr) ? 1 : 0;
    }
  }
  s_cnt[tid] = ties;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) { if (tid < off) s_cnt[tid] += s_cnt[tid + off]; __syncthreads(); }
  if (tid == 0) {
    int n, L, r;
    o.w_best = eval(dbest, n, L, r);
    o.d_best = dbest; o.n_best = n; o.len_best = L; o.r_best = r;
    o.band_best = (int)(pre(dbest + r + lt) - pre(dbest - r + lt - 1));
    o.tie = s_cnt[0];
    // the diagonal of the first row of the table (k-mer asc, i asc, j asc)
    const int df = d_first[p];
    (void)eval(df, n, L, r);
    o.d_first = df; o.n_first = n; o.len_first = L; o.r_first = r;
    o.band_first = (int)(pre(df + r + lt) - pre(df - r + lt - 1));
    out[p] = o;
  }
}

// pair-list path: the first row = the first sorted S element with a match, paired with the first j of its run
__global__ __launch_bounds__(256) void k_first_d(const uint32_t* __restrict__ first_e, int64_t npairs, const uint64_t* __restrict__ ks,
                                                 const uint32_t* __restrict__ ps, const uint64_t* __restrict__ kt,
                                                 const uint32_t* __restrict__ pt, int64_t nt, int32_t* __restrict__ d_first) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= npairs) return;
  const uint32_t fe = first_e[p];
  if (fe == 0xffffffffu) { d_first[p] = 0; return; }
  const int64_t lo = lb_u64(kt, nt, ks[fe]);
  d_first[p] = (int)ps[fe] - (int)pt[lo];
}

// ---- all reads against all reads through ONE index (K9) -----------------------------------------------------------
// K9a: k-mer of every position of every read; value = (read << 32) | pos.  rstart[r] = first k-mer index of read r.
__global__ __launch_bounds__(256) void k_enc_reads(const uint8_t* __restrict__ arena, const uint64_t* __restrict__ roff,
                                                   const uint64_t* __restrict__ rstart, int64_t nreads, int64_t total, int k, int L,
                                                   uint64_t* __restrict__ keys, uint64_t* __restrict__ vals) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= total) return;
  const int64_t r = ub_u64(rstart, nreads + 1, (uint64_t)g) - 1;
  const uint32_t q = (uint32_t)(g - (int64_t)rstart[r]);
  const uint8_t* __restrict__ s = arena + roff[r] + q;
  uint64_t v = 0;
  for (int t = 0; t < k; t++) v = v * (uint64_t)L + s[t];
  keys[g] = v;
  vals[g] = ((uint64_t)r << 32) | q;
}
// K9b: self join.  Inside a run of equal k-mers the entries are ordered by (read, pos); element e pairs with every
// later entry of a DIFFERENT read: those start at fs[e].
__global__ __launch_bounds__(256) void k_self_count(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ vals, int64_t n,
                                                    int shard_rank, int shard_world, uint32_t* __restrict__ fs, uint64_t* __restrict__ cnt) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  // multi-GPU: the pairs (a, b), a < b, are dealt round-robin by a; every rank joins only its own share
  if ((int)((vals[e] >> 32) % (uint64_t)shard_world) != shard_rank) { fs[e] = (uint32_t)e; cnt[e] = 0; return; }
  const int64_t hi = ub_u64(keys, n, keys[e]);
  const uint64_t next_read = ((vals[e] >> 32) + 1) << 32;
  const int64_t f = e + 1 + lb_u64(vals + e + 1, hi - e - 1, next_read);
  fs[e] = (uint32_t)f;
  cnt[e] = (uint64_t)(hi - f);
}
// one seed per thread: key = rank of the pair (ra < rb) = ra * nreads + rb, value = d = pos_a - pos_b.  Seeds are
// produced in (k-mer, i, j) order per pair, which a STABLE sort by the pair key keeps.
__global__ __launch_bounds__(256) void k_self_expand(const uint64_t* __restrict__ off, int64_t n, int64_t nseeds,
                                                     const uint64_t* __restrict__ vals, const uint32_t* __restrict__ fs,
                                                     uint64_t nreads, uint64_t* __restrict__ pkey, int32_t* __restrict__ dval) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= nseeds) return;
  const int64_t e = ub_u64(off, n, (uint64_t)o) - 1;
  const int64_t f = (int64_t)fs[e] + (o - (int64_t)off[e]);
  const uint64_t va = vals[e], vb = vals[f];
  pkey[o] = (va >> 32) * nreads + (vb >> 32);
  dval[o] = (int32_t)(uint32_t)va - (int32_t)(uint32_t)vb;
}
// per candidate pair (unique key u, seeds [soff[u], soff[u] + scount[u])): its reads, histogram base, first diagonal
__global__ __launch_bounds__(256) void k_cand_pairs(const uint64_t* __restrict__ ukeys, const uint64_t* __restrict__ soff,
                                                    const int32_t* __restrict__ dval, int64_t np, uint64_t nreads,
                                                    const uint64_t* __restrict__ roff, const int32_t* __restrict__ rlen,
                                                    const uint64_t* __restrict__ hbase, DPair* __restrict__ pairs,
                                                    int32_t* __restrict__ d_first, int32_t* __restrict__ pa, int32_t* __restrict__ pb) {
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (u >= np) return;
  const uint64_t a = ukeys[u] / nreads, b = ukeys[u] % nreads;
  pairs[u] = DPair{roff[a], roff[b], rlen[a], rlen[b], hbase[u]};
  d_first[u] = dval[soff[u]];
  pa[u] = (int32_t)a; pb[u] = (int32_t)b;
}
constexpr int kSmallPair = 64;     // pairs with at most this many seeds are scored by one wavefront, without a histogram
constexpr int kMediumPair = 2048;  // all-pairs path: up to this many seeds by one workgroup from the seed list (k_band_medium)
__global__ __launch_bounds__(256) void k_pair_hsize(const uint64_t* __restrict__ ukeys, const unsigned long long* __restrict__ cnt,
                                                    int64_t np, uint64_t nreads, const int32_t* __restrict__ rlen,
                                                    uint64_t* __restrict__ hsize, int sparse_max) {
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (u >= np) return;
  hsize[u] = cnt[u] <= (unsigned long long)sparse_max ? 0ull
                                                      : (uint64_t)rlen[ukeys[u] / nreads] + (uint64_t)rlen[ukeys[u] % nreads] + 1;
}
// seeds [s0, s1) belong to the pairs [u0, u1) of this chunk; hbase is relative to the chunk's first histogram entry
__global__ __launch_bounds__(256) void k_scatter_hist(const int32_t* __restrict__ dval, int64_t s0, int64_t s1,
                                                      const uint64_t* __restrict__ soff, int64_t u0, int64_t u1,
                                                      const DPair* __restrict__ pairs, uint64_t hchunk0, uint32_t* __restrict__ hist,
                                                      int sparse_max) {
  const int64_t o = s0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= s1) return;
  const int64_t u = u0 + ub_u64(soff + u0, u1 - u0, (uint64_t)o) - 1;
  if ((u + 1 < u1 ? soff[u + 1] : (uint64_t)s1) - soff[u] <= (uint64_t)sparse_max) return;     // no histogram for sparse pairs
  const DPair pr = pairs[u];
  atomicAdd(&hist[pr.hbase - hchunk0 + (uint64_t)(dval[o] + pr.t_len)], 1u);
}

// K8d: a pair with at most 64 seeds is scored by ONE wavefront straight from its seed list (lane l = seed l, in table
// order): same L, r, window, n, w as k_band_select's eval(), the neighbour count by a shuffle loop over the lanes.
// Seeds of pair u: dval[soff[u] + l] in table order (all-pairs path, soff != nullptr) or dval[64 u + l] in arbitrary order
// with the first row's diagonal given in d_first (pair-list path).
__global__ __launch_bounds__(256) void k_band_small(const DPair* __restrict__ pairs, const uint64_t* __restrict__ soff,
                                                    const unsigned long long* __restrict__ cnt, const int32_t* __restrict__ dval,
                                                    const int32_t* __restrict__ d_first, int64_t np, BandConst c,
                                                    pw_overlap_band* __restrict__ out) {
  const int64_t u = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int l = (int)(threadIdx.x & 63u);
  if (u >= np) return;
  const int ns = (int)(cnt[u] > 64ull ? 65 : cnt[u]);
  if (ns > kSmallPair) return;
  const DPair pr = pairs[u];
  const int ls = pr.s_len, lt = pr.t_len;
  const bool valid = l < ns;
  const int d = valid ? dval[(soff ? soff[u] : (uint64_t)u * 64ull) + (uint64_t)l] : 0x7fffffff;
  if (ns == 0) return;
  int L = 1, r = 1, first = 0, last = 0;
  if (valid) {
    L = ov_len(d, ls, lt, c.q); r = ov_rad(L, c.C);
    const double x = (double)d / (double)r;
    int lo = -lt, hi = d;
    while (lo < hi) { const int mid = lo + ((hi - lo) >> 1); if (!(x - ov_x(mid, ls, lt, c) <= 1.0)) lo = mid + 1; else hi = mid; }
    first = lo;
    lo = d; hi = ls;
    while (lo < hi) { const int mid = lo + ((hi - lo + 1) >> 1); if (ov_x(mid, ls, lt, c) - x <= 1.0) lo = mid; else hi = mid - 1; }
    last = lo;
  }
  int n = -1; bool rep = valid;
  for (int j = 0; j < ns; j++) {
    const int dj = __shfl(d, j, 64);
    n += (valid && dj >= first && dj <= last) ? 1 : 0;
    rep = rep && !(j < l && dj == d);              // the first lane of every occupied diagonal represents it
  }
  const double w = valid ? ((double)(n + 1) - (double)(2ll * r * L) * c.p0) / (double)L : 0.0;
  // best occupied diagonal: largest w, then smallest d
  double bw = w; int bd = rep ? d : 0x7fffffff;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double ow = __shfl_xor(bw, off, 64); const int od = __shfl_xor(bd, off, 64);
    if (od != 0x7fffffff && (bd == 0x7fffffff || ow > bw || (ow == bw && od < bd))) { bw = ow; bd = od; }
  }
  const double wcap = bw >= 1.0 ? 1.0 : bw;
  const double thr = wcap - fabs(wcap) * 1e-9;
  const unsigned long long tie_mask = __ballot(rep && (bw <= 0.0 || w >= thr));
  const unsigned long long win_mask = __ballot(rep && d == bd);
  const int wl = __ffsll((long long)win_mask) - 1;
  const int rb = __shfl(r, wl, 64), Lb = __shfl(L, wl, 64), nb = __shfl(n, wl, 64);
  const int d0 = d_first ? d_first[u] : __shfl(d, 0, 64);
  const int fl = __ffsll((long long)__ballot(valid && d == d0)) - 1;      // a seed on the first row's diagonal
  const int r0 = __shfl(r, fl, 64), L0 = __shfl(L, fl, 64), n0 = __shfl(n, fl, 64);
  const unsigned long long band_b = __ballot(valid && d >= bd - rb && d <= bd + rb);
  const unsigned long long band_f = __ballot(valid && d >= d0 - r0 && d <= d0 + r0);
  if (l == 0) {
    pw_overlap_band o;
    memset(&o, 0, sizeof o);
    o.n_seeds = ns; o.w_best = bw; o.d_best = bd; o.n_best = nb; o.r_best = rb; o.len_best = Lb;
    o.band_best = __popcll(band_b); o.tie = __popcll(tie_mask);
    o.d_first = d0; o.n_first = n0; o.r_first = r0; o.len_first = L0; o.band_first = __popcll(band_f);
    out[u] = o;
  }
}

// L, r and the window [first, last] of diagonals within 1 of d on the axis d / r(d): the values k_band_select's eval() uses
__device__ __forceinline__ void ov_window(int d, int ls, int lt, BandConst c, int& L, int& r, int& first, int& last) {
  L = ov_len(d, ls, lt, c.q); r = ov_rad(L, c.C);
  const double x = (double)d / (double)r;
  first = d - r < -lt ? -lt : d - r;
  if (x - ov_x(first, ls, lt, c) <= 1.0) { while (first > -lt && x - ov_x(first - 1, ls, lt, c) <= 1.0) first--; }
  else { do first++; while (first < d && !(x - ov_x(first, ls, lt, c) <= 1.0)); }
  last = d + r > ls ? ls : d + r;
  if (ov_x(last, ls, lt, c) - x <= 1.0) { while (last < ls && ov_x(last + 1, ls, lt, c) - x <= 1.0) last++; }
  else { do last--; while (last > d && !(ov_x(last, ls, lt, c) - x <= 1.0)); }
}

// K8e: a pair with 65 .. kMediumPair seeds is scored by ONE workgroup straight from its seed list in LDS (all-pairs path):
// the same L, r, window, n, w per occupied diagonal as k_band_select -- the neighbour count by a loop over the list instead
// of a prefix-summed histogram of |S| + |T| + 1 counters per pair (10 001 for 5 kb reads: reading and summing them was the
// bulk of the dense kernel's time, for the ~250 seeds of a true overlap).  The first seed of every occupied diagonal
// represents it.
__global__ __launch_bounds__(256) void k_list_medium(const unsigned long long* __restrict__ cnt, int64_t np, unsigned long long* __restrict__ n_list,
                                                     int64_t* __restrict__ list) {
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (u >= np) return;
  if (cnt[u] > (unsigned long long)kSmallPair && cnt[u] <= (unsigned long long)kMediumPair) list[atomicAdd(n_list, 1ull)] = u;
}
__global__ __launch_bounds__(256) void k_band_medium(const DPair* __restrict__ pairs, const int64_t* __restrict__ list,
                                                     const uint64_t* __restrict__ soff, const unsigned long long* __restrict__ cnt,
                                                     const int32_t* __restrict__ dval, const int32_t* __restrict__ d_first, BandConst c,
                                                     pw_overlap_band* __restrict__ out) {
  constexpr int PER = kMediumPair / 256;
  __shared__ int s_dv[kMediumPair];
  __shared__ double s_w[256];
  __shared__ int s_d[256], s_cnt[256], s_b1[256], s_b2[256];
  __shared__ int s_best[3], s_first[3];
  const int64_t u = list[blockIdx.x];
  const int tid = (int)threadIdx.x;
  const int ns = (int)cnt[u];
  const DPair pr = pairs[u];
  const int ls = pr.s_len, lt = pr.t_len;
  const uint64_t base = soff[u];
  for (int l = tid; l < ns; l += 256) s_dv[l] = dval[base + (uint64_t)l];
  __syncthreads();
  const int d0 = d_first[u];
  double wq[PER];
  unsigned repmask = 0;
  double bw = 0.0; int bd = 0x7fffffff;
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const int l = tid + 256 * q;
    wq[q] = 0.0;
    if (l < ns) {
      const int d = s_dv[l];
      int L, r, first, last;
      ov_window(d, ls, lt, c, L, r, first, last);
      int n = -1; bool rp = true;
      for (int j = 0; j < ns; j++) {
        const int dj = s_dv[j];
        n += (dj >= first && dj <= last) ? 1 : 0;
        rp = rp && !(j < l && dj == d);
      }
      const double w = ((double)(n + 1) - (double)(2ll * r * L) * c.p0) / (double)L;
      wq[q] = w;
      if (rp) {
        repmask |= 1u << q;
        if (bd == 0x7fffffff || w > bw || (w == bw && d < bd)) { bw = w; bd = d; }
      }
      if (d == d0 && rp) { s_first[0] = n; s_first[1] = L; s_first[2] = r; }       // (one representative per diagonal)
    }
  }
  s_w[tid] = bw; s_d[tid] = bd;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (tid < off) {
      const double ow = s_w[tid + off]; const int od = s_d[tid + off];
      const bool have = s_d[tid] != 0x7fffffff, ohave = od != 0x7fffffff;
      if (ohave && (!have || ow > s_w[tid] || (ow == s_w[tid] && od < s_d[tid]))) { s_w[tid] = ow; s_d[tid] = od; }
    }
    __syncthreads();
  }
  const double wbest = s_w[0]; const int dbest = s_d[0];
  __syncthreads();
  const double wcap = wbest >= 1.0 ? 1.0 : wbest;
  const double thr = wcap - fabs(wcap) * 1e-9;
  int ties = 0;
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const int l = tid + 256 * q;
    if (l < ns && (repmask >> q & 1u)) {
      ties += (wbest <= 0.0 || wq[q] >= thr) ? 1 : 0;
      if (s_dv[l] == dbest) {                                                       // the best diagonal's representative
        int L, r, first, last;
        ov_window(dbest, ls, lt, c, L, r, first, last);
        int n = -1;
        for (int j = 0; j < ns; j++) n += (s_dv[j] >= first && s_dv[j] <= last) ? 1 : 0;
        s_best[0] = n; s_best[1] = L; s_best[2] = r;
      }
    }
  }
  s_cnt[tid] = ties;
  __syncthreads();
  const int rb = s_best[2], r0 = s_first[2];
  int b1 = 0, b2 = 0;
  for (int l = tid; l < ns; l += 256) {
    const int d = s_dv[l];
    b1 += (d >= dbest - rb && d <= dbest + rb) ? 1 : 0;
    b2 += (d >= d0 - r0 && d <= d0 + r0) ? 1 : 0;
  }
  s_b1[tid] = b1; s_b2[tid] = b2;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (tid < off) { s_cnt[tid] += s_cnt[tid + off]; s_b1[tid] += s_b1[tid + off]; s_b2[tid] += s_b2[tid + off]; }
    __syncthreads();
  }
  if (tid == 0) {
    pw_overlap_band o;
    memset(&o, 0, sizeof o);
    o.n_seeds = ns; o.w_best = wbest; o.d_best = dbest; o.n_best = s_best[0]; o.len_best = s_best[1]; o.r_best = s_best[2];
    o.band_best = s_b1[0]; o.tie = s_cnt[0];
    o.d_first = d0; o.n_first = s_first[0]; o.len_first = s_first[1]; o.r_first = s_first[2]; o.band_first = s_b2[0];
    out[u] = o;
  }
}

struct Ev {
  hipEvent_t e = nullptr;
  int make() { OV_CHECK(hipEventCreate(&e)); return 0; }
  ~Ev() { if (e) (void)hipEventDestroy(e); }
};

struct Buf {
  void* p = nullptr;
  int alloc(size_t bytes) { OV_CHECK(hipMalloc(&p, bytes ? bytes : 16)); return 0; }
  ~Buf() { if (p) (void)hipFree(p); }
};

int run_chunk(const uint8_t* d_arena, const pw_read_pair* pairs, int64_t n, int L, int k, int kbits, BandConst bc,
              pw_overlap_band* out, hipEvent_t ev0, hipEvent_t ev1, float* ms) {
  std::vector<DPair> hp((size_t)n);
  std::vector<uint64_t> ss((size_t)n + 1), ts((size_t)n + 1);
  uint64_t hb = 0, cs = 0, ct = 0;
  for (int64_t p = 0; p < n; p++) {
    hp[(size_t)p] = DPair{pairs[p].s_off, pairs[p].t_off, pairs[p].s_len, pairs[p].t_len, hb};
    hb += (uint64_t)pairs[p].s_len + (uint64_t)pairs[p].t_len + 1;
    ss[(size_t)p] = cs; ts[(size_t)p] = ct;
    cs += pairs[p].s_len >= k ? (uint64_t)(pairs[p].s_len - k + 1) : 0;
    ct += pairs[p].t_len >= k ? (uint64_t)(pairs[p].t_len - k + 1) : 0;
  }
  ss[(size_t)n] = cs; ts[(size_t)n] = ct;
  Buf dp, dss, dts, kin, pin, ksb, psb, ktb, ptb, hist, rows, first, dout, tmp, dlist;
  if (dp.alloc(sizeof(DPair) * (size_t)n) || dss.alloc(8 * ((size_t)n + 1)) || dts.alloc(8 * ((size_t)n + 1)) ||
      kin.alloc(8 * (size_t)std::max(cs, ct)) || pin.alloc(4 * (size_t)std::max(cs, ct)) || ksb.alloc(8 * (size_t)cs) ||
      psb.alloc(4 * (size_t)cs) || ktb.alloc(8 * (size_t)ct) || ptb.alloc(4 * (size_t)ct) || hist.alloc(4 * (size_t)hb) ||
      rows.alloc(8 * (size_t)n) || first.alloc(4 * (size_t)n) || dout.alloc(sizeof(pw_overlap_band) * (size_t)n) ||
      dlist.alloc(256 * (size_t)n)) return -1;
  OV_CHECK(hipMemcpy(dp.p, hp.data(), sizeof(DPair) * (size_t)n, hipMemcpyHostToDevice));
  OV_CHECK(hipMemcpy(dss.p, ss.data(), 8 * ((size_t)n + 1), hipMemcpyHostToDevice));
  OV_CHECK(hipMemcpy(dts.p, ts.data(), 8 * ((size_t)n + 1), hipMemcpyHostToDevice));
  OV_CHECK(hipEventRecord(ev0, nullptr));
  OV_CHECK(hipMemsetAsync(hist.p, 0, 4 * (size_t)hb, nullptr));
  OV_CHECK(hipMemsetAsync(rows.p, 0, 8 * (size_t)n, nullptr));
  OV_CHECK(hipMemsetAsync(first.p, 0xff, 4 * (size_t)n, nullptr));
  int pbits = 1; while (((uint64_t)n >> pbits) != 0) pbits++;
  size_t tb = 0, tb2 = 0;
  for (int side = 0; side < 2; side++) {
    const uint64_t tot = side ? ct : cs;
    if (tot == 0) continue;
    hipLaunchKernelGGL(k_enc_batch, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, nullptr, d_arena, (const DPair*)dp.p,
                       (const uint64_t*)(side ? dts.p : dss.p), n, (int64_t)tot, side, k, L, kbits, (uint64_t*)kin.p, (uint32_t*)pin.p);
    uint64_t* ko = (uint64_t*)(side ? ktb.p : ksb.p); uint32_t* po = (uint32_t*)(side ? ptb.p : psb.p);
    OV_CHECK(rocprim::radix_sort_pairs(nullptr, tb2, (const uint64_t*)kin.p, ko, (const uint32_t*)pin.p, po, (size_t)tot, 0u,
                                       (unsigned)(kbits + pbits), (hipStream_t) nullptr));
    if (tb2 > tb) { if (tmp.p) { (void)hipFree(tmp.p); tmp.p = nullptr; } if (tmp.alloc(tb2)) return -1; tb = tb2; }
    OV_CHECK(rocprim::radix_sort_pairs(tmp.p, tb2, (const uint64_t*)kin.p, ko, (const uint32_t*)pin.p, po, (size_t)tot, 0u,
                                       (unsigned)(kbits + pbits), (hipStream_t) nullptr));
  }
  if (cs && ct)
    hipLaunchKernelGGL(k_join_hist, dim3((unsigned)((cs + 255) / 256)), dim3(256), 0, nullptr, (const uint64_t*)ksb.p,
                       (const uint32_t*)psb.p, (int64_t)cs, (const uint64_t*)ktb.p, (const uint32_t*)ptb.p, (int64_t)ct,
                       (const DPair*)dp.p, kbits, (uint32_t*)hist.p, (unsigned long long*)rows.p, (uint32_t*)first.p, (int32_t*)dlist.p);
  Buf dfirst;
  if (dfirst.alloc(4 * (size_t)n)) return -1;
  hipLaunchKernelGGL(k_first_d, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (const uint32_t*)first.p, n,
                     (const uint64_t*)ksb.p, (const uint32_t*)psb.p, (const uint64_t*)ktb.p, (const uint32_t*)ptb.p, (int64_t)ct,
                     (int32_t*)dfirst.p);
  hipLaunchKernelGGL(k_band_small, dim3((unsigned)(((uint64_t)n * 64 + 255) / 256)), dim3(256), 0, nullptr, (const DPair*)dp.p,
                     (const uint64_t*)nullptr, (const unsigned long long*)rows.p, (const int32_t*)dlist.p, (const int32_t*)dfirst.p, n, bc,
                     (pw_overlap_band*)dout.p);
  hipLaunchKernelGGL(k_band_select, dim3((unsigned)n), dim3(256), 0, nullptr, (const DPair*)dp.p, (uint32_t*)hist.p,
                     (const unsigned long long*)rows.p, (const int32_t*)dfirst.p, (uint64_t)0, kSmallPair, bc, (pw_overlap_band*)dout.p);
  OV_CHECK(hipEventRecord(ev1, nullptr));
  OV_CHECK(hipMemcpy(out, dout.p, sizeof(pw_overlap_band) * (size_t)n, hipMemcpyDeviceToHost));
  OV_CHECK(hipGetLastError());
  float t = 0.f;
  OV_CHECK(hipEventElapsedTime(&t, ev0, ev1));
  *ms += t;
  return 0;
}

int run_all_pairs(const uint8_t* d_arena, const uint64_t* read_off, const int32_t* read_len, int64_t R, int L, int k, int kbits,
                  BandConst bc, int shard_rank, int shard_world, int64_t max_pairs, int32_t* pair_a, int32_t* pair_b, pw_overlap_band* out,
                  int64_t* n_out, float* ms) {
  std::vector<uint64_t> rstart((size_t)R + 1);
  uint64_t K = 0;
  for (int64_t r = 0; r < R; r++) { rstart[(size_t)r] = K; K += read_len[r] >= k ? (uint64_t)(read_len[r] - k + 1) : 0; }
  rstart[(size_t)R] = K;
  *n_out = 0;
  if (K == 0) return 0;
  if (K >= (1ull << 32)) { set_err("more than 2^32 k-mers in one index: split the read set"); return -1; }
  Ev e0, e1;
  if (e0.make() || e1.make()) return -1;
  const hipEvent_t ev0 = e0.e, ev1 = e1.e;
  Buf droff, drlen, drstart, kin, vin, ks, vs, fs, cnt, off, scal, tmp;
  if (droff.alloc(8 * (size_t)R) || drlen.alloc(4 * (size_t)R) || drstart.alloc(8 * ((size_t)R + 1)) || kin.alloc(8 * (size_t)K) ||
      vin.alloc(8 * (size_t)K) || ks.alloc(8 * (size_t)K) || vs.alloc(8 * (size_t)K) || fs.alloc(4 * (size_t)K) ||
      cnt.alloc(8 * (size_t)K) || off.alloc(8 * (size_t)K) || scal.alloc(64)) return -1;
  OV_CHECK(hipMemcpy(droff.p, read_off, 8 * (size_t)R, hipMemcpyHostToDevice));
  OV_CHECK(hipMemcpy(drlen.p, read_len, 4 * (size_t)R, hipMemcpyHostToDevice));
  OV_CHECK(hipMemcpy(drstart.p, rstart.data(), 8 * ((size_t)R + 1), hipMemcpyHostToDevice));
  OV_CHECK(hipEventRecord(ev0, nullptr));
  const dim3 blk(256), gK((unsigned)((K + 255) / 256));
  hipLaunchKernelGGL(k_enc_reads, gK, blk, 0, nullptr, d_arena, (const uint64_t*)droff.p, (const uint64_t*)drstart.p, R, (int64_t)K, k, L,
                     (uint64_t*)kin.p, (uint64_t*)vin.p);
  size_t tb = 0;
  OV_CHECK(rocprim::radix_sort_pairs(nullptr, tb, (const uint64_t*)kin.p, (uint64_t*)ks.p, (const uint64_t*)vin.p, (uint64_t*)vs.p, (size_t)K,
                                     0u, (unsigned)kbits, (hipStream_t) nullptr));
  if (tmp.alloc(tb)) return -1;
  OV_CHECK(rocprim::radix_sort_pairs(tmp.p, tb, (const uint64_t*)kin.p, (uint64_t*)ks.p, (const uint64_t*)vin.p, (uint64_t*)vs.p, (size_t)K,
                                     0u, (unsigned)kbits, (hipStream_t) nullptr));
  hipLaunchKernelGGL(k_self_count, gK, blk, 0, nullptr, (const uint64_t*)ks.p, (const uint64_t*)vs.p, (int64_t)K, shard_rank, shard_world,
                     (uint32_t*)fs.p, (uint64_t*)cnt.p);
  size_t tb2 = 0;
  OV_CHECK(rocprim::exclusive_scan(nullptr, tb2, (const uint64_t*)cnt.p, (uint64_t*)off.p, (uint64_t)0, (size_t)K, rocprim::plus<uint64_t>(), (hipStream_t) nullptr));
  Buf tmp2;
  if (tmp2.alloc(tb2)) return -1;
  OV_CHECK(rocprim::exclusive_scan(tmp2.p, tb2, (const uint64_t*)cnt.p, (uint64_t*)off.p, (uint64_t)0, (size_t)K, rocprim::plus<uint64_t>(), (hipStream_t) nullptr));
  uint64_t last_off = 0, last_cnt = 0;
  OV_CHECK(hipMemcpy(&last_off, (uint64_t*)off.p + (K - 1), 8, hipMemcpyDeviceToHost));
  OV_CHECK(hipMemcpy(&last_cnt, (uint64_t*)cnt.p + (K - 1), 8, hipMemcpyDeviceToHost));
  const uint64_t NS = last_off + last_cnt;
  if (NS == 0) return 0;
  if (NS >= (1ull << 32)) { set_err("more than 2^32 seeds between the reads: use a longer word"); return -1; }
  // seeds -> (pair key, d), stably sorted by the pair key
  Buf pk_in, dv_in, pk, dv;
  if (pk_in.alloc(8 * (size_t)NS) || dv_in.alloc(4 * (size_t)NS) || pk.alloc(8 * (size_t)NS) || dv.alloc(4 * (size_t)NS)) return -1;
  hipLaunchKernelGGL(k_self_expand, dim3((unsigned)((NS + 255) / 256)), blk, 0, nullptr, (const uint64_t*)off.p, (int64_t)K, (int64_t)NS,
                     (const uint64_t*)vs.p, (const uint32_t*)fs.p, (uint64_t)R, (uint64_t*)pk_in.p, (int32_t*)dv_in.p);
  int pbits = 1; while ((((uint64_t)R * (uint64_t)R) >> pbits) != 0) pbits++;
  size_t tb3 = 0;
  OV_CHECK(rocprim::radix_sort_pairs(nullptr, tb3, (const uint64_t*)pk_in.p, (uint64_t*)pk.p, (const int32_t*)dv_in.p, (int32_t*)dv.p, (size_t)NS,
                                     0u, (unsigned)pbits, (hipStream_t) nullptr));
  Buf tmp3;
  if (tmp3.alloc(tb3)) return -1;
  OV_CHECK(rocprim::radix_sort_pairs(tmp3.p, tb3, (const uint64_t*)pk_in.p, (uint64_t*)pk.p, (const int32_t*)dv_in.p, (int32_t*)dv.p, (size_t)NS,
                                     0u, (unsigned)pbits, (hipStream_t) nullptr));
  // unique pair keys with their seed counts
  Buf uk, uc, nruns;
  if (uk.alloc(8 * (size_t)NS) || uc.alloc(8 * (size_t)NS) || nruns.alloc(16)) return -1;
  size_t tb4 = 0;
  OV_CHECK(rocprim::run_length_encode(nullptr, tb4, (const uint64_t*)pk.p, (unsigned int)NS, (uint64_t*)uk.p, (unsigned long long*)uc.p,
                                      (unsigned long long*)nruns.p, (hipStream_t) nullptr));
  Buf tmp4;
  if (tmp4.alloc(tb4)) return -1;
  OV_CHECK(rocprim::run_length_encode(tmp4.p, tb4, (const uint64_t*)pk.p, (unsigned int)NS, (uint64_t*)uk.p, (unsigned long long*)uc.p,
                                      (unsigned long long*)nruns.p, (hipStream_t) nullptr));
  unsigned long long NP = 0;
  OV_CHECK(hipMemcpy(&NP, nruns.p, 8, hipMemcpyDeviceToHost));
  if ((int64_t)NP > max_pairs) {
    char msg[160];
    snprintf(msg, sizeof msg, "%llu pairs of reads share a seed (capacity %lld): raise max_pairs or the word length", NP, (long long)max_pairs);
    set_err(msg);
    return -1;
  }
  // seed offsets and histogram bases of the candidate pairs
  Buf soff, hsize, hbase, dpairs, dfirst, dpa, dpb, dout;
  if (soff.alloc(8 * (size_t)NP) || hsize.alloc(8 * (size_t)NP) || hbase.alloc(8 * (size_t)NP) || dpairs.alloc(sizeof(DPair) * (size_t)NP) ||
      dfirst.alloc(4 * (size_t)NP) || dpa.alloc(4 * (size_t)NP) || dpb.alloc(4 * (size_t)NP) || dout.alloc(sizeof(pw_overlap_band) * (size_t)NP)) return -1;
  const dim3 gP((unsigned)((NP + 255) / 256));
  size_t tb5 = 0;
  OV_CHECK(rocprim::exclusive_scan(nullptr, tb5, (const uint64_t*)uc.p, (uint64_t*)soff.p, (uint64_t)0, (size_t)NP, rocprim::plus<uint64_t>(), (hipStream_t) nullptr));
  Buf tmp5;
  if (tmp5.alloc(tb5)) return -1;
  OV_CHECK(rocprim::exclusive_scan(tmp5.p, tb5, (const uint64_t*)uc.p, (uint64_t*)soff.p, (uint64_t)0, (size_t)NP, rocprim::plus<uint64_t>(), (hipStream_t) nullptr));
  hipLaunchKernelGGL(k_pair_hsize, gP, blk, 0, nullptr, (const uint64_t*)uk.p, (const unsigned long long*)uc.p, (int64_t)NP, (uint64_t)R,
                     (const int32_t*)drlen.p, (uint64_t*)hsize.p, kMediumPair);
  OV_CHECK(rocprim::exclusive_scan(tmp5.p, tb5, (const uint64_t*)hsize.p, (uint64_t*)hbase.p, (uint64_t)0, (size_t)NP, rocprim::plus<uint64_t>(), (hipStream_t) nullptr));
  hipLaunchKernelGGL(k_cand_pairs, gP, blk, 0, nullptr, (const uint64_t*)uk.p, (const uint64_t*)soff.p, (const int32_t*)dv.p, (int64_t)NP, (uint64_t)R,
                     (const uint64_t*)droff.p, (const int32_t*)drlen.p, (const uint64_t*)hbase.p, (DPair*)dpairs.p, (int32_t*)dfirst.p,
                     (int32_t*)dpa.p, (int32_t*)dpb.p);
  hipLaunchKernelGGL(k_band_small, dim3((unsigned)((NP * 64 + 255) / 256)), blk, 0, nullptr, (const DPair*)dpairs.p, (const uint64_t*)soff.p,
                     (const unsigned long long*)uc.p, (const int32_t*)dv.p, (const int32_t*)nullptr, (int64_t)NP, bc, (pw_overlap_band*)dout.p);
  // pairs with 65 .. kMediumPair seeds: one workgroup each, from the seed list
  {
    Buf mlist, mcount;
    if (mlist.alloc(8 * (size_t)NP) || mcount.alloc(16)) return -1;
    OV_CHECK(hipMemsetAsync(mcount.p, 0, 16, nullptr));
    hipLaunchKernelGGL(k_list_medium, gP, blk, 0, nullptr, (const unsigned long long*)uc.p, (int64_t)NP, (unsigned long long*)mcount.p, (int64_t*)mlist.p);
    unsigned long long NM = 0;
    OV_CHECK(hipMemcpy(&NM, mcount.p, 8, hipMemcpyDeviceToHost));
    if (NM) hipLaunchKernelGGL(k_band_medium, dim3((unsigned)NM), blk, 0, nullptr, (const DPair*)dpairs.p, (const int64_t*)mlist.p, (const uint64_t*)soff.p,
                               (const unsigned long long*)uc.p, (const int32_t*)dv.p, (const int32_t*)dfirst.p, bc, (pw_overlap_band*)dout.p);
    OV_CHECK(hipDeviceSynchronize());                    // (mlist is freed at the end of this scope)
  }
  // chunks of pairs whose histograms fit 2^30 counters (none at all when no pair has more than kMediumPair seeds)
  uint64_t last_base = 0, last_size = 0;
  if (NP) {
    OV_CHECK(hipMemcpy(&last_base, (const uint64_t*)hbase.p + (NP - 1), 8, hipMemcpyDeviceToHost));
    OV_CHECK(hipMemcpy(&last_size, (const uint64_t*)hsize.p + (NP - 1), 8, hipMemcpyDeviceToHost));
  }
  const bool any_hist = last_base + last_size > 0;
  const size_t NPh = any_hist ? (size_t)NP : 0;
  std::vector<uint64_t> h_hbase(NPh), h_soff(NPh), h_hsize(NPh);
  if (any_hist) {
  OV_CHECK(hipMemcpy(h_hbase.data(), hbase.p, 8 * (size_t)NP, hipMemcpyDeviceToHost));
  OV_CHECK(hipMemcpy(h_soff.data(), soff.p, 8 * (size_t)NP, hipMemcpyDeviceToHost));
  OV_CHECK(hipMemcpy(h_hsize.data(), hsize.p, 8 * (size_t)NP, hipMemcpyDeviceToHost));
  }
  const uint64_t cap = 1ull << 30;
  Buf hist;
  uint64_t hist_cap = 0;
  for (uint64_t u0 = 0; any_hist && u0 < NP;) {
    uint64_t u1 = u0, tot = 0;
    while (u1 < NP && (u1 == u0 || tot + h_hsize[(size_t)u1] <= cap)) { tot += h_hsize[(size_t)u1]; u1++; }
    if (tot == 0) { u0 = u1; continue; }                 // no pair of this chunk needs a histogram
    if (tot > hist_cap) { if (hist.p) { (void)hipFree(hist.p); hist.p = nullptr; } if (hist.alloc(4 * (size_t)tot)) return -1; hist_cap = tot; }
    OV_CHECK(hipMemsetAsync(hist.p, 0, 4 * (size_t)tot, nullptr));
    const uint64_t s0 = h_soff[(size_t)u0], s1 = u1 < NP ? h_soff[(size_t)u1] : NS;
    hipLaunchKernelGGL(k_scatter_hist, dim3((unsigned)((s1 - s0 + 255) / 256)), blk, 0, nullptr, (const int32_t*)dv.p, (int64_t)s0, (int64_t)s1,
                       (const uint64_t*)soff.p, (int64_t)u0, (int64_t)u1, (const DPair*)dpairs.p, h_hbase[(size_t)u0], (uint32_t*)hist.p,
                       kMediumPair);
    hipLaunchKernelGGL(k_band_select, dim3((unsigned)(u1 - u0)), blk, 0, nullptr, (const DPair*)dpairs.p + u0, (uint32_t*)hist.p,
                       (const unsigned long long*)uc.p + u0, (const int32_t*)dfirst.p + u0, h_hbase[(size_t)u0], kMediumPair, bc,
                       (pw_overlap_band*)dout.p + u0);
    u0 = u1;
  }
  OV_CHECK(hipEventRecord(ev1, nullptr));
  OV_CHECK(hipMemcpy(out, dout.p, sizeof(pw_overlap_band) * (size_t)NP, hipMemcpyDeviceToHost));
  OV_CHECK(hipMemcpy(pair_a, dpa.p, 4 * (size_t)NP, hipMemcpyDeviceToHost));
  OV_CHECK(hipMemcpy(pair_b, dpb.p, 4 * (size_t)NP, hipMemcpyDeviceToHost));
  OV_CHECK(hipGetLastError());
  float t = 0.f;
  OV_CHECK(hipEventElapsedTime(&t, ev0, ev1));
  *ms = t;
  *n_out = (int64_t)NP;
  return 0;
}

}  // namespace

extern "C" {

int pw_overlap_all_pairs(int device, const uint8_t* arena, uint64_t arena_bytes, const uint64_t* read_off, const int32_t* read_len,
                         int64_t n_reads, int alphabet_len, int wordlen, double len_coeff, double radius_coeff, double word_p_null,
                         int shard_rank, int shard_world, int64_t max_pairs, int32_t* pair_a, int32_t* pair_b, pw_overlap_band* out,
                         int64_t* n_out) {
  if (alphabet_len < 1 || alphabet_len > 36 || wordlen < 1 || wordlen > 31) { set_err("alphabet_len 1..36, wordlen 1..31"); return -1; }
  if (n_reads < 0 || n_reads >= (1ll << 31) || !n_out || (n_reads && (!read_off || !read_len))) { set_err("bad arguments"); return -1; }
  if (!(len_coeff > 0) || !(radius_coeff > 0) || !(word_p_null > 0)) { set_err("coefficients must be positive"); return -1; }
  if (shard_world < 1 || shard_rank < 0 || shard_rank >= shard_world) { set_err("bad shard"); return -1; }
  uint64_t kmax = 1; int kbits = 0;
  for (int i = 0; i < wordlen; i++) { if (kmax > (1ull << 62) / (uint64_t)alphabet_len) { set_err("alphabet_len ^ wordlen must be below 2^62"); return -1; } kmax *= (uint64_t)alphabet_len; }
  while (((kmax - 1) >> kbits) != 0) kbits++;
  if (kbits == 0) kbits = 1;
  for (int64_t r = 0; r < n_reads; r++)
    if (read_len[r] < 0 || read_off[r] + (uint64_t)read_len[r] > arena_bytes) { set_err("a read lies outside the arena"); return -1; }
  for (uint64_t i = 0; i < arena_bytes; i++) if (arena[i] >= alphabet_len) { set_err("letter outside the alphabet"); return -1; }
  g_ms = 0.0; *n_out = 0;
  if (n_reads < 2) return 0;
  OV_CHECK(hipSetDevice(device));
  Buf d_arena;
  if (d_arena.alloc((size_t)arena_bytes + 64)) return -1;
  OV_CHECK(hipMemcpy(d_arena.p, arena, (size_t)arena_bytes, hipMemcpyHostToDevice));
  float ms = 0.f;
  const int rc = run_all_pairs((const uint8_t*)d_arena.p, read_off, read_len, n_reads, alphabet_len, wordlen, kbits,
                               BandConst{len_coeff, radius_coeff, word_p_null}, shard_rank, shard_world, max_pairs, pair_a, pair_b, out,
                               n_out, &ms);
  g_ms = (double)ms;
  return rc;
}

const char* pw_overlap_last_error(void) { return g_err.c_str(); }
double pw_overlap_last_ms(void) { return g_ms; }

int pw_overlap_bands(int device, const uint8_t* arena, uint64_t arena_bytes, const pw_read_pair* pairs, int64_t n_pairs,
                     int alphabet_len, int wordlen, double len_coeff, double radius_coeff, double word_p_null,
                     pw_overlap_band* out) {
  static_assert(sizeof(pw_overlap_band) == 64, "pw_overlap_band is 64 bytes");
  if (alphabet_len < 1 || alphabet_len > 36 || wordlen < 1 || wordlen > 31) { set_err("alphabet_len 1..36, wordlen 1..31"); return -1; }
  if (n_pairs < 0 || (n_pairs && (!pairs || !out))) { set_err("bad arguments"); return -1; }
  if (!(len_coeff > 0) || !(radius_coeff > 0) || !(word_p_null > 0)) { set_err("coefficients must be positive"); return -1; }
  uint64_t kmax = 1; int kbits = 0;
  for (int i = 0; i < wordlen; i++) { if (kmax > (1ull << 62) / (uint64_t)alphabet_len) { set_err("alphabet_len ^ wordlen must be below 2^62"); return -1; } kmax *= (uint64_t)alphabet_len; }
  while (((kmax - 1) >> kbits) != 0) kbits++;
  if (kbits == 0) kbits = 1;
  for (int64_t p = 0; p < n_pairs; p++) {
    const pw_read_pair& r = pairs[p];
    if (r.s_len < 0 || r.t_len < 0 || r.s_off + (uint64_t)r.s_len > arena_bytes || r.t_off + (uint64_t)r.t_len > arena_bytes) {
      set_err("a read lies outside the arena"); return -1;
    }
  }
  for (uint64_t i = 0; i < arena_bytes; i++) if (arena[i] >= alphabet_len) { set_err("letter outside the alphabet"); return -1; }
  g_ms = 0.0;
  if (n_pairs == 0) return 0;
  OV_CHECK(hipSetDevice(device));
  Buf d_arena;
  if (d_arena.alloc((size_t)arena_bytes + 64)) return -1;
  OV_CHECK(hipMemcpy(d_arena.p, arena, (size_t)arena_bytes, hipMemcpyHostToDevice));
  Ev e0, e1;
  if (e0.make() || e1.make()) return -1;
  const hipEvent_t ev0 = e0.e, ev1 = e1.e;
  // chunks: at most 2^31 histogram entries, 2^31 k-mers per side and pair ids that fit beside the k-mer
  const uint64_t lim = 1ull << 31;
  const int64_t max_pairs_bits = 62 - kbits;
  float ms = 0.f;
  int rc = 0;
  for (int64_t p0 = 0; p0 < n_pairs && rc == 0;) {
    uint64_t hb = 0, cs = 0, ct = 0; int64_t p1 = p0;
    while (p1 < n_pairs) {
      const uint64_t h = (uint64_t)pairs[p1].s_len + (uint64_t)pairs[p1].t_len + 1;
      if (p1 > p0 && (hb + h > lim || cs + (uint64_t)pairs[p1].s_len > lim || ct + (uint64_t)pairs[p1].t_len > lim ||
                      (max_pairs_bits < 40 && (p1 - p0 + 1) >= (1ll << max_pairs_bits)))) break;
      hb += h; cs += (uint64_t)pairs[p1].s_len; ct += (uint64_t)pairs[p1].t_len; p1++;
    }
    rc = run_chunk((const uint8_t*)d_arena.p, pairs + p0, p1 - p0, alphabet_len, wordlen, kbits,
                   BandConst{len_coeff, radius_coeff, word_p_null}, out + p0, ev0, ev1, &ms);
    p0 = p1;
  }
  g_ms = (double)ms;
  return rc;
}

}  // extern "C"
