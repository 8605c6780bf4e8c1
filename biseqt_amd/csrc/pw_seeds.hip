// pw_seeds.hip -- exact-match k-mer seeds of one pair of sequences on gfx950 (C ABI: include/pw_seeds.h).
//
// What the reference does with SQLite tables and Python generators (kmers.py:437-509, seeds.py:117-237) is a
// sort-merge join:
//   K5a k_encode   one thread per position: the k-mer as an integer in base L (kmers.py:164-210), or the
//                  "masked" key L^k when its letter set equals a mask set (kmers.py:232-236)
//       sort       (k-mer, position) of S and of T by k-mer -- stable LSD radix sort (rocPRIM), so positions stay
//                  ascending inside a k-mer, which is the reference's (seqid, pos) hit order
//   K5b k_match    one thread per sorted S element: its run of equal k-mers in T (two binary searches) -> the
//                  number of rows it contributes; exclusive scan -> row offsets
//   K5c k_expand   one thread per row: (d, a) = (i - j, i + j).  Rows come out in the reference's rowid order
//                  (k-mer asc, i asc, j asc) by construction, with no further sort
//   K5d k_count    COUNT(*) in a (d, a) band: predicate + wave reduction + one atomic per wave
// A self comparison (seeds.py:33,141-143) joins S with itself: per k-mer run the pairs i < j in combination
// order, then the trivial pairs -- an element contributes the pairs with the later elements of its run, and the
// LAST element of a run contributes the run's trivial rows.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <cstring>
#include <string>
#include <vector>
#include <rocprim/rocprim.hpp>

#include "../../include/pw_seeds.h"

namespace {

thread_local std::string g_err;
void set_err(const std::string& s) { g_err = s; }
#define SD_CHECK(call)                                                                       \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      set_err(std::string(#call) + ": " + hipGetErrorString(e_));                            \
      return -1;                                                                             \
    }                                                                                        \
  } while (0)

constexpr int kMaxMasks = 16;
struct MaskSets { uint64_t set[kMaxMasks]; int n; };

// ---- K5a ------------------------------------------------------------------------------------------------
// Key type K: 32 bits whenever L^k (the masked key included) fits -- DNA words up to k = 15 -- which cuts the sort's
// traffic from 12 to 8 bytes per k-mer and pass; 64 bits otherwise.
template <typename K>
__global__ __launch_bounds__(256) void k_encode(const uint8_t* __restrict__ seq, int64_t n, int k, int L,
                                                uint64_t kinv, MaskSets ms, K* __restrict__ keys,
                                                uint32_t* __restrict__ pos) {
  __shared__ uint8_t tile[256 + 64];
  const int64_t base = (int64_t)blockIdx.x * 256;
  const int64_t nk = n - k + 1;
  for (int t = (int)threadIdx.x; t < 256 + k - 1; t += 256) {
    const int64_t p = base + t;
    tile[t] = p < n ? seq[p] : 0;
  }
  __syncthreads();
  const int64_t p = base + threadIdx.x;
  if (p >= nk) return;
  uint64_t v = 0, lets = 0;
  for (int t = 0; t < k; t++) {
    const uint32_t c = tile[threadIdx.x + t];
    v = v * (uint64_t)L + c;
    lets |= 1ull << c;
  }
  bool masked = false;
  for (int i = 0; i < ms.n; i++) masked |= lets == ms.set[i];
  keys[p] = (K)(masked ? kinv : v);
  pos[p] = (uint32_t)p;
}

template <typename K>
__device__ __forceinline__ int64_t lower_bound_k(const K* __restrict__ a, int64_t n, K key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}
template <typename K>
__device__ __forceinline__ int64_t upper_bound_k(const K* __restrict__ a, int64_t n, K key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] <= key) lo = mid + 1; else hi = mid; }
  return lo;
}
__device__ __forceinline__ int64_t lower_bound_u64(const uint64_t* __restrict__ a, int64_t n, uint64_t key) { return lower_bound_k<uint64_t>(a, n, key); }
__device__ __forceinline__ int64_t upper_bound_u64(const uint64_t* __restrict__ a, int64_t n, uint64_t key) { return upper_bound_k<uint64_t>(a, n, key); }

// ---- K5b ------------------------------------------------------------------------------------------------
// other = sorted keys of T (or of S itself for a self comparison).  Two ways to find an element's run [lo, hi) in it:
//   * a direct-address table tab[key] = first index of `key` in `other` (k_table_fill) when the key space is small
//     (DNA, k <= 13: <= 256 MB of table, resident in the Infinity Cache): S is sorted too, so neighbouring threads read
//     neighbouring table entries -- two coalesced 4-byte reads instead of 2 log2(n) dependent ones;
//   * two binary searches otherwise.
template <typename K>
__global__ __launch_bounds__(256) void k_match(const K* __restrict__ ks, int64_t ns,
                                               const K* __restrict__ other, int64_t no, uint64_t kinv,
                                               int self, const uint32_t* __restrict__ tab, uint32_t* __restrict__ lo_out,
                                               uint64_t* __restrict__ cnt) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= ns) return;
  const K key = ks[e];
  if ((uint64_t)key >= kinv) { lo_out[e] = 0; cnt[e] = 0; return; }
  int64_t lo, hi;
  if (tab != nullptr) { lo = tab[(uint64_t)key]; hi = tab[(uint64_t)key + 1]; }
  else { lo = lower_bound_k<K>(other, no, key); hi = upper_bound_k<K>(other, no, key); }
  lo_out[e] = (uint32_t)lo;
  if (!self) cnt[e] = (uint64_t)(hi - lo);
  else cnt[e] = (uint64_t)(hi - 1 - e) + (e == hi - 1 ? (uint64_t)(hi - lo) : 0ull);
}
// tab[q] = number of elements of `other` below q, for q = 0 .. kinv + 1: element i (the first of its run) fills the
// keys after the previous run's key up to its own; one extra thread fills the tail.  (Used only when the keys are dense
// enough that these gaps are short: see pw_seeds_build.)
template <typename K>
__global__ __launch_bounds__(256) void k_table_fill(const K* __restrict__ other, int64_t no, uint64_t kinv, uint32_t* __restrict__ tab) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i > no) return;
  const int64_t prev = i > 0 ? (int64_t)other[i - 1] : -1;
  const int64_t cur = i < no ? (int64_t)other[i] : (int64_t)kinv + 1;
  for (int64_t q = prev + 1; q <= cur; q++) tab[q] = (uint32_t)i;
}

// ---- K5c ------------------------------------------------------------------------------------------------
// One thread per row, kExpRows rows per workgroup.  The element a row belongs to is found by a binary search over the
// row offsets -- inside the window of elements that the workgroup's rows span (two searches per workgroup over the whole
// array, then ~10 cache-resident steps per row instead of log2(n) scattered ones).
constexpr int kExpRows = 2048;
template <typename K>
__global__ __launch_bounds__(256) void k_expand(const uint64_t* __restrict__ off, int64_t ns, int64_t nrows,
                                                const uint32_t* __restrict__ ps, const uint32_t* __restrict__ po,
                                                const uint32_t* __restrict__ lo_in, const K* __restrict__ ks,
                                                int self, int2* __restrict__ rows) {
  __shared__ int64_t win[2];
  const int64_t o0 = (int64_t)blockIdx.x * kExpRows;
  const int64_t olast = (o0 + kExpRows < nrows ? o0 + kExpRows : nrows) - 1;
  if (threadIdx.x == 0) win[0] = upper_bound_u64(off, ns, (uint64_t)o0) - 1;
  if (threadIdx.x == 64) win[1] = upper_bound_u64(off, ns, (uint64_t)olast) - 1;
  __syncthreads();
  const int64_t e0 = win[0], nwin = win[1] - win[0] + 1;
#pragma unroll 1
  for (int q = 0; q < kExpRows / 256; q++) {
    const int64_t o = o0 + q * 256 + threadIdx.x;
    if (o >= nrows) return;
    const int64_t e = e0 + upper_bound_u64(off + e0, nwin, (uint64_t)o) - 1;      // last element whose first row is <= o
    const int64_t r = o - (int64_t)off[e];
    int32_t i, j;
    if (!self) {
      i = (int32_t)ps[e];
      j = (int32_t)po[(int64_t)lo_in[e] + r];
    } else {
      // the run of e is [lo, hi): e pairs with e + 1 .. hi - 1; the last element of the run then lists (x, x)
      const int64_t lo = lo_in[e];
      const K key = ks[e];
      const bool last = e + 1 >= ns || ks[e + 1] != key;
      if (!last) { i = (int32_t)ps[e]; j = (int32_t)ps[e + 1 + r]; }
      else { i = (int32_t)ps[lo + r]; j = i; }
    }
    rows[o] = make_int2(i - j, i + j);
  }
}

// ---- K5d ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_count(const int2* __restrict__ rows, int64_t nrows, int have_d, int dmin,
                                               int dmax, int have_a, int amin, int amax,
                                               unsigned long long* __restrict__ out) {
  unsigned long long c = 0;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < nrows; o += (int64_t)gridDim.x * 256) {
    const int2 r = rows[o];
    const bool ok = (!have_d || (r.x >= dmin && r.x <= dmax)) && (!have_a || (r.y >= amin && r.y <= amax));
    c += ok ? 1ull : 0ull;
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) c += __shfl_xor(c, s, 64);
  if ((threadIdx.x & 63u) == 0 && c) atomicAdd(out, c);
}

// ---- K6: neighbours on the scaled diagonal axis (blot.py:521-527) -------------------------------------------
__global__ __launch_bounds__(256) void k_scale(const int2* __restrict__ rows, int64_t nrows, const double* __restrict__ radius,
                                               int nT, double* __restrict__ x) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= nrows) return;
  const int d = rows[o].x;
  x[o] = (double)d / radius[d + nT];
}
// xs sorted ascending.  The two predicates are the KD-tree's own test fl(|x - x'|) <= 1, one side each; both are
// monotone along xs (rounding is monotone), so a binary search on the predicate itself is exact.
__global__ __launch_bounds__(256) void k_neigh(const double* __restrict__ x, const double* __restrict__ xs, int64_t nrows,
                                               int32_t* __restrict__ counts) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= nrows) return;
  const double v = x[o];
  int64_t lo = 0, hi = nrows;                     // first index with v - xs[idx] <= 1
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (!(v - xs[mid] <= 1.0)) lo = mid + 1; else hi = mid; }
  const int64_t first = lo;
  lo = 0; hi = nrows;                             // first index with NOT (xs[idx] - v <= 1)
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (xs[mid] - v <= 1.0) lo = mid + 1; else hi = mid; }
  counts[o] = (int32_t)(lo - first - 1);
}

// Self comparison: the seeds a WordBlot iterates are SeedIndex.seeds(exclude_trivial=True) (seeds.py:186-197): every
// non-trivial row (i < j, d < 0) followed by its mirror image, trivial rows (d = 0) dropped.
__global__ __launch_bounds__(256) void k_self_flag(const int2* __restrict__ rows, int64_t n, uint64_t* __restrict__ flag) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o < n) flag[o] = rows[o].x != 0 ? 1ull : 0ull;
}
__global__ __launch_bounds__(256) void k_self_points(const int2* __restrict__ rows, int64_t n, const uint64_t* __restrict__ pos,
                                                     int2* __restrict__ pts) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= n) return;
  const int2 r = rows[o];
  if (r.x == 0) return;
  pts[2 * pos[o]] = r;
  pts[2 * pos[o] + 1] = make_int2(-r.x, r.y);
}

// ---- K7: the neighbourhood graph of the seeds (blot.py:343-374) and its connected components (:452-468) ----------
__global__ __launch_bounds__(256) void k_graph_keys(const int2* __restrict__ rows, int64_t n, int nT, uint64_t* __restrict__ keys,
                                                    uint32_t* __restrict__ vals) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= n) return;
  const int2 r = rows[o];
  keys[o] = ((uint64_t)(uint32_t)(r.x + nT) << 32) | (uint32_t)r.y;
  vals[o] = (uint32_t)o;
}
__global__ __launch_bounds__(256) void k_graph_dstart(const uint64_t* __restrict__ keys, int64_t n, int64_t nd, uint32_t* __restrict__ dstart) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q > nd) return;
  dstart[q] = (uint32_t)lower_bound_u64(keys, n, (uint64_t)q << 32);
}
// One thread per seed in (d, a) order.  For every diagonal d' that passes the KD-tree's test on the scaled axis,
// fl(|fl(d c) - fl(d' c)|) <= R, the seeds with |a - a'| <= R form one contiguous piece of that diagonal's run.
template <bool FILL>
__global__ __launch_bounds__(256) void k_graph_scan(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ order, int64_t n,
                                                    const uint32_t* __restrict__ dstart, int nd, int nT, double c, double R, int win,
                                                    uint32_t* __restrict__ cnt, const uint64_t* __restrict__ off,
                                                    uint32_t* __restrict__ adj) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= n) return;
  const uint64_t key = keys[s];
  const uint32_t o = order[s];
  const int q = (int)(key >> 32);
  const int64_t a = (int64_t)(uint32_t)key;
  const double X = (double)(q - nT) * c;
  const int64_t ra = (int64_t)floor(R);           // |a - a'| <= R for integers
  uint64_t w = FILL ? off[o] : 0;
  uint32_t total = 0;
  const int q0 = q - win < 0 ? 0 : q - win, q1 = q + win > nd - 1 ? nd - 1 : q + win;
  for (int qq = q0; qq <= q1; qq++) {
    const double Xp = (double)(qq - nT) * c;
    if (!(fabs(X - Xp) <= R)) continue;
    const int64_t b = dstart[qq], e = dstart[qq + 1];
    if (b == e) continue;
    const int64_t alo = a - ra < 0 ? 0 : a - ra, ahi = a + ra;
    const uint64_t klo = ((uint64_t)(uint32_t)qq << 32) | (uint64_t)alo;
    const uint64_t khi = ((uint64_t)(uint32_t)qq << 32) | (uint64_t)(ahi > 0xffffffffll ? 0xffffffffll : ahi);
    const int64_t lo = b + lower_bound_u64(keys + b, e - b, klo);
    const int64_t hi = b + upper_bound_u64(keys + b, e - b, khi);
    if (!FILL) total += (uint32_t)(hi - lo);
    else for (int64_t t = lo; t < hi; t++) { const uint32_t v = order[t]; if (v != o) adj[w++] = v; }
  }
  if (!FILL) cnt[o] = total - 1;                  // its own entry is removed (blot.py:371-372)
}
__global__ __launch_bounds__(256) void k_widen(const uint32_t* __restrict__ in, int64_t n, uint64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[i];
}
__device__ __forceinline__ int cc_root(const int* __restrict__ parent, int v) {
  int p = parent[v];
  while (p != v) { v = p; p = parent[v]; }
  return v;
}
__global__ __launch_bounds__(256) void k_cc_init(const uint8_t* __restrict__ avail, int64_t n, int* __restrict__ parent) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) parent[i] = avail[i] ? (int)i : -1;
}
__global__ __launch_bounds__(256) void k_cc_hook(const uint64_t* __restrict__ off, const uint32_t* __restrict__ cnt,
                                                 const uint32_t* __restrict__ adj, int64_t n, int* __restrict__ parent,
                                                 int* __restrict__ changed) {
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (u >= n || parent[u] < 0) return;
  const uint64_t b = off[u], e = b + cnt[u];
  for (uint64_t t = b; t < e; t++) {
    const int v = (int)adj[t];
    if (parent[v] < 0) continue;
    const int ru = cc_root(parent, (int)u), rv = cc_root(parent, v);
    if (ru != rv) { atomicMin(&parent[ru > rv ? ru : rv], ru > rv ? rv : ru); *changed = 1; }
  }
}
__global__ __launch_bounds__(256) void k_cc_compress(int64_t n, int* __restrict__ parent) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && parent[i] >= 0) parent[i] = cc_root(parent, (int)i);
}

__global__ void k_total(const uint64_t* __restrict__ off, const uint64_t* __restrict__ cnt, int64_t ns,
                        unsigned long long* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = ns > 0 ? off[ns - 1] + cnt[ns - 1] : 0ull;
}

struct DevBuf {
  void* p = nullptr; size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    SD_CHECK(hipMalloc(&p, bytes ? bytes : 16));
    cap = bytes;
    return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct pw_seed_index {
  int device = 0, L = 0, k = 0, self = 0, bits = 0;
  bool key32 = false;                   // L^k fits 32 bits: 4-byte keys
  DevBuf tab;                           // direct-address table of the join (small key spaces)
  int64_t nS = 0, nT = 0, nkS = 0, nkT = 0, nrows = -1;
  uint64_t kinv = 0;
  MaskSets ms;
  DevBuf dS, dT, keys_in, keys_s, keys_t, pos_in, pos_s, pos_t, lo, cnt, off, rows, tmp, scalar;
  DevBuf g_keys, g_order, g_dstart, g_cnt, g_off, g_adj, g_pts;     // neighbourhood graph (K7)
  int64_t g_edges = -1, g_npts = -1;    // g_npts: points of the graph (= rows, or the mirrored non-trivial rows of a self comparison)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float ms_build = 0.f;
};

template <typename K>
static int encode_sort(pw_seed_index* x, const uint8_t* seq, int64_t n, int64_t nk, DevBuf& keys_out, DevBuf& pos_out,
                       hipStream_t st) {
  if (keys_out.ensure((size_t)std::max<int64_t>(nk, 1) * sizeof(K)) != 0 || pos_out.ensure((size_t)std::max<int64_t>(nk, 1) * 4) != 0) return -1;
  if (nk <= 0) return 0;
  if (x->keys_in.ensure((size_t)nk * sizeof(K)) != 0 || x->pos_in.ensure((size_t)nk * 4) != 0) return -1;
  hipLaunchKernelGGL((k_encode<K>), dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st, seq, n, x->k, x->L, x->kinv, x->ms,
                     (K*)x->keys_in.p, (uint32_t*)x->pos_in.p);
  size_t tb = 0;
  SD_CHECK(rocprim::radix_sort_pairs(nullptr, tb, (const K*)x->keys_in.p, (K*)keys_out.p,
                                     (const uint32_t*)x->pos_in.p, (uint32_t*)pos_out.p, (size_t)nk, 0u,
                                     (unsigned)x->bits, st));
  if (x->tmp.ensure(tb) != 0) return -1;
  SD_CHECK(rocprim::radix_sort_pairs(x->tmp.p, tb, (const K*)x->keys_in.p, (K*)keys_out.p,
                                     (const uint32_t*)x->pos_in.p, (uint32_t*)pos_out.p, (size_t)nk, 0u,
                                     (unsigned)x->bits, st));
  return 0;
}

// encode + sort both sequences, join them: everything of pw_seeds_build that depends on the key type
template <typename K>
static int build_join(pw_seed_index* x, hipStream_t st, int64_t ns, bool count_only, unsigned long long total) {
  if (count_only) {
    if (encode_sort<K>(x, (const uint8_t*)x->dS.p, x->nS, x->nkS, x->keys_s, x->pos_s, st) != 0) return -1;
    if (!x->self && encode_sort<K>(x, (const uint8_t*)x->dT.p, x->nT, x->nkT, x->keys_t, x->pos_t, st) != 0) return -1;
    if (ns <= 0) return 0;
    const K* other = x->self ? (const K*)x->keys_s.p : (const K*)x->keys_t.p;
    const int64_t no = x->self ? ns : x->nkT;
    // the direct-address table pays when the key space is small and dense enough: at most 2^26 keys (256 MB of table) and on
    // average no more than 64 keys between two consecutive elements of `other` (k_table_fill walks those gaps serially)
    const uint32_t* tab = nullptr;
    if (x->key32 && x->kinv <= (1ull << 26) && no > 0 && x->kinv / (uint64_t)no <= 64) {
      if (x->tab.ensure((size_t)(x->kinv + 2) * 4) != 0) return -1;
      hipLaunchKernelGGL((k_table_fill<K>), dim3((unsigned)((no + 256) / 256)), dim3(256), 0, st, other, no, x->kinv, (uint32_t*)x->tab.p);
      tab = (const uint32_t*)x->tab.p;
    }
    hipLaunchKernelGGL((k_match<K>), dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, st, (const K*)x->keys_s.p, ns,
                       other, no, x->kinv, x->self, tab, (uint32_t*)x->lo.p, (uint64_t*)x->cnt.p);
    return 0;
  }
  hipLaunchKernelGGL((k_expand<K>), dim3((unsigned)((total + kExpRows - 1) / kExpRows)), dim3(256), 0, st, (const uint64_t*)x->off.p, ns,
                     (int64_t)total, (const uint32_t*)x->pos_s.p,
                     x->self ? (const uint32_t*)x->pos_s.p : (const uint32_t*)x->pos_t.p, (const uint32_t*)x->lo.p,
                     (const K*)x->keys_s.p, x->self, (int2*)x->rows.p);
  return 0;
}

extern "C" {

const char* pw_seeds_last_error(void) { return g_err.c_str(); }

pw_seed_index* pw_seeds_create(int device, const uint8_t* S, int64_t nS, const uint8_t* T, int64_t nT,
                               int alphabet_len, int wordlen, const uint64_t* mask_sets, int n_masks,
                               int self_comp) {
  if (alphabet_len < 1 || alphabet_len > 36) { set_err("alphabet_len must be 1..36 (kmers.py:266)"); return nullptr; }
  if (wordlen < 1 || wordlen > 31) { set_err("wordlen must be 1..31 (kmers.py:269)"); return nullptr; }
  if (n_masks < 0 || n_masks > kMaxMasks) { set_err("at most 16 mask sets"); return nullptr; }
  if (nS < 0 || nT < 0 || nS >= (1ll << 31) || nT >= (1ll << 31)) { set_err("sequence length out of range"); return nullptr; }
  // L^k must fit: the masked key is L^k itself
  long double lk = 1; for (int i = 0; i < wordlen; i++) lk *= alphabet_len;
  if (lk >= (long double)(1ull << 62)) { set_err("alphabet_len ^ wordlen must be below 2^62"); return nullptr; }
  for (int64_t i = 0; i < nS; i++) if (S[i] >= alphabet_len) { set_err("letter outside the alphabet in S"); return nullptr; }
  if (self_comp < 0) self_comp = (nS == nT && (nS == 0 || memcmp(S, T, (size_t)nS) == 0)) ? 1 : 0;
  if (!self_comp) for (int64_t i = 0; i < nT; i++) if (T[i] >= alphabet_len) { set_err("letter outside the alphabet in T"); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { set_err("hipSetDevice failed"); return nullptr; }
  pw_seed_index* x = new pw_seed_index();
  x->device = device; x->L = alphabet_len; x->k = wordlen; x->self = self_comp;
  x->nS = nS; x->nT = self_comp ? nS : nT;
  x->nkS = nS >= wordlen ? nS - wordlen + 1 : 0;
  x->nkT = self_comp ? x->nkS : (nT >= wordlen ? nT - wordlen + 1 : 0);
  uint64_t kinv = 1; for (int i = 0; i < wordlen; i++) kinv *= (uint64_t)alphabet_len;
  x->kinv = kinv;
  // sort width: the largest key that occurs -- L^k - 1, or the masked key L^k when mask sets are given (for DNA words
  // without masks that is 2k bits: k = 12 sorts in three 8-bit passes instead of four)
  const uint64_t kmax = n_masks > 0 ? kinv : (kinv > 1 ? kinv - 1 : 1);
  x->bits = 1; while ((kmax >> x->bits) != 0) x->bits++;
  x->key32 = kinv < 0xffffffffull;
  x->ms.n = n_masks;
  for (int i = 0; i < kMaxMasks; i++) x->ms.set[i] = i < n_masks ? mask_sets[i] : 0;
  auto fail = [&](const char* what) { if (g_err.empty()) set_err(what); pw_seeds_destroy(x); return (pw_seed_index*)nullptr; };
  if (x->dS.ensure((size_t)nS + 64) != 0) return fail("hipMalloc");
  if (nS && hipMemcpy(x->dS.p, S, (size_t)nS, hipMemcpyHostToDevice) != hipSuccess) return fail("H2D of S failed");
  if (!self_comp) {
    if (x->dT.ensure((size_t)nT + 64) != 0) return fail("hipMalloc");
    if (nT && hipMemcpy(x->dT.p, T, (size_t)nT, hipMemcpyHostToDevice) != hipSuccess) return fail("H2D of T failed");
  }
  if (hipEventCreate(&x->ev0) != hipSuccess || hipEventCreate(&x->ev1) != hipSuccess) return fail("hipEventCreate");
  return x;
}

int pw_seeds_build(pw_seed_index* x, int64_t max_rows, void* stream) {
  if (!x) { set_err("null index"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  SD_CHECK(hipSetDevice(x->device));
  if (max_rows <= 0) max_rows = (1ll << 31) - 1;
  x->nrows = -1; x->g_edges = -1; x->g_npts = -1;
  SD_CHECK(hipEventRecord(x->ev0, st));
  const int64_t ns = x->nkS;
  if (x->scalar.ensure(16) != 0) return -1;
  if (ns > 0 && (x->lo.ensure((size_t)ns * 4) != 0 || x->cnt.ensure((size_t)ns * 8) != 0 || x->off.ensure((size_t)ns * 8) != 0)) return -1;
  if ((x->key32 ? build_join<uint32_t>(x, st, ns, true, 0) : build_join<uint64_t>(x, st, ns, true, 0)) != 0) return -1;
  unsigned long long total = 0;
  if (ns > 0) {
    size_t tb = 0;
    SD_CHECK(rocprim::exclusive_scan(nullptr, tb, (const uint64_t*)x->cnt.p, (uint64_t*)x->off.p, (uint64_t)0, (size_t)ns,
                                     rocprim::plus<uint64_t>(), st));
    if (x->tmp.ensure(tb) != 0) return -1;
    SD_CHECK(rocprim::exclusive_scan(x->tmp.p, tb, (const uint64_t*)x->cnt.p, (uint64_t*)x->off.p, (uint64_t)0, (size_t)ns,
                                     rocprim::plus<uint64_t>(), st));
    hipLaunchKernelGGL(k_total, dim3(1), dim3(64), 0, st, (const uint64_t*)x->off.p, (const uint64_t*)x->cnt.p, ns,
                       (unsigned long long*)x->scalar.p);
    SD_CHECK(hipMemcpyAsync(&total, x->scalar.p, 8, hipMemcpyDeviceToHost, st));
    SD_CHECK(hipStreamSynchronize(st));
  }
  if ((int64_t)total > max_rows) {
    char msg[160];
    snprintf(msg, sizeof msg, "the seeds table would hold %llu rows (limit %lld): raise max_rows or the word length", total, (long long)max_rows);
    set_err(msg);
    return -1;
  }
  if (x->rows.ensure((size_t)std::max<unsigned long long>(total, 1) * 8) != 0) return -1;
  if (total > 0 && (x->key32 ? build_join<uint32_t>(x, st, ns, false, total) : build_join<uint64_t>(x, st, ns, false, total)) != 0) return -1;
  SD_CHECK(hipEventRecord(x->ev1, st));
  SD_CHECK(hipEventSynchronize(x->ev1));
  SD_CHECK(hipEventElapsedTime(&x->ms_build, x->ev0, x->ev1));
  SD_CHECK(hipGetLastError());
  x->nrows = (int64_t)total;
  return 0;
}

int64_t pw_seeds_num_rows(const pw_seed_index* x) { return x ? x->nrows : -1; }
int pw_seeds_is_self(const pw_seed_index* x) { return x ? x->self : -1; }
const int32_t* pw_seeds_rows_device(const pw_seed_index* x) { return (x && x->nrows >= 0) ? (const int32_t*)x->rows.p : nullptr; }
double pw_seeds_build_ms(const pw_seed_index* x) { return x ? (double)x->ms_build : -1.0; }
int64_t pw_seeds_algorithmic_bytes(const pw_seed_index* x) {
  if (!x || x->nrows < 0) return -1;
  return x->nS + (x->self ? 0 : x->nT) + 8 * x->nrows;
}

int pw_seeds_rows(const pw_seed_index* x, int32_t* da, int64_t cap) {
  if (!x || x->nrows < 0) { set_err("pw_seeds_rows before a successful pw_seeds_build"); return -1; }
  if (cap < x->nrows) { set_err("pw_seeds_rows: capacity too small"); return -1; }
  SD_CHECK(hipSetDevice(x->device));
  if (x->nrows) SD_CHECK(hipMemcpy(da, x->rows.p, (size_t)x->nrows * 8, hipMemcpyDeviceToHost));
  return 0;
}

int64_t pw_seeds_count(const pw_seed_index* x, int have_d, int32_t dmin, int32_t dmax, int have_a, int32_t amin,
                       int32_t amax) {
  if (!x || x->nrows < 0) { set_err("pw_seeds_count before a successful pw_seeds_build"); return -1; }
  if (!have_d && !have_a) return x->nrows;
  if (x->nrows == 0) return 0;
  SD_CHECK(hipSetDevice(x->device));
  unsigned long long* out = (unsigned long long*)x->scalar.p + 1;
  SD_CHECK(hipMemsetAsync(out, 0, 8, nullptr));
  const int64_t blocks = std::min<int64_t>((x->nrows + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(k_count, dim3((unsigned)blocks), dim3(256), 0, nullptr, (const int2*)x->rows.p, x->nrows, have_d,
                     dmin, dmax, have_a, amin, amax, out);
  unsigned long long c = 0;
  SD_CHECK(hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost));
  return (int64_t)c;
}

int64_t pw_seeds_kmers(const pw_seed_index* xc, int which, int64_t* out, int64_t cap) {
  pw_seed_index* x = const_cast<pw_seed_index*>(xc);
  if (!x) { set_err("null index"); return -1; }
  const bool t = which != 0 && !x->self;
  const int64_t n = t ? x->nT : x->nS, nk = t ? x->nkT : x->nkS;
  if (cap < nk) { set_err("pw_seeds_kmers: capacity too small"); return -1; }
  if (nk <= 0) return 0;
  SD_CHECK(hipSetDevice(x->device));
  DevBuf keys, pos;
  if (keys.ensure((size_t)nk * 8) != 0 || pos.ensure((size_t)nk * 4) != 0) { keys.release(); pos.release(); return -1; }
  hipLaunchKernelGGL((k_encode<uint64_t>), dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, nullptr,
                     (const uint8_t*)(t ? x->dT.p : x->dS.p), n, x->k, x->L, x->kinv, x->ms, (uint64_t*)keys.p, (uint32_t*)pos.p);
  std::vector<uint64_t> h((size_t)nk);
  const hipError_t e = hipMemcpy(h.data(), keys.p, (size_t)nk * 8, hipMemcpyDeviceToHost);
  keys.release(); pos.release();
  if (e != hipSuccess) { set_err("D2H of the k-mers failed"); return -1; }
  for (int64_t i = 0; i < nk; i++) out[i] = h[(size_t)i] >= x->kinv ? -1 : (int64_t)h[(size_t)i];
  return nk;
}

int pw_seeds_band_neighbours(const pw_seed_index* xc, const double* radius, int64_t n_radius, int32_t* counts, int64_t cap) {
  pw_seed_index* x = const_cast<pw_seed_index*>(xc);
  if (!x || x->nrows < 0) { set_err("pw_seeds_band_neighbours before a successful pw_seeds_build"); return -1; }
  if (x->self) { set_err("pw_seeds_band_neighbours is not defined for a self comparison"); return -1; }
  if (n_radius != x->nS + x->nT + 1) { set_err("radius table must hold nS + nT + 1 entries (d = -nT .. nS)"); return -1; }
  if (cap < x->nrows) { set_err("pw_seeds_band_neighbours: capacity too small"); return -1; }
  for (int64_t i = 0; i < n_radius; i++) if (!(radius[i] > 0)) { set_err("band radii must be positive"); return -1; }
  const int64_t n = x->nrows;
  if (n == 0) return 0;
  SD_CHECK(hipSetDevice(x->device));
  DevBuf rad, xu, xs, cn;
  int rc = -1;
  do {
    if (rad.ensure((size_t)n_radius * 8) != 0 || xu.ensure((size_t)n * 8) != 0 || xs.ensure((size_t)n * 8) != 0 || cn.ensure((size_t)n * 4) != 0) break;
    if (hipMemcpy(rad.p, radius, (size_t)n_radius * 8, hipMemcpyHostToDevice) != hipSuccess) { set_err("H2D of the radius table failed"); break; }
    hipLaunchKernelGGL(k_scale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (const int2*)x->rows.p, n,
                       (const double*)rad.p, (int)x->nT, (double*)xu.p);
    size_t tb = 0;
    if (rocprim::radix_sort_keys(nullptr, tb, (const double*)xu.p, (double*)xs.p, (size_t)n, 0u, 64u, (hipStream_t) nullptr) != hipSuccess) { set_err("radix_sort_keys (size) failed"); break; }
    if (x->tmp.ensure(tb) != 0) break;
    if (rocprim::radix_sort_keys(x->tmp.p, tb, (const double*)xu.p, (double*)xs.p, (size_t)n, 0u, 64u, (hipStream_t) nullptr) != hipSuccess) { set_err("radix_sort_keys failed"); break; }
    hipLaunchKernelGGL(k_neigh, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (const double*)xu.p, (const double*)xs.p, n, (int32_t*)cn.p);
    if (hipMemcpy(counts, cn.p, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) { set_err("D2H of the neighbour counts failed"); break; }
    rc = 0;
  } while (0);
  rad.release(); xu.release(); xs.release(); cn.release();
  return rc;
}

int64_t pw_seeds_graph_build(pw_seed_index* x, double d_coeff, double radius) {
  if (!x || x->nrows < 0) { set_err("pw_seeds_graph_build before a successful pw_seeds_build"); return -1; }
  if (!(d_coeff > 0) || !(radius >= 0)) { set_err("d_coeff must be positive and radius non-negative"); return -1; }
  x->g_edges = -1; x->g_npts = -1;
  SD_CHECK(hipSetDevice(x->device));
  const int2* pts = (const int2*)x->rows.p;
  int64_t n = x->nrows;
  if (x->self && n > 0) {
    // the point list of a self comparison: non-trivial rows and their mirror images, in table order
    DevBuf flag, pos;
    if (flag.ensure((size_t)n * 8) != 0 || pos.ensure((size_t)n * 8) != 0) { flag.release(); pos.release(); return -1; }
    const dim3 g((unsigned)((n + 255) / 256)), bl(256);
    hipLaunchKernelGGL(k_self_flag, g, bl, 0, nullptr, (const int2*)x->rows.p, n, (uint64_t*)flag.p);
    size_t tbs = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, tbs, (const uint64_t*)flag.p, (uint64_t*)pos.p, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), (hipStream_t) nullptr);
    if (e == hipSuccess && x->tmp.ensure(tbs) != 0) e = hipErrorOutOfMemory;
    if (e == hipSuccess) e = rocprim::exclusive_scan(x->tmp.p, tbs, (const uint64_t*)flag.p, (uint64_t*)pos.p, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), (hipStream_t) nullptr);
    uint64_t lp = 0, lf = 0;
    if (e == hipSuccess) e = hipMemcpy(&lp, (uint64_t*)pos.p + (n - 1), 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&lf, (uint64_t*)flag.p + (n - 1), 8, hipMemcpyDeviceToHost);
    const int64_t np = (int64_t)(2 * (lp + lf));
    if (e == hipSuccess && x->g_pts.ensure((size_t)std::max<int64_t>(np, 1) * 8) != 0) e = hipErrorOutOfMemory;
    if (e == hipSuccess) hipLaunchKernelGGL(k_self_points, g, bl, 0, nullptr, (const int2*)x->rows.p, n, (const uint64_t*)pos.p, (int2*)x->g_pts.p);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    flag.release(); pos.release();
    if (e != hipSuccess) { set_err("building the point list of the self comparison failed"); return -1; }
    pts = (const int2*)x->g_pts.p; n = np;
  }
  x->g_npts = n;
  if (n == 0) { x->g_edges = 0; return 0; }
  const int64_t nd = x->nS + x->nT + 1;
  const double wd = floor(radius / d_coeff) + 2;
  const int win = wd > (double)nd ? (int)nd : (int)wd;
  DevBuf kin, vin;
  int64_t rc = -1;
  do {
    if (kin.ensure((size_t)n * 8) != 0 || vin.ensure((size_t)n * 4) != 0) break;
    if (x->g_keys.ensure((size_t)n * 8) != 0 || x->g_order.ensure((size_t)n * 4) != 0 || x->g_dstart.ensure((size_t)(nd + 1) * 4) != 0 ||
        x->g_cnt.ensure((size_t)n * 4) != 0 || x->g_off.ensure((size_t)(n + 1) * 8) != 0) break;
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    hipLaunchKernelGGL(k_graph_keys, grid, blk, 0, nullptr, pts, n, (int)x->nT, (uint64_t*)kin.p, (uint32_t*)vin.p);
    int dbits = 1; while (((uint64_t)nd >> dbits) != 0) dbits++;
    size_t tb = 0;
    if (rocprim::radix_sort_pairs(nullptr, tb, (const uint64_t*)kin.p, (uint64_t*)x->g_keys.p, (const uint32_t*)vin.p, (uint32_t*)x->g_order.p,
                                  (size_t)n, 0u, (unsigned)(32 + dbits), (hipStream_t) nullptr) != hipSuccess) { set_err("radix_sort_pairs (size) failed"); break; }
    if (x->tmp.ensure(tb) != 0) break;
    if (rocprim::radix_sort_pairs(x->tmp.p, tb, (const uint64_t*)kin.p, (uint64_t*)x->g_keys.p, (const uint32_t*)vin.p, (uint32_t*)x->g_order.p,
                                  (size_t)n, 0u, (unsigned)(32 + dbits), (hipStream_t) nullptr) != hipSuccess) { set_err("radix_sort_pairs failed"); break; }
    hipLaunchKernelGGL(k_graph_dstart, dim3((unsigned)((nd + 256) / 256)), blk, 0, nullptr, (const uint64_t*)x->g_keys.p, n, nd, (uint32_t*)x->g_dstart.p);
    hipLaunchKernelGGL((k_graph_scan<false>), grid, blk, 0, nullptr, (const uint64_t*)x->g_keys.p, (const uint32_t*)x->g_order.p, n,
                       (const uint32_t*)x->g_dstart.p, (int)nd, (int)x->nT, d_coeff, radius, win, (uint32_t*)x->g_cnt.p,
                       (const uint64_t*)nullptr, (uint32_t*)nullptr);
    // offsets = exclusive scan of the counts (64-bit)
    uint64_t* wide = (uint64_t*)kin.p;            // reuse: n x 8 bytes
    hipLaunchKernelGGL(k_widen, grid, blk, 0, nullptr, (const uint32_t*)x->g_cnt.p, n, wide);
    tb = 0;
    if (rocprim::exclusive_scan(nullptr, tb, (const uint64_t*)wide, (uint64_t*)x->g_off.p, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), (hipStream_t) nullptr) != hipSuccess) { set_err("exclusive_scan (size) failed"); break; }
    if (x->tmp.ensure(tb) != 0) break;
    if (rocprim::exclusive_scan(x->tmp.p, tb, (const uint64_t*)wide, (uint64_t*)x->g_off.p, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), (hipStream_t) nullptr) != hipSuccess) { set_err("exclusive_scan failed"); break; }
    hipLaunchKernelGGL(k_total, dim3(1), dim3(64), 0, nullptr, (const uint64_t*)x->g_off.p, (const uint64_t*)wide, n, (unsigned long long*)x->scalar.p);
    unsigned long long total = 0;
    if (hipMemcpy(&total, x->scalar.p, 8, hipMemcpyDeviceToHost) != hipSuccess) { set_err("D2H of the edge count failed"); break; }
    if (total >= (1ull << 32)) { set_err("the neighbourhood graph has more than 2^32 edges: use a smaller radius"); break; }
    if (hipMemcpy((uint64_t*)x->g_off.p + n, &total, 8, hipMemcpyHostToDevice) != hipSuccess) { set_err("H2D failed"); break; }
    if (x->g_adj.ensure((size_t)std::max<unsigned long long>(total, 1) * 4) != 0) break;
    if (total) hipLaunchKernelGGL((k_graph_scan<true>), grid, blk, 0, nullptr, (const uint64_t*)x->g_keys.p, (const uint32_t*)x->g_order.p, n,
                                  (const uint32_t*)x->g_dstart.p, (int)nd, (int)x->nT, d_coeff, radius, win, (uint32_t*)nullptr,
                                  (const uint64_t*)x->g_off.p, (uint32_t*)x->g_adj.p);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { set_err("the graph kernels failed"); break; }
    rc = (int64_t)total;
  } while (0);
  kin.release(); vin.release();
  x->g_edges = rc;
  return rc;
}

int64_t pw_seeds_graph_num_points(const pw_seed_index* x) { return (x && x->g_edges >= 0) ? x->g_npts : -1; }

int pw_seeds_graph_points(const pw_seed_index* x, int32_t* da, int64_t cap) {
  if (!x || x->g_edges < 0) { set_err("pw_seeds_graph_points before a successful pw_seeds_graph_build"); return -1; }
  if (cap < x->g_npts) { set_err("pw_seeds_graph_points: capacity too small"); return -1; }
  SD_CHECK(hipSetDevice(x->device));
  if (x->g_npts) SD_CHECK(hipMemcpy(da, x->self ? x->g_pts.p : x->rows.p, (size_t)x->g_npts * 8, hipMemcpyDeviceToHost));
  return 0;
}

int pw_seeds_graph_counts(const pw_seed_index* x, int32_t* counts, int64_t cap) {
  if (!x || x->g_edges < 0) { set_err("pw_seeds_graph_counts before a successful pw_seeds_graph_build"); return -1; }
  if (cap < x->g_npts) { set_err("pw_seeds_graph_counts: capacity too small"); return -1; }
  SD_CHECK(hipSetDevice(x->device));
  if (x->g_npts) SD_CHECK(hipMemcpy(counts, x->g_cnt.p, (size_t)x->g_npts * 4, hipMemcpyDeviceToHost));
  return 0;
}

int pw_seeds_graph_fetch(const pw_seed_index* x, int64_t* offsets, int32_t* neighbours) {
  if (!x || x->g_edges < 0) { set_err("pw_seeds_graph_fetch before a successful pw_seeds_graph_build"); return -1; }
  SD_CHECK(hipSetDevice(x->device));
  if (x->g_npts == 0) { offsets[0] = 0; return 0; }
  SD_CHECK(hipMemcpy(offsets, x->g_off.p, (size_t)(x->g_npts + 1) * 8, hipMemcpyDeviceToHost));
  if (x->g_edges) SD_CHECK(hipMemcpy(neighbours, x->g_adj.p, (size_t)x->g_edges * 4, hipMemcpyDeviceToHost));
  return 0;
}

int pw_seeds_graph_components(const pw_seed_index* x, const uint8_t* avail, int32_t* labels) {
  if (!x || x->g_edges < 0) { set_err("pw_seeds_graph_components before a successful pw_seeds_graph_build"); return -1; }
  const int64_t n = x->g_npts;
  if (n == 0) return 0;
  SD_CHECK(hipSetDevice(x->device));
  DevBuf av, par, flag;
  int rc = -1;
  do {
    if (av.ensure((size_t)n) != 0 || par.ensure((size_t)n * 4) != 0 || flag.ensure(16) != 0) break;
    if (hipMemcpy(av.p, avail, (size_t)n, hipMemcpyHostToDevice) != hipSuccess) { set_err("H2D of avail failed"); break; }
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    hipLaunchKernelGGL(k_cc_init, grid, blk, 0, nullptr, (const uint8_t*)av.p, n, (int*)par.p);
    bool ok = true;
    for (int it = 0; it < 10000; it++) {          // every round at least halves the number of roots still to merge
      if (hipMemsetAsync(flag.p, 0, 4, nullptr) != hipSuccess) { ok = false; break; }
      hipLaunchKernelGGL(k_cc_hook, grid, blk, 0, nullptr, (const uint64_t*)x->g_off.p, (const uint32_t*)x->g_cnt.p,
                         (const uint32_t*)x->g_adj.p, n, (int*)par.p, (int*)flag.p);
      hipLaunchKernelGGL(k_cc_compress, grid, blk, 0, nullptr, n, (int*)par.p);
      int changed = 0;
      if (hipMemcpy(&changed, flag.p, 4, hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
      if (!changed) break;
    }
    if (!ok) { set_err("the component kernels failed"); break; }
    if (hipMemcpy(labels, par.p, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) { set_err("D2H of the labels failed"); break; }
    rc = 0;
  } while (0);
  av.release(); par.release(); flag.release();
  return rc;
}

void pw_seeds_destroy(pw_seed_index* x) {
  if (!x) return;
  (void)hipSetDevice(x->device);
  DevBuf* bufs[] = {&x->tab, &x->dS, &x->dT, &x->keys_in, &x->keys_s, &x->keys_t, &x->pos_in, &x->pos_s, &x->pos_t, &x->lo, &x->cnt,
                    &x->off, &x->rows, &x->tmp, &x->scalar, &x->g_keys, &x->g_order, &x->g_dstart, &x->g_cnt, &x->g_off, &x->g_adj, &x->g_pts};
  for (DevBuf* b : bufs) b->release();
  if (x->ev0) (void)hipEventDestroy(x->ev0);
  if (x->ev1) (void)hipEventDestroy(x->ev1);
  delete x;
}

}  // extern "C"
