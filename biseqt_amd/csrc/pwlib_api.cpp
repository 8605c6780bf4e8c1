// pwlib_api.cpp -- host side of libpwlib (pwlib.so): the batch C ABI of include/pw_batch.h and, on
// top of it, the four drop-in functions of include/pwlib.h (reference biseqt/pwlib/pw.c).
//
// Everything that computes runs in the gfx950 kernels of pw_device.h / pw_trace.hip; this file plans
// (pw_plan.h), moves bytes and launches.  There is no CPU fallback: what the kernels do not support is
// reported as an error.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/pw_batch.h"
#include "pw_launch.h"
#include "pw_plan.h"
#include "pw_model.h"
#define PW_FN inline
#include "pw_strip.h"
#include <atomic>

extern "C" {
#include "../../include/pwlib.h"
}

static_assert(sizeof(pw::Result) == 32 && sizeof(pw_result) == 32, "result record must be 32 bytes");
static_assert(sizeof(pw::PairDesc) == 96, "PairDesc layout");

namespace {
int env_int(const char* name, int dflt);

// A small cache of the big device buffers (tie-mask planes, transcript slots): freeing and re-allocating tens of GB per
// batch stalls in the driver for seconds at a time (measured in the config-4 alignment stage).  Buffers of at least
// 1 MB are kept on pw_batch_destroy, per device up to PWLIB_POOL_GB (default: a quarter of the device's memory; 0
// disables), and handed to the next batch on that device that fits.  pw_pool_trim() releases everything.
struct PoolEntry { void* p; size_t bytes; int device; };
std::mutex g_pool_mutex;
std::vector<PoolEntry> g_pool;
constexpr int kMaxDevices = 64;
size_t g_pool_held[kMaxDevices] = {0};     // bytes parked per device
size_t g_pool_cap[kMaxDevices] = {0};      // 0 = not computed yet

// Cap of the bytes parked on one device: PWLIB_POOL_GB if set (0 disables the pool), otherwise a quarter of the
// device's memory (72 GB on an MI355X), read once per device -- and, at the moment a buffer is parked, never more than
// half of what is FREE on the device then (parked bytes included): several processes sharing a device, or another
// allocator in this process (torch, RCCL), then see the pool shrink instead of an out-of-memory error.
size_t pool_cap(int device) {
  if (device < 0 || device >= kMaxDevices) return 0;
  if (g_pool_cap[device] == 0) {
    const char* v = getenv("PWLIB_POOL_GB");
    size_t cap;
    if (v && *v) cap = (size_t)atoi(v) << 30;
    else {
      size_t fr = 0, tot = 0;
      cap = (hipMemGetInfo(&fr, &tot) == hipSuccess) ? tot / 4 : ((size_t)16 << 30);
    }
    g_pool_cap[device] = cap + 1;            // + 1: "computed" even when the cap itself is 0
  }
  return g_pool_cap[device] - 1;
}
// ... the free-memory half of the rule (the caller holds no lock; the device is current)
size_t pool_room_now(int device, size_t held) {
  static const bool fixed = [] { const char* v = getenv("PWLIB_POOL_GB"); return v && *v; }();
  const size_t cap = pool_cap(device);
  if (fixed) return cap;
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) return cap;
  return std::min(cap, (fr + held) / 2);
}

void* pool_take(int device, size_t bytes, size_t* got) {
  std::lock_guard<std::mutex> lk(g_pool_mutex);
  int best = -1;
  // a parked buffer may be up to 25 % (+ 1 MB) larger than asked for: batches of one chunked job differ by less
  for (int i = 0; i < (int)g_pool.size(); i++)
    if (g_pool[i].device == device && g_pool[i].bytes >= bytes && g_pool[i].bytes <= bytes + bytes / 4 + (1u << 20) &&
        (best < 0 || g_pool[i].bytes < g_pool[best].bytes)) best = i;
  if (best < 0) return nullptr;
  void* p = g_pool[best].p; *got = g_pool[best].bytes;
  if (device >= 0 && device < kMaxDevices) g_pool_held[device] -= g_pool[best].bytes;
  g_pool.erase(g_pool.begin() + best);
  return p;
}
// (the caller has made sure no kernel still uses p: batch_free_device synchronises first)
void pool_give(int device, void* p, size_t bytes) {
  if (!p) return;
  if (device >= 0 && device < kMaxDevices && bytes >= (1u << 20)) {
    size_t held_now;
    { std::lock_guard<std::mutex> lk(g_pool_mutex); held_now = g_pool_held[device]; }
    const size_t cap = pool_room_now(device, held_now);
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    if (g_pool_held[device] + bytes <= cap && g_pool.size() < 64) {
      g_pool.push_back(PoolEntry{p, bytes, device});
      g_pool_held[device] += bytes;
      return;
    }
  }
  (void)hipFree(p);
}
void pool_drop(int device /* < 0: every device */) {
  std::vector<PoolEntry> drop;
  {
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    std::vector<PoolEntry> keep;
    for (auto& e : g_pool) (device < 0 || e.device == device ? drop : keep).push_back(e);
    g_pool.swap(keep);
    for (int d = 0; d < kMaxDevices; d++) if (device < 0 || d == device) g_pool_held[d] = 0;
  }
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (auto& d : drop) { (void)hipSetDevice(d.device); (void)hipFree(d.p); }
  (void)hipSetDevice(cur);
}
hipError_t pool_alloc(int device, void** p, size_t bytes, size_t* got) {
  // size classes, 8 per octave: consecutive batches of similar size then reuse each other's buffers
  if (bytes >= (1u << 20)) {
    int lg = 0; while ((bytes >> (lg + 1)) != 0) lg++;
    const size_t step = (size_t)1 << (lg - 3);
    bytes = (bytes + step - 1) / step * step;
  }
  *p = pool_take(device, bytes, got);
  if (*p) return hipSuccess;
  *got = bytes;
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {           // out of memory with buffers parked in the pool: release them and retry once
    (void)hipGetLastError();
    pool_drop(device);
    e = hipMalloc(p, bytes);
  }
  return e;
}

thread_local std::string g_err;

int fail(const std::string& msg) { g_err = msg; return -1; }

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));          \
  } while (0)

struct BkClass {
  int bk = 0;
  int nw = 1;                   // wavefronts per pair (> 1: multi-wavefront kernel for wide bands)
  std::vector<int32_t> order;   // pair indices, largest table first
  int32_t* d_order = nullptr;
};

}  // namespace

struct pw_batch {
  int device = 0;
  int32_t n = 0;
  uint32_t flags = 0;
  int mode = 0, type = 0, L = 0;
  double go = 0, ge = 0;
  std::vector<double> subst;
  bool simple = false, use_f64 = false;
  double plan_ms = 0;
  int scale_shift = 0;                   // dyadic scaling: every score is held times 2^scale_shift by the integer kernels
  double score_mul = 1.0;                // 2^-scale_shift: what the kernels multiply a reported score with
  int variant = 0, brule = 0, endrule = 0, gosign = 0;
  std::vector<pw_pair> pairs;
  std::vector<pw::Plan> plans;
  std::vector<pw::PairDesc> descs;
  std::vector<BkClass> classes;
  std::vector<pw::WaveDesc> waves;      // lane-packed kernel: one per wavefront
  std::vector<int32_t> strips;          // standard-mode pairs wider than a workgroup: the strip pipeline (K2c, pw_strip.h)
  std::vector<uint32_t> strip_ctl_init;                   // per strip pair: the 16 dwords its control block starts from
  uint64_t* d_fifo = nullptr; size_t fifo_alloc = 0, fifo_bytes = 0;   // FIFO rows of the largest strip pair (pairs run one after another)
  pw::StripBest* d_sbest = nullptr;      // [max strips]
  uint32_t* d_ctl = nullptr;             // [strip pairs][2]: work queue head, abort flag
  std::vector<int32_t> tiled;           // pairs that go through the time-blocked tiled kernel (K2b)
  void* d_state[2] = {nullptr, nullptr}; // their per-diagonal state, double buffered (shared: pairs run one after another)
  int32_t st_pitch = 0;
  int packed_seg = 0, packed_rule = 0, packed_nw = 1;     // packed_nw: wavefronts per pair (K2a with the 16-bit body)
  int packed_mat = 0;                                     // the packed kernel reads a substitution matrix (WaveFill16<.., MAT>)
  pw::WaveDesc* d_waves = nullptr;
  int64_t cells = 0, alg_bytes = 0;
  // device
  uint8_t* d_arena = nullptr; uint64_t arena_bytes = 0;
  bool arena_shared = false;             // PW_FLAG_SHARED_ARENA: d_arena belongs to the caller (pw_batch_share_arena)
  pw::PairDesc* d_pairs = nullptr;
  uint32_t* d_masks = nullptr; uint64_t mask_words = 0; size_t masks_alloc = 0, arena_alloc = 0, pairs_alloc = 0, results_alloc = 0;
  void* d_hdump = nullptr; uint64_t h_elems = 0;
  pw::Result* d_results = nullptr;
  uint8_t* d_tx = nullptr; uint64_t tx_bytes = 0; size_t tx_alloc = 0;
  void* d_subst = nullptr;
  int32_t* d_ends = nullptr;
  uint8_t* d_txpacked = nullptr; size_t txpacked_alloc = 0;    // pw_batch_pack_transcripts: the ops back to back
  uint64_t* d_txoffsets = nullptr;                              // [n + 1]
  hipEvent_t ev_fill0 = nullptr, ev_fill1 = nullptr, ev_tr0 = nullptr, ev_tr1 = nullptr;
  bool fill_timed = false, trace_timed = false;
  // scores as the caller gave them (batch_build may scale b->subst / go / ge by a power of two)
  std::vector<double> subst_in; double go_in = 0, ge_in = 0;
  // Strip pairs whose pipeline gave up waiting (PW_ST_BADPATH with no end cell) are solved again by a batch of one with the
  // strips disabled (repair_strip_pair); the replacement lives as long as the batch and serves every later traceback
  std::vector<std::pair<int32_t, pw_batch*>> repaired;
  // one event per stream the batch was launched on, re-recorded behind every launch sequence: destroying the batch waits
  // for exactly these (not for the device: other batches' streams keep running)
  std::vector<std::pair<hipStream_t, hipEvent_t>> done_events;
  bool traced = false, traced_from = false;          // what the last traceback call was (a repaired pair repeats it)
  std::vector<int32_t> last_ends;
};

namespace {

int batch_free_device(pw_batch* b) {
  if (!b) return 0;
  for (auto& r : b->repaired) pw_batch_destroy(r.second);
  b->repaired.clear();
  (void)hipSetDevice(b->device);
  // the buffers are parked for the next batch, not freed (hipFree would synchronise by itself): make sure nothing that
  // was launched for THIS batch still reads or writes them -- the events recorded behind its launches, stream by stream
  // (a device-wide synchronisation would stall every other batch in flight)
  for (auto& se : b->done_events) { (void)hipEventSynchronize(se.second); (void)hipEventDestroy(se.second); }
  b->done_events.clear();
  for (auto& c : b->classes) if (c.d_order) (void)hipFree(c.d_order);
  if (!b->arena_shared) pool_give(b->device, b->d_arena, b->arena_alloc);
  pool_give(b->device, b->d_pairs, b->pairs_alloc);
  pool_give(b->device, b->d_masks, b->masks_alloc);
  if (b->d_hdump) (void)hipFree(b->d_hdump);
  pool_give(b->device, b->d_results, b->results_alloc);
  pool_give(b->device, b->d_tx, b->tx_alloc);
  if (b->d_subst) (void)hipFree(b->d_subst);
  if (b->d_ends) (void)hipFree(b->d_ends);
  pool_give(b->device, b->d_txpacked, b->txpacked_alloc);
  if (b->d_txoffsets) (void)hipFree(b->d_txoffsets);
  if (b->d_waves) (void)hipFree(b->d_waves);
  pool_give(b->device, b->d_fifo, b->fifo_alloc);
  if (b->d_sbest) (void)hipFree(b->d_sbest);
  if (b->d_ctl) (void)hipFree(b->d_ctl);
  if (b->d_state[0]) (void)hipFree(b->d_state[0]);
  if (b->d_state[1]) (void)hipFree(b->d_state[1]);
  if (b->ev_fill0) (void)hipEventDestroy(b->ev_fill0);
  if (b->ev_fill1) (void)hipEventDestroy(b->ev_fill1);
  if (b->ev_tr0) (void)hipEventDestroy(b->ev_tr0);
  if (b->ev_tr1) (void)hipEventDestroy(b->ev_tr1);
  return 0;
}

bool is_integral(double v) { return v == floor(v) && fabs(v) < 1e9; }

// Planning: host arithmetic only (dptable_init per pair, score type, kernel variant and geometry, launch classes) -- no
// device call, so pw_plan_only can run it anywhere.  batch_alloc then creates the device buffers it sized.
int batch_plan(pw_batch* b) {
  const double t_build0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  // ---- scoring analysis ----
  const int L = b->L;
  bool integral = is_integral(b->go) && is_integral(b->ge);
  double maxabs = std::max(fabs(b->go), std::max(fabs(b->ge), fabs(b->go + b->ge)));
  b->simple = true;
  {
    const double mt0 = b->subst[0], mm0 = L > 1 ? b->subst[1] : b->subst[0];
    for (int i = 0; i < L; i++) for (int j = 0; j < L; j++) {
      const double v = b->subst[(size_t)i * L + j];
      if (!(v == v) || fabs(v) > 1e300) return fail("substitution scores must be finite");
      integral = integral && is_integral(v);
      maxabs = std::max(maxabs, fabs(v));
      if (v != (i == j ? mt0 : mm0)) b->simple = false;
    }
  }
  // Dyadic scaling: scores that are all multiples of 2^-k (k <= 10; e.g. config 5's extension scores 0.25 / -1 / 0 / -1,
  // reference experiments/blot_stats.py:365-372) are held times 2^k and run on the integer kernels.  Every partial sum of
  // such scores is exact in the reference's doubles (the planner keeps them far below 2^53 / 2^k), scaling by a power of
  // two preserves every comparison and tie, and the kernels report value * 2^-k, which is exact again: bit-identical
  // results, no f64 kernel.  PW_FLAG_FORCE_F64 and PWLIB_NO_DYADIC=1 keep the scores as given.
  if (!integral && !(b->flags & PW_FLAG_FORCE_F64) && !env_int("PWLIB_NO_DYADIC", 0) && maxabs < 1e6) {
    for (int sh = 1; sh <= 10 && !b->scale_shift; sh++) {
      const double f = (double)(1 << sh);
      bool ok = is_integral(b->go * f) && is_integral(b->ge * f);
      for (size_t i = 0; ok && i < b->subst.size(); i++) ok = is_integral(b->subst[i] * f);
      if (ok) b->scale_shift = sh;
    }
    if (b->scale_shift) {
      const double f = (double)(1 << b->scale_shift);
      for (auto& v : b->subst) v *= f;
      b->go *= f; b->ge *= f;
      b->score_mul = 1.0 / f;
      integral = true; maxabs *= f;
    }
  }
  const double mt = b->subst[0], mm = L > 1 ? b->subst[1] : b->subst[0];
  // the best and the worst substitution (any of them may be: the API accepts mismatch > match, and a matrix)
  double smax = b->subst[0], smin = b->subst[0];
  for (double v : b->subst) { smax = std::max(smax, v); smin = std::min(smin, v); }
  if (b->simple) { smax = std::max(mt, mm); smin = std::min(mt, mm); }      // (L = 1: the mismatch score never occurs)
  // An integer substitution matrix over at most 4 letters goes into the packed kernels as rows of bytes
  // (WaveFill16<.., MAT>, _alnchoice_M reads subst_scores[o][m], _pw_internals.c:217-245): bytes subst - min, at most 127
  // (times 4 under the scores-times-4 rule), and min <= 0 -- letters outside a sequence score the minimum and must not
  // lift a cell that has not started.
  const bool mat_ok = !b->simple && L <= 4 && integral && smin <= 0 && smax - smin <= 127 && !env_int("PWLIB_NO_PACKED_MAT", 0);
  pw::plan_rules(b->mode, b->type, &b->brule, &b->endrule);
  b->gosign = b->go < 0 ? -1 : (b->go > 0 ? 1 : 0);
  // ---- pass 1: per-pair plans (dptable_init arithmetic) and batch statistics ----
  int64_t maxspan = 0, maxmin = 0; int maxnd = 0; int64_t sumnd = 0; int nsolv = 0;
  // a batch of few pairs: kernels that take the pairs one after another (strips, tiles: each pair gets the whole chip)
  // against kernels that take all of them at once -- estimated times of both from the one table of pw_model.h
  const pw::PlanModel& model = pw::kPlanModel;
  pw::BatchEstimates est;
  int min_x = 0x7fffffff;
  b->plans.resize(b->n); b->descs.resize(b->n);
  uint64_t mask_words = 0, h_elems = 0, tx_bytes = 0;
  for (int32_t k = 0; k < b->n; k++) {
    const pw_pair& p = b->pairs[k];
    if (p.origin_len < 0 || p.mutant_len < 0) return fail("negative sequence length");
    if (p.origin_off + (uint64_t)p.origin_len > b->arena_bytes || p.mutant_off + (uint64_t)p.mutant_len > b->arena_bytes)
      return fail("pair frame outside the arena");
    if ((int64_t)p.origin_len + p.mutant_len > (1 << 30)) return fail("sequences too long");
    if ((p.origin_off & 3) || (p.mutant_off & 3)) return fail("frames must start on a 4-byte boundary of the arena");
    pw::Plan pl = pw::plan_problem(b->mode, b->type, p.origin_len, p.mutant_len, p.dmin, p.dmax);
    b->plans[k] = pl;
    pw::PairDesc d;
    memset(&d, 0, sizeof d);
    d.o_off = p.origin_off; d.m_off = p.mutant_off;
    d.X = p.origin_len; d.Y = p.mutant_len;
    d.solvable = (pl.rc == 0 && pl.ndiag > 0) ? 1 : 0;
    d.tx_off = tx_bytes;
    d.tx_cap = p.origin_len + p.mutant_len + 1;
    tx_bytes += ((uint64_t)d.tx_cap + 15) / 16 * 16;
    if (d.solvable) {
      d.dmin = pl.dmin; d.ndiag = pl.ndiag; d.s0 = pl.s0; d.nblocks = pl.nblocks;
      d.steady_b0 = pl.steady_b0; d.steady_b1 = pl.steady_b1;
      d.h_pitch = std::min(p.origin_len, p.mutant_len) + 1;
      b->cells += pl.cells;
      b->alg_bytes += pl.cells / 2 + p.origin_len + p.mutant_len + 32;
      maxspan = std::max<int64_t>(maxspan, (int64_t)p.origin_len + p.mutant_len + 2);
      maxmin = std::max<int64_t>(maxmin, std::min(p.origin_len, p.mutant_len));
      maxnd = std::max(maxnd, pl.ndiag); sumnd += pl.ndiag; nsolv++;
      est.add_pair(model, (double)pl.nblocks * 16.0 /* anti-diagonals of the (banded) table */, pl.ndiag);
      min_x = std::min(min_x, (int)p.origin_len);
      if (b->mode == pw::STD_MODE) est.add_std_pair(model, p.origin_len, p.mutant_len, pl.ndiag);
    }
    b->descs[k] = d;
  }
  // ---- score type and kernel variant ----
  // int32 is exact iff every score is an integer and no partial sum can leave +-2^27 (pw_wave.h)
  b->use_f64 = (b->flags & PW_FLAG_FORCE_F64) || !integral || (double)maxspan * maxabs >= (double)(1 << 27);
  const bool bany = b->brule == pw::BRULE_ANY;
  const bool track = b->endrule == pw::END_STD_LOCAL || b->endrule == pw::END_BANDED_LOCAL;
  // A substitution matrix needs no kernel of its own: the wavefront kernels read every substitution score from a table in
  // LDS (pw_wave.h, TAB), the packed kernels take small integer matrices as rows of bytes (mat_ok).  The generic kernel is
  // left with what is decided at run time: go > 0, the score-plane dump, and alphabets beyond the LDS copy (32 letters).
  if ((b->flags & (PW_FLAG_FORCE_GENERIC | PW_FLAG_DUMP_SCORES)) || b->go > 0 || L > 32) b->variant = pw::VAR_GENERIC;
  else if (bany) b->variant = pw::VAR_FAST_ANY_TRACK;
  else if (track) b->variant = pw::VAR_FAST_TRACK;
  else b->variant = pw::VAR_FAST;
  // latency mode: with at most 256 pairs the batch is a few hundred wavefronts on a 1024-SIMD chip, so the time is
  // the length of one pair's dependency chain, not throughput (PWLIB_LATENCY_MODE=0 / 1 overrides)
  const int lat_env = env_int("PWLIB_LATENCY_MODE", -1);
  const bool latency_mode = lat_env >= 0 ? lat_env != 0 : nsolv <= 256;
  // lane-packed 16-bit kernel (pw_wave.h, WaveFill16): LOCAL / B_LOCAL, every running value fits int16.  The
  // score bound is 8000, not 16000: the first diagonal above the band is computed like any other and its offer
  // into the band is lowered by only 8192 (the sentinel), so no score -- in or out of the band -- may reach that
  // (regression: test_band_edge_never_leaks_long_pairs).
  // Rules 1 / 2 (B_OVERLAP / B_GLOBAL, WaveFill16<.., RULE>): scores go negative, the sentinel is -24000 and every
  // real score must stay within [-23000, 30000].  Lower bound of any in-band cell: the straight run down its own diagonal
  // from the table edge (min(X,Y) substitutions) -- for B_GLOBAL after the gap run from (0, 0) to that diagonal.
  int pbk = 0, pnl = 0, pseg = 0, prule = -1;
  if (b->variant == pw::VAR_FAST_ANY_TRACK && track) prule = 0;
  // (B_OVERLAP, and standard-mode OVERLAP: the same begin rule, the best last cell of a diagonal, another order of ties)
  else if (b->variant == pw::VAR_FAST && b->brule == pw::BRULE_EDGE &&
           (b->endrule == pw::END_BANDED_OVERLAP || b->endrule == pw::END_STD_OVERLAP)) prule = 1;
  // (the two mixed standard-mode types, begin and end rule read at run time by the rule-1 body: START_ANCHORED_OVERLAP begins
  //  at (0, 0) and ends like OVERLAP, END_ANCHORED_OVERLAP begins like OVERLAP and ends at (X, Y))
  else if (b->variant == pw::VAR_FAST && ((b->brule == pw::BRULE_ORIGIN && b->endrule == pw::END_STD_OVERLAP) ||
                                         (b->brule == pw::BRULE_EDGE && b->endrule == pw::END_CORNER))) prule = 1;
  // (B_GLOBAL, and standard-mode GLOBAL: the same begin / end rule on the band [-Y, X])
  else if (b->variant == pw::VAR_FAST && b->brule == pw::BRULE_ORIGIN && b->endrule == pw::END_CORNER) prule = 2;
  // (END_ANCHORED: begin anywhere like LOCAL, end at (X, Y) -- the captured last cell of one diagonal, nothing tracked)
  else if (b->variant == pw::VAR_FAST_ANY_TRACK && b->endrule == pw::END_CORNER) prule = 4;
  // (START_ANCHORED: begin at (0, 0) like GLOBAL, end at the first best cell, which must beat 0)
  else if (b->variant == pw::VAR_FAST_TRACK && b->brule == pw::BRULE_ORIGIN && b->endrule == pw::END_STD_LOCAL) prule = 5;
  if (!b->simple && (prule > 2 || !mat_ok)) prule = -1;   // (the packed matrix form: rules 0 .. 3, matrices it admits)
  // The packed kernels keep cells that have not started (and cells beyond a diagonal's end) at a shallow 16-bit sentinel, pinned
  // from below by a maximum; what keeps them from creeping UP is that letters outside a sequence "match nothing" and that
  // scores nothing -- true only while the mismatch score (what the plain form gives such letters; with one letter the match
  // score) is <= 0.  With mismatch > 0, which the API accepts, a diagonal that waits ~1400 steps for its first cell starts
  // from a positive phantom score (found by the fuzz on a 3673-diagonal band, scores 1 / 6 / -5 / -2: a wrong end cell, and the
  // walk from it left the mask plane).  Such scores take the matrix form where it applies (its off-table letters score the
  // matrix MINIMUM, required <= 0) and the 32-bit kernels otherwise.
  bool force_simple_mat = false;
  if (b->simple && prule >= 0 && mm > 0) {
    const bool can_mat = prule <= 2 && L >= 2 && L <= 4 && integral && smin <= 0 && smax - smin <= 127 && !env_int("PWLIB_NO_PACKED_MAT", 0);
    if (can_mat) force_simple_mat = true; else prule = -1;
  }
  if (prule >= 4 && env_int("PWLIB_NO_PACKED_ANCHORED", 0)) prule = -1;
  bool pfits = false;
  // (any substitution may be the best one: the API accepts mismatch > match)
  if (prule == 0 || prule == 4) pfits = (double)maxmin * std::max(0.0, smax) <= 8000;
  else if (prule > 0) {
    // real scores must stay above the values derived from the sentinel (<= -24000 + 100) and below int16's top -- rule 5
    // below 8192, the range of its running-best key
    const double worst = std::max(0.0, -smin);
    const double lowest = (double)maxmin * worst + fabs(b->go) + fabs(b->ge) * (maxnd + 2);
    const double highest = (double)maxmin * std::max(0.0, smax);
    pfits = lowest <= 23000 && highest <= (prule == 5 ? 8000 : 30000) && b->go <= 0 && !env_int("PWLIB_NO_PACKED_OVERLAP", 0);
  }
  // (a few standard-mode pairs: the strips, one pair after another, when they are estimated to finish before the 16-bit body
  //  on several wavefronts per pair would -- tests/micro/few_pairs.py: 2 kb x 2 kb, one pair 0.6 ms on the strips, 1.5 ms
  //  there; four pairs 2.3 ms and 1.6 ms)
  const bool strip_scores = b->simple || (integral && L <= 4 && smax <= 127 && smin >= -128 && !env_int("PWLIB_STRIP_NO_BYTE_ROWS", 0));
  const bool strips_win = latency_mode && strip_scores && b->mode == pw::STD_MODE && min_x >= 127 && !(b->flags & PW_FLAG_DUMP_SCORES) &&
                          !env_int("PWLIB_NO_STRIP", 0) && !env_int("PWLIB_NO_SMALL_STRIP", 0) && !(b->flags & PW_FLAG_NO_STRIP) &&
                          (double)maxspan * maxabs < (double)(1 << 25) && est.strips_beat_packed_workgroups(model);
  if (prule >= 0 && pfits && !b->use_f64 &&
      !(b->flags & (PW_FLAG_NO_PACKED16 | PW_FLAG_FORCE_TILED | PW_FLAG_FORCE_STRIP)) && maxnd > 2048 && maxnd <= 64 * pw::kMaxWavesPerPair * 32 &&
      nsolv > 0 && maxabs <= 100 && maxspan < 32000 && b->ge <= 0 && !env_int("PWLIB_NO_PACKED_MW", 0) && !strips_win) {
    // bands wider than one wavefront holds, many pairs (standard-mode tables of 1 .. 8 kb, say): the 16-bit body on a
    // workgroup of up to 8 wavefronts per pair, as few diagonals per lane as 8 wavefronts allow
    for (int i = 0; i < pw::kNumPackedBK; i++) {
      const int bk = pw::kPackedBK[i];
      if ((int64_t)64 * pw::kMaxWavesPerPair * bk >= maxnd) { pbk = bk; break; }
    }
    if (pbk) {
      b->packed_nw = (maxnd + 64 * pbk - 1) / (64 * pbk);
      pnl = 64 * b->packed_nw; pseg = 0;
      b->variant = pw::VAR_FAST16;
    }
  }
  else if (prule >= 0 && pfits && !b->use_f64 &&
      !(b->flags & (PW_FLAG_NO_PACKED16 | PW_FLAG_FORCE_TILED | PW_FLAG_FORCE_STRIP)) && maxnd <= 2048 &&
      nsolv > 0 && maxabs <= 100 && maxspan < 32000 && b->ge <= 0) {
    // Diagonals per lane and pairs per wavefront.  One pair per wave keeps the pair descriptor in scalar registers (measured
    // ~7 % cheaper per cell); several pairs per wave (lane packing) keep more of the 64 x BK diagonal slots busy.  Each layout is
    // priced (pw_model.h): what a slot-step costs at that lane width (the wide lanes pay for their registers) / the share of
    // busy slots x a factor for the last, partly filled round of wavefronts over the SIMDs.  Round 3 found config 4's overlap
    // batches -- bands of 9 .. 111 diagonals, 20 000 pairs -- on 28 diagonals per lane (69 % of the slots busy, but 1250
    // wavefronts on 1024 SIMDs: 7.4 ms) where 8 per lane take 4.9 ms; with 50 000 pairs per batch 16 per lane win
    // (profiles/round3_n_lane_width.txt).  Packing is taken when it is priced 5 % below one pair per wavefront.
    const char* forced = getenv("PWLIB_PACKED_BK");        // tuning / A-B: "<bk>" or "<bk>s" (force packing)
    if (forced && !*forced) forced = nullptr;
    const double meannd = (double)sumnd / nsolv;
    double cost1 = 1e300, costp = 1e300; int bk1 = 0, bkp = 0, nlp = 0;
    for (int i = 0; i < pw::kNumPackedBK; i++) {
      const int bk = pw::kPackedBK[i];
      if (forced && atoi(forced) != bk) continue;
      const int nl = (maxnd + bk - 1) / bk;
      if (nl > 64) continue;
      if (!bk1) {                                                             // smallest BK that fits: one pair per wavefront
        bk1 = bk;
        cost1 = model.one_pair_discount * model.seg_slot_cost[i] * pw::last_round_factor((double)nsolv) / (meannd / (64.0 * bk));
      }
      const int ppw = 64 / nl;
      const int64_t nwv = ((int64_t)nsolv + ppw - 1) / ppw;
      // (packing must leave at least one wavefront per SIMD: 20 000 pairs with a 21-diagonal band packed 64 to a wavefront are
      //  313 wavefronts with 12 cells per lane and step -- 0.69 ms against 0.41 ms for 10 to a wavefront)
      const bool enough = forced || nwv >= 1024;
      const double cp = model.seg_slot_cost[i] * pw::last_round_factor((double)nwv) / ((double)ppw * meannd / (64.0 * bk));
      if (ppw >= 2 && enough && cp < costp - 1e-9) { costp = cp; bkp = bk; nlp = nl; }      // ties: the narrower lanes
    }
    const bool want_seg = bkp && (!bk1 || costp < 0.95 * cost1 || (forced && strchr(forced, 's')));
    // pairs that fit one wavefront side by side at the narrowest lanes
    const int ppw1 = bk1 ? std::max(1, 64 / ((maxnd + bk1 - 1) / bk1)) : 1;
    if ((latency_mode || nsolv < 1024 * ppw1) && bk1 && !forced) {
      // fewer wavefronts than SIMDs (side by side): the time is one wavefront's chain of steps, so as few diagonals per
      // lane as the band allows -- several pairs side by side where they fit, which changes the number of wavefronts, not
      // the chain (2 kb pairs, band radius 20: 1.41 -> 0.49 ms; radius 50: 0.87 -> 0.49 ms).  Round 3 (found by
      // tests/micro/planner_check.py): this also holds for 1024 ... 1024 x ppw1 pairs, which used to fall between the two
      // rules and ran one pair per wavefront, two wavefronts per SIMD -- 2000 pairs of 1 kb with a 21-diagonal band
      // 0.48 ms, side by side 0.37 ms.
      pbk = bk1; pnl = (maxnd + bk1 - 1) / bk1; pseg = 64 / pnl >= 2 ? 1 : 0;
    }
    else if (want_seg) { pbk = bkp; pnl = nlp; pseg = 1; }
    else if (bk1) { pbk = bk1; pnl = (maxnd + bk1 - 1) / bk1; pseg = 0; }
    // one pair per wavefront with 16+ diagonals per lane is a long serial chain: small batches go multi-wavefront --
    // the strips if they win, else the 16-bit body on up to 8 wavefronts with 4 or 8 diagonals per lane
    if (latency_mode && pbk >= 16 && !pseg) {
      pbk = 0;
      if (!strips_win && !forced && !env_int("PWLIB_NO_PACKED_MW", 0)) {
        for (int i = 0; i < pw::kNumPackedBK; i++) {
          const int bk = pw::kPackedBK[i];
          if ((int64_t)64 * pw::kMaxWavesPerPair * bk >= maxnd) { pbk = bk; break; }
        }
        if (pbk) { b->packed_nw = (maxnd + 64 * pbk - 1) / (64 * pbk); pnl = 64 * b->packed_nw; pseg = 0; }
        if (b->packed_nw <= 1) { pbk = 0; b->packed_nw = 1; }      // (one wavefront would do: not this case)
      }
    }
    if (pbk) b->variant = pw::VAR_FAST16;
  }
  if (!b->simple && b->variant == pw::VAR_FAST16) b->packed_mat = 1;
  // Match / mismatch scoring over at most 4 letters IS such a matrix, and the matrix form's cell pair is shorter -- one
  // v_perm_b32 instead of xor, min and multiply-add, and under the local rule the bias comes off with a saturating subtract
  // that makes the maximum with the begin candidate 0 unnecessary -- at the price of registers (3 wavefronts per SIMD instead
  // of 5).  Taken where an A/B on the GPU showed it faster (tests/micro/ab_simple_matrix.py,
  // profiles/round3_h_ab_simple_matrix.txt): the local rule at 8 diagonals per lane, one pair per wavefront (config 2's shape:
  // 3.50 -> 3.29 ms) and at 16 (2.23 -> 2.15 ms), standard-mode GLOBAL at 32 per lane (4.38 -> 4.27 ms); slower lane-packed
  // (+4 %) and under the overlap rule (+4 %).  PWLIB_SIMPLE_AS_MATRIX=0 / 1: never / wherever the matrix form exists (A/B).
  if (b->simple && b->variant == pw::VAR_FAST16 && prule >= 0 && prule <= 2 && L >= 2 && L <= 4 && integral && smin <= 0 &&
      smax - smin <= 127 && !env_int("PWLIB_NO_PACKED_MAT", 0)) {
    const int knob = env_int("PWLIB_SIMPLE_AS_MATRIX", -1);
    const bool measured = !pseg && b->packed_nw <= 1 && (((pbk == 8 || pbk == 16) && prule == 0) || (pbk == 32 && prule == 2));
    if (force_simple_mat || knob > 0 || (knob < 0 && measured)) b->packed_mat = 1;
  }
  // ---- pass 2: kernel geometry per pair, mask planes, launch classes ----
  for (int32_t k = 0; k < b->n; k++) {
    pw::PairDesc& d = b->descs[k];
    if (!d.solvable) continue;
    int bk, nl, nw = 1;
    bool tiled = false;
    // (scores within +-2^25: the strip kernel tracks a row's best as 32 * H + step)
    const bool strip_ok = b->mode == pw::STD_MODE && !b->use_f64 && (double)maxspan * maxabs < (double)(1 << 25) &&
                          b->variant != pw::VAR_GENERIC && b->variant != pw::VAR_FAST16 && strip_scores &&
                          !(b->flags & (PW_FLAG_DUMP_SCORES | PW_FLAG_FORCE_TILED | PW_FLAG_NO_STRIP)) && !env_int("PWLIB_NO_STRIP", 0);
    // ... always for tables wider than a workgroup holds; and for batches of a few pairs (at most 256: latency mode) when the
    // strips of all pairs, one pair after another, are estimated to finish before the slowest workgroup would (2 kb x 2 kb:
    // 0.6 ms per pair against 13.6 ms for one workgroup of 32-diagonal lanes -- up to 16 such pairs; 1 kb x 1 kb: 0.3 ms
    // against 1.0 ms -- up to 2), for tables that span at least two strips
    const bool few = latency_mode && d.X >= 127 && est.strips_beat_workgroups(model);
    if (strip_ok && ((b->flags & PW_FLAG_FORCE_STRIP) || d.ndiag > 2048 * pw::kMaxWavesPerPair || (few && !env_int("PWLIB_NO_SMALL_STRIP", 0)))) {
      // one pair wider than a workgroup, integer scores, simple scoring: rows in strips of 64, a pipeline of wavefronts
      const int nstrips = (d.X + 1 + 63) / 64, nkq = (d.Y + 64 + pw::kStripBlock - 1) / pw::kStripBlock;
      d.layout = 1; d.bk = 0; d.nl = 64;
      d.mask_off = mask_words;
      mask_words += (uint64_t)nstrips * nkq * 64 * 4;
      d.h_off = h_elems;
      b->strips.push_back(k);
      b->fifo_bytes = std::max<size_t>(b->fifo_bytes, (size_t)nstrips * (size_t)pw::strip_fifo_pitch(d.Y) * 8);
      continue;
    }
    // a few pairs with bands wider than a wavefront holds, not served by the strips (f64 scores, a substitution matrix,
    // go > 0, banded): the tiled kernel when all pairs, one after another, are estimated to finish before the slowest workgroup
    const bool few_tiled = latency_mode && d.ndiag > 1024 && !(b->flags & PW_FLAG_DUMP_SCORES) && !env_int("PWLIB_NO_SMALL_TILED", 0) &&
                           est.tiles_beat_workgroups(model, b->use_f64);
    if (b->variant == pw::VAR_FAST16) { bk = pbk; nl = pnl; }
    else if ((b->flags & PW_FLAG_FORCE_TILED) || d.ndiag > 2048 * pw::kMaxWavesPerPair || few_tiled) {
      // wider than a workgroup holds (or forced): time-blocked tiles of the band, one pair after another
      if (b->flags & PW_FLAG_DUMP_SCORES) return fail("the score plane is not available for tiled (very wide) tables");
      bk = pw::kTileBKHost;
      nl = (d.ndiag + bk - 1) / bk;
      tiled = true;
    } else {
      bk = pw::plan_pick_bk(d.ndiag, pw::kSupportedBK, pw::kNumSupportedBK);
      nl = 64;
      if (bk == 0) {
        // wider than one wavefront holds: a workgroup of nw wavefronts, 2048 diagonals each -- in latency mode as many
        // wavefronts as a workgroup takes, with as few diagonals per lane as that allows (2 kb x 2 kb: 8 x 8 instead of
        // 2 x 32 diagonals per lane)
        bk = 32;
        if (latency_mode || !env_int("PWLIB_MW_WIDE_LANES", 0)) {
          // (also with many pairs: at 32 diagonals per lane the kernel spills 1.4 - 3 KB of registers per lane; 300 pairs of
          //  10 kb with a 3001-diagonal band: 208 ms with 2 x 32, measured below with 6 x 8)
          for (int cand : {8, 16, 32}) {
            if ((int64_t)64 * pw::kMaxWavesPerPair * cand >= d.ndiag) { bk = cand; break; }
          }
        }
        nw = (d.ndiag + 64 * bk - 1) / (64 * bk);
        nl = 64 * nw;
      } else if ((latency_mode && bk >= 16) || (b->use_f64 && bk >= 32)) {
        // (f64 with 32 diagonals per lane needs more registers than a wavefront has: 3000 pairs with a 1201-diagonal band
        //  take 166 ms on one wavefront each, 47 ms on five wavefronts of 4 diagonals per lane)
        // a handful of pairs cannot fill the chip anyway: spread each over up to 8 wavefronts with few diagonals
        // per lane (the step count is fixed by X + Y; the work per step shrinks 4-8x, the exchange costs ~0.3 us)
        for (int cand : {4, 8, 16}) {
          if ((int64_t)64 * pw::kMaxWavesPerPair * cand >= d.ndiag) { bk = cand; break; }
        }
        nw = (d.ndiag + 64 * bk - 1) / (64 * bk);
        nl = 64 * nw;
      }
    }
    d.bk = bk; d.nl = nl;
    d.mask_off = mask_words;
    mask_words += (uint64_t)(d.nblocks + 1) * nl * bk;   // + one spare row: the branch-free stores of idle lanes land there
    d.h_off = h_elems;
    if (b->flags & PW_FLAG_DUMP_SCORES) h_elems += (uint64_t)d.ndiag * d.h_pitch;
    if (tiled) { b->tiled.push_back(k); b->st_pitch = std::max(b->st_pitch, (d.ndiag + 63) / 64 * 64); continue; }
    size_t ci = 0;
    for (; ci < b->classes.size(); ci++) if (b->classes[ci].bk == bk && b->classes[ci].nw == nw) break;
    if (ci == b->classes.size()) { b->classes.emplace_back(); b->classes.back().bk = bk; b->classes.back().nw = nw; }
    b->classes[ci].order.push_back(k);
  }
  for (auto& c : b->classes)
    std::stable_sort(c.order.begin(), c.order.end(), [&](int32_t x, int32_t y) {
      return b->descs[x].nblocks > b->descs[y].nblocks;
    });
  if (b->variant == pw::VAR_FAST16) {
    // consecutive (similar length) pairs share a wavefront
    const int ppw = pseg ? 64 / pnl : 1;
    b->packed_seg = pseg; b->packed_rule = prule;
    // rule 0, scores below 2048: the kernel that holds every score times 4 (WaveFill16, RULE 3)
    if (prule == 0 && (double)maxmin * std::max(0.0, smax) <= 2047 && !env_int("PWLIB_NO_SCALED16", 0) &&
        (!b->packed_mat || 4 * (smax - smin) <= 127))
      b->packed_rule = 3;
    BkClass& c = b->classes[0];
    for (size_t i = 0; i < c.order.size(); i += ppw) {
      pw::WaveDesc wd;
      memset(&wd, 0, sizeof wd);
      wd.first = (int32_t)i; wd.count = (int32_t)std::min<size_t>(ppw, c.order.size() - i);
      wd.nl = (pseg || b->packed_nw > 1) ? pnl : 64;      // lanes per pair in the wave / workgroup (the mask plane rows stay pnl wide)
      wd.nblocks = 0; wd.steady_b0 = 0; wd.steady_b1 = 0x7fffffff;
      for (int q = 0; q < wd.count; q++) {
        const pw::PairDesc& d = b->descs[c.order[i + q]];
        wd.nblocks = std::max(wd.nblocks, d.nblocks);
        wd.steady_b0 = std::max(wd.steady_b0, d.steady_b0);
        wd.steady_b1 = std::min(wd.steady_b1, d.steady_b1);
      }
      if (wd.steady_b1 < wd.steady_b0) wd.steady_b1 = wd.steady_b0;
      b->waves.push_back(wd);
    }
  }
  b->mask_words = mask_words; b->h_elems = h_elems; b->tx_bytes = tx_bytes;
  b->arena_shared = (b->flags & PW_FLAG_SHARED_ARENA) != 0;
  b->plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_build0;
  return 0;
}

int batch_alloc(pw_batch* b) {
  const int L = b->L;
  const uint64_t mask_words = b->mask_words, h_elems = b->h_elems, tx_bytes = b->tx_bytes;
  // ---- device buffers ----
  const bool tim = env_int("PWLIB_TIMING", 0) != 0;
  auto tnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_plan = tnow();
  HIP_TRY(hipSetDevice(b->device));
  if (!b->arena_shared)
    HIP_TRY(pool_alloc(b->device, (void**)&b->d_arena, b->arena_bytes + 16, &b->arena_alloc));   // kernels read whole dwords: slack past the last frame
  HIP_TRY(pool_alloc(b->device, (void**)&b->d_pairs, sizeof(pw::PairDesc) * std::max<int32_t>(b->n, 1), &b->pairs_alloc));
  HIP_TRY(pool_alloc(b->device, (void**)&b->d_masks, 4 * mask_words + 64, &b->masks_alloc));   // slack: the walker reads whole 16-byte groups
  HIP_TRY(pool_alloc(b->device, (void**)&b->d_results, sizeof(pw::Result) * std::max<int32_t>(b->n, 1), &b->results_alloc));
  HIP_TRY(pool_alloc(b->device, (void**)&b->d_tx, std::max<uint64_t>(tx_bytes, 16), &b->tx_alloc));
  const size_t esz = b->use_f64 ? 8 : 4;
  if (h_elems) HIP_TRY(hipMalloc(&b->d_hdump, esz * h_elems));
  HIP_TRY(hipMalloc(&b->d_subst, esz * (size_t)L * L));
  if (b->use_f64) {
    HIP_TRY(hipMemcpy(b->d_subst, b->subst.data(), 8 * (size_t)L * L, hipMemcpyHostToDevice));
  } else {
    std::vector<int32_t> si((size_t)L * L);
    for (size_t i = 0; i < si.size(); i++) si[i] = (int32_t)b->subst[i];
    HIP_TRY(hipMemcpy(b->d_subst, si.data(), 4 * si.size(), hipMemcpyHostToDevice));
  }
  if (b->n) HIP_TRY(hipMemcpy(b->d_pairs, b->descs.data(), sizeof(pw::PairDesc) * b->n, hipMemcpyHostToDevice));
  if (!b->strips.empty()) {
    int maxs = 0;
    for (int32_t k : b->strips) maxs = std::max(maxs, (b->descs[k].X + 1 + 63) / 64);
    HIP_TRY(pool_alloc(b->device, (void**)&b->d_fifo, b->fifo_bytes, &b->fifo_alloc));
    // granules are recognised by their epoch tag: whatever the (possibly recycled) buffer holds must never look like one
    HIP_TRY(hipMemset(b->d_fifo, 0, b->fifo_bytes));
    HIP_TRY(hipMalloc((void**)&b->d_sbest, sizeof(pw::StripBest) * (size_t)maxs));
    HIP_TRY(hipMalloc((void**)&b->d_ctl, 64 * b->strips.size()));
  }
  if (!b->tiled.empty()) {
    HIP_TRY(hipMalloc(&b->d_state[0], (size_t)5 * b->st_pitch * 8));
    HIP_TRY(hipMalloc(&b->d_state[1], (size_t)5 * b->st_pitch * 8));
  }
  if (!b->waves.empty()) {
    HIP_TRY(hipMalloc((void**)&b->d_waves, sizeof(pw::WaveDesc) * b->waves.size()));
    HIP_TRY(hipMemcpy(b->d_waves, b->waves.data(), sizeof(pw::WaveDesc) * b->waves.size(), hipMemcpyHostToDevice));
  }
  for (auto& c : b->classes) {
    HIP_TRY(hipMalloc((void**)&c.d_order, 4 * c.order.size()));
    HIP_TRY(hipMemcpy(c.d_order, c.order.data(), 4 * c.order.size(), hipMemcpyHostToDevice));
  }
  {  // records of pairs no kernel will touch
    std::vector<pw::Result> init(std::max<int32_t>(b->n, 1));
    for (auto& r : init) { r.score = 0; r.opt_i = r.opt_j = -1; r.origin_idx = r.mutant_idx = 0; r.tx_len = 0; r.status = 0; }
    HIP_TRY(hipMemcpy(b->d_results, init.data(), sizeof(pw::Result) * init.size(), hipMemcpyHostToDevice));
  }
  if (b->flags & PW_FLAG_PROFILE) {
    HIP_TRY(hipEventCreate(&b->ev_fill0)); HIP_TRY(hipEventCreate(&b->ev_fill1));
    HIP_TRY(hipEventCreate(&b->ev_tr0)); HIP_TRY(hipEventCreate(&b->ev_tr1));
  }
  if (tim && b->n > 1000) fprintf(stderr, "pwlib timing: batch of %d pairs: planning %.1f ms, device buffers + descriptors %.1f ms\n", b->n, b->plan_ms, tnow() - t_plan);
  return 0;
}

int batch_build(pw_batch* b) {
  if (batch_plan(b) != 0) return -1;
  return batch_alloc(b);
}

// Records "everything launched for this batch on `st` so far" (see pw_batch::done_events).
int mark_done(pw_batch* b, hipStream_t st) {
  for (auto& se : b->done_events)
    if (se.first == st) { HIP_TRY(hipEventRecord(se.second, st)); return 0; }
  hipEvent_t ev = nullptr;
  HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  b->done_events.emplace_back(st, ev);
  HIP_TRY(hipEventRecord(ev, st));
  return 0;
}

std::atomic<uint32_t> g_strip_epoch{1};

// XCC ids present on a device (the strip pipeline keeps runs of consecutive strips on one XCD): counted once per device
// by a census kernel.  xcc_queue[id] = queue index or -1; returns the number of queues (0 on error).
int xcc_queues(int device, int32_t* xcc_queue) {
  static std::mutex mu;
  static int cached_n[kMaxDevices] = {0};
  static int32_t cached_map[kMaxDevices][8];
  std::lock_guard<std::mutex> lk(mu);
  if (device < 0 || device >= kMaxDevices) return 0;
  if (cached_n[device] == 0) {
    uint32_t* d = nullptr; uint32_t h[8] = {0};
    if (hipMalloc((void**)&d, sizeof h) != hipSuccess) return 0;
    bool ok = hipMemset(d, 0, sizeof h) == hipSuccess && pw::launch_xcc_census(d, nullptr) == hipSuccess &&
              hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    if (!ok) return 0;
    int n = 0;
    for (int i = 0; i < 8; i++) cached_map[device][i] = h[i] ? n++ : -1;
    cached_n[device] = n;
  }
  memcpy(xcc_queue, cached_map[device], sizeof cached_map[device]);
  return cached_n[device];
}

// The strip kernel's byte rows (pw_strip.h, BROW): at most 4 letters, every score an integer that fits a signed byte
bool strip_byte_rows_ok(const pw_batch* b) {
  if (b->L > 4 || env_int("PWLIB_STRIP_NO_BYTE_ROWS", 0)) return false;
  for (double v : b->subst) if (v != std::floor(v) || v < -128 || v > 127) return false;
  return true;
}

// K2c: the strip pairs of a batch, one after another (each one fills the chip by itself)
int launch_strip_fills(pw_batch* b, hipStream_t st) {
  static const int workers = std::max(1, env_int("PWLIB_STRIP_WAVES", 1024));
  static const int lds_kb = std::max(0, env_int("PWLIB_STRIP_LDS_KB", 0));
  size_t q = 0;
  if (b->strip_ctl_init.size() != 16 * b->strips.size()) b->strip_ctl_init.assign(16 * b->strips.size(), 0u);
  for (int32_t k : b->strips) {
    const pw::PairDesc& d = b->descs[k];
    pw::StripParams a;
    memset(&a, 0, sizeof a);
    a.arena = b->d_arena; a.o_off = d.o_off; a.m_off = d.m_off;
    a.fifo = b->d_fifo; a.masks = b->d_masks + d.mask_off; a.sbest = b->d_sbest; a.ctl = b->d_ctl + 16 * q++;
    a.result = b->d_results + k;
    a.X = d.X; a.Y = d.Y;
    a.nstrips = (d.X + 1 + 63) / 64; a.nkq = (d.Y + 64 + pw::kStripBlock - 1) / pw::kStripBlock;
    a.fifo_pitch = pw::strip_fifo_pitch(d.Y);
    uint32_t e = g_strip_epoch.fetch_add(1) & 0xffffffu;          // 24 bits: the tag's other 8 are the column's low bits
    if (e == 0) e = g_strip_epoch.fetch_add(1) & 0xffffffu;
    a.epoch = e;
    a.brule = b->brule; a.endrule = b->endrule;
    a.match = (int32_t)b->subst[0]; a.mismatch = (int32_t)(b->L > 1 ? b->subst[1] : b->subst[0]);
    a.go = (int32_t)b->go; a.ge = (int32_t)b->ge;
    a.score_mul = b->score_mul;
    a.spin_limit = env_int("PWLIB_STRIP_SPIN_LIMIT", 1 << 21);
    a.nq = xcc_queues(b->device, a.xcc_queue);
    if (a.nq <= 0) return fail("could not determine the XCDs of the device");
    a.run_len = std::max(1, env_int("PWLIB_STRIP_RUN", std::max(1, workers / a.nq)));
    const bool track = b->endrule != pw::END_CORNER;
    // tuning aid: PWLIB_STRIP_TRACE=<file> dumps per-strip clock stamps (100 MHz; [8 + i]: shader-clock counts at the same points) of the first strip pair: dequeue, set-up
    // done, first granules seen, steps 64 / 96 reached, end; [7] = XCC id | workgroup << 8
    const char* trace = getenv("PWLIB_STRIP_TRACE");
    uint64_t* d_stamps = nullptr;
    if (trace && *trace && q == 1) {
      HIP_TRY(hipMalloc((void**)&d_stamps, (size_t)a.nstrips * 128));
      HIP_TRY(hipMemset(d_stamps, 0, (size_t)a.nstrips * 128));
      a.stamps = d_stamps;
    }
    const bool byte_rows = strip_byte_rows_ok(b);
    if (!byte_rows && !b->simple) return fail("internal: a substitution matrix on the strips needs their byte rows");
    // (the 16 dwords the control block starts from: zeros and, behind them, the byte rows; kept in the batch until it dies)
    uint32_t* ci = b->strip_ctl_init.data() + 16 * (q - 1);
    uint32_t rows[4] = {0u, 0u, 0u, 0u};
    if (byte_rows)
      for (int o = 0; o < 4; o++)
        for (int m = 0; m < 4; m++)
          if (o < b->L && m < b->L) rows[o] |= ((uint32_t)(int32_t)b->subst[(size_t)o * b->L + m] & 0xffu) << (8 * m);
    // (written once per batch in effect: every solve stores the same values, also while an earlier copy may still be reading them)
    for (int z = 0; z < 16; z++) ci[z] = (z >= pw::kStripRows && z < pw::kStripRows + 4) ? rows[z - pw::kStripRows] : 0u;
    HIP_TRY(pw::launch_strip_fill(a, track, byte_rows, ci, workers, lds_kb << 10, st));
    if (d_stamps) {
      std::vector<uint64_t> h((size_t)a.nstrips * 16);
      HIP_TRY(hipStreamSynchronize(st));
      HIP_TRY(hipMemcpy(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost));
      (void)hipFree(d_stamps);
      if (FILE* f = fopen(trace, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
  }
  return 0;
}

template <typename T>
int launch_all_fills(pw_batch* b, hipStream_t st) {
  pw::FillParams<T> a;
  memset(&a, 0, sizeof a);
  a.pairs = b->d_pairs; a.arena = b->d_arena; a.masks = b->d_masks;
  a.hdump = (T*)b->d_hdump; a.results = b->d_results; a.subst = (const T*)b->d_subst;
  a.npairs = b->n; a.L = b->L; a.brule = b->brule; a.endrule = b->endrule;
  a.banded = b->mode == pw::BANDED_MODE;
  a.match = (T)b->subst[0]; a.mismatch = (T)(b->L > 1 ? b->subst[1] : b->subst[0]);
  a.go = (T)b->go; a.ge = (T)b->ge;
  a.score_mul = b->score_mul;
  for (auto& c : b->classes) {
    a.order = c.d_order;
    if (c.nw > 1) HIP_TRY(pw::launch_fill_mw(a, b->variant, c.bk, c.nw, (int)c.order.size(), st));
    else HIP_TRY(pw::launch_fill(a, b->variant, c.bk, (int)c.order.size(), st));
  }
  // K2b: tiled pairs, one after another; per pair one launch per time block, then the end-cell search
  a.order = nullptr;
  a.st_pitch = b->st_pitch;
  for (int32_t k : b->tiled) {
    const pw::PairDesc& d = b->descs[k];
    const int ntiles = (d.nl + pw::kTileCentralLanes - 1) / pw::kTileCentralLanes;
    int cur = 0;
    for (int tb = 0; tb < d.nblocks; tb += pw::kTileBlocks) {
      a.tile_b0 = tb; a.tile_nb = std::min(pw::kTileBlocks, d.nblocks - tb);
      a.st_in = (const T*)b->d_state[cur]; a.st_out = (T*)b->d_state[cur ^ 1];
      HIP_TRY(pw::launch_tile(a, b->variant, (int)k, ntiles, st));
      cur ^= 1;
    }
    a.st_in = (const T*)b->d_state[cur];
    HIP_TRY(pw::launch_tile_finish(a, (int)k, st));
  }
  return 0;
}

int launch_packed_fill(pw_batch* b, hipStream_t st) {
  pw::FillParams<int32_t> a;
  memset(&a, 0, sizeof a);
  a.pairs = b->d_pairs; a.arena = b->d_arena; a.masks = b->d_masks; a.results = b->d_results;
  a.npairs = b->n; a.L = b->L; a.brule = b->brule; a.endrule = b->endrule;
  a.banded = b->mode == pw::BANDED_MODE;
  a.match = (int32_t)b->subst[0]; a.mismatch = (int32_t)(b->L > 1 ? b->subst[1] : b->subst[0]);
  a.go = (int32_t)b->go; a.ge = (int32_t)b->ge;
  a.score_mul = b->score_mul;
  a.order = b->classes[0].d_order; a.waves = b->d_waves;
  if (b->packed_mat) {
    // rows of bytes scale * (subst[o][m] - min), m = 0 .. 3 from the low byte up; letters beyond L never occur
    const int scale = b->packed_rule == 3 ? 4 : 1;
    double smin = b->subst[0];
    for (double v : b->subst) smin = std::min(smin, v);
    for (int o = 0; o < 4; o++) {
      uint32_t row = 0;
      for (int m = 0; m < 4; m++)
        if (o < b->L && m < b->L) row |= (uint32_t)(scale * (int)(b->subst[(size_t)o * b->L + m] - smin)) << (8 * m);
      a.mat_rows[o] = row;
    }
    a.mat_bias = scale * (int)(-smin);
  }
  if (b->packed_nw > 1) HIP_TRY(pw::launch_fill16_mw(a, b->classes[0].bk, b->packed_rule, b->packed_mat, b->packed_nw, (int)b->waves.size(), st));
  else HIP_TRY(pw::launch_fill16(a, b->classes[0].bk, b->packed_seg, b->packed_rule, b->packed_mat, (int)b->waves.size(), st));
  return 0;
}

}  // namespace

extern "C" {

const char* pw_last_error(void) { return g_err.c_str(); }

void pw_pool_trim(void) { pool_drop(-1); }

int pw_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes) {
  size_t fr = 0, tot = 0;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMemGetInfo(&fr, &tot));
  if (free_bytes) *free_bytes = fr;
  if (total_bytes) *total_bytes = tot;
  return 0;
}

int pw_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

pw_batch* pw_batch_create(int device, const pw_scoring* sc, int32_t n_pairs, const pw_pair* pairs,
                          uint64_t arena_bytes, uint32_t flags) {
  if (!sc || n_pairs < 0 || (n_pairs > 0 && !pairs)) { fail("pw_batch_create: bad arguments"); return nullptr; }
  if (sc->mode != pw::STD_MODE && sc->mode != pw::BANDED_MODE) { fail("unknown alignment mode"); return nullptr; }
  if (sc->type < 0 || sc->type > (sc->mode == pw::STD_MODE ? 6 : 2)) { fail("unknown alignment type"); return nullptr; }
  if (sc->alphabet_len < 1 || sc->alphabet_len > 256 || !sc->subst) { fail("alphabet_len must be 1..256 with a score matrix"); return nullptr; }
  pw_batch* b = new pw_batch();
  b->device = device; b->n = n_pairs; b->flags = flags;
  b->mode = sc->mode; b->type = sc->type; b->L = sc->alphabet_len; b->go = sc->go; b->ge = sc->ge;
  b->subst.assign(sc->subst, sc->subst + (size_t)b->L * b->L);
  b->subst_in = b->subst; b->go_in = b->go; b->ge_in = b->ge;
  b->pairs.assign(pairs, pairs + n_pairs);
  b->arena_bytes = arena_bytes;
  if (batch_build(b) != 0) { batch_free_device(b); delete b; return nullptr; }
  return b;
}

void pw_batch_destroy(pw_batch* b) {
  if (!b) return;
  batch_free_device(b);
  delete b;
}

int pw_plan_only(const pw_scoring* sc, int32_t n_pairs, const pw_pair* pairs, uint64_t arena_bytes, uint32_t flags,
                 char* kernel, int32_t kernel_cap, int32_t* info) {
  if (!sc || n_pairs < 0 || (n_pairs > 0 && !pairs) || !sc->subst) return fail("pw_plan_only: bad arguments");
  if (sc->mode != pw::STD_MODE && sc->mode != pw::BANDED_MODE) return fail("unknown alignment mode");
  if (sc->type < 0 || sc->type > (sc->mode == pw::STD_MODE ? 6 : 2)) return fail("unknown alignment type");
  if (sc->alphabet_len < 1 || sc->alphabet_len > 256) return fail("alphabet_len must be 1..256 with a score matrix");
  pw_batch b;
  b.device = -1; b.n = n_pairs; b.flags = flags;
  b.mode = sc->mode; b.type = sc->type; b.L = sc->alphabet_len; b.go = sc->go; b.ge = sc->ge;
  b.subst.assign(sc->subst, sc->subst + (size_t)b.L * b.L);
  b.pairs.assign(pairs, pairs + n_pairs);
  b.arena_bytes = arena_bytes;
  if (batch_plan(&b) != 0) return -1;
  if (kernel && kernel_cap > 0) { strncpy(kernel, pw_batch_kernel_name(&b), (size_t)kernel_cap - 1); kernel[kernel_cap - 1] = 0; }
  if (info) {
    int64_t one = 0, wg = 0;
    for (const auto& c : b.classes) (c.nw > 1 || (b.variant == pw::VAR_FAST16 && b.packed_nw > 1) ? wg : one) += (int64_t)c.order.size();
    info[0] = b.use_f64 ? 1 : 0; info[1] = b.scale_shift; info[2] = (int32_t)one; info[3] = (int32_t)wg;
    info[4] = (int32_t)b.tiled.size(); info[5] = (int32_t)b.strips.size();
    info[6] = b.variant == pw::VAR_FAST16 ? b.packed_rule : -1; info[7] = b.packed_mat;
  }
  return 0;
}

int pw_batch_init_rc(const pw_batch* b, int32_t k) { return (k < 0 || k >= b->n) ? -1 : b->plans[k].rc; }

int pw_batch_band(const pw_batch* b, int32_t k, int32_t* dmin, int32_t* dmax, int32_t* num_rows) {
  if (k < 0 || k >= b->n) return -1;
  if (dmin) *dmin = b->plans[k].dmin;
  if (dmax) *dmax = b->plans[k].dmax;
  if (num_rows) *num_rows = b->plans[k].num_rows;
  return b->plans[k].clamped;
}

int64_t pw_batch_pair_cells(const pw_batch* b, int32_t k) { return (k < 0 || k >= b->n) ? -1 : b->plans[k].cells; }
int64_t pw_batch_cells(const pw_batch* b) { return b->cells; }
int64_t pw_batch_algorithmic_bytes(const pw_batch* b) { return b->alg_bytes; }
int pw_batch_score_type(const pw_batch* b) { return b->use_f64 ? 1 : 0; }

const char* pw_batch_kernel_name(const pw_batch* b) {
  static thread_local char name[96];
  int bk = 0, nw = 1; size_t most = 0;
  for (const auto& c : b->classes) if (c.order.size() > most) { most = c.order.size(); bk = c.bk; nw = c.nw; }
  if (b->classes.empty() && b->tiled.empty() && !b->strips.empty()) {
    snprintf(name, sizeof name, "k_fill_strip<%s> x row strips", b->endrule != pw::END_CORNER ? "true" : "false");
    return name;
  }
  if (b->classes.empty() && !b->tiled.empty()) { snprintf(name, sizeof name, "k_fill_tile<%s> x time blocks", b->use_f64 ? "double" : "int"); return name; }
  const char* t = b->use_f64 ? "double" : "int";
  if (nw > 1) { snprintf(name, sizeof name, "k_fill_mw<%s, %d, ...> x %d wavefronts", t, bk, nw); return name; }
  switch (b->variant) {
    case pw::VAR_FAST16:
      if (b->packed_nw > 1) snprintf(name, sizeof name, "k_fill16_mw<%d, %d> x %d wavefronts", bk, b->packed_rule, b->packed_nw);
      else if (b->packed_rule == 3) snprintf(name, sizeof name, "k_fill16<%d, %s> x4", bk, b->packed_seg ? "true" : "false");
      else if (b->packed_rule) snprintf(name, sizeof name, "k_fill16<%d, %s, %d>", bk, b->packed_seg ? "true" : "false", b->packed_rule);
      else snprintf(name, sizeof name, "k_fill16<%d, %s>", bk, b->packed_seg ? "true" : "false");
      if (b->packed_mat) strncat(name, " matrix", sizeof name - strlen(name) - 1);
      break;
    case pw::VAR_FAST_ANY_TRACK: snprintf(name, sizeof name, "k_fill<%s, %d, true, true, false>", t, bk); break;
    case pw::VAR_FAST_TRACK: snprintf(name, sizeof name, "k_fill<%s, %d, false, true, false>", t, bk); break;
    case pw::VAR_FAST: snprintf(name, sizeof name, "k_fill<%s, %d, false, false, false>", t, bk); break;
    default: snprintf(name, sizeof name, "k_fill<%s, %d, false, true, true>", t, bk); break;
  }
  return name;
}

void* pw_arena_upload(int device, const uint8_t* host, uint64_t bytes) {
  void* p = nullptr;
  if (hipSetDevice(device) != hipSuccess || hipMalloc(&p, bytes + 16) != hipSuccess) { (void)hipGetLastError(); fail("pw_arena_upload: allocation failed"); return nullptr; }
  if (hipMemset((uint8_t*)p + bytes, 0, 16) != hipSuccess || (bytes && hipMemcpy(p, host, bytes, hipMemcpyHostToDevice) != hipSuccess)) {
    (void)hipGetLastError(); (void)hipFree(p); fail("pw_arena_upload: copy failed"); return nullptr;
  }
  return p;
}
void pw_arena_free(int device, void* p) { if (p) { (void)hipSetDevice(device); (void)hipFree(p); } }
int pw_batch_share_arena(pw_batch* b, void* dev) {
  if (!b->arena_shared) return fail("pw_batch_share_arena: the batch was not created with PW_FLAG_SHARED_ARENA");
  if (!dev) return fail("pw_batch_share_arena: null arena");
  b->d_arena = (uint8_t*)dev;
  return 0;
}

int pw_batch_upload_arena(pw_batch* b, const uint8_t* host, uint64_t bytes) {
  if (b->arena_shared) return fail("the batch shares a caller-owned arena (PW_FLAG_SHARED_ARENA): nothing to upload");
  if (bytes > b->arena_bytes) return fail("arena upload larger than the arena");
  HIP_TRY(hipSetDevice(b->device));
  if (bytes) HIP_TRY(hipMemcpy(b->d_arena, host, bytes, hipMemcpyHostToDevice));
  return 0;
}

void* pw_batch_arena_device(pw_batch* b) { return b->d_arena; }

void* pw_host_alloc(uint64_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); fail("hipHostMalloc failed"); return nullptr; }
  return p;
}
void pw_host_free(void* p) { if (p) (void)hipHostFree(p); }

int pw_batch_upload_arena_async(pw_batch* b, const uint8_t* host, uint64_t bytes, void* stream) {
  if (b->arena_shared) return fail("the batch shares a caller-owned arena (PW_FLAG_SHARED_ARENA): nothing to upload");
  if (bytes > b->arena_bytes) return fail("arena upload larger than the arena");
  HIP_TRY(hipSetDevice(b->device));
  if (bytes) HIP_TRY(hipMemcpyAsync(b->d_arena, host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return mark_done(b, (hipStream_t)stream);
}
int pw_batch_results_async(pw_batch* b, pw_result* out, void* stream) {
  HIP_TRY(hipSetDevice(b->device));
  if (b->n) HIP_TRY(hipMemcpyAsync(out, b->d_results, sizeof(pw_result) * (size_t)b->n, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return mark_done(b, (hipStream_t)stream);
}
int pw_batch_transcripts_async(pw_batch* b, uint8_t* out, void* stream) {
  HIP_TRY(hipSetDevice(b->device));
  if (b->tx_bytes) HIP_TRY(hipMemcpyAsync(out, b->d_tx, b->tx_bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return mark_done(b, (hipStream_t)stream);
}

int pw_batch_solve(pw_batch* b, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (b->arena_shared && !b->d_arena) return fail("pw_batch_solve: the batch was created with PW_FLAG_SHARED_ARENA and has no arena yet");
  HIP_TRY(hipSetDevice(b->device));
  b->traced = false; b->traced_from = false;
  if (!b->repaired.empty()) {          // a new solve: the strips get another chance
    HIP_TRY(hipDeviceSynchronize());
    for (auto& r : b->repaired) pw_batch_destroy(r.second);
    b->repaired.clear();
  }
  if (b->flags & PW_FLAG_PROFILE) HIP_TRY(hipEventRecord(b->ev_fill0, st));
  int rc = b->variant == pw::VAR_FAST16 ? launch_packed_fill(b, st)
           : b->use_f64 ? launch_all_fills<double>(b, st) : launch_all_fills<int32_t>(b, st);
  if (rc == 0 && !b->strips.empty()) rc = launch_strip_fills(b, st);
  if (rc != 0) return rc;
  if (b->flags & PW_FLAG_PROFILE) { HIP_TRY(hipEventRecord(b->ev_fill1, st)); b->fill_timed = true; }
  return mark_done(b, st);
}

static pw_batch* repaired_sub(pw_batch* b, int32_t k) {
  for (auto& r : b->repaired) if (r.first == k) return r.second;
  return nullptr;
}

// The replacement of a repaired strip pair repeats the batch's last traceback call; its record and transcript slot (same
// capacity, ops right-aligned) are copied over the pair's own, device to device, on the same stream.
static int replay_trace(pw_batch* b, int32_t k, pw_batch* sub, hipStream_t st) {
  if (b->traced_from) {
    const int32_t e[2] = {b->last_ends[2 * (size_t)k], b->last_ends[2 * (size_t)k + 1]};
    if (pw_batch_traceback_from(sub, e, st) != 0) return -1;
  } else if (pw_batch_traceback(sub, st) != 0) return -1;
  const pw::PairDesc& d = b->descs[k];
  HIP_TRY(hipMemcpyAsync(b->d_results + k, sub->d_results, sizeof(pw::Result), hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(b->d_tx + d.tx_off, sub->d_tx + sub->descs[0].tx_off, (size_t)d.tx_cap, hipMemcpyDeviceToDevice, st));
  return 0;
}

// A strip pair whose pipeline was abandoned (a wavefront waited longer than PWLIB_STRIP_SPIN_LIMIT polls for the strip above
// it: a starved or contended device) is solved once more inside the library, as a batch of one that shares the arena and
// may not use the strips (tiled or workgroup kernels).  Synchronous; called from pw_batch_results.
static int repair_strip_pair(pw_batch* b, int32_t k) {
  pw_scoring sc;
  sc.mode = b->mode; sc.type = b->type; sc.alphabet_len = b->L; sc.subst = b->subst_in.data(); sc.go = b->go_in; sc.ge = b->ge_in;
  const uint32_t keep = b->flags & (PW_FLAG_FORCE_F64 | PW_FLAG_FORCE_GENERIC | PW_FLAG_NO_PACKED16);
  pw_batch* sub = pw_batch_create(b->device, &sc, 1, &b->pairs[k], b->arena_bytes, keep | PW_FLAG_SHARED_ARENA | PW_FLAG_NO_STRIP);
  if (!sub) return -1;
  if (!sub->strips.empty()) { pw_batch_destroy(sub); return fail("internal: the replacement of a strip pair took the strips again"); }
  if (pw_batch_share_arena(sub, b->d_arena) != 0 || pw_batch_solve(sub, nullptr) != 0) { pw_batch_destroy(sub); return -1; }
  b->repaired.emplace_back(k, sub);
  if (b->traced) { if (replay_trace(b, k, sub, nullptr) != 0) return -1; }
  else HIP_TRY(hipMemcpyAsync(b->d_results + k, sub->d_results, sizeof(pw::Result), hipMemcpyDeviceToDevice, nullptr));
  HIP_TRY(hipStreamSynchronize(nullptr));
  static const bool verbose = env_int("PWLIB_TIMING", 0) != 0;
  if (verbose) fprintf(stderr, "pwlib: strip pipeline of pair %d abandoned; solved again on %s\n", (int)k, pw_batch_kernel_name(sub));
  return 0;
}

static int do_trace(pw_batch* b, const int32_t* d_ends, hipStream_t st) {
  pw::TraceParams p;
  memset(&p, 0, sizeof p);
  p.pairs = b->d_pairs; p.arena = b->d_arena; p.masks = b->d_masks; p.results = b->d_results;
  p.transcripts = b->d_tx; p.npairs = b->n; p.gosign = b->gosign;
  p.banded = b->mode == pw::BANDED_MODE; p.ends = d_ends;
  // long transcripts (strip-pipeline pairs) are fixed up by several wavefronts each: ~2000 ops per segment
  if (!b->strips.empty()) {
    int64_t longest = 0;
    for (int32_t k : b->strips) longest = std::max<int64_t>(longest, b->descs[k].tx_cap);
    p.fix_segments = (int32_t)std::min<int64_t>(256, std::max<int64_t>(1, longest / 2048));
  }
  if (b->flags & PW_FLAG_PROFILE) HIP_TRY(hipEventRecord(b->ev_tr0, st));
  for (int32_t k : b->strips) {        // strip-layout pairs: one wavefront each (before the fix-up pass below)
    if (repaired_sub(b, k)) continue;  // (its mask plane is that of an abandoned fill: the replacement walks its own)
    const pw::PairDesc& d = b->descs[k];
    pw::StripTraceParams sp;
    memset(&sp, 0, sizeof sp);
    sp.masks = b->d_masks + d.mask_off; sp.result = b->d_results + k; sp.tx = b->d_tx + d.tx_off;
    sp.ends = d_ends ? d_ends + 2 * k : nullptr;
    sp.X = d.X; sp.Y = d.Y; sp.nkq = (d.Y + 64 + pw::kStripBlock - 1) / pw::kStripBlock; sp.tx_cap = d.tx_cap; sp.gosign = b->gosign;
    HIP_TRY(pw::launch_strip_trace(sp, st));
  }
  HIP_TRY(pw::launch_trace(p, st));
  for (auto& r : b->repaired) if (replay_trace(b, r.first, r.second, st) != 0) return -1;
  if (b->flags & PW_FLAG_PROFILE) { HIP_TRY(hipEventRecord(b->ev_tr1, st)); b->trace_timed = true; }
  return mark_done(b, st);
}

int pw_batch_traceback(pw_batch* b, void* stream) {
  HIP_TRY(hipSetDevice(b->device));
  b->traced = true; b->traced_from = false;
  return do_trace(b, nullptr, (hipStream_t)stream);
}

int pw_batch_traceback_from(pw_batch* b, const int32_t* ends_ij, void* stream) {
  HIP_TRY(hipSetDevice(b->device));
  hipStream_t st = (hipStream_t)stream;
  if (!b->d_ends) HIP_TRY(hipMalloc((void**)&b->d_ends, 8 * std::max<int32_t>(b->n, 1)));
  // validate on the host: an end cell outside the table would send the walker out of the mask plane
  for (int32_t k = 0; k < b->n; k++) {
    if (!b->descs[k].solvable) continue;
    const int i = ends_ij[2 * k], j = ends_ij[2 * k + 1];
    if (i < 0 && j < 0) continue;
    const pw::Plan& pl = b->plans[k];
    const int X = b->pairs[k].origin_len, Y = b->pairs[k].mutant_len;
    bool ok;
    if (b->mode == pw::STD_MODE) ok = i >= 0 && i <= X && j >= 0 && j <= Y;
    else ok = i >= 0 && i < pl.num_rows && j >= 0 && j < pw::plan_len(X, Y, pl.dmin + i);
    if (!ok) return fail("traceback end cell outside the table");
  }
  HIP_TRY(hipMemcpyAsync(b->d_ends, ends_ij, 8 * (size_t)b->n, hipMemcpyHostToDevice, st));
  b->traced = true; b->traced_from = true;
  b->last_ends.assign(ends_ij, ends_ij + 2 * (size_t)b->n);
  return do_trace(b, b->d_ends, st);
}

int pw_batch_sync(pw_batch* b, void* stream) {
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

void* pw_batch_results_device(pw_batch* b) { return b->d_results; }
void* pw_batch_transcripts_device(pw_batch* b) { return b->d_tx; }
uint64_t pw_batch_transcripts_bytes(const pw_batch* b) { return b->tx_bytes; }

int pw_batch_tx_slot(const pw_batch* b, int32_t k, uint64_t* off, int32_t* cap) {
  if (k < 0 || k >= b->n) return -1;
  if (off) *off = b->descs[k].tx_off;
  if (cap) *cap = b->descs[k].tx_cap;
  return 0;
}

int pw_batch_results(pw_batch* b, pw_result* out) {
  HIP_TRY(hipSetDevice(b->device));
  if (b->n) HIP_TRY(hipMemcpy(out, b->d_results, sizeof(pw_result) * (size_t)b->n, hipMemcpyDeviceToHost));
  // (the D2H copy above has waited for everything launched on the default stream; callers that launched on another stream
  //  synchronise it first, as for any read of the records)
  for (int32_t k : b->strips)
    if ((out[k].status & PW_ST_BADPATH) && out[k].opt_i < 0 && !repaired_sub(b, k)) {
      if (env_int("PWLIB_NO_STRIP_REPAIR", 0))
        return fail("the strip pipeline of a wide pair was abandoned (a wavefront waited too long for the strip above it)");
      if (repair_strip_pair(b, k) != 0)
        return fail("the strip pipeline of a wide pair was abandoned and solving the pair again without it failed: " + std::string(g_err));
      HIP_TRY(hipMemcpy(out + k, b->d_results + k, sizeof(pw_result), hipMemcpyDeviceToHost));
    }
  return 0;
}

int pw_batch_transcripts(pw_batch* b, uint8_t* out) {
  HIP_TRY(hipSetDevice(b->device));
  if (b->tx_bytes) HIP_TRY(hipMemcpy(out, b->d_tx, b->tx_bytes, hipMemcpyDeviceToHost));
  return 0;
}

int pw_batch_pack_transcripts(pw_batch* b, void* stream) {
  HIP_TRY(hipSetDevice(b->device));
  if (!b->d_txoffsets) {
    HIP_TRY(pool_alloc(b->device, (void**)&b->d_txpacked, std::max<uint64_t>(b->tx_bytes, 16), &b->txpacked_alloc));
    HIP_TRY(hipMalloc((void**)&b->d_txoffsets, 8 * ((size_t)b->n + 1)));
  }
  HIP_TRY(pw::launch_tx_pack(b->d_pairs, b->d_results, b->d_tx, b->n, b->d_txoffsets, b->d_txpacked, (hipStream_t)stream));
  return mark_done(b, (hipStream_t)stream);
}
void* pw_batch_packed_device(pw_batch* b) { return b->d_txpacked; }
void* pw_batch_packed_offsets_device(pw_batch* b) { return b->d_txoffsets; }
int pw_batch_packed_total_async(pw_batch* b, uint64_t* host_out, void* stream) {
  if (!b->d_txoffsets) return fail("pw_batch_packed_total_async before pw_batch_pack_transcripts");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipMemcpyAsync(host_out, b->d_txoffsets + b->n, 8, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return mark_done(b, (hipStream_t)stream);
}
int pw_batch_packed(pw_batch* b, uint8_t* out, uint64_t cap, uint64_t* offsets_out) {
  if (!b->d_txoffsets) return fail("pw_batch_packed before pw_batch_pack_transcripts");
  HIP_TRY(hipSetDevice(b->device));
  std::vector<uint64_t> off((size_t)b->n + 1);
  HIP_TRY(hipMemcpy(off.data(), b->d_txoffsets, 8 * off.size(), hipMemcpyDeviceToHost));
  if (offsets_out) memcpy(offsets_out, off.data(), 8 * off.size());
  if (out) {
    if (off[b->n] > cap) return fail("packed transcripts: buffer too small");
    if (off[b->n]) HIP_TRY(hipMemcpy(out, b->d_txpacked, off[b->n], hipMemcpyDeviceToHost));
  }
  return 0;
}

int pw_batch_scores(pw_batch* b, int32_t k, double* out, int64_t n) {
  if (!(b->flags & PW_FLAG_DUMP_SCORES) || k < 0 || k >= b->n || !b->descs[k].solvable) return fail("no score plane");
  const pw::PairDesc& d = b->descs[k];
  const int64_t want = (int64_t)d.ndiag * d.h_pitch;
  if (n < want) return fail("score buffer too small");
  HIP_TRY(hipSetDevice(b->device));
  if (b->use_f64) {
    HIP_TRY(hipMemcpy(out, (double*)b->d_hdump + d.h_off, 8 * (size_t)want, hipMemcpyDeviceToHost));
  } else {
    std::vector<int32_t> tmp((size_t)want);
    HIP_TRY(hipMemcpy(tmp.data(), (int32_t*)b->d_hdump + d.h_off, 4 * (size_t)want, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < want; i++) out[i] = (double)tmp[(size_t)i] * b->score_mul;
  }
  return 0;
}

int pw_batch_table(pw_batch* b, int32_t k, double* out, int64_t n) {
  if (!(b->flags & PW_FLAG_DUMP_SCORES) || k < 0 || k >= b->n || !b->descs[k].solvable) return fail("no score plane");
  if (b->mode != pw::STD_MODE) return fail("pw_batch_table: standard mode only");
  const pw::PairDesc& d = b->descs[k];
  const int64_t want = (int64_t)(d.X + 1) * (d.Y + 1);
  if (n < want) return fail("table buffer too small");
  HIP_TRY(hipSetDevice(b->device));
  double* dev = nullptr;
  HIP_TRY(hipMalloc((void**)&dev, 8 * (size_t)want));
  const void* plane = b->use_f64 ? (const void*)((double*)b->d_hdump + d.h_off) : (const void*)((int32_t*)b->d_hdump + d.h_off);
  hipError_t e = pw::launch_table_rowmajor(plane, b->use_f64, d.X, d.Y, d.h_pitch, b->score_mul, dev, nullptr);
  if (e == hipSuccess) e = hipMemcpy(out, dev, 8 * (size_t)want, hipMemcpyDeviceToHost);
  (void)hipFree(dev);
  if (e != hipSuccess) return fail(hipGetErrorString(e));
  return 0;
}

float pw_batch_fill_ms(pw_batch* b) {
  if (!b->fill_timed) return -1.f;
  float ms = -1.f;
  (void)hipSetDevice(b->device);
  if (hipEventSynchronize(b->ev_fill1) != hipSuccess) return -1.f;
  if (hipEventElapsedTime(&ms, b->ev_fill0, b->ev_fill1) != hipSuccess) return -1.f;
  return ms;
}

float pw_batch_trace_ms(pw_batch* b) {
  if (!b->trace_timed) return -1.f;
  float ms = -1.f;
  (void)hipSetDevice(b->device);
  if (hipEventSynchronize(b->ev_tr1) != hipSuccess) return -1.f;
  if (hipEventElapsedTime(&ms, b->ev_tr0, b->ev_tr1) != hipSuccess) return -1.f;
  return ms;
}

}  // extern "C"

// =================================================================================================
// The four drop-in functions (include/pwlib.h): one problem per dptable, solved as a batch of one.
// =================================================================================================
namespace {

const uint64_t kMagic = 0x70776c69622d6869ull;   // "pwlib-hi"

// Lives immediately in front of the row-pointer array T->cells points at.
struct Hidden {
  uint64_t magic;
  pw_batch* batch;
  dpcell* cell_slab;          // all rows, back to back (NULL for huge tables: lazy)
  dpcell* opt_row;            // lazy tables: the one row that holds the optimal cell
  bool lazy;
  alnchoice* choice_slab;     // materialised choices
  int64_t ncells;
  int L;
  int dmin_c;
  std::vector<alignment*>* alns;
};

Hidden* hidden_of(dptable* T) {
  if (!T || !T->cells) return nullptr;
  Hidden* h = (Hidden*)((char*)T->cells - sizeof(Hidden));
  return h->magic == kMagic ? h : nullptr;
}

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

}  // namespace

extern "C" {

int dptable_init(dptable* T) {
  if (!T || !T->prob || !T->prob->frame || !T->prob->scores) return -1;
  alnprob* prob = T->prob;
  alnframe* fr = prob->frame;
  const int X = fr->origin_range.j - fr->origin_range.i, Y = fr->mutant_range.j - fr->mutant_range.i;
  if (prob->mode != STD_MODE && prob->mode != BANDED_MODE) {
    printf("Panick in %s (%d): %s\n", "pwlib", __LINE__, "Shouldn't have happened!");   // pw.c:22-23 exits here
    return -1;
  }
  if (X < 0 || Y < 0) { fprintf(stderr, "pwlib: negative frame length\n"); return -1; }
  if (prob->max_new_mins > 0) {
    fprintf(stderr, "pwlib: max_new_mins > 0 is not supported (the reference reads uninitialised memory there)\n");
    return -1;
  }
  int type, dmin = 0, dmax = 0;
  if (prob->mode == STD_MODE) type = (int)prob->std_params->type;
  else { type = (int)prob->banded_params->type; dmin = prob->banded_params->dmin; dmax = prob->banded_params->dmax; }
  pw::Plan pl = pw::plan_problem((int)prob->mode, type, X, Y, dmin, dmax);
  if (prob->mode == BANDED_MODE) {
    if (pl.clamped) {   // _pw_internals.c:32-35: message on stdout, caller's struct updated
      printf("Band [%d, %d] exceeds table limits, reduced it to [%d, %d].\n", dmin, dmax, pl.dmin, pl.dmax);
      prob->banded_params->dmin = pl.dmin; prob->banded_params->dmax = pl.dmax;
    }
    if (pl.rc != 0) {
      if (type == B_GLOBAL && (X - Y > pl.dmax || X - Y < pl.dmin || (int64_t)pl.dmax * pl.dmin > 0))
        printf("End points not within band for global alignment!\n");            // :41
      else printf("Invalid band: [%d, %d]!\n", pl.dmin, pl.dmax);                // :47
      return -1;
    }
  }
  if (pl.ndiag > (1 << 21)) {
    fprintf(stderr, "pwlib: %d diagonals exceed the widest GPU fill kernel (%d); refusing (no CPU fallback)\n",
            pl.ndiag, 1 << 21);
    return -1;
  }
  T->num_rows = pl.num_rows;
  T->row_lens = (int*)malloc(sizeof(int) * (size_t)std::max(pl.num_rows, 1));
  char* blk = (char*)malloc(sizeof(Hidden) + sizeof(dpcell*) * (size_t)std::max(pl.num_rows, 1));
  if (!T->row_lens || !blk) { fprintf(stderr, "pwlib: out of memory\n"); free(T->row_lens); free(blk); return -1; }
  Hidden* h = (Hidden*)blk;
  h->magic = kMagic; h->batch = nullptr; h->choice_slab = nullptr; h->L = 0; h->dmin_c = pl.dmin;
  h->alns = new std::vector<alignment*>();
  h->ncells = pl.cells;
  // Tables beyond 2^26 cells (1 GiB of dpcell) do not get their cells on the host: the row-pointer array is
  // there, rows are NULL, and dptable_solve allocates only the row of the optimal cell (what pw.py:272 reads).
  h->lazy = pl.cells > ((int64_t)1 << 26);
  h->opt_row = nullptr;
  h->cell_slab = h->lazy ? nullptr : (dpcell*)calloc((size_t)std::max<int64_t>(pl.cells, 1), sizeof(dpcell));   // all empty (:64-74)
  if (!h->lazy && !h->cell_slab) { fprintf(stderr, "pwlib: out of memory\n"); delete h->alns; free(T->row_lens); free(blk); return -1; }
  T->cells = (dpcell**)(blk + sizeof(Hidden));
  int64_t off = 0;
  for (int i = 0; i < pl.num_rows; i++) {
    const int len = prob->mode == STD_MODE ? Y + 1 : pw::plan_len(X, Y, pl.dmin + i);
    T->row_lens[i] = len;
    T->cells[i] = h->lazy ? nullptr : h->cell_slab + off;
    off += len;
  }
  return 0;
}

void dptable_free(dptable* T) {
  Hidden* h = hidden_of(T);
  if (!h) return;
  pw_batch_destroy(h->batch);
  for (alignment* a : *h->alns) { free(a->transcript); free(a); }
  delete h->alns;
  free(h->choice_slab);
  free(h->cell_slab);
  free(h->opt_row);
  h->magic = 0;
  free((char*)T->cells - sizeof(Hidden));
  free(T->row_lens);
  T->cells = NULL; T->row_lens = NULL; T->num_rows = -1;
}

intpair dptable_solve(dptable* T) {
  const intpair none = {-1, -1};
  Hidden* h = hidden_of(T);
  if (!h) { fprintf(stderr, "pwlib: dptable_solve on an uninitialised table\n"); return none; }
  alnprob* prob = T->prob;
  alnframe* fr = prob->frame;
  const int X = fr->origin_range.j - fr->origin_range.i, Y = fr->mutant_range.j - fr->mutant_range.i;
  if (T->num_rows <= 0) return none;
  // letters -> one byte each; the alphabet size is not part of the ABI: use the largest letter seen
  const size_t moff = ((size_t)X + 3) / 4 * 4;     // frames start on 4-byte boundaries
  std::vector<uint8_t> arena(moff + (size_t)Y + 16, 0);
  int maxlet = 0;
  for (int i = 0; i < X; i++) { const int c = fr->origin[fr->origin_range.i + i]; if (c < 0 || c > 255) { fprintf(stderr, "pwlib: letter %d out of range 0..255\n", c); return none; } arena[i] = (uint8_t)c; maxlet = std::max(maxlet, c); }
  for (int i = 0; i < Y; i++) { const int c = fr->mutant[fr->mutant_range.i + i]; if (c < 0 || c > 255) { fprintf(stderr, "pwlib: letter %d out of range 0..255\n", c); return none; } arena[moff + i] = (uint8_t)c; maxlet = std::max(maxlet, c); }
  const int L = maxlet + 1;
  std::vector<double> subst((size_t)L * L);
  for (int i = 0; i < L; i++) for (int j = 0; j < L; j++) subst[(size_t)i * L + j] = prob->scores->subst_scores[i][j];
  pw_scoring sc;
  sc.mode = (int)prob->mode;
  sc.type = prob->mode == STD_MODE ? (int)prob->std_params->type : (int)prob->banded_params->type;
  sc.alphabet_len = L; sc.subst = subst.data();
  sc.go = prob->scores->gap_open_score; sc.ge = prob->scores->gap_extend_score;
  pw_pair pr;
  pr.origin_off = 0; pr.mutant_off = (uint64_t)moff; pr.origin_len = X; pr.mutant_len = Y;
  pr.dmin = prob->mode == BANDED_MODE ? prob->banded_params->dmin : 0;
  pr.dmax = prob->mode == BANDED_MODE ? prob->banded_params->dmax : 0;
  const bool timing = env_int("PWLIB_TIMING", 0) != 0;
  auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_a = now();
  // the full table (host: 48 B per cell) is materialised for tables up to 2^24 cells; beyond that only the
  // optimal cell is (table_scores-style callers need PWLIB_NO_TABLE unset and a table that size)
  const bool want_table = prob->mode == STD_MODE && !env_int("PWLIB_NO_TABLE", 0) && h->ncells <= (1 << 24);
  if (h->batch) { pw_batch_destroy(h->batch); h->batch = nullptr; }
  h->batch = pw_batch_create(env_int("PWLIB_DEVICE", 0), &sc, 1, &pr, arena.size(), want_table ? PW_FLAG_DUMP_SCORES : 0);
  if (!h->batch) { fprintf(stderr, "pwlib: %s\n", pw_last_error()); return none; }
  pw_result res;
  if (pw_batch_upload_arena(h->batch, arena.data(), arena.size()) != 0 || pw_batch_solve(h->batch, nullptr) != 0 ||
      pw_batch_sync(h->batch, nullptr) != 0 || pw_batch_results(h->batch, &res) != 0) {
    fprintf(stderr, "pwlib: %s\n", pw_last_error());
    return none;
  }
  const double t_b = now();
  double t_plane = 0.0;
  // ---- materialise what the reference's callers read out of C memory ----
  free(h->choice_slab); h->choice_slab = nullptr;
  if (want_table) {
    // every cell's choices[0].score (Aligner.table_scores, pw.py:278-285): the table comes back row-major
    double* table = (double*)malloc(sizeof(double) * (size_t)h->ncells);
    h->choice_slab = (alnchoice*)malloc(sizeof(alnchoice) * (size_t)h->ncells);
    if (!table || !h->choice_slab) { fprintf(stderr, "pwlib: out of memory\n"); free(table); return none; }
    if (pw_batch_table(h->batch, 0, table, h->ncells) != 0) { fprintf(stderr, "pwlib: %s\n", pw_last_error()); free(table); return none; }
    t_plane = now() - t_b;
    const int mins_cd = prob->max_new_mins;
    // 48 bytes per cell of freshly allocated host memory: first touch and stores are spread over a few threads
    auto fill_cells = [&](int64_t c0, int64_t c1) {
      for (int64_t c = c0; c < c1; c++) {
        alnchoice* ch = &h->choice_slab[c];
        ch->op = 0; ch->score = table[c]; ch->base = NULL; ch->mins_cd = mins_cd; ch->cur_min = 0;
        h->cell_slab[c].num_choices = 1; h->cell_slab[c].choices = ch;
      }
    };
    const int nth = h->ncells >= (1 << 16) ? std::max(1, std::min(8, (int)std::thread::hardware_concurrency())) : 1;
    if (nth <= 1) fill_cells(0, h->ncells);
    else {
      std::vector<std::thread> th;
      const int64_t per = (h->ncells + nth - 1) / nth;
      for (int t = 0; t < nth; t++) {
        const int64_t c0 = t * per, c1 = std::min<int64_t>(h->ncells, c0 + per);
        if (c0 < c1) th.emplace_back(fill_cells, c0, c1);
      }
      for (auto& t : th) t.join();
    }
    free(table);
  } else if (res.opt_i >= 0 && res.opt_j >= 0) {
    if (h->lazy) {
      free(h->opt_row);
      h->opt_row = (dpcell*)calloc((size_t)T->row_lens[res.opt_i], sizeof(dpcell));
      if (!h->opt_row) { fprintf(stderr, "pwlib: out of memory\n"); return none; }
      T->cells[res.opt_i] = h->opt_row;
    }
    h->choice_slab = (alnchoice*)calloc(1, sizeof(alnchoice));
    if (!h->choice_slab) { fprintf(stderr, "pwlib: out of memory\n"); return none; }
    h->choice_slab->score = res.score; h->choice_slab->base = NULL; h->choice_slab->mins_cd = prob->max_new_mins;
    T->cells[res.opt_i][res.opt_j].num_choices = 1;
    T->cells[res.opt_i][res.opt_j].choices = h->choice_slab;
  }
  if (timing) fprintf(stderr, "pwlib timing: create+upload+solve+sync %.2f ms, materialise %.2f ms (score plane D2H %.2f ms)\n", t_b - t_a, now() - t_b, t_plane);
  intpair opt = {res.opt_i, res.opt_j};
  return opt;
}

alignment* dptable_traceback(dptable* T, intpair end) {
  Hidden* h = hidden_of(T);
  if (!h || !h->batch) { fprintf(stderr, "pwlib: dptable_traceback before dptable_solve\n"); return NULL; }
  const int32_t ends[2] = {end.i, end.j};
  pw_result res;
  if (pw_batch_traceback_from(h->batch, ends, nullptr) != 0 || pw_batch_sync(h->batch, nullptr) != 0 ||
      pw_batch_results(h->batch, &res) != 0) {
    fprintf(stderr, "pwlib: %s\n", pw_last_error());
    return NULL;
  }
  if (!(res.status & PW_ST_TRACED)) return NULL;
  if (res.status & PW_ST_PANICK) {   // the reference exits the process here (pw.c:132-134)
    printf("Panick in %s (%d): %s\n", "pwlib", __LINE__, "Shouldn't have happened!");
    return NULL;
  }
  if (res.tx_len == 0) return NULL;  // empty transcript (pw.c:135-138)
  uint64_t off; int32_t cap;
  pw_batch_tx_slot(h->batch, 0, &off, &cap);
  std::vector<uint8_t> tx(pw_batch_transcripts_bytes(h->batch));
  if (pw_batch_transcripts(h->batch, tx.data()) != 0) { fprintf(stderr, "pwlib: %s\n", pw_last_error()); return NULL; }
  alignment* a = (alignment*)malloc(sizeof(alignment));
  a->transcript = (char*)malloc((size_t)res.tx_len + 1);
  memcpy(a->transcript, tx.data() + off + cap - res.tx_len, (size_t)res.tx_len);
  a->transcript[res.tx_len] = 0;
  a->origin_idx = res.origin_idx + T->prob->frame->origin_range.i;
  a->mutant_idx = res.mutant_idx + T->prob->frame->mutant_range.i;
  // score of the END cell given (pw.c:148): the solve score for the optimal cell, the materialised
  // table for any other cell (standard mode), otherwise unknown
  if (end.i == res.opt_i && end.j == res.opt_j) a->score = res.score;
  else if (T->cells[end.i] && T->cells[end.i][end.j].num_choices > 0) a->score = T->cells[end.i][end.j].choices[0].score;
  else a->score = NAN;
  h->alns->push_back(a);
  return a;
}

}  // extern "C"
