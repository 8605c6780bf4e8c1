// tests/emu/emu_wave.cpp -- TEST INFRASTRUCTURE ONLY: a lockstep CPU emulator of one 64-lane wavefront.
//
// It compiles the *same* lane program the gfx950 kernels are built from (biseqt_amd/csrc/pw_wave.h)
// and the same host planner (pw_plan.h), and runs the 64 lanes as 64 ucontext fibers that meet at
// every cross-lane operation.  Purpose: debug the indexing / tie-break logic of the kernel on the CPU
// (this container has no GPU) by comparing it with the oracle.  It is never linked into the product
// library and the product never falls back to it.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <functional>
#include <vector>

#define PW_FN inline
#include "../../biseqt_amd/csrc/pw_wave.h"
#include "../../biseqt_amd/csrc/pw_strip.h"
#include "../../biseqt_amd/csrc/pw_plan.h"

namespace {

struct Emu {
  static const int N = 64;          // lanes of a wavefront
  static const int MAXN = 512;      // ... of a workgroup of up to 8 wavefronts (the multi-wavefront kernels: `n` = 64 x wavefronts)
  int n = N;
  ucontext_t main_ctx, ctx[MAXN];
  std::vector<char> stacks[MAXN];
  bool done[MAXN];
  int cur;
  int xch[2][MAXN];
  unsigned phase[MAXN];
  std::function<void()> body;
  static Emu* self;

  static void tramp() {
    Emu* e = self;
    e->body();
    e->done[e->cur] = true;
    swapcontext(&e->ctx[e->cur], &e->main_ctx);
  }
  void run(std::function<void()> fn) {
    body = fn;
    self = this;
    for (int l = 0; l < n; l++) {
      stacks[l].resize(1 << 18);
      done[l] = false; phase[l] = 0;
      getcontext(&ctx[l]);
      ctx[l].uc_stack.ss_sp = stacks[l].data();
      ctx[l].uc_stack.ss_size = stacks[l].size();
      ctx[l].uc_link = &main_ctx;
      makecontext(&ctx[l], (void (*)())tramp, 0);
    }
    bool any = true;
    while (any) {
      any = false;
      for (int l = 0; l < n; l++) {
        if (done[l]) continue;
        cur = l;
        swapcontext(&main_ctx, &ctx[l]);
        any = true;
      }
    }
  }
  void barrier() { int l = cur; swapcontext(&ctx[l], &main_ctx); }
  int exchange(int v, int src_lane, int old) {
    const int l = cur;
    const unsigned ph = phase[l]++ & 1u;
    xch[ph][l] = v;
    barrier();
    cur = l;   // (the scheduler sets cur before resuming; keep the local view explicit)
    return (src_lane >= 0 && src_lane < n) ? xch[ph][src_lane] : old;
  }
};
Emu* Emu::self = nullptr;

struct EmuP {
  static uint32_t const_dword(const uint8_t* base, int idx) { uint32_t v; memcpy(&v, base + 4 * (size_t)idx, 4); return v; }
  template <typename T, bool A> static T tab_read(const T* tab, uint32_t, uint32_t off, bool) { return *(const T*)((const char*)tab + off); }
  static uint32_t shl1_in(uint32_t m, bool flag) { return (m << 1) | (flag ? 1u : 0u); }
  static int lane() { return Emu::self->cur; }
  static int32_t shr1(int32_t v, int32_t old) { Emu* e = Emu::self; return e->exchange(v, e->cur - 1, old); }
  static int32_t shl1(int32_t v, int32_t old) { Emu* e = Emu::self; return e->exchange(v, e->cur + 1, old); }
  static int32_t shr1z(int32_t v) { Emu* e = Emu::self; return e->exchange(v, e->cur - 1, 0); }
  static int32_t shl1z(int32_t v) { Emu* e = Emu::self; return e->exchange(v, e->cur + 1, 0); }
  static int32_t shfl_xor(int32_t v, int m) { Emu* e = Emu::self; return e->exchange(v, e->cur ^ m, 0); }
  static int32_t uniform(int32_t v) { return v; }
  static int nlanes() { return 64; }
  static int nwaves() { return 1; }
  static int32_t wave_bcast(int32_t v, int) { return v; }
  static int lane0() { return 0; }
  static bool central() { return true; }
  static constexpr bool kVirtualLanes = false;
  static constexpr bool kBatchedShifts = false;
};

// A workgroup of several wavefronts as ONE long row of lanes (pw_device.h, DevPM: k_fill_mw / k_fill16_mw): the value that
// crosses a wavefront boundary goes through LDS there; here every lane simply has a neighbour.
struct EmuPM : EmuP {
  static int nlanes() { return Emu::self->n; }
  static int nwaves() { return Emu::self->n / 64; }
  static constexpr bool kBatchedShifts = true;
  template <int N> static void shrv(int32_t* v, const int32_t* old, int) {
    Emu* e = Emu::self;
    for (int i = 0; i < N; i++) v[i] = e->exchange(v[i], e->cur - 1, old[i]);
  }
  template <int N> static void shlv(int32_t* v, const int32_t* old, int) {
    Emu* e = Emu::self;
    for (int i = 0; i < N; i++) v[i] = e->exchange(v[i], e->cur + 1, old[i]);
  }
  static int32_t wave_bcast(int32_t v, int w) { Emu* e = Emu::self; return e->exchange(v, 64 * w, 0); }
};
int g_waves = 1;
extern "C" void emu_set_waves(int w) { g_waves = w < 1 ? 1 : (w > 8 ? 8 : w); }

// the strip pipeline (pw_strip.h): same lanes; the FIFO between strips is plain memory (strips run one after another)
struct EmuPS : EmuP {
  static bool all(bool p) {
    int v = p ? 1 : 0;
    for (int off = 32; off >= 1; off >>= 1) v &= shfl_xor(v, off);
    return v != 0;
  }
  static uint32_t readlane(uint32_t v, int i) { Emu* e = Emu::self; return (uint32_t)e->exchange((int)v, i, 0); }
  static int32_t shl1z(int32_t v) { Emu* e = Emu::self; return e->exchange(v, e->cur + 1, 0); }
  static uint32_t lane_from(uint32_t from, uint32_t v) { Emu* e = Emu::self; return (uint32_t)e->exchange((int)v, (int)from, 0); }
  static uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t bytes) {
    return (uint32_t)(((((uint64_t)hi) << 32) | (uint64_t)lo) >> (8 * (bytes & 3u)));
  }
  static int32_t twice_plus(int32_t h, bool b) { return h + h + (b ? 1 : 0); }
  static uint32_t byte_of(uint32_t v, int i) { return (v >> (8 * i)) & 0xffu; }
  static int32_t sbyte_of(uint32_t v, int i) { return (int32_t)(int8_t)((v >> (8 * i)) & 0xffu); }
  static uint32_t perm_bytes(uint32_t row, uint32_t sel) {
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) r |= ((row >> (8 * ((sel >> (8 * i)) & 3u))) & 0xffu) << (8 * i);
    return r;
  }
  static void issue_here() {}
  static uint64_t fifo_load(const uint64_t* p) { return *p; }
  static uint64_t fifo_poll(const uint64_t* p) { return *p; }
  static uint64_t fifo_poll_local(const uint64_t* p) { return *p; }
  static uint64_t fifo_load_local(const uint64_t* p) { return *p; }
  static uint64_t* slot(int s) { static uint64_t slots[2][Emu::N]; return &slots[s][Emu::self->cur]; }
  template <int SLOT> static void slot_zero() { *slot(SLOT) = 0; }
  template <int SLOT> static void fifo_load_async(const uint64_t* p, bool) { *slot(SLOT) = *p; }
  template <int ROW> static int32_t rowmov(int32_t old, int32_t v) { const int l = Emu::self->cur; return (l >= 16 * ROW && l < 16 * ROW + 16) ? v : old; }
  template <int SLOT, int N> static uint64_t wait_vm() { return *slot(SLOT); }
  static uint32_t letters_dword(const uint8_t* m, int idx) {
    return (uint32_t)m[4 * idx] | ((uint32_t)m[4 * idx + 1] << 8) | ((uint32_t)m[4 * idx + 2] << 16) | ((uint32_t)m[4 * idx + 3] << 24);
  }
  static void letters_x8(const uint8_t* m, int idx, uint32_t (&win)[8]) { for (int d = 0; d < 8; d++) win[d] = letters_dword(m, idx + d); }
  static uint32_t flag_poll(const uint32_t* p) { return *p; }
  static void fifo_store(uint64_t* p, uint64_t v) { *p = v; }
  static void fifo_store_local(uint64_t* p, uint64_t v) { *p = v; }
  static void sleep() {}
  // (an abandoned wait: the lane's fiber ends here, like the wavefront on the device)
  static void exit_wave() { Emu* e = Emu::self; e->done[e->cur] = true; swapcontext(&e->ctx[e->cur], &e->main_ctx); }
  static uint64_t ballot(bool p) {
    Emu* e = Emu::self;
    const int l = e->cur;
    const unsigned ph = e->phase[l]++ & 1u;
    e->xch[ph][l] = p ? 1 : 0;
    e->barrier();
    e->cur = l;
    uint64_t m = 0;
    for (int j = 0; j < Emu::N; j++) if (e->xch[ph][j]) m |= (uint64_t)1 << j;
    return m;
  }
  static void wave_sync() { Emu* e = Emu::self; (void)e->exchange(0, e->cur, 0); }
  static uint64_t clock() { return 0; }
  static uint64_t cycles() { return 0; }
  static int32_t in_vgpr(int32_t v) { return v; }
  static uint32_t flag_load(const uint32_t* p) { return *p; }
  static void flag_set(uint32_t* p) { *p = 1u; }
};

template <typename T, int BK, bool BANY, bool TRACK, bool GENERIC>
void run_fill(const pw::FillParams<T>& a, const pw::PairDesc& pd, const T* subst) {
  Emu emu;
  if (g_waves > 1) {
    emu.n = 64 * g_waves;
    emu.run([&]() {
      pw::WaveFill<EmuPM, T, BK, BANY, TRACK, GENERIC> w(a, pd, subst);
      w.pair_slot = 0;
      w.run();
    });
    return;
  }
  emu.run([&]() {
    pw::WaveFill<EmuP, T, BK, BANY, TRACK, GENERIC> w(a, pd, subst);
    w.pair_slot = 0;
    w.run();
  });
}

int g_packed_mode = 0;
bool g_packed_mat = false;

template <typename T, int BK> struct Run16 {
  static bool go(const pw::FillParams<T>&, const pw::PairDesc&) { return false; }
};
template <int BK> struct Run16<int32_t, BK> {
  template <bool SEG, int RULE, bool MAT>
  static void one(const pw::FillParams<int32_t>& a, const pw::WaveDesc& wd) {
    Emu emu;
    if constexpr (!SEG) {
      if (g_waves > 1) {
        emu.n = 64 * g_waves;
        emu.run([&]() { pw::WaveFill16<EmuPM, BK, false, RULE, MAT> w(a, wd); w.run(); });
        return;
      }
    }
    emu.run([&]() { pw::WaveFill16<EmuP, BK, SEG, RULE, MAT> w(a, wd); w.run(); });
  }
  template <int RULE, bool MAT>
  static void seg_or_not(const pw::FillParams<int32_t>& a, const pw::WaveDesc& wd, bool seg) {
    if (seg) one<true, RULE, MAT>(a, wd); else one<false, RULE, MAT>(a, wd);
  }
  static bool go(const pw::FillParams<int32_t>& a, const pw::PairDesc& pd) {
    if constexpr (BK % 4 == 0) {
      // one pair in the wave, but on nl < 64 lanes: exercises the pair-boundary overrides of the lane-packed kernel
      pw::WaveDesc wd;
      memset(&wd, 0, sizeof wd);
      wd.first = 0; wd.count = 1; wd.nblocks = pd.nblocks; wd.steady_b0 = pd.steady_b0; wd.steady_b1 = pd.steady_b1;
      // packed16 == 1: lane-packed form on nl < 64 lanes; packed16 == 2: one pair per wave, uniform form
      // packed16 == 3: the scores-times-4 form of rule 0 (the caller keeps the scores below 2048); 4: the same, lane-packed
      const bool seg = g_packed_mode == 1 || g_packed_mode == 4;
      const bool x4 = g_packed_mode == 3 || g_packed_mode == 4;
      wd.nl = seg ? pd.nl : 64 * g_waves;
      const bool local_end = a.endrule == pw::END_STD_LOCAL || a.endrule == pw::END_BANDED_LOCAL;
      int rule;
      if (a.brule == pw::BRULE_ANY) rule = local_end ? (x4 ? 3 : 0) : 4;                 // LOCAL / B_LOCAL; END_ANCHORED
      else if (local_end) rule = 5;                                                     // START_ANCHORED
      else if (a.endrule == pw::END_BANDED_OVERLAP || a.endrule == pw::END_STD_OVERLAP) rule = 1;
      else rule = a.brule == pw::BRULE_ORIGIN ? 2 : 1;                                  // END_CORNER: global, or END_ANCHORED_OVERLAP
      const bool mat = g_packed_mat;
      switch (rule * 2 + (mat ? 1 : 0)) {
        case 0: seg_or_not<0, false>(a, wd, seg); break;
        case 1: seg_or_not<0, true>(a, wd, seg); break;
        case 2: seg_or_not<1, false>(a, wd, seg); break;
        case 3: seg_or_not<1, true>(a, wd, seg); break;
        case 4: seg_or_not<2, false>(a, wd, seg); break;
        case 5: seg_or_not<2, true>(a, wd, seg); break;
        case 6: seg_or_not<3, false>(a, wd, seg); break;
        case 7: seg_or_not<3, true>(a, wd, seg); break;
        case 8: seg_or_not<4, false>(a, wd, seg); break;
        case 10: seg_or_not<5, false>(a, wd, seg); break;
        default: return false;
      }
      return true;
    } else {
      return false;
    }
  }
};

template <typename T, int BK>
void dispatch_variant(const pw::FillParams<T>& a, const pw::PairDesc& pd, const T* subst, int generic,
                      int bany, int track, int packed16) {
  if (packed16 && Run16<T, BK>::go(a, pd)) return;
  if (generic) run_fill<T, BK, false, true, true>(a, pd, subst);
  else if (bany && track) run_fill<T, BK, true, true, false>(a, pd, subst);
  else if (!bany && track) run_fill<T, BK, false, true, false>(a, pd, subst);
  else if (!bany && !track) run_fill<T, BK, false, false, false>(a, pd, subst);
  else run_fill<T, BK, true, true, false>(a, pd, subst);   // (bany, !track): END_ANCHORED rides on (1,1)
}

template <typename T>
int solve_T(int mode, int type, const int* origin, int X, const int* mutant, int Y, int L,
            const double* subst, double go, double ge, int dmin_in, int dmax_in, int force_generic, int bk, int packed16,
            int* info, double* score, char* txbuf, int txcap, double* hdump) {
  pw::Plan pl = pw::plan_problem(mode, type, X, Y, dmin_in, dmax_in);
  info[0] = pl.rc; info[1] = pl.dmin; info[2] = pl.dmax; info[3] = pl.num_rows;
  info[4] = -1; info[5] = -1; info[6] = 0; info[7] = 0; info[8] = 0; info[9] = 0;
  if (pl.rc != 0) return 0;
  if (pl.ndiag <= 0) return 0;
  if ((int64_t)64 * g_waves * bk < pl.ndiag) return -3;
  if (g_waves > 1 && (packed16 == 1 || packed16 == 4)) return -6;          // (lane packing is a one-wavefront layout)
  // arena: origin at 0, mutant after it, both padded
  const int opad = ((X > 0 ? X : 1) + 31) / 16 * 16;
  const int mpad = ((Y > 0 ? Y : 1) + 31) / 16 * 16 + 16;
  std::vector<uint8_t> arena(opad + mpad, 0);
  for (int i = 0; i < X; i++) arena[i] = (uint8_t)origin[i];
  for (int i = 0; i < Y; i++) arena[opad + i] = (uint8_t)mutant[i];
  pw::PairDesc pd;
  memset(&pd, 0, sizeof pd);
  pd.o_off = 0; pd.m_off = opad; pd.mask_off = 0; pd.h_off = 0; pd.tx_off = 0;
  pd.X = X; pd.Y = Y; pd.dmin = pl.dmin; pd.ndiag = pl.ndiag; pd.s0 = pl.s0;
  pd.nblocks = pl.nblocks; pd.steady_b0 = pl.steady_b0; pd.steady_b1 = pl.steady_b1;
  pd.h_pitch = (X < Y ? X : Y) + 1;
  pd.tx_cap = X + Y + 1; pd.bk = bk; pd.solvable = 1;
  pd.nl = 64 * g_waves;
  std::vector<uint32_t> masks((size_t)(pl.nblocks + 1) * pd.nl * bk + 64, 0xdeadbeefu);   // + spare row and slack like the product
  std::vector<T> hd;
  if (hdump) hd.assign((size_t)pl.ndiag * pd.h_pitch, T(0));
  std::vector<T> sub((size_t)L * L);
  bool simple = true;
  for (int i = 0; i < L; i++) for (int j = 0; j < L; j++) {
    sub[(size_t)i * L + j] = (T)subst[(size_t)i * L + j];
    if (subst[(size_t)i * L + j] != (i == j ? subst[0] : (L > 1 ? subst[1] : subst[0]))) simple = false;
  }
  pw::Result res;
  memset(&res, 0, sizeof res);
  pw::FillParams<T> a;
  memset(&a, 0, sizeof a);
  a.pairs = &pd; a.order = nullptr; a.arena = arena.data(); a.masks = masks.data();
  a.hdump = hdump ? hd.data() : nullptr; a.results = &res; a.subst = sub.data();
  a.npairs = 1; a.L = L; a.brule = pl.brule; a.endrule = pl.endrule; a.banded = (mode == pw::BANDED_MODE);
  a.match = sub[0]; a.mismatch = L > 1 ? sub[1] : sub[0]; a.go = (T)go; a.ge = (T)ge;
  a.score_mul = 1.0;
  // a small integer matrix may go through the packed kernels' matrix form (the product's admission: pwlib_api.cpp)
  double smin = subst[0], smax = subst[0];
  bool integral = true;
  for (int i = 0; i < L * L; i++) { smin = subst[i] < smin ? subst[i] : smin; smax = subst[i] > smax ? subst[i] : smax; integral = integral && subst[i] == (double)(int)subst[i]; }
  const bool x4mode = packed16 == 3 || packed16 == 4;
  const bool mat16 = packed16 && !simple && L <= 4 && integral && smin <= 0 && (x4mode ? 4 : 1) * (smax - smin) <= 127 && sizeof(T) == 4;
  const int generic = force_generic || (!simple && !mat16) || go > 0 || hdump != nullptr;
  const int bany = pl.brule == pw::BRULE_ANY;
  const int track = pl.endrule == pw::END_STD_LOCAL || pl.endrule == pw::END_BANDED_LOCAL;
  const bool rule_local = bany && track;
  const bool rule_overlap = (pl.brule == pw::BRULE_EDGE && (pl.endrule == pw::END_BANDED_OVERLAP || pl.endrule == pw::END_STD_OVERLAP)) ||
                            (pl.brule == pw::BRULE_ORIGIN && pl.endrule == pw::END_STD_OVERLAP) ||
                            (pl.brule == pw::BRULE_EDGE && pl.endrule == pw::END_CORNER);
  const bool rule_global = pl.brule == pw::BRULE_ORIGIN && pl.endrule == pw::END_CORNER;      // B_GLOBAL, and GLOBAL on the full band
  const bool rule_anchored = !mat16 && ((bany && pl.endrule == pw::END_CORNER) ||             // END_ANCHORED
                                        (pl.brule == pw::BRULE_ORIGIN && pl.endrule == pw::END_STD_LOCAL));   // START_ANCHORED
  const int use16 = packed16 && !generic && (rule_local || rule_overlap || rule_global || rule_anchored) && bk % 4 == 0 && sizeof(T) == 4;
  g_packed_mode = packed16;
  g_packed_mat = mat16;
  if (mat16) {
    const int scale = x4mode ? 4 : 1;
    for (int o = 0; o < 4; o++) {
      uint32_t row = 0;
      for (int m = 0; m < 4; m++) if (o < L && m < L) row |= (uint32_t)(scale * (int)(subst[o * L + m] - smin)) << (8 * m);
      a.mat_rows[o] = row;
    }
    a.mat_bias = scale * (int)(-smin);
  }
  if (packed16 && !generic && !use16) return -5;                   // the caller asked for a packed kernel that does not exist
  if (use16 && g_waves == 1) pd.nl = (pl.ndiag + bk - 1) / bk;
  switch (bk) {
    case 2: dispatch_variant<T, 2>(a, pd, sub.data(), generic, bany, track, use16); break;
    case 4: dispatch_variant<T, 4>(a, pd, sub.data(), generic, bany, track, use16); break;
    case 8: dispatch_variant<T, 8>(a, pd, sub.data(), generic, bany, track, use16); break;
    case 16: dispatch_variant<T, 16>(a, pd, sub.data(), generic, bany, track, use16); break;
    case 32: dispatch_variant<T, 32>(a, pd, sub.data(), generic, bany, track, use16); break;
    // (lane widths only the packed kernels are built for)
    case 12: if (!use16 || !Run16<T, 12>::go(a, pd)) return -4; break;
    case 20: if (!use16 || !Run16<T, 20>::go(a, pd)) return -4; break;
    default: return -4;
  }
  // traceback by "one lane"
  std::vector<uint8_t> tx((size_t)pd.tx_cap + 1, 0);
  pw::TraceParams tp;
  memset(&tp, 0, sizeof tp);
  tp.pairs = &pd; tp.arena = arena.data(); tp.masks = masks.data(); tp.results = &res;
  tp.transcripts = tx.data(); tp.npairs = 1;
  tp.gosign = go < 0 ? -1 : (go > 0 ? 1 : 0); tp.banded = a.banded; tp.ends = nullptr;
  uint32_t win[pw::WIN_WORDS];
  pw::trace_walk(tp, 0, win);
  pw::trace_fixup_serial(tp, 0);
  info[4] = res.opt_i; info[5] = res.opt_j; info[6] = res.origin_idx; info[7] = res.mutant_idx;
  info[8] = res.tx_len; info[9] = res.status;
  *score = res.score;
  if (res.tx_len > 0 && res.tx_len < txcap) {
    memcpy(txbuf, tx.data() + pd.tx_cap - res.tx_len, (size_t)res.tx_len);
    txbuf[res.tx_len] = 0;
  } else if (txcap > 0) txbuf[0] = 0;
  if (hdump) for (size_t i = 0; i < hd.size(); i++) hdump[i] = (double)hd[i];
  return 0;
}

}  // namespace

int g_strip_byte_rows = 1;
extern "C" void emu_set_strip_byte_rows(int v) { g_strip_byte_rows = v; }
// a substitution matrix for the next emu_solve_strip calls (L <= 4, integer entries within a signed byte; L = 0: match / mismatch)
int g_strip_L = 0;
double g_strip_subst[16];
extern "C" void emu_set_strip_matrix(const double* S, int L) {
  g_strip_L = (S != nullptr && L >= 1 && L <= 4) ? L : 0;
  for (int i = 0; i < g_strip_L * g_strip_L; i++) g_strip_subst[i] = S[i];
}

// Standard-mode problem through the strip pipeline: strips in index order, end-cell reduction, strip walker, fix-up.
extern "C" int emu_solve_strip(int type, const int* origin, int X, const int* mutant, int Y, double match, double mismatch,
                               double go, double ge, unsigned epoch, int* info, double* score, char* txbuf, int txcap) {
  pw::Plan pl = pw::plan_problem(pw::STD_MODE, type, X, Y, 0, 0);
  info[0] = pl.rc; info[1] = pl.dmin; info[2] = pl.dmax; info[3] = pl.num_rows;
  info[4] = -1; info[5] = -1; info[6] = 0; info[7] = 0; info[8] = 0; info[9] = 0;
  if (pl.rc != 0) return 0;
  const int opad = ((X > 0 ? X : 1) + 31) / 16 * 16;
  const int mpad = ((Y > 0 ? Y : 1) + 31) / 16 * 16;
  std::vector<uint8_t> arena(opad + mpad, 0);
  for (int i = 0; i < X; i++) arena[i] = (uint8_t)origin[i];
  for (int i = 0; i < Y; i++) arena[opad + i] = (uint8_t)mutant[i];
  pw::StripParams a;
  memset(&a, 0, sizeof a);
  a.arena = arena.data(); a.o_off = 0; a.m_off = opad;
  a.X = X; a.Y = Y;
  a.nstrips = (X + 1 + 63) / 64;
  a.nkq = (Y + 64 + pw::kStripBlock - 1) / pw::kStripBlock;
  a.fifo_pitch = pw::strip_fifo_pitch(Y);
  std::vector<uint64_t> fifo((size_t)a.nstrips * a.fifo_pitch, 0x00000000deadbeefull);   // stale granules of "earlier solves"
  for (size_t i = 0; i < fifo.size(); i += 3) fifo[i] = ((uint64_t)(((epoch - 1) << 8) | (i & 0xffu)) << 32) | 12345u;
  std::vector<uint32_t> masks((size_t)a.nstrips * a.nkq * 64 * 4, 0xdeadbeefu);
  std::vector<pw::StripBest> sbest(a.nstrips);
  uint32_t ctl[16] = {0};
  pw::Result res;
  memset(&res, 0, sizeof res);
  a.fifo = fifo.data(); a.masks = masks.data(); a.sbest = sbest.data(); a.ctl = ctl; a.result = &res;
  a.epoch = epoch; a.brule = pl.brule; a.endrule = pl.endrule;
  a.match = (int32_t)match; a.mismatch = (int32_t)mismatch; a.go = (int32_t)go; a.ge = (int32_t)ge;
  a.spin_limit = 4;
  a.score_mul = 1.0;
  const bool track = pl.endrule != pw::END_CORNER;
  // byte rows (pw_strip.h, BROW) under the library's admission rule: letters 0 .. 3, both scores a signed byte
  bool brow = g_strip_byte_rows != 0 && a.match >= -128 && a.match <= 127 && a.mismatch >= -128 && a.mismatch <= 127;
  for (int i = 0; i < X; i++) brow = brow && origin[i] < 4;
  for (int i = 0; i < Y; i++) brow = brow && mutant[i] < 4;
  if (g_strip_L > 0) {
    if (!brow && g_strip_byte_rows == 0) return -8;         // a matrix needs the byte rows
    brow = true;
    for (int o = 0; o < g_strip_L; o++)
      for (int m = 0; m < g_strip_L; m++)
        ctl[pw::kStripRows + o] |= ((uint32_t)(int32_t)g_strip_subst[o * g_strip_L + m] & 0xffu) << (8 * m);
  } else if (brow) {
    for (int o = 0; o < 4; o++)
      for (int m = 0; m < 4; m++) ctl[pw::kStripRows + o] |= ((uint32_t)(o == m ? a.match : a.mismatch) & 0xffu) << (8 * m);
  }
  for (int w = 0; w < a.nstrips; w++) {
    Emu emu;
    bool ok = true;
    if (track && brow) emu.run([&]() { pw::StripFill<EmuPS, true, true> f(a); f.run(w, (w & 1) == 0, (w & 1) != 0); });
    else if (track) emu.run([&]() { pw::StripFill<EmuPS, true, false> f(a); f.run(w, (w & 1) == 0, (w & 1) != 0); });
    else if (brow) emu.run([&]() { pw::StripFill<EmuPS, false, true> f(a); f.run(w, (w & 1) == 0, (w & 1) != 0); });
    else emu.run([&]() { pw::StripFill<EmuPS, false, false> f(a); f.run(w, (w & 1) == 0, (w & 1) != 0); });
    if (ctl[pw::kStripAbort] != 0u) ok = false;
    if (!ok) return -7;
  }
  { Emu emu; emu.run([&]() { pw::strip_reduce<EmuPS>(a); }); }
  std::vector<uint8_t> tx((size_t)(X + Y + 1) + 1, 0);
  pw::StripTraceParams tp;
  memset(&tp, 0, sizeof tp);
  tp.masks = masks.data(); tp.result = &res; tp.tx = tx.data(); tp.ends = nullptr;
  tp.X = X; tp.Y = Y; tp.nkq = a.nkq; tp.tx_cap = X + Y + 1; tp.gosign = go < 0 ? -1 : (go > 0 ? 1 : 0);
  std::vector<uint32_t> win(pw::kWalkWinWords, 0);
  { Emu emu; emu.run([&]() { pw::strip_walk<EmuPS>(tp, win.data()); }); }
  pw::PairDesc pd;
  memset(&pd, 0, sizeof pd);
  pd.o_off = 0; pd.m_off = opad; pd.X = X; pd.Y = Y; pd.tx_off = 0; pd.tx_cap = X + Y + 1; pd.solvable = 1;
  pw::TraceParams fp;
  memset(&fp, 0, sizeof fp);
  fp.pairs = &pd; fp.arena = arena.data(); fp.results = &res; fp.transcripts = tx.data(); fp.npairs = 1;
  pw::trace_fixup_serial(fp, 0);
  info[4] = res.opt_i; info[5] = res.opt_j; info[6] = res.origin_idx; info[7] = res.mutant_idx;
  info[8] = res.tx_len; info[9] = res.status;
  *score = res.score;
  if (res.tx_len > 0 && res.tx_len < txcap) {
    memcpy(txbuf, tx.data() + pd.tx_cap - res.tx_len, (size_t)res.tx_len);
    txbuf[res.tx_len] = 0;
  } else if (txcap > 0) txbuf[0] = 0;
  return 0;
}

extern "C" int emu_solve(int mode, int type, const int* origin, int X, const int* mutant, int Y, int L,
                         const double* subst, double go, double ge, int dmin, int dmax, int use_double,
                         int force_generic, int bk, int* info, double* score, char* txbuf, int txcap,
                         double* hdump, int packed16) {
  if (use_double)
    return solve_T<double>(mode, type, origin, X, mutant, Y, L, subst, go, ge, dmin, dmax, force_generic, bk, 0,
                           info, score, txbuf, txcap, hdump);
  return solve_T<int32_t>(mode, type, origin, X, mutant, Y, L, subst, go, ge, dmin, dmax, force_generic, bk, packed16,
                          info, score, txbuf, txcap, hdump);
}
