// pw_device.h -- gfx950 (CDNA4, wave64) device side of the pairwise-alignment engine.
//
//   k_fill<T, BK, ...>  K1/K3: one wavefront per sequence pair runs the anti-diagonal wavefront fill of
//                       pw_wave.h with the band resident in registers, cross-lane traffic on DPP wave
//                       shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1, full VALU rate, no LDS), tie
//                       masks streamed to HBM as 16 B/lane stores, end-cell search fused in the epilogue.
//
// Integer stencil: no MFMA anywhere.  Written for gfx950 only.
#ifndef PW_DEVICE_H
#define PW_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PW_FN __device__ __forceinline__
#include "pw_wave.h"
#include "pw_launch.h"

namespace pw {

// m + m + flag as ONE v_addc_co_u32 whose carry-in is the flag's lane mask (the compiler would select a 0 / 1 and shift-or)
PW_FN uint32_t dev_shl1_in(uint32_t m, bool flag) {
  const uint64_t mask = __builtin_amdgcn_ballot_w64(flag);
  uint32_t r;
  uint64_t cout;
  asm("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(r), "=s"(cout) : "v"(m), "s"(mask));
  return r;
}

// A substitution score out of the table (pw_wave.h, TAB): `off` is a byte offset.  ALWAYS_LDS: the table is known to sit in
// LDS at compile time -- a ds_read by 32-bit address, no flat load, no 64-bit address arithmetic; otherwise asked at run time
// (wave-uniform) and read from LDS or from global memory.
template <typename T, bool ALWAYS_LDS> PW_FN T dev_tab_read(const T* tab, uint32_t handle, uint32_t off, bool in_lds) {
  typedef __attribute__((address_space(3))) const T lds_T;
  if (ALWAYS_LDS || in_lds) return *(lds_T*)(uintptr_t)(handle + off);
  return *(const T*)((const char*)tab + off);
}
PW_FN uint32_t dev_lds_handle(const void* p) {      // the LDS address of a __shared__ object
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// Platform policy of the lane program on a CDNA wave64.
// A dword of a read-only frame (the letter arena) at a WAVE-UNIFORM index: from the constant address space, i.e. a scalar
// load (lgkmcnt).  A plain load would be a vector one, and on this target vector loads share vmcnt with the stores: the
// compiler's wait for the load then waits for the mask stores in front of it as well.
PW_FN uint32_t dev_const_dword(const uint8_t* base, int idx) {
  typedef const __attribute__((address_space(4))) uint32_t c_u32;
  return ((c_u32*)base)[idx];
}

struct DevP {
  PW_FN static uint32_t const_dword(const uint8_t* base, int idx) { return dev_const_dword(base, idx); }
  template <typename T, bool A> PW_FN static T tab_read(const T* tab, uint32_t h, uint32_t off, bool in_lds) { return dev_tab_read<T, A>(tab, h, off, in_lds); }
  PW_FN static uint32_t shl1_in(uint32_t m, bool flag) { return dev_shl1_in(m, flag); }
  PW_FN static int lane() { return (int)(threadIdx.x & 63u); }
  // DPP wave shifts: lane i receives lane i-1 (shr) / i+1 (shl); the lane with no source keeps `old`.
  PW_FN static int32_t shr1(int32_t v, int32_t old) {
    return __builtin_amdgcn_update_dpp(old, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  }
  PW_FN static int32_t shl1(int32_t v, int32_t old) {
    return __builtin_amdgcn_update_dpp(old, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
  }
  // the same shifts with zero fill (bound_ctrl): no dependence on an old value, so no copy in front of the DPP move
  PW_FN static int32_t shr1z(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true); }
  PW_FN static int32_t shl1z(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true); }
  PW_FN static int32_t shfl_xor(int32_t v, int m) { return __shfl_xor(v, m, 64); }
  PW_FN static int32_t uniform(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
  PW_FN static int nlanes() { return 64; }
  PW_FN static int nwaves() { return 1; }
  PW_FN static int32_t wave_bcast(int32_t v, int) { return v; }
  PW_FN static int lane0() { return 0; }
  PW_FN static bool central() { return true; }
  static constexpr bool kVirtualLanes = false;
  static constexpr bool kBatchedShifts = false;
};

// N values moved by one lane across a workgroup that is one long row of lanes: DPP inside each wavefront, one
// LDS slot set per wavefront edge and ONE barrier for the group.  `phase` alternates between two slot sets, so
// a wavefront that is already writing exchange e + 1 never overwrites what a slower one still reads from e
// (exchange e + 1's own barrier orders e's reads before e + 2's writes).
template <int N, int MAXW> PW_FN void wg_shift_right(int32_t* v, const int32_t* old, int phase) {
  __shared__ int32_t edge[2][MAXW][N];
  const int w = (int)(threadIdx.x >> 6), l = (int)(threadIdx.x & 63u);
  int32_t r[N];
#pragma unroll
  for (int i = 0; i < N; i++) r[i] = __builtin_amdgcn_update_dpp(old[i], v[i], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  if (l == 63) {
#pragma unroll
    for (int i = 0; i < N; i++) edge[phase][w][i] = v[i];
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (l == 0 && w > 0) {
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = edge[phase][w - 1][i];
  }
#pragma unroll
  for (int i = 0; i < N; i++) v[i] = r[i];
}
template <int N, int MAXW> PW_FN void wg_shift_left(int32_t* v, const int32_t* old, int phase) {
  __shared__ int32_t edge[2][MAXW][N];
  const int w = (int)(threadIdx.x >> 6), l = (int)(threadIdx.x & 63u), nw = (int)(blockDim.x >> 6);
  int32_t r[N];
#pragma unroll
  for (int i = 0; i < N; i++) r[i] = __builtin_amdgcn_update_dpp(old[i], v[i], 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
  if (l == 0) {
#pragma unroll
    for (int i = 0; i < N; i++) edge[phase][w][i] = v[i];
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (l == 63 && w + 1 < nw) {
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = edge[phase][w + 1][i];
  }
#pragma unroll
  for (int i = 0; i < N; i++) v[i] = r[i];
}

// Platform policy for bands wider than one wavefront holds: a workgroup of blockDim.x / 64 wavefronts is one
// long row of lanes.  Inside a wavefront the shifts are the same DPP moves; the value that crosses a wavefront
// boundary goes through LDS (one slot per wavefront, written by its edge lane) between two workgroup barriers.
// Every wavefront runs the same sequence of shifts, so the barriers are reached uniformly.
struct DevPM {
  PW_FN static uint32_t const_dword(const uint8_t* base, int idx) { return dev_const_dword(base, idx); }
  template <typename T, bool A> PW_FN static T tab_read(const T* tab, uint32_t h, uint32_t off, bool in_lds) { return dev_tab_read<T, A>(tab, h, off, in_lds); }
  PW_FN static uint32_t shl1_in(uint32_t m, bool flag) { return dev_shl1_in(m, flag); }
  PW_FN static int lane() { return (int)threadIdx.x; }
  PW_FN static int lane0() { return 0; }
  PW_FN static bool central() { return true; }
  static constexpr bool kVirtualLanes = false;
  static constexpr bool kBatchedShifts = true;
  template <int N> PW_FN static void shrv(int32_t* v, const int32_t* old, int phase) { wg_shift_right<N, 16>(v, old, phase); }
  template <int N> PW_FN static void shlv(int32_t* v, const int32_t* old, int phase) { wg_shift_left<N, 16>(v, old, phase); }
  PW_FN static int nlanes() { return (int)blockDim.x; }
  PW_FN static int nwaves() { return (int)(blockDim.x >> 6); }
  PW_FN static int32_t shr1(int32_t v, int32_t old) {
    __shared__ int32_t edge[16];
    const int w = (int)(threadIdx.x >> 6), l = (int)(threadIdx.x & 63u);
    int32_t r = __builtin_amdgcn_update_dpp(old, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    if (l == 63) edge[w] = v;
    __syncthreads();
    if (l == 0 && w > 0) r = edge[w - 1];
    __syncthreads();
    return r;
  }
  PW_FN static int32_t shl1(int32_t v, int32_t old) {
    __shared__ int32_t edge[16];
    const int w = (int)(threadIdx.x >> 6), l = (int)(threadIdx.x & 63u);
    int32_t r = __builtin_amdgcn_update_dpp(old, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
    if (l == 0) edge[w] = v;
    __syncthreads();
    if (l == 63 && w + 1 < (int)(blockDim.x >> 6)) r = edge[w + 1];
    __syncthreads();
    return r;
  }
  PW_FN static int32_t shfl_xor(int32_t v, int m) { return __shfl_xor(v, m, 64); }
  PW_FN static int32_t uniform(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
  PW_FN static int32_t wave_bcast(int32_t v, int w) {      // v is wave-uniform; returns wavefront w's value
    __shared__ int32_t slot[16];
    if ((threadIdx.x & 63u) == 0) slot[threadIdx.x >> 6] = v;
    __syncthreads();
    const int32_t r = slot[w];
    __syncthreads();
    return r;
  }
};

constexpr int kMaxLdsL = 32;   // substitution tables up to 32 x 32 are staged in LDS

#ifndef PW_FILL_ATTR
#define PW_FILL_ATTR
#endif
template <typename T, int BK, bool BANY, bool TRACK, bool GENERIC>
__global__ __launch_bounds__(64) PW_FILL_ATTR void k_fill(const FillParams<T> a) {
  constexpr bool TAB = true;      // (pw_wave.h: the substitution score is read from a table)
  __shared__ T sub_lds[TAB ? kMaxLdsL * kMaxLdsL : 1];
  const T* tab = a.subst;
  if (TAB) {
    if (a.L <= kMaxLdsL) {
      for (int i = (int)threadIdx.x; i < a.L * a.L; i += 64) sub_lds[i] = a.subst[i];
      __syncthreads();
      tab = sub_lds;
    }
  }
  const int slot = (int)blockIdx.x;
  const int pair = a.order ? a.order[slot] : slot;
  const PairDesc pd = a.pairs[pair];
  WaveFill<DevP, T, BK, BANY, TRACK, GENERIC> w(a, pd, tab);
  w.tab_handle = dev_lds_handle(sub_lds); w.tab_in_lds = tab == sub_lds;
  w.pair_slot = pair;
  w.run();
}

// Lane-packed 16-bit kernel: one wavefront = WaveDesc.count pairs side by side (pw_wave.h, WaveFill16).
// Occupancy: left to itself the max-ilp schedule spends registers freely (the BK = 8 one-pair-per-wavefront body takes
// 157-164 VGPRs: 3 wavefronts per SIMD); held to 5 per SIMD it needs 87-96 without a spill and config 2 runs 1.3-2 %
// faster (tests/micro/ab_k1.sh: 2132 -> 2160-2178 GCUPS; 4 per SIMD: 2139-2147).  build.py passes the bound of each
// instantiation (PW_FILL16_WAVES: one pair per wavefront, PW_FILL16_WAVES_SEG: lane-packed; 0 = the compiler's default)
// -- only where the held schedule does not spill and was measured faster.
#ifndef PW_FILL16_WAVES
#define PW_FILL16_WAVES 0
#endif
#ifndef PW_FILL16_WAVES_SEG
#define PW_FILL16_WAVES_SEG 0
#endif
template <int BK, bool SEG, int RULE> struct Fill16Occupancy {
  static constexpr unsigned w = SEG ? PW_FILL16_WAVES_SEG : PW_FILL16_WAVES;
  static constexpr unsigned lo = w ? w : 1, hi = w ? w : 8;
};
template <int BK, bool SEG, int RULE, bool MAT = false>
__global__ __launch_bounds__(64) PW_FILL_ATTR
__attribute__((amdgpu_waves_per_eu(Fill16Occupancy<BK, SEG, RULE>::lo, Fill16Occupancy<BK, SEG, RULE>::hi)))
void k_fill16(const FillParams<int32_t> a) {
  const WaveDesc wd = a.waves[blockIdx.x];
  WaveFill16<DevP, BK, SEG, RULE, MAT> w(a, wd);
  w.run();
}

// The same 16-bit body on a workgroup of up to 8 wavefronts per pair (K2a's exchange through LDS): bands of 2049 .. 16 384
// diagonals, e.g. standard-mode tables of 1 .. 8 kb, whose scores fit the packed kernel.  One WaveDesc per pair,
// nl = 64 x wavefronts.
template <int BK, int RULE, bool MAT = false>
__global__ __launch_bounds__(512) void k_fill16_mw(const FillParams<int32_t> a) {
  const WaveDesc wd = a.waves[blockIdx.x];
  WaveFill16<DevPM, BK, false, RULE, MAT> w(a, wd);
  w.run();
}

// Platform policy of a TILE of the time-blocked single-pair kernel (K2b): a workgroup of kTileLanes lanes whose
// lane indices are global (tile * kTileCentral - kTileGhost + thread); the first kTileGhost and the last
// kTileGhost lanes are ghost copies of the neighbouring tiles' lanes.
constexpr int kTileLanes = PW_TILE_LANES, kTileGhost = PW_TILE_GHOST, kTileCentral = kTileLanes - 2 * kTileGhost;
struct DevPT {
  PW_FN static uint32_t const_dword(const uint8_t* base, int idx) { return dev_const_dword(base, idx); }
  template <typename T, bool A> PW_FN static T tab_read(const T* tab, uint32_t h, uint32_t off, bool in_lds) { return dev_tab_read<T, A>(tab, h, off, in_lds); }
  PW_FN static uint32_t shl1_in(uint32_t m, bool flag) { return dev_shl1_in(m, flag); }
  // Workgroups are dealt to the 8 XCDs round-robin; neighbouring tiles exchange their state through memory between
  // launches, so tile t = (b mod 8) * (grid / 8) + b / 8 keeps runs of consecutive tiles on one XCD (and its L2).
  // (The grid is a multiple of 8; tiles beyond the band find no diagonal and do nothing.)
  PW_FN static int tile() {
#ifdef PW_TILE_NO_XCD_REMAP
    return (int)blockIdx.x;
#else
    return (int)(blockIdx.x & 7u) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
#endif
  }
  PW_FN static int lane0() { return tile() * kTileCentral - kTileGhost; }
  PW_FN static int lane() { return lane0() + (int)threadIdx.x; }
  PW_FN static int nlanes() { return lane0() + kTileLanes; }      // global index one past this tile's last lane
  PW_FN static int nwaves() { return kTileLanes / 64; }
  PW_FN static bool central() { return (int)threadIdx.x >= kTileGhost && (int)threadIdx.x < kTileLanes - kTileGhost; }
  static constexpr bool kVirtualLanes = true;
  static constexpr bool kBatchedShifts = true;
  template <int N> PW_FN static void shrv(int32_t* v, const int32_t* old, int phase) { wg_shift_right<N, kTileLanes / 64>(v, old, phase); }
  template <int N> PW_FN static void shlv(int32_t* v, const int32_t* old, int phase) { wg_shift_left<N, kTileLanes / 64>(v, old, phase); }
  // workgroup-wide maxima of four integers
  PW_FN static void wg_max4(int (&v)[4]) {
    __shared__ int red[4];
    if (threadIdx.x < 4) red[threadIdx.x] = -0x7fffffff;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) atomicMax(&red[i], v[i]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = red[i];
  }
  PW_FN static int32_t shr1(int32_t v, int32_t old) {
    __shared__ int32_t edge[kTileLanes / 64];
    const int w = (int)(threadIdx.x >> 6), l = (int)(threadIdx.x & 63u);
    int32_t r = __builtin_amdgcn_update_dpp(old, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    if (l == 63) edge[w] = v;
    __syncthreads();
    if (l == 0 && w > 0) r = edge[w - 1];
    __syncthreads();
    return r;
  }
  PW_FN static int32_t shl1(int32_t v, int32_t old) {
    __shared__ int32_t edge[kTileLanes / 64];
    const int w = (int)(threadIdx.x >> 6), l = (int)(threadIdx.x & 63u);
    int32_t r = __builtin_amdgcn_update_dpp(old, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
    if (l == 0) edge[w] = v;
    __syncthreads();
    if (l == 63 && w + 1 < kTileLanes / 64) r = edge[w + 1];
    __syncthreads();
    return r;
  }
  PW_FN static int32_t shfl_xor(int32_t v, int m) { return __shfl_xor(v, m, 64); }
  PW_FN static int32_t uniform(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
  PW_FN static int32_t wave_bcast(int32_t v, int) { return v; }     // (the tile kernel has no in-kernel end search)
};

constexpr int kTileBK = PW_TILE_BK;     // diagonals per lane of the tiled kernel (default 2 x 128 centre lanes = 256 centre diagonals, 64 ghost lanes = 128 steps per launch)
static_assert(kTileBK == kTileBKHost && kTileCentral == kTileCentralLanes && kTileBlocks * 16 <= kTileGhost * kTileBK,
              "tile geometry: host and device must agree, and a time block must fit the ghost zone");

// K2b: one launch = one time block (tile_nb <= kTileGhost * kTileBK / 16 blocks) of ONE pair; grid = tiles.
template <typename T, bool BANY, bool TRACK, bool GENERIC>
__global__ __launch_bounds__(kTileLanes) void k_fill_tile(const FillParams<T> a, const int pair) {
  constexpr bool TAB = true;      // (pw_wave.h: the substitution score is read from a table)
  __shared__ T sub_lds[TAB ? kMaxLdsL * kMaxLdsL : 1];
  const T* tab = a.subst;
  if (TAB) {
    if (a.L <= kMaxLdsL) {
      for (int i = (int)threadIdx.x; i < a.L * a.L; i += (int)blockDim.x) sub_lds[i] = a.subst[i];
      __syncthreads();
      tab = sub_lds;
    }
  }
  const PairDesc pd = a.pairs[pair];
  WaveFill<DevPT, T, kTileBK, BANY, TRACK, GENERIC> w(a, pd, tab);
  w.tab_handle = dev_lds_handle(sub_lds); w.tab_in_lds = tab == sub_lds;
  w.pair_slot = pair;
  w.run_tile();
}

// End-cell search of a tiled pair from the final per-diagonal state (same rules as WaveFill::finish).
template <typename T>
__global__ __launch_bounds__(256) void k_tile_finish(const FillParams<T> a, const int pair) {
  using Tr = ScoreTraits<T>;
  __shared__ T s_s[256]; __shared__ unsigned long long s_k[256]; __shared__ int s_x[256], s_y[256], s_h[256];
  const PairDesc pd = a.pairs[pair];
  const int X = pd.X, Y = pd.Y, endrule = a.endrule, pitch = a.st_pitch;
  T cs = Tr::neg(); unsigned long long ck = ~0ull; int cx = -1, cy = -1, have = 0;
  for (int dd = (int)threadIdx.x; dd < pd.ndiag; dd += 256) {
    const int d = pd.dmin + dd;
    const bool ends_right = d < X - Y;
    const int lx = ends_right ? d + Y : X, ly = ends_right ? Y : X - d;
    T s; unsigned long long k; int x, y; bool ok = true;
    if (endrule == END_CORNER) { ok = d == X - Y; s = a.st_in[dd]; k = 0; x = X; y = Y; }
    else if (endrule == END_STD_OVERLAP) { s = a.st_in[dd]; x = lx; y = ly; k = ends_right ? (unsigned)lx : (unsigned)(X + ly); }
    else if (endrule == END_BANDED_OVERLAP) { s = a.st_in[dd]; x = lx; y = ly; k = (unsigned)dd; }
    else {
      const int tfirst = (d < 0 ? -d : d) - pd.s0;
      const int aa = ((int)a.st_in[4 * pitch + dd] - tfirst) >> 1;
      s = a.st_in[3 * pitch + dd]; x = aa + (d > 0 ? d : 0); y = aa - (d < 0 ? d : 0);
      if (endrule == END_STD_LOCAL) k = (unsigned long long)(unsigned)x * (unsigned long long)(unsigned)(Y + 1) + (unsigned)y;
      else k = ((unsigned long long)(unsigned)dd << 32) | (unsigned)aa;
    }
    if (ok && (!have || s > cs || (s == cs && k < ck))) { cs = s; ck = k; cx = x; cy = y; have = 1; }
  }
  s_s[threadIdx.x] = cs; s_k[threadIdx.x] = ck; s_x[threadIdx.x] = cx; s_y[threadIdx.x] = cy; s_h[threadIdx.x] = have;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      const int o = (int)threadIdx.x + off;
      if (s_h[o] && (!s_h[threadIdx.x] || s_s[o] > s_s[threadIdx.x] || (s_s[o] == s_s[threadIdx.x] && s_k[o] < s_k[threadIdx.x]))) {
        s_s[threadIdx.x] = s_s[o]; s_k[threadIdx.x] = s_k[o]; s_x[threadIdx.x] = s_x[o]; s_y[threadIdx.x] = s_y[o]; s_h[threadIdx.x] = 1;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    Result r;
    r.score = (double)s_s[0] * a.score_mul;
    r.opt_i = a.banded ? s_x[0] - s_y[0] - pd.dmin : s_x[0];
    r.opt_j = a.banded ? (s_x[0] < s_y[0] ? s_x[0] : s_y[0]) : s_y[0];
    r.origin_idx = 0; r.mutant_idx = 0; r.tx_len = 0; r.status = 0;
    if (!s_h[0] || (endrule == END_STD_LOCAL && !(s_s[0] > T(0)))) { r.opt_i = -1; r.opt_j = -1; r.score = 0.0; }
    a.results[pair] = r;
  }
}

template <typename T>
hipError_t launch_tile_T(const FillParams<T>& a, int variant, int pair, int ntiles, hipStream_t st) {
  const dim3 grid((unsigned)((ntiles + 7) / 8 * 8)), block(kTileLanes);      // a multiple of 8: see DevPT::tile()
  switch (variant) {
    case VAR_FAST_ANY_TRACK: hipLaunchKernelGGL((k_fill_tile<T, true, true, false>), grid, block, 0, st, a, pair); break;
    case VAR_FAST_TRACK: hipLaunchKernelGGL((k_fill_tile<T, false, true, false>), grid, block, 0, st, a, pair); break;
    case VAR_FAST: hipLaunchKernelGGL((k_fill_tile<T, false, false, false>), grid, block, 0, st, a, pair); break;
    case VAR_GENERIC: hipLaunchKernelGGL((k_fill_tile<T, false, true, true>), grid, block, 0, st, a, pair); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
template <typename T>
hipError_t launch_tile_finish_T(const FillParams<T>& a, int pair, hipStream_t st) {
  hipLaunchKernelGGL((k_tile_finish<T>), dim3(1), dim3(256), 0, st, a, pair);
  return hipGetLastError();
}

// K2a: one WORKGROUP (up to 8 wavefronts, 64 * BK diagonals each) per pair: bands wider than 2048 diagonals
// (BK = 32), and -- with BK = 4 / 8 / 16 -- the low-latency layout for batches of a few pairs, where spreading one
// pair over 8 wavefronts beats one wavefront working through 16 or 32 diagonals per lane.
template <typename T, int BK, bool BANY, bool TRACK, bool GENERIC>
__global__ __launch_bounds__(512) void k_fill_mw(const FillParams<T> a) {
  constexpr bool TAB = true;      // (pw_wave.h: the substitution score is read from a table)
  __shared__ T sub_lds[TAB ? kMaxLdsL * kMaxLdsL : 1];
  const T* tab = a.subst;
  if (TAB) {
    if (a.L <= kMaxLdsL) {
      for (int i = (int)threadIdx.x; i < a.L * a.L; i += (int)blockDim.x) sub_lds[i] = a.subst[i];
      __syncthreads();
      tab = sub_lds;
    }
  }
  const int slot = (int)blockIdx.x;
  const int pair = a.order ? a.order[slot] : slot;
  const PairDesc pd = a.pairs[pair];
  WaveFill<DevPM, T, BK, BANY, TRACK, GENERIC> w(a, pd, tab);
  w.tab_handle = dev_lds_handle(sub_lds); w.tab_in_lds = tab == sub_lds;
  w.pair_slot = pair;
  w.run();
}

template <typename T, int BK>
hipError_t launch_variant_mw(const FillParams<T>& a, int variant, int nw, int nblocks, hipStream_t st) {
  const dim3 grid((unsigned)nblocks), block((unsigned)(64 * nw));
  switch (variant) {
    case VAR_FAST_ANY_TRACK: hipLaunchKernelGGL((k_fill_mw<T, BK, true, true, false>), grid, block, 0, st, a); break;
    case VAR_FAST_TRACK: hipLaunchKernelGGL((k_fill_mw<T, BK, false, true, false>), grid, block, 0, st, a); break;
    case VAR_FAST: hipLaunchKernelGGL((k_fill_mw<T, BK, false, false, false>), grid, block, 0, st, a); break;
    case VAR_GENERIC: hipLaunchKernelGGL((k_fill_mw<T, BK, false, true, true>), grid, block, 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <typename T, int BK>
hipError_t launch_variant(const FillParams<T>& a, int variant, int nblocks, hipStream_t st) {
  const dim3 grid((unsigned)nblocks), block(64);
  switch (variant) {
    case VAR_FAST_ANY_TRACK: hipLaunchKernelGGL((k_fill<T, BK, true, true, false>), grid, block, 0, st, a); break;
    case VAR_FAST_TRACK: hipLaunchKernelGGL((k_fill<T, BK, false, true, false>), grid, block, 0, st, a); break;
    case VAR_FAST: hipLaunchKernelGGL((k_fill<T, BK, false, false, false>), grid, block, 0, st, a); break;
    case VAR_GENERIC: hipLaunchKernelGGL((k_fill<T, BK, false, true, true>), grid, block, 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace pw
#endif
