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

// Platform policy of the lane program on a CDNA wave64.
struct DevP {
  PW_FN static int lane() { return (int)(threadIdx.x & 63u); }
  // DPP wave shifts: lane i receives lane i-1 (shr) / i+1 (shl); the lane with no source keeps `old`.
  PW_FN static int32_t shr1(int32_t v, int32_t old) {
    return __builtin_amdgcn_update_dpp(old, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  }
  PW_FN static int32_t shl1(int32_t v, int32_t old) {
    return __builtin_amdgcn_update_dpp(old, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
  }
  PW_FN static int32_t shfl_xor(int32_t v, int m) { return __shfl_xor(v, m, 64); }
  PW_FN static int32_t uniform(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
  PW_FN static int nlanes() { return 64; }
  PW_FN static int nwaves() { return 1; }
  PW_FN static int32_t wave_bcast(int32_t v, int) { return v; }
  PW_FN static int lane0() { return 0; }
  PW_FN static bool central() { return true; }
  static constexpr bool kVirtualLanes = false;
};

// Platform policy for bands wider than one wavefront holds: a workgroup of blockDim.x / 64 wavefronts is one
// long row of lanes.  Inside a wavefront the shifts are the same DPP moves; the value that crosses a wavefront
// boundary goes through LDS (one slot per wavefront, written by its edge lane) between two workgroup barriers.
// Every wavefront runs the same sequence of shifts, so the barriers are reached uniformly.
struct DevPM {
  PW_FN static int lane() { return (int)threadIdx.x; }
  PW_FN static int lane0() { return 0; }
  PW_FN static bool central() { return true; }
  static constexpr bool kVirtualLanes = false;
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
  __shared__ T sub_lds[GENERIC ? kMaxLdsL * kMaxLdsL : 1];
  const T* tab = a.subst;
  if (GENERIC) {
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
  w.pair_slot = pair;
  w.run();
}

// Lane-packed 16-bit kernel: one wavefront = WaveDesc.count pairs side by side (pw_wave.h, WaveFill16).
template <int BK, bool SEG>
__global__ __launch_bounds__(64) PW_FILL_ATTR void k_fill16(const FillParams<int32_t> a) {
  const WaveDesc wd = a.waves[blockIdx.x];
  WaveFill16<DevP, BK, SEG> w(a, wd);
  w.run();
}

// K2a: one WORKGROUP (up to 8 wavefronts, 2048 diagonals each) per pair for bands wider than 2048 diagonals.
template <typename T, bool BANY, bool TRACK, bool GENERIC>
__global__ __launch_bounds__(512) void k_fill_mw(const FillParams<T> a) {
  __shared__ T sub_lds[GENERIC ? kMaxLdsL * kMaxLdsL : 1];
  const T* tab = a.subst;
  if (GENERIC) {
    if (a.L <= kMaxLdsL) {
      for (int i = (int)threadIdx.x; i < a.L * a.L; i += (int)blockDim.x) sub_lds[i] = a.subst[i];
      __syncthreads();
      tab = sub_lds;
    }
  }
  const int slot = (int)blockIdx.x;
  const int pair = a.order ? a.order[slot] : slot;
  const PairDesc pd = a.pairs[pair];
  WaveFill<DevPM, T, 32, BANY, TRACK, GENERIC> w(a, pd, tab);
  w.pair_slot = pair;
  w.run();
}

template <typename T>
hipError_t launch_variant_mw(const FillParams<T>& a, int variant, int nw, int nblocks, hipStream_t st) {
  const dim3 grid((unsigned)nblocks), block((unsigned)(64 * nw));
  switch (variant) {
    case VAR_FAST_ANY_TRACK: hipLaunchKernelGGL((k_fill_mw<T, true, true, false>), grid, block, 0, st, a); break;
    case VAR_FAST_TRACK: hipLaunchKernelGGL((k_fill_mw<T, false, true, false>), grid, block, 0, st, a); break;
    case VAR_FAST: hipLaunchKernelGGL((k_fill_mw<T, false, false, false>), grid, block, 0, st, a); break;
    case VAR_GENERIC: hipLaunchKernelGGL((k_fill_mw<T, false, true, true>), grid, block, 0, st, a); break;
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
