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
