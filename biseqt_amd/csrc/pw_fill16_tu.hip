// pw_fill16_tu.hip -- one translation unit per (diagonals-per-lane, rule) of the lane-packed 16-bit fill kernel:
// compiled with -DPW_BK=<4|8|...|32> -DPW_RULE=<0|1|2> (see build.py; rule 0 = LOCAL / B_LOCAL, 1 = B_OVERLAP,
// 2 = B_GLOBAL).  Exports pw::launch_fill16_bk<BK>_r<RULE>.
#include "pw_device.h"

#define PW_CAT2(a, b) a##b
#define PW_CAT(a, b) PW_CAT2(a, b)
#ifndef PW_RULE
#define PW_RULE 0
#endif

namespace pw {
hipError_t PW_CAT(PW_CAT(launch_fill16_bk, PW_BK), PW_CAT(_r, PW_RULE))(const FillParams<int32_t>& a, int seg, int nwaves, hipStream_t st) {
  if (seg) hipLaunchKernelGGL((k_fill16<PW_BK, true, PW_RULE>), dim3((unsigned)nwaves), dim3(64), 0, st, a);
  else hipLaunchKernelGGL((k_fill16<PW_BK, false, PW_RULE>), dim3((unsigned)nwaves), dim3(64), 0, st, a);
  return hipGetLastError();
}
// ... on `nw` wavefronts per pair (one workgroup each)
hipError_t PW_CAT(PW_CAT(launch_fill16mw_bk, PW_BK), PW_CAT(_r, PW_RULE))(const FillParams<int32_t>& a, int nw, int npairs, hipStream_t st) {
  hipLaunchKernelGGL((k_fill16_mw<PW_BK, PW_RULE>), dim3((unsigned)npairs), dim3((unsigned)(64 * nw)), 0, st, a);
  return hipGetLastError();
}
}  // namespace pw
