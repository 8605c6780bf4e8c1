// pw_fill16_tu.hip -- one translation unit per (diagonals-per-lane, rule, matrix or not) of the lane-packed 16-bit fill
// kernel: compiled with -DPW_BK=<4|8|...|32> -DPW_RULE=<0..5> -DPW_MAT=<0|1> (see build.py; rule 0 = LOCAL / B_LOCAL,
// 1 = B_OVERLAP, 2 = B_GLOBAL, 3 = rule 0 with scores held times 4, 4 = END_ANCHORED, 5 = START_ANCHORED; PW_MAT = 1: scores
// from a substitution matrix of up to 4 x 4 letters).  Exports pw::launch_fill16_bk<BK>_r<RULE>_m<MAT> and
// pw::launch_fill16mw_bk<BK>_r<RULE>_m<MAT>.
#include "pw_device.h"

#define PW_CAT2(a, b) a##b
#define PW_CAT(a, b) PW_CAT2(a, b)
#ifndef PW_RULE
#define PW_RULE 0
#endif
#ifndef PW_MAT
#define PW_MAT 0
#endif
#define PW_SUFFIX PW_CAT(PW_CAT(PW_CAT(_bk, PW_BK), PW_CAT(_r, PW_RULE)), PW_CAT(_m, PW_MAT))

namespace pw {
hipError_t PW_CAT(launch_fill16, PW_SUFFIX)(const FillParams<int32_t>& a, int seg, int nwaves, hipStream_t st) {
  if (seg) hipLaunchKernelGGL((k_fill16<PW_BK, true, PW_RULE, PW_MAT != 0>), dim3((unsigned)nwaves), dim3(64), 0, st, a);
  else hipLaunchKernelGGL((k_fill16<PW_BK, false, PW_RULE, PW_MAT != 0>), dim3((unsigned)nwaves), dim3(64), 0, st, a);
  return hipGetLastError();
}
// ... on `nw` wavefronts per pair (one workgroup each)
hipError_t PW_CAT(launch_fill16mw, PW_SUFFIX)(const FillParams<int32_t>& a, int nw, int npairs, hipStream_t st) {
  hipLaunchKernelGGL((k_fill16_mw<PW_BK, PW_RULE, PW_MAT != 0>), dim3((unsigned)npairs), dim3((unsigned)(64 * nw)), 0, st, a);
  return hipGetLastError();
}
}  // namespace pw
