// pw_fill16_tu.hip -- one translation unit per diagonals-per-lane value of the lane-packed 16-bit fill
// kernel: compiled with -DPW_BK=<4|8|...|32> (see build.py).  Exports pw::launch_fill16_bk<BK>.
#include "pw_device.h"

#define PW_CAT2(a, b) a##b
#define PW_CAT(a, b) PW_CAT2(a, b)

namespace pw {
hipError_t PW_CAT(launch_fill16_bk, PW_BK)(const FillParams<int32_t>& a, int seg, int nwaves, hipStream_t st) {
  if (seg) hipLaunchKernelGGL((k_fill16<PW_BK, true>), dim3((unsigned)nwaves), dim3(64), 0, st, a);
  else hipLaunchKernelGGL((k_fill16<PW_BK, false>), dim3((unsigned)nwaves), dim3(64), 0, st, a);
  return hipGetLastError();
}
}  // namespace pw
