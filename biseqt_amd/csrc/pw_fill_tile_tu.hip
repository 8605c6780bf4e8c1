// pw_fill_tile_tu.hip -- the time-blocked tiled single-pair kernels (K2b), one translation unit per score
// type: compiled with -DPW_T=<int32_t|double> -DPW_TNAME=<i32|f64>.
#include "pw_device.h"

#define PW_CAT2(a, b) a##b
#define PW_CAT(a, b) PW_CAT2(a, b)

namespace pw {
hipError_t PW_CAT(launch_tile_, PW_TNAME)(const FillParams<PW_T>& a, int variant, int pair, int ntiles, hipStream_t st) {
  return launch_tile_T<PW_T>(a, variant, pair, ntiles, st);
}
hipError_t PW_CAT(launch_tile_finish_, PW_TNAME)(const FillParams<PW_T>& a, int pair, hipStream_t st) {
  return launch_tile_finish_T<PW_T>(a, pair, st);
}
}  // namespace pw
