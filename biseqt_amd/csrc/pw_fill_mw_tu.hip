// pw_fill_mw_tu.hip -- the multi-wavefront (wide band) fill kernels, one translation unit per score type:
// compiled with -DPW_T=<int32_t|double> -DPW_TNAME=<i32|f64>.  Exports pw::launch_fill_mw_<TNAME>.
#include "pw_device.h"

#define PW_CAT2(a, b) a##b
#define PW_CAT(a, b) PW_CAT2(a, b)

namespace pw {
hipError_t PW_CAT(launch_fill_mw_, PW_TNAME)(const FillParams<PW_T>& a, int variant, int bk, int nw, int nblocks, hipStream_t st) {
  switch (bk) {
    case 4: return launch_variant_mw<PW_T, 4>(a, variant, nw, nblocks, st);
    case 8: return launch_variant_mw<PW_T, 8>(a, variant, nw, nblocks, st);
    case 16: return launch_variant_mw<PW_T, 16>(a, variant, nw, nblocks, st);
    case 32: return launch_variant_mw<PW_T, 32>(a, variant, nw, nblocks, st);
    default: return hipErrorInvalidValue;
  }
}
}  // namespace pw
