// pw_fill_tu.hip -- one translation unit per (score type, diagonals-per-lane): compiled several times
// with -DPW_T=<int32_t|double> -DPW_TNAME=<i32|f64> -DPW_BK=<2|4|8|16|32> so the instantiations build in
// parallel (see build.py).  Each TU exports one launcher, pw::launch_fill_<TNAME>_bk<BK>.
#include "pw_device.h"

#define PW_CAT2(a, b, c, d) a##b##c##d
#define PW_CAT(a, b, c, d) PW_CAT2(a, b, c, d)

namespace pw {
hipError_t PW_CAT(launch_fill_, PW_TNAME, _bk, PW_BK)(const FillParams<PW_T>& a, int variant, int nblocks,
                                                      hipStream_t st) {
  return launch_variant<PW_T, PW_BK>(a, variant, nblocks, st);
}
}  // namespace pw
