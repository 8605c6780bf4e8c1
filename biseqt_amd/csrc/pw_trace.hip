// pw_trace.hip -- K4 traceback kernel (one lane per pair walks the tie masks back from the end cell)
// and the run-time dispatch over the per-(type, BK) fill translation units.
#include "pw_device.h"

namespace pw {

__global__ __launch_bounds__(64) void k_trace(const TraceParams p) {
  const int pair = (int)(blockIdx.x * 64u + threadIdx.x);
  if (pair < p.npairs) trace_pair(p, pair);
}

hipError_t launch_trace(const TraceParams& p, hipStream_t st) {
  if (p.npairs <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_trace, dim3((unsigned)((p.npairs + 63) / 64)), dim3(64), 0, st, p);
  return hipGetLastError();
}

#define PW_DECL(TN, T) \
  hipError_t launch_fill_##TN##_bk2(const FillParams<T>&, int, int, hipStream_t);  \
  hipError_t launch_fill_##TN##_bk4(const FillParams<T>&, int, int, hipStream_t);  \
  hipError_t launch_fill_##TN##_bk8(const FillParams<T>&, int, int, hipStream_t);  \
  hipError_t launch_fill_##TN##_bk16(const FillParams<T>&, int, int, hipStream_t); \
  hipError_t launch_fill_##TN##_bk32(const FillParams<T>&, int, int, hipStream_t);
PW_DECL(i32, int32_t)
PW_DECL(f64, double)
#undef PW_DECL

hipError_t launch_fill(const FillParams<int32_t>& a, int variant, int bk, int nblocks, hipStream_t st) {
  switch (bk) {
    case 2: return launch_fill_i32_bk2(a, variant, nblocks, st);
    case 4: return launch_fill_i32_bk4(a, variant, nblocks, st);
    case 8: return launch_fill_i32_bk8(a, variant, nblocks, st);
    case 16: return launch_fill_i32_bk16(a, variant, nblocks, st);
    case 32: return launch_fill_i32_bk32(a, variant, nblocks, st);
    default: return hipErrorInvalidValue;
  }
}
hipError_t launch_fill(const FillParams<double>& a, int variant, int bk, int nblocks, hipStream_t st) {
  switch (bk) {
    case 2: return launch_fill_f64_bk2(a, variant, nblocks, st);
    case 4: return launch_fill_f64_bk4(a, variant, nblocks, st);
    case 8: return launch_fill_f64_bk8(a, variant, nblocks, st);
    case 16: return launch_fill_f64_bk16(a, variant, nblocks, st);
    case 32: return launch_fill_f64_bk32(a, variant, nblocks, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace pw
