// Microbenchmark (diagnostic, not product): issue cost of the 64-bit VALU ops the f64 fill kernel is made of, on gfx950.
// 4 and 8 waves per SIMD, 8 independent accumulators, one instruction kind per kernel (same method as valu_table.hip).
#include <hip/hip_runtime.h>
#include <stdio.h>

#define BODY8(I0, I1, I2, I3, I4, I5, I6, I7) \
  asm volatile(I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n" \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c2) : "vcc", "s20", "s21");
#define SAME8(T) BODY8(T(0), T(1), T(2), T(3), T(4), T(5), T(6), T(7))
#define KERNEL(NAME, MACRO)                                                          \
  __global__ __launch_bounds__(64) void NAME(double* out, int n, double b, double c2) { \
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21");                    \
    for (int i = 0; i < n; i++) { SAME8(MACRO) SAME8(MACRO) SAME8(MACRO) SAME8(MACRO) SAME8(MACRO) SAME8(MACRO) SAME8(MACRO) SAME8(MACRO) } \
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;     \
  }
#define I_ADD(k) "v_add_f64 %" #k ", %" #k ", %8"
#define I_MAX(k) "v_max_f64 %" #k ", %" #k ", %8"
#define I_FMA(k) "v_fma_f64 %" #k ", %" #k ", %8, %9"
#define I_CMPF(k) "v_cmp_eq_f64 s[20:21], %" #k ", %8"
#define I_CMPGT(k) "v_cmp_gt_f64 vcc, %" #k ", %8"
#define I_CMPU(k) "v_cmp_eq_u64 s[20:21], %" #k ", %8"
#define I_CMPNEU(k) "v_cmp_ne_u64 vcc, %" #k ", %8"
#define I_LSHL64(k) "v_lshlrev_b64 %" #k ", 1, %" #k
#define LIST(X) X(add_f64, I_ADD) X(max_f64, I_MAX) X(fma_f64, I_FMA) X(cmp_eq_f64, I_CMPF) X(cmp_gt_f64, I_CMPGT) X(cmp_eq_u64, I_CMPU) \
  X(cmp_ne_u64, I_CMPNEU) X(lshlrev_b64, I_LSHL64)
#define X(name, macro) KERNEL(k_##name, macro)
LIST(X)
#undef X
typedef void (*kern_t)(double*, int, double, double);
struct Entry { const char* name; kern_t fn; };
int main() {
  double* d_out; hipMalloc(&d_out, 8 * 64 * 256 * 4 * 8);
  Entry tab[] = {
#define X(name, macro) {#name, k_##name},
    LIST(X)
#undef X
  };
  const int n = 300;
  printf("%-16s %12s %12s   (ns per wave-instruction per SIMD)\n", "instr", "4 waves/SIMD", "8 waves/SIMD");
  for (auto& e : tab) {
    double res[2]; int wi = 0;
    for (int w : {4, 8}) {
      const int blocks = 256 * 4 * w;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(64), 0, 0, d_out, 5, 1.0, 3.0);
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(64), 0, 0, d_out, n, 1.0, 3.0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      res[wi++] = ms * 1e6 / ((double)n * 64 * w);
    }
    printf("%-16s %12.3f %12.3f\n", e.name, res[0], res[1]);
  }
  return 0;
}
