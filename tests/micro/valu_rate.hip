// Microbenchmark (diagnostic, not product): integer VALU issue rate on gfx950 at different occupancies.
// Each wave runs a long stream of INDEPENDENT 32-bit integer ops (8 accumulators) of one kind.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(64) void k(int* out, int n, int c) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b = c;
  for (int i = 0; i < n; i++) {
    if (KIND == 0) {   // v_add_u32
      REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                        "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (KIND == 1) {   // v_max3_i32
      REP8(asm volatile("v_max3_i32 %0, %0, %8, %1\n v_max3_i32 %1, %1, %8, %2\n v_max3_i32 %2, %2, %8, %3\n v_max3_i32 %3, %3, %8, %4\n"
                        "v_max3_i32 %4, %4, %8, %5\n v_max3_i32 %5, %5, %8, %6\n v_max3_i32 %6, %6, %8, %7\n v_max3_i32 %7, %7, %8, %0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (KIND == 2) {   // v_cmp + v_cndmask pairs (vcc)
      REP8(asm volatile("v_cmp_eq_u32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_eq_u32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        "v_cmp_eq_u32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_eq_u32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    } else if (KIND == 3) {   // packed 16-bit: v_pk_add_i16 / v_pk_max_i16
      REP8(asm volatile("v_pk_add_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_add_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n"
                        "v_pk_add_i16 %4, %4, %8\n v_pk_max_i16 %5, %5, %8\n v_pk_add_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (KIND == 4) {   // dependent chain of v_add_u32 (one accumulator)
      REP8(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                        "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                        : "+v"(a0) : "v"(b));)
    } else if (KIND == 5) {   // dpp mov + add
      REP8(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32 %2, %2, %8\n v_mov_b32_dpp %3, %4 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_add_u32 %5, %5, %8\n"
                        "v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32 %2, %2, %8\n v_mov_b32_dpp %1, %4 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_add_u32 %5, %5, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (KIND == 6) {   // v_cmp to SGPR pair + v_cndmask e64 with the same pair (dependent through SGPR)
      REP8(asm volatile("v_cmp_eq_u32 s[20:21], %0, %8\n v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cmp_eq_u32 s[22:23], %1, %8\n v_cndmask_b32 %1, %1, %8, s[22:23]\n"
                        "v_cmp_eq_u32 s[24:25], %2, %8\n v_cndmask_b32 %2, %2, %8, s[24:25]\n v_cmp_eq_u32 s[26:27], %3, %8\n v_cndmask_b32 %3, %3, %8, s[26:27]\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20","s21","s22","s23","s24","s25","s26","s27");)
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
void run(const char* name, int* d_out, int waves_per_simd) {
  const int n = 2000;                       // 2000 * 64 instr per wave
  const int blocks = 256 * 4 * waves_per_simd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d_out, 10, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d_out, n, 1);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)n * 64 * waves_per_simd;
  printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
         ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}

int main() {
  int* d_out; hipMalloc(&d_out, 4 * 64 * 256 * 4 * 8);
  for (int w : {1, 2, 3, 4, 8}) {
    run<0>("v_add_u32 indep", d_out, w);
    run<1>("v_max3_i32 semi-dep", d_out, w);
    run<2>("v_cmp+v_cndmask vcc", d_out, w);
    run<3>("v_pk_add/max_i16", d_out, w);
    run<4>("v_add_u32 dependent chain", d_out, w);
    run<5>("dpp mov + add", d_out, w);
    run<6>("v_cmp->sgpr->v_cndmask dep", d_out, w);
  }
  return 0;
}
