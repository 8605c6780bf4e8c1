// Microbenchmark (diagnostic, not product): per-instruction issue cost of candidate VALU ops on gfx950.
// 4 and 8 waves per SIMD, 8 independent accumulators, one instruction kind per kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define BODY8(I0, I1, I2, I3, I4, I5, I6, I7) \
  asm volatile(I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n" \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c2) : "vcc", "s20", "s21");
#define SAME8(T) BODY8(T(0), T(1), T(2), T(3), T(4), T(5), T(6), T(7))

#define KERNEL(NAME, MACRO)                                                          \
  __global__ __launch_bounds__(64) void NAME(int* out, int n, int b, int c2) {       \
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21");                    \
    for (int i = 0; i < n; i++) { SAME8(MACRO) SAME8(MACRO) SAME8(MACRO) SAME8(MACRO) SAME8(MACRO) SAME8(MACRO) SAME8(MACRO) SAME8(MACRO) } \
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;     \
  }

#define I_ADD(k) "v_add_u32 %" #k ", %" #k ", %8"
#define I_SUB(k) "v_sub_u32 %" #k ", %" #k ", %8"
#define I_MAX(k) "v_max_i32 %" #k ", %" #k ", %8"
#define I_MINU(k) "v_min_u32 %" #k ", %" #k ", %8"
#define I_XOR(k) "v_xor_b32 %" #k ", %" #k ", %8"
#define I_LSHL(k) "v_lshlrev_b32 %" #k ", 4, %" #k
#define I_CMP32(k) "v_cmp_eq_u32 vcc, %" #k ", %8"
#define I_CMP64(k) "v_cmp_eq_u32 s[20:21], %" #k ", %8"
#define I_CND32(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc"
#define I_CND64(k) "v_cndmask_b32 %" #k ", %" #k ", %8, s[20:21]"
#define I_ADDC(k) "v_addc_co_u32 %" #k ", vcc, %" #k ", %" #k ", vcc"
#define I_ADDC64(k) "v_addc_co_u32 %" #k ", vcc, %" #k ", %" #k ", s[20:21]"
#define I_MAX3(k) "v_max3_i32 %" #k ", %" #k ", %8, %9"
#define I_ADD3(k) "v_add3_u32 %" #k ", %" #k ", %8, %9"
#define I_LSHLOR(k) "v_lshl_or_b32 %" #k ", %" #k ", 4, %8"
#define I_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 1, %8"
#define I_OR3(k) "v_or3_b32 %" #k ", %" #k ", %8, %9"
#define I_ANDOR(k) "v_and_or_b32 %" #k ", %" #k ", %8, %9"
#define I_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 8, 8"
#define I_ALIGNBIT(k) "v_alignbit_b32 %" #k ", %" #k ", %8, 8"
#define I_PERM(k) "v_perm_b32 %" #k ", %" #k ", %8, %9"
#define I_MAD24(k) "v_mad_i32_i24 %" #k ", %" #k ", %8, %9"
#define I_MADU24(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %9"
#define I_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8"
#define I_PKADD(k) "v_pk_add_i16 %" #k ", %" #k ", %8"
#define I_PKMAX(k) "v_pk_max_i16 %" #k ", %" #k ", %8"
#define I_PKMINU(k) "v_pk_min_u16 %" #k ", %" #k ", %8"
#define I_PKMAD(k) "v_pk_mad_i16 %" #k ", %" #k ", %8, %9"
#define I_PKSUB(k) "v_pk_sub_i16 %" #k ", %" #k ", %8"
#define I_PKLSHL(k) "v_pk_lshlrev_b16 %" #k ", 4, %" #k
#define I_DPPSHR(k) "v_mov_b32_dpp %" #k ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf"
#define I_DPPROW(k) "v_mov_b32_dpp %" #k ", %8 row_shr:1 row_mask:0xf bank_mask:0xf"
#define I_ADDDPP(k) "v_add_u32_dpp %" #k ", %8, %" #k " row_shr:1 row_mask:0xf bank_mask:0xf"
#define I_ADDSDWA(k) "v_add_u32_sdwa %" #k ", %" #k ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"
#define I_CMPSDWA(k) "v_cmp_eq_u32_sdwa vcc, %" #k ", %8 src0_sel:BYTE_0 src1_sel:BYTE_2"
#define I_MAXF(k) "v_max_f32 %" #k ", %" #k ", %8"
#define I_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9"
#define I_ADDF64(k) "v_add_u32 %" #k ", %" #k ", %8"
#define I_MOV(k) "v_mov_b32 %" #k ", %8"
#define I_ADDI(k) "v_add_u32 %" #k ", 5, %" #k
#define I_MAXI(k) "v_max_i32 %" #k ", 0, %" #k
#define I_SUBREV(k) "v_subrev_u32 %" #k ", %8, %" #k
#define I_MED3(k) "v_med3_i32 %" #k ", %" #k ", %8, %9"
#define I_SAD(k) "v_sad_u32 %" #k ", %" #k ", %8, %9"
#define I_CMPGT(k) "v_cmp_gt_i32 vcc, %" #k ", %8"
#define I_CMPSWAP(k) "v_cmp_eq_u32 vcc, %" #k ", %8\n v_addc_co_u32 %" #k ", vcc, %" #k ", %" #k ", vcc"
#define I_MAX3_SAME(k) "v_max3_i32 %" #k ", %" #k ", %" #k ", %8"

#define LIST(X) \
  X(add, I_ADD) X(sub, I_SUB) X(max_i32, I_MAX) X(min_u32, I_MINU) X(xor_, I_XOR) X(lshlrev, I_LSHL) X(mov, I_MOV) X(add_imm, I_ADDI) X(max_imm0, I_MAXI) \
  X(cmp_e32, I_CMP32) X(cmp_gt_e32, I_CMPGT) X(cmp_e64_sgpr, I_CMP64) X(cndmask_vcc, I_CND32) X(cndmask_sgpr, I_CND64) \
  X(addc_vcc, I_ADDC) X(addc_sgpr, I_ADDC64) X(cmp_then_addc, I_CMPSWAP) \
  X(max3, I_MAX3) X(max3_2regs, I_MAX3_SAME) X(add3, I_ADD3) X(lshl_or, I_LSHLOR) X(lshl_add, I_LSHLADD) X(or3, I_OR3) X(and_or, I_ANDOR) X(bfe, I_BFE) \
  X(alignbit, I_ALIGNBIT) X(perm, I_PERM) X(mad_i24, I_MAD24) X(mad_u24, I_MADU24) X(mul_lo, I_MULLO) X(med3, I_MED3) X(sad, I_SAD) \
  X(pk_add_i16, I_PKADD) X(pk_max_i16, I_PKMAX) X(pk_min_u16, I_PKMINU) X(pk_mad_i16, I_PKMAD) X(pk_sub_i16, I_PKSUB) X(pk_lshl, I_PKLSHL) \
  X(dpp_wave_shr, I_DPPSHR) X(dpp_row_shr, I_DPPROW) X(add_dpp_row, I_ADDDPP) X(add_sdwa, I_ADDSDWA) X(cmp_sdwa, I_CMPSDWA) \
  X(max_f32, I_MAXF) X(fma_f32, I_FMA)

#define X(name, macro) KERNEL(k_##name, macro)
LIST(X)
#undef X

typedef void (*kern_t)(int*, int, int, int);
struct Entry { const char* name; kern_t fn; int per_slot; };

int main() {
  int* d_out; hipMalloc(&d_out, 4 * 64 * 256 * 4 * 8);
  Entry tab[] = {
#define X(name, macro) {#name, k_##name, 1},
    LIST(X)
#undef X
  };
  const int n = 300;
  printf("%-16s %12s %12s   (ns per wave-instruction per SIMD; cmp_then_addc counts 2 instr per slot)\n", "instr", "4 waves/SIMD", "8 waves/SIMD");
  for (auto& e : tab) {
    double res[2];
    int wi = 0;
    for (int w : {4, 8}) {
      const int blocks = 256 * 4 * w;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(64), 0, 0, d_out, 5, 1, 3);
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(64), 0, 0, d_out, n, 1, 3);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      res[wi++] = ms * 1e6 / ((double)n * 64 * w);
    }
    printf("%-16s %12.3f %12.3f\n", e.name, res[0], res[1]);
  }
  return 0;
}
