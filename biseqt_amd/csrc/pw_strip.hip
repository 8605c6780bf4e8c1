// pw_strip.hip -- gfx950 kernels of the strip pipeline (pw_strip.h, K2c): one pair wider than a workgroup.
//
//   k_fill_strip<TRACK>   persistent wavefronts (one per workgroup) draw strips of 64 rows from a work queue in index
//                         order; a strip takes the row above it from a FIFO in memory (8-byte granules written and read
//                         with agent-scope relaxed atomics: `global_store/load_dwordx2 ... sc1`, the data-tagged
//                         hand-off of MI355X_MICROARCH.md -- no fence, no flag) and everything else from its own lanes
//                         by DPP wave shifts.  Every wait is bounded: a wave that waits too long raises the abort flag
//                         and every wave leaves at its next look at it, so the grid always drains.
//   k_strip_finish        end cell of the pair from the per-strip candidates.
//   k_trace_strip         traceback over the strip layout, one wavefront per pair.
#include "pw_device.h"
#include "pw_strip.h"

namespace pw {

struct DevPS {
  PW_FN static uint32_t shl1_in(uint32_t m, bool flag) { return dev_shl1_in(m, flag); }
  PW_FN static int lane() { return (int)(threadIdx.x & 63u); }
  PW_FN static int32_t shr1(int32_t v, int32_t old) {
    return __builtin_amdgcn_update_dpp(old, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  }
  PW_FN static int32_t shl1(int32_t v, int32_t old) {
    return __builtin_amdgcn_update_dpp(old, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
  }
  // wave_shl:1 with zero fill (bound_ctrl): no dependence on an old value, so no copy in front of the DPP move
  PW_FN static int32_t shl1z(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true); }
  // value of lane `from` (any lane, through the LDS crossbar: ds_bpermute_b32, counted by lgkmcnt)
  PW_FN static uint32_t lane_from(uint32_t from, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from << 2), (int)v);
  }
  PW_FN static uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t bytes) { return __builtin_amdgcn_alignbyte(hi, lo, bytes); }
  PW_FN static uint32_t byte_of(uint32_t v, int i) { return __builtin_amdgcn_ubfe(v, 8u * (uint32_t)i, 8u); }   // folds into an SDWA byte select
  PW_FN static int32_t sbyte_of(uint32_t v, int i) { return __builtin_amdgcn_sbfe((int32_t)v, 8u * (uint32_t)i, 8u); }   // ... with sign extension
  // byte i of the result = byte sel.byte[i] (0 .. 3) of `row` (v_perm_b32: selector values 0 .. 3 pick from the second source)
  PW_FN static uint32_t perm_bytes(uint32_t row, uint32_t sel) { return __builtin_amdgcn_perm(row, row, sel); }
  PW_FN static void issue_here() { __builtin_amdgcn_sched_barrier(0); }
  // 2 h + b as ONE add-with-carry whose carry-in is the comparison's lane mask (the compiler would select and shift-or)
  PW_FN static int32_t twice_plus(int32_t h, bool b) {
    const uint64_t m = __builtin_amdgcn_ballot_w64(b);
    int32_t r;
    uint64_t cout;
    asm("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(r), "=s"(cout) : "v"(h), "s"(m));
    return r;
  }
  PW_FN static int32_t shfl_xor(int32_t v, int m) { return __shfl_xor(v, m, 64); }
  PW_FN static int32_t uniform(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
  PW_FN static bool all(bool p) { return __all(p ? 1 : 0) != 0; }
  PW_FN static uint64_t ballot(bool p) { return __ballot(p ? 1 : 0); }
  PW_FN static void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
  PW_FN static uint32_t readlane(uint32_t v, int i) { return (uint32_t)__builtin_amdgcn_readlane((int)v, i); }
  // FIFO granules: 8-byte agent-scope relaxed atomics on GLOBAL addresses (global_load/store_dwordx2 ... sc1; a generic
  // pointer would make these flat_* operations, which complete out of order and force vmcnt(0) waits on everything)
  typedef __attribute__((address_space(1))) uint64_t g_u64;
  typedef __attribute__((address_space(1))) uint32_t g_u32;
  PW_FN static uint64_t fifo_load(const uint64_t* p) {
    return __hip_atomic_load((const g_u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  PW_FN static void fifo_store(uint64_t* p, uint64_t v) {
    __hip_atomic_store((g_u64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // inside one XCD: a plain store stays in the shared L2, where the reader's L1-bypassing load finds it
  // ... with a load that bypasses the L1 and is served by that L2 (`nt`)
  PW_FN static uint64_t fifo_load_local(const uint64_t* p) { return __builtin_nontemporal_load((const g_u64*)p); }
  // (NOT volatile: a volatile store is emitted write-through and followed by s_waitcnt vmcnt(0))
  PW_FN static void fifo_store_local(uint64_t* p, uint64_t v) { *(g_u64*)p = v; }
  // The hand-over loads of the steady loop.  On gfx9-family targets loads and stores share vmcnt and the compiler waits for
  // vmcnt(0) as soon as both kinds are in flight -- here a full drain of the mask / FIFO stores (~1.2 us) at every
  // hand-over.  So these loads are issued by an asm statement the compiler does not count, and waited for by hand with a
  // count (wait_vm<SLOT, N>: at most N younger vector memory operations may remain in flight).  Their destination must
  // stay untouched until the data arrives, which no register the compiler allocates can promise (a copy frees the
  // original, the late data then lands in whatever lives there next): they go to ACCUMULATION registers a[2 SLOT],
  // a[2 SLOT + 1], which this kernel uses for nothing else (gfx950: unified VGPR / AGPR file; vector memory instructions
  // can target them).  The pair is zeroed first, so "not arrived yet" reads as a granule without a tag.
  // (the zeroing runs in EVERY lane -- call it outside the branch that picks the loading lanes -- so that a lane that
  //  loads nothing reads as "no granule, tag 0" too: the fast hand-over compares all 64 lanes at once)
  template <int SLOT> PW_FN static void slot_zero() {
    static_assert(SLOT == 0 || SLOT == 1, "two hand-over slots");
    if (SLOT == 0) asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0" ::: "memory", "a0", "a1");
    else asm volatile("v_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0" ::: "memory", "a2", "a3");
  }
  template <int SLOT> PW_FN static void fifo_load_async(const uint64_t* p, bool cross) {
    static_assert(SLOT == 0 || SLOT == 1, "two hand-over slots");
    if (SLOT == 0) {
      if (cross) asm volatile("global_load_dwordx2 a[0:1], %0, off sc1" :: "v"(p) : "memory", "a0", "a1");
      else asm volatile("global_load_dwordx2 a[0:1], %0, off nt" :: "v"(p) : "memory", "a0", "a1");
    } else {
      if (cross) asm volatile("global_load_dwordx2 a[2:3], %0, off sc1" :: "v"(p) : "memory", "a2", "a3");
      else asm volatile("global_load_dwordx2 a[2:3], %0, off nt" :: "v"(p) : "memory", "a2", "a3");
    }
  }
  // `v` into lanes 16 ROW .. 16 ROW + 15 of `old`, the other lanes keep theirs: ONE DPP move whose row mask enables that
  // row only (no lane mask in scalar registers, no select)
  template <int ROW> PW_FN static int32_t rowmov(int32_t old, int32_t v) {
    return __builtin_amdgcn_update_dpp(old, v, 0xe4 /* quad_perm:[0,1,2,3] */, 1 << ROW, 0xf, false);
  }
  template <int SLOT, int N> PW_FN static uint64_t wait_vm() {
    uint32_t lo, hi;
    if (SLOT == 0) asm volatile("s_waitcnt vmcnt(%2)\n\tv_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1" : "=v"(lo), "=v"(hi) : "n"(N) : "memory");
    else asm volatile("s_waitcnt vmcnt(%2)\n\tv_accvgpr_read_b32 %0, a2\n\tv_accvgpr_read_b32 %1, a3" : "=v"(lo), "=v"(hi) : "n"(N) : "memory");
    return ((uint64_t)hi << 32) | lo;
  }
  // mutant letters: scalar loads from the constant address space (lgkmcnt, not vmcnt; the frame is 4-byte aligned)
  PW_FN static uint32_t letters_dword(const uint8_t* m, int idx) {
    typedef const __attribute__((address_space(4))) uint32_t c_u32;
    return ((c_u32*)m)[idx];
  }
  PW_FN static void letters_x8(const uint8_t* m, int idx, uint32_t (&win)[8]) {
    typedef const __attribute__((address_space(4))) uint32_t c_u32;
    typedef uint32_t u8x __attribute__((ext_vector_type(8), aligned(4)));
    typedef const __attribute__((address_space(4))) u8x c_u8x;
    const u8x v = *(c_u8x*)((c_u32*)m + idx);
#pragma unroll
    for (int d = 0; d < 8; d++) win[d] = v[d];
  }
  // Loads of the waiting loops: issued and waited for inside one asm statement, so that the compiler's count of
  // outstanding memory operations -- which lets the fast path wait for exactly the load it needs (vmcnt(N)) -- never
  // meets a load inside a loop of unknown length (that would turn every wait into vmcnt(0)).
  PW_FN static uint64_t fifo_poll(const uint64_t* p) {
    uint64_t v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
  }
  // ... when the writer ran on this XCD: the streaming load (nt) that the hand-over loads use, served by the shared L2,
  // which has the plain store.  (sc0 alone is workgroup scope and may be served by this CU's L1 for ever; callers mix in
  // an agent-scope poll every few rounds, so that nothing depends on how a cache treats the hint.)
  PW_FN static uint64_t fifo_poll_local(const uint64_t* p) {
    uint64_t v;
    asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
  }
  PW_FN static uint32_t flag_poll(const uint32_t* p) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
  }
  PW_FN static void sleep() { __builtin_amdgcn_s_sleep(1); }
  PW_FN static void exit_wave() { __builtin_amdgcn_endpgm(); }
  PW_FN static uint64_t clock() { return __builtin_amdgcn_s_memrealtime(); }      // 100 MHz
  PW_FN static uint64_t cycles() { return __builtin_amdgcn_s_memtime(); }         // shader clock
  PW_FN static uint32_t flag_load(const uint32_t* p) { return __hip_atomic_load((const g_u32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  PW_FN static void flag_set(uint32_t* p) { __hip_atomic_store((g_u32*)p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  PW_FN static int32_t in_vgpr(int32_t v) { asm volatile("" : "+v"(v)); return v; }
};

// XCC id of the executing wavefront (HW_REG_XCC_ID = 20, bits 3:0).  Used for placement, which decides speed and the
// flavour of a store, never whether a strip gets done: a queue nobody serves ends in the bounded waits' abort.
PW_FN int xcc_id() { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu); }

template <bool TRACK, bool BROW>
__global__ __launch_bounds__(64) void k_fill_strip(const StripParams a) {
  const int qi = a.xcc_queue[xcc_id() & 7];
  if (qi < 0) return;
  for (;;) {
    int t = 0;
    if (threadIdx.x == 0) t = (int)atomicAdd(a.ctl + qi, 1u);     // this XCD's strips, in index order
    t = __builtin_amdgcn_readfirstlane(t);
    const int run = qi + a.nq * (t / a.run_len), i = t % a.run_len;
    const int w = run * a.run_len + i;
    if (w >= a.nstrips) return;
    if (DevPS::flag_load(a.ctl + kStripAbort) != 0u) return;
    if (a.stamps != nullptr && threadIdx.x == 0) {
      a.stamps[(uint64_t)w * 16] = DevPS::clock();
      // [7]: XCC id | workgroup << 8 | HW_REG_HW_ID << 32 (wave slot, SIMD, CU, SE: which strips shared a SIMD)
      a.stamps[(uint64_t)w * 16 + 7] = (uint64_t)xcc_id() | ((uint64_t)blockIdx.x << 8) |
                                      ((uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32);
    }
    StripFill<DevPS, TRACK, BROW> f(a);
    f.run(w, i == 0, i == a.run_len - 1);      // (an abandoned wait ends the wavefront inside)
  }
}

// Which XCC ids does this device have?  (One workgroup per slot of a grid large enough to reach every XCD.)
__global__ __launch_bounds__(64) void k_xcc_census(uint32_t* seen) {
  if (threadIdx.x == 0) atomicOr(seen + (xcc_id() & 7), 1u);
}

__global__ __launch_bounds__(64) void k_strip_finish(const StripParams a) { strip_reduce<DevPS>(a); }

__global__ __launch_bounds__(64) void k_trace_strip(const StripTraceParams p) {
  __shared__ uint32_t win[kWalkWinWords];
  strip_walk<DevPS>(p, win);
}

hipError_t launch_strip_fill(const StripParams& a, bool track, bool byte_rows, const uint32_t* ctl_init, int nworkers, int lds_bytes,
                             hipStream_t st) {
  // the work queue heads and the abort flag start at zero, the byte rows at their values (16 dwords the caller keeps alive)
  hipError_t e = hipMemcpyAsync(a.ctl, ctl_init, 16 * sizeof(uint32_t), hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return e;
  const dim3 grid((unsigned)nworkers), block(64);
  if (track && byte_rows) hipLaunchKernelGGL((k_fill_strip<true, true>), grid, block, (size_t)lds_bytes, st, a);
  else if (track) hipLaunchKernelGGL((k_fill_strip<true, false>), grid, block, (size_t)lds_bytes, st, a);
  else if (byte_rows) hipLaunchKernelGGL((k_fill_strip<false, true>), grid, block, (size_t)lds_bytes, st, a);
  else hipLaunchKernelGGL((k_fill_strip<false, false>), grid, block, (size_t)lds_bytes, st, a);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_strip_finish, dim3(1), dim3(64), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_xcc_census(uint32_t* d_seen8, hipStream_t st) {
  hipLaunchKernelGGL(k_xcc_census, dim3(4096), dim3(64), 0, st, d_seen8);
  return hipGetLastError();
}

hipError_t launch_strip_trace(const StripTraceParams& p, hipStream_t st) {
  hipLaunchKernelGGL(k_trace_strip, dim3(1), dim3(64), 0, st, p);
  return hipGetLastError();
}

}  // namespace pw
