// pw_trace.hip -- K4 traceback kernel (one lane per pair walks the tie masks back from the end cell)
// and the run-time dispatch over the per-(type, BK) fill translation units.
#include "pw_device.h"

#include <stdlib.h>

namespace pw {

// `walkers` lanes of each wavefront walk one pair each (tuning knob PWLIB_WALKERS_PER_WAVE; measured on
// MI355X: 1..64 all land within 1.0-1.4 ms for 10 000 2 kb pairs -- the walk is bound by instructions per
// step, not by latency -- so the default packs 64).
__global__ __launch_bounds__(64) void k_trace(const TraceParams p, const int walkers) {
  // each walker's window of the mask plane; the +1 keeps the lanes' windows on different LDS banks
  __shared__ uint32_t win[64][WIN_WORDS + 1];
  const int pair = (int)blockIdx.x * walkers + (int)threadIdx.x;
  if ((int)threadIdx.x < walkers && pair < p.npairs) trace_walk(p, pair, win[threadIdx.x]);
}

// K4b, wave-parallel: 64 transcript positions per pass.  The (x, y) each position refers to is the start cell plus the
// number of origin- / mutant-consuming ops in front of it: a ballot and a popcount below the lane.  A long transcript (a
// strip-pipeline pair: 10^5 ops) is cut into gridDim.y segments, one wavefront each; a segment's start cell is the
// alignment's start plus the ops in front of the segment, which the wavefront counts itself (bytes equal to 'I' / 'D',
// four per lane and pass) -- no pass over the segments, no second kernel.
__device__ inline uint32_t bytes_equal(uint32_t w, uint32_t c4) {      // 0x80 in every byte of w that equals the byte of c4
  const uint32_t v = w ^ c4;
  const uint32_t t = (v & 0x7f7f7f7fu) + 0x7f7f7f7fu;
  return ~(t | v | 0x7f7f7f7fu);
}
__global__ __launch_bounds__(64) void k_trace_fixup(const TraceParams p) {
  const int pair = (int)blockIdx.x;
  const PairDesc& pd = p.pairs[pair];
  if (!pd.solvable) return;
  const Result r = p.results[pair];
  if (!(r.status & ST_TRACED) || r.tx_len <= 0) return;
  const uint8_t* oseq = p.arena + pd.o_off;
  const uint8_t* mseq = p.arena + pd.m_off;
  uint8_t* tx = p.transcripts + pd.tx_off + pd.tx_cap - r.tx_len;
  const int lane = (int)(threadIdx.x & 63u);
  const unsigned long long below = (1ull << lane) - 1ull;
  const int nseg = (int)gridDim.y, seg = (int)blockIdx.y;
  const int seglen = ((r.tx_len + nseg - 1) / nseg + 63) & ~63;
  const int s0 = seg * seglen;
  if (s0 >= r.tx_len) return;
  const int s1 = r.tx_len < s0 + seglen ? r.tx_len : s0 + seglen;
  int x = r.origin_idx, y = r.mutant_idx;
  if (s0 > 0) {
    // ops in [0, s0): every op but I consumes an origin letter, every op but D a mutant letter
    int nI = 0, nD = 0;
    const int head = (int)((4u - (uint32_t)((uintptr_t)tx & 3u)) & 3u);       // bytes up to the first aligned dword
    if (lane < head && lane < s0) { nI += tx[lane] == 'I'; nD += tx[lane] == 'D'; }
    const int nwords = s0 > head ? (s0 - head) >> 2 : 0;
    const uint32_t* tw = (const uint32_t*)(tx + head);
    for (int q = lane; q < nwords; q += 64) {
      const uint32_t w = tw[q];
      nI += __popc(bytes_equal(w, 0x49494949u)); nD += __popc(bytes_equal(w, 0x44444444u));
    }
    const int tail0 = head + 4 * nwords;
    if (tail0 + lane < s0) { nI += tx[tail0 + lane] == 'I'; nD += tx[tail0 + lane] == 'D'; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { nI += __shfl_xor(nI, off, 64); nD += __shfl_xor(nD, off, 64); }
    x += s0 - nI; y += s0 - nD;
  }
  for (int k0 = s0; k0 < s1; k0 += 64) {
    const int k = k0 + lane;
    const uint8_t ch = k < s1 ? tx[k] : (uint8_t)0;
    const bool isx = ch == 'X';
    const unsigned long long bo = __ballot(isx || ch == 'D'), bm = __ballot(isx || ch == 'I');
    if (isx) {
      const int mx = x + __popcll(bo & below), my = y + __popcll(bm & below);
      tx[k] = (oseq[mx] == mseq[my]) ? 'M' : 'S';
    }
    x += __popcll(bo); y += __popcll(bm);
  }
}

// ---- compaction of the transcripts (the bytes a multi-GPU gather moves) ---------------------------------------------
// The slots are X + Y + 1 bytes per pair, the ops themselves about half of that.  k_tx_offsets: exclusive prefix sum of
// tx_len over the pairs (one workgroup: a chunk per thread, the chunk sums scanned in LDS), offsets[n] = total bytes.
// k_tx_pack: one wavefront per pair copies its ops (right-aligned in the slot) to packed + offsets[pair].
__global__ __launch_bounds__(1024) void k_tx_offsets(const Result* __restrict__ results, int n, unsigned long long* __restrict__ offsets) {
  __shared__ unsigned long long part[1024];
  const int t = (int)threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int k0 = t * per, k1 = k0 + per < n ? k0 + per : n;
  unsigned long long s = 0;
  for (int k = k0; k < k1; k++) { const int len = results[k].tx_len; s += len > 0 ? (unsigned long long)len : 0ull; }
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {              // inclusive scan of the chunk sums
    const unsigned long long v = t >= off ? part[t - off] : 0ull;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  unsigned long long run = t > 0 ? part[t - 1] : 0ull;
  for (int k = k0; k < k1; k++) {
    offsets[k] = run;
    const int len = results[k].tx_len;
    run += len > 0 ? (unsigned long long)len : 0ull;
  }
  if (t == 1023) offsets[n] = part[1023];
}
__global__ __launch_bounds__(64) void k_tx_pack(const PairDesc* __restrict__ pairs, const Result* __restrict__ results,
                                                const uint8_t* __restrict__ slots, const unsigned long long* __restrict__ offsets,
                                                uint8_t* __restrict__ packed) {
  const int pair = (int)blockIdx.x;
  const int len = results[pair].tx_len;
  if (len <= 0) return;
  const PairDesc& pd = pairs[pair];
  const uint8_t* src = slots + pd.tx_off + pd.tx_cap - len;
  uint8_t* dst = packed + offsets[pair];
  const int lane = (int)threadIdx.x;
  // bytes up to the destination's first aligned dword, then dwords (the source read byte-wise: its alignment is another), then the tail
  const int head = (int)((4u - (uint32_t)((uintptr_t)dst & 3u)) & 3u);
  if (lane < head && lane < len) dst[lane] = src[lane];
  const int nwords = len > head ? (len - head) >> 2 : 0;
  for (int q = lane; q < nwords; q += 64) {
    const uint8_t* sp = src + head + 4 * q;
    PackedU32 v = *(const PackedU32*)sp;
    *(uint32_t*)(dst + head + 4 * q) = v.v;
  }
  const int tail0 = head + 4 * nwords;
  if (tail0 + lane < len) dst[tail0 + lane] = src[tail0 + lane];
}
hipError_t launch_tx_pack(const PairDesc* pairs, const Result* results, const uint8_t* slots, int n, uint64_t* offsets,
                          uint8_t* packed, hipStream_t st) {
  hipLaunchKernelGGL(k_tx_offsets, dim3(1), dim3(1024), 0, st, results, n, (unsigned long long*)offsets);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || n <= 0) return e;
  hipLaunchKernelGGL(k_tx_pack, dim3((unsigned)n), dim3(64), 0, st, pairs, results, slots, (const unsigned long long*)offsets, packed);
  return hipGetLastError();
}

// Standard-mode score plane [d - dmin][a] -> the reference's row-major table [x][y] as doubles (what the drop-in
// materialises for Aligner.table_scores, pw.py:278-285): coalesced writes, the strided reads stay on the device.
template <typename T>
__global__ __launch_bounds__(256) void k_table_rowmajor(const T* __restrict__ plane, int X, int Y, int pitch, double mul, double* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= (int64_t)(X + 1) * (Y + 1)) return;
  const int x = (int)(c / (Y + 1)), y = (int)(c - (int64_t)x * (Y + 1));
  out[c] = (double)plane[(int64_t)(x - y + Y) * pitch + (x < y ? x : y)] * mul;   // (mul: dyadic scaling, a power of two)
}
hipError_t launch_table_rowmajor(const void* plane, bool f64, int X, int Y, int pitch, double mul, double* out, hipStream_t st) {
  const int64_t n = (int64_t)(X + 1) * (Y + 1);
  const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
  if (f64) hipLaunchKernelGGL((k_table_rowmajor<double>), grid, blk, 0, st, (const double*)plane, X, Y, pitch, mul, out);
  else hipLaunchKernelGGL((k_table_rowmajor<int32_t>), grid, blk, 0, st, (const int32_t*)plane, X, Y, pitch, mul, out);
  return hipGetLastError();
}

hipError_t launch_trace(const TraceParams& p, hipStream_t st) {
  if (p.npairs <= 0) return hipSuccess;
  static int walkers = 0;
  if (walkers == 0) {
    const char* e = getenv("PWLIB_WALKERS_PER_WAVE");      // tuning knob; default measured on MI355X
    walkers = e ? atoi(e) : 64;
    if (walkers < 1 || walkers > 64) walkers = 64;
  }
  hipLaunchKernelGGL(k_trace, dim3((unsigned)((p.npairs + walkers - 1) / walkers)), dim3(64), 0, st, p, walkers);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // (segments: 1 for batches of short transcripts; with strip-pipeline pairs in the batch their transcripts are cut)
  hipLaunchKernelGGL(k_trace_fixup, dim3((unsigned)p.npairs, (unsigned)(p.fix_segments > 0 ? p.fix_segments : 1)), dim3(64), 0, st, p);
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
// the packed kernels: one launcher pair per (diagonals per lane, rule, matrix) translation unit (pw_fill16_tu.hip);
// rules 0 .. 5 without a matrix, rules 0 .. 3 with one
#define PW_BKS(M, R, MT) M(4, R, MT) M(8, R, MT) M(12, R, MT) M(16, R, MT) M(20, R, MT) M(24, R, MT) M(28, R, MT) M(32, R, MT)
#define PW_ALL16(M) PW_BKS(M, 0, 0) PW_BKS(M, 1, 0) PW_BKS(M, 2, 0) PW_BKS(M, 3, 0) PW_BKS(M, 4, 0) PW_BKS(M, 5, 0) \
                    PW_BKS(M, 0, 1) PW_BKS(M, 1, 1) PW_BKS(M, 2, 1) PW_BKS(M, 3, 1)
#define PW_DECL16(BK, R, MT) hipError_t launch_fill16_bk##BK##_r##R##_m##MT(const FillParams<int32_t>&, int, int, hipStream_t); \
  hipError_t launch_fill16mw_bk##BK##_r##R##_m##MT(const FillParams<int32_t>&, int, int, hipStream_t);
PW_ALL16(PW_DECL16)
#undef PW_DECL16

hipError_t launch_tile_i32(const FillParams<int32_t>&, int, int, int, hipStream_t);
hipError_t launch_tile_f64(const FillParams<double>&, int, int, int, hipStream_t);
hipError_t launch_tile_finish_i32(const FillParams<int32_t>&, int, hipStream_t);
hipError_t launch_tile_finish_f64(const FillParams<double>&, int, hipStream_t);
hipError_t launch_tile(const FillParams<int32_t>& a, int variant, int pair, int ntiles, hipStream_t st) { return launch_tile_i32(a, variant, pair, ntiles, st); }
hipError_t launch_tile(const FillParams<double>& a, int variant, int pair, int ntiles, hipStream_t st) { return launch_tile_f64(a, variant, pair, ntiles, st); }
hipError_t launch_tile_finish(const FillParams<int32_t>& a, int pair, hipStream_t st) { return launch_tile_finish_i32(a, pair, st); }
hipError_t launch_tile_finish(const FillParams<double>& a, int pair, hipStream_t st) { return launch_tile_finish_f64(a, pair, st); }

hipError_t launch_fill_mw_i32(const FillParams<int32_t>&, int, int, int, int, hipStream_t);
hipError_t launch_fill_mw_f64(const FillParams<double>&, int, int, int, int, hipStream_t);
hipError_t launch_fill_mw(const FillParams<int32_t>& a, int variant, int bk, int nw, int nblocks, hipStream_t st) {
  return launch_fill_mw_i32(a, variant, bk, nw, nblocks, st);
}
hipError_t launch_fill_mw(const FillParams<double>& a, int variant, int bk, int nw, int nblocks, hipStream_t st) {
  return launch_fill_mw_f64(a, variant, bk, nw, nblocks, st);
}

hipError_t launch_fill16(const FillParams<int32_t>& a, int bk, int seg, int rule, int mat, int nwaves, hipStream_t st) {
#define PW_CASE16(BK, R, MT) case ((BK) * 8 + (R)) * 2 + (MT): return launch_fill16_bk##BK##_r##R##_m##MT(a, seg, nwaves, st);
  switch ((bk * 8 + rule) * 2 + (mat ? 1 : 0)) {
    PW_ALL16(PW_CASE16)
    default: return hipErrorInvalidValue;
  }
#undef PW_CASE16
}

hipError_t launch_fill16_mw(const FillParams<int32_t>& a, int bk, int rule, int mat, int nw, int npairs, hipStream_t st) {
#define PW_CASE16(BK, R, MT) case ((BK) * 8 + (R)) * 2 + (MT): return launch_fill16mw_bk##BK##_r##R##_m##MT(a, nw, npairs, st);
  switch ((bk * 8 + rule) * 2 + (mat ? 1 : 0)) {
    PW_ALL16(PW_CASE16)
    default: return hipErrorInvalidValue;
  }
#undef PW_CASE16
}

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
