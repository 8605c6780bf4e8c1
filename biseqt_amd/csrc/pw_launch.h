// pw_launch.h -- host-visible launchers of the gfx950 kernels in pw_kernels.hip.
#ifndef PW_LAUNCH_H
#define PW_LAUNCH_H

#include <hip/hip_runtime_api.h>

#include "pw_types.h"

namespace pw {

// Fill-kernel variants (template switches of WaveFill, pw_wave.h).
enum { VAR_FAST_ANY_TRACK = 0,  // begin anywhere + per-diagonal best: LOCAL, B_LOCAL (END_ANCHORED rides along)
       VAR_FAST_TRACK = 1,      // begin at origin/edges + per-diagonal best: START_ANCHORED
       VAR_FAST = 2,            // begin at origin/edges, end on the table edge: GLOBAL, *OVERLAP, B_GLOBAL, B_OVERLAP
       VAR_GENERIC = 3,         // substitution matrix / go > 0 / score-plane dump: everything at run time
       VAR_FAST16 = 4 };        // packed 16-bit (LOCAL / B_LOCAL, B_OVERLAP, B_GLOBAL), several pairs per wavefront (launch_fill16)

static const int kSupportedBK[] = {2, 4, 8, 16, 32};
static const int kNumSupportedBK = 5;

hipError_t launch_fill(const FillParams<int32_t>& a, int variant, int bk, int nblocks, hipStream_t st);
hipError_t launch_fill(const FillParams<double>& a, int variant, int bk, int nblocks, hipStream_t st);
// lane-packed 16-bit kernel (VAR_FAST16): one block per WaveDesc
static const int kPackedBK[] = {4, 8, 12, 16, 20, 24, 28, 32};
static const int kNumPackedBK = 8;
// seg != 0: several pairs per wavefront (WaveDesc.nl lanes each); seg == 0: one pair per wavefront, WaveDesc.nl == 64
// rule: 0 .. 5 (pw_wave.h, WaveFill16); mat != 0: scores from FillParams::mat_rows (rules 0 .. 3 only)
hipError_t launch_fill16(const FillParams<int32_t>& a, int bk, int seg, int rule, int mat, int nwaves, hipStream_t st);
hipError_t launch_fill16_mw(const FillParams<int32_t>& a, int bk, int rule, int mat, int nw, int npairs, hipStream_t st);   // nw wavefronts per pair
// wide bands: one workgroup of nw wavefronts (2048 diagonals each, nw <= kMaxWavesPerPair) per pair
static const int kMaxWavesPerPair = 8;
hipError_t launch_fill_mw(const FillParams<int32_t>& a, int variant, int bk, int nw, int nblocks, hipStream_t st);
hipError_t launch_fill_mw(const FillParams<double>& a, int variant, int bk, int nw, int nblocks, hipStream_t st);
// tiled single-pair kernel (K2b): tiles of kTileCentralDiags diagonals + ghosts, time blocks of kTileBlocks blocks
// (geometry overridable at build time for tuning runs: -DPW_TILE_LANES= -DPW_TILE_BK= -DPW_TILE_GHOST= -DPW_TILE_BLOCKS=)
#ifndef PW_TILE_LANES
#define PW_TILE_LANES 256
#endif
#ifndef PW_TILE_BK
#define PW_TILE_BK 2
#endif
#ifndef PW_TILE_GHOST
#define PW_TILE_GHOST 64
#endif
#ifndef PW_TILE_BLOCKS
#define PW_TILE_BLOCKS (PW_TILE_GHOST * PW_TILE_BK / 16)
#endif
static const int kTileBKHost = PW_TILE_BK, kTileCentralLanes = PW_TILE_LANES - 2 * PW_TILE_GHOST, kTileBlocks = PW_TILE_BLOCKS;
hipError_t launch_tile(const FillParams<int32_t>& a, int variant, int pair, int ntiles, hipStream_t st);
hipError_t launch_tile(const FillParams<double>& a, int variant, int pair, int ntiles, hipStream_t st);
hipError_t launch_tile_finish(const FillParams<int32_t>& a, int pair, hipStream_t st);
hipError_t launch_tile_finish(const FillParams<double>& a, int pair, hipStream_t st);
hipError_t launch_trace(const TraceParams& p, hipStream_t st);
// compaction of the transcripts: offsets[n + 1] = exclusive prefix sum of tx_len, packed = the ops back to back
hipError_t launch_tx_pack(const PairDesc* pairs, const Result* results, const uint8_t* slots, int n, uint64_t* offsets,
                          uint8_t* packed, hipStream_t st);
// strip pipeline (pw_strip.h / pw_strip.hip): one standard-mode pair wider than a workgroup
struct StripParams;
struct StripTraceParams;
hipError_t launch_strip_fill(const StripParams& a, bool track, bool byte_rows, const uint32_t* ctl_init, int nworkers, int lds_bytes,
                             hipStream_t st);
hipError_t launch_strip_trace(const StripTraceParams& p, hipStream_t st);
hipError_t launch_xcc_census(uint32_t* d_seen8, hipStream_t st);
hipError_t launch_table_rowmajor(const void* plane, bool f64, int X, int Y, int pitch, double mul, double* out, hipStream_t st);

}  // namespace pw
#endif
