/* pw_batch.h -- batch entry points of libpwlib (pwlib.so): many independent alignment problems per call.
 *
 * NEW SURFACE (no reference counterpart): the reference solves one pair per dptable (pw.c:47-114) and
 * has no batch call.  These functions sit beside the four drop-in functions of pwlib.h
 * in the same shared object; per pair they compute exactly what
 *     dptable_init -> dptable_solve -> dptable_traceback(T, opt)
 * computes (reference pw.c:10-26, 47-114, 116-151), for a whole batch in a handful of kernel launches.
 *
 * Plain C ABI: pointers and sizes only.  Sequences are passed as one byte per letter in a single
 * "arena"; pairs reference frames of it by offset and length (the reference's alnframe
 * origin_range/mutant_range, pwlib.h:84-89, already applied).  `stream` arguments are a hipStream_t
 * passed as void* (NULL = the default stream); device pointers are ordinary device addresses, so a
 * caller may hand in memory it owns (e.g. a torch tensor's data_ptr()).
 */
#ifndef PW_BATCH_H
#define PW_BATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pw_batch pw_batch;

/* Scoring shared by all pairs of a batch (the reference's alnscores, pwlib.h:71-78, plus mode/type). */
typedef struct {
  int mode;             /* alnmode:  0 STD_MODE, 1 BANDED_MODE */
  int type;             /* std_alntype 0..6 or banded_alntype 0..2 */
  int alphabet_len;     /* L; letters are 0 .. L-1, L <= 256 */
  const double* subst;  /* row-major L x L, subst[o*L + m] = subst_scores[o][m] */
  double go, ge;        /* gap open / gap extend scores */
} pw_scoring;

/* One problem. */
typedef struct {
  uint64_t origin_off;  /* byte offset of the origin frame in the arena; must be a multiple of 4 */
  uint64_t mutant_off;  /* byte offset of the mutant frame in the arena; must be a multiple of 4 */
  int32_t origin_len;   /* X */
  int32_t mutant_len;   /* Y */
  int32_t dmin, dmax;   /* banded mode: diag_range as given (it is clamped like _pw_internals.c:29-36) */
} pw_pair;

/* One result; 32 bytes.  Identical on host and device: this is the record a multi-GPU gather moves. */
typedef struct {
  double score;         /* cells[opt].choices[0].score */
  int32_t opt_i, opt_j; /* what dptable_solve returns: (x,y) in STD mode, (d-dmin, a) in banded mode; -1,-1 = none */
  int32_t origin_idx;   /* alignment start relative to the frame start (add origin_range.i) */
  int32_t mutant_idx;
  int32_t tx_len;       /* transcript length (0 if none) */
  int32_t status;       /* PW_ST_* bits */
} pw_result;

#define PW_ST_TRACED 1  /* traceback ran for this pair */
#define PW_ST_EMPTY 2   /* empty transcript: the reference's dptable_traceback returns NULL (pw.c:135-138) */
#define PW_ST_PANICK 4  /* the reference would exit(1) here (pw.c:132-134) */
#define PW_ST_BADPATH 8 /* internal error: the traceback left the table (never expected) */

/* flags of pw_batch_create */
#define PW_FLAG_DUMP_SCORES 1   /* also write the score of every cell (for table_scores-style callers) */
#define PW_FLAG_FORCE_F64 2     /* compute in double even if all scores are small integers */
#define PW_FLAG_FORCE_GENERIC 4 /* use the run-time-everything kernel (testing) */
#define PW_FLAG_PROFILE 8       /* bracket every fill launch with HIP events (pw_batch_fill_ms) */
#define PW_FLAG_NO_PACKED16 16   /* never use the packed 16-bit steady-phase kernel (testing / A-B) */
#define PW_FLAG_FORCE_TILED 32   /* run every pair through the time-blocked tiled kernel (testing) */
#define PW_FLAG_FORCE_STRIP 64   /* run every pair that the strip pipeline supports through it (testing) */
#define PW_FLAG_SHARED_ARENA 128 /* the batch owns no arena: pw_batch_share_arena gives it a caller-owned device copy */
#define PW_FLAG_NO_STRIP 256     /* never use the strip pipeline (K2c): wide pairs take the tiled / workgroup kernels */

const char* pw_last_error(void);
int pw_device_count(void);
int pw_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes);   /* hipMemGetInfo of the runtime the library uses */

/* Plans the batch (band clamp / feasibility per pair exactly as dptable_init), picks the kernel variants,
 * allocates every device buffer and uploads the descriptors.  The arena is `arena_bytes` long; its
 * contents are supplied later (pw_batch_upload_arena or pw_batch_arena_device).  NULL on error. */
/* The library keeps the big device buffers of destroyed batches (>= 1 MB each) for the next batch on that device: per
 * device up to PWLIB_POOL_GB if set (0 disables), otherwise up to a quarter of the device's memory and never more than half
 * of what is free at the moment a buffer is parked.  pw_pool_trim releases them all -- call it before handing the GPU's
 * memory to another allocator in the process (torch, RCCL). */
void pw_pool_trim(void);

pw_batch* pw_batch_create(int device, const pw_scoring* scoring, int32_t n_pairs, const pw_pair* pairs,
                          uint64_t arena_bytes, uint32_t flags);
/* Waits for what was launched FOR THIS BATCH (an event per stream it was used on, recorded behind every launch of the
 * library) before releasing or parking its buffers, so a batch may be destroyed while its last launches are still in
 * flight, without stalling other batches' streams.  Work the caller itself queued on the batch's buffers (its own kernels
 * reading pw_batch_results_device, say) is the caller's to wait for. */
void pw_batch_destroy(pw_batch* b);

/* The planner alone: what pw_batch_create would choose for these problems, computed on the host with no device call and
 * nothing allocated (works on a machine without a GPU; the tests pin the kernel choices with it).  `kernel` receives the
 * name pw_batch_kernel_name would report (NUL-terminated, at most kernel_cap bytes); `info`, if not NULL, 8 ints: score
 * type (0 int32 / 1 double), dyadic scale shift (scores are held times 2^shift), pairs on one-wavefront kernels, on
 * workgroup kernels, on the tiled kernel, on the strip pipeline, packed rule (-1: not a packed kernel), matrix form (0 / 1).
 * 0, or -1 with pw_last_error set. */
int pw_plan_only(const pw_scoring* scoring, int32_t n_pairs, const pw_pair* pairs, uint64_t arena_bytes, uint32_t flags,
                 char* kernel, int32_t kernel_cap, int32_t* info);

/* per-pair planning results (host side, available right after create) */
int pw_batch_init_rc(const pw_batch* b, int32_t k);                         /* 0 or -1, as dptable_init */
int pw_batch_band(const pw_batch* b, int32_t k, int32_t* dmin, int32_t* dmax, int32_t* num_rows);
int64_t pw_batch_pair_cells(const pw_batch* b, int32_t k);                   /* cells the reference allocates */
int64_t pw_batch_cells(const pw_batch* b);                                   /* sum over solvable pairs */
int64_t pw_batch_algorithmic_bytes(const pw_batch* b);                       /* 0.5 B/cell + X+Y + 32 per pair */
int pw_batch_score_type(const pw_batch* b);                                  /* 0 int32, 1 double */
const char* pw_batch_kernel_name(const pw_batch* b);                         /* fill kernel of the largest pair class */

int pw_batch_upload_arena(pw_batch* b, const uint8_t* host_arena, uint64_t bytes);   /* synchronous H2D */
void* pw_batch_arena_device(pw_batch* b);
/* One device copy of an arena for MANY batches (overlap pipelines: every read takes part in dozens of pairs, the pairs are
 * solved in several batches): pw_arena_upload returns a device copy of `bytes` bytes (+ the slack the kernels read), owned
 * by the caller until pw_arena_free; a batch created with PW_FLAG_SHARED_ARENA reads its frames from the copy it is given
 * by pw_batch_share_arena (at least the batch's arena_bytes long) instead of allocating and uploading its own. */
void* pw_arena_upload(int device, const uint8_t* host_arena, uint64_t bytes);
void pw_arena_free(int device, void* dev_arena);
int pw_batch_share_arena(pw_batch* b, void* dev_arena);
/* Pinned host memory for the asynchronous transfers below (hipHostMalloc / hipHostFree). */
void* pw_host_alloc(uint64_t bytes);
void pw_host_free(void* p);
/* H2D of the arena / D2H of the records and transcript slots, asynchronous on `stream` (ordered with the kernels
 * launched on it); host buffers should come from pw_host_alloc, the caller synchronises (pw_batch_sync). */
int pw_batch_upload_arena_async(pw_batch* b, const uint8_t* host_arena, uint64_t bytes, void* stream);
int pw_batch_results_async(pw_batch* b, pw_result* host_out, void* stream);
int pw_batch_transcripts_async(pw_batch* b, uint8_t* host_out, void* stream);

/* K1 (+ end-cell search): asynchronous on `stream`. */
int pw_batch_solve(pw_batch* b, void* stream);
/* K4 from each pair's optimal cell: asynchronous on `stream`, after pw_batch_solve on the same stream. */
int pw_batch_traceback(pw_batch* b, void* stream);
/* K4 from explicit end cells (table coordinates i,j per pair, host array of 2*n_pairs ints). */
int pw_batch_traceback_from(pw_batch* b, const int32_t* ends_ij, void* stream);
int pw_batch_sync(pw_batch* b, void* stream);

/* results: device-resident buffers and D2H copies */
void* pw_batch_results_device(pw_batch* b);            /* pw_result[n_pairs] */
void* pw_batch_transcripts_device(pw_batch* b);        /* transcript slots, see pw_batch_tx_slot */
uint64_t pw_batch_transcripts_bytes(const pw_batch* b);
int pw_batch_tx_slot(const pw_batch* b, int32_t k, uint64_t* off, int32_t* cap); /* ops of pair k end at off+cap */
/* Synchronous D2H.  A wide pair whose strip pipeline (K2c) gave up waiting (a starved or contended device; its record then
 * carries PW_ST_BADPATH and no end cell) is solved again here, once, with the strips disabled, and every later traceback of
 * the batch serves it from that replacement: callers of this function never see the abandoned state.  The *_async readers
 * and pw_batch_results_device see the status bit only -- call pw_batch_results once to trigger the repair.  An error is
 * returned only if the second solve fails as well. */
int pw_batch_results(pw_batch* b, pw_result* host_out);
int pw_batch_transcripts(pw_batch* b, uint8_t* host_out);                        /* synchronous D2H of all slots */
/* Compaction (what a multi-GPU gather should move: the slots are X + Y + 1 bytes per pair, the ops about half of that).
 * After a traceback on `stream`: the ops of all pairs back to back in pair order, no padding, and their exclusive offsets
 * (uint64[n_pairs + 1]; [n_pairs] = total bytes), both device-resident; asynchronous on `stream`.  Pairs without a
 * transcript take no bytes.  pw_batch_packed: synchronous D2H of offsets (n_pairs + 1) and / or bytes (either may be NULL);
 * pw_batch_packed_total_async: D2H of the total alone (8 bytes, into pinned memory), ordered on `stream`. */
int pw_batch_pack_transcripts(pw_batch* b, void* stream);
void* pw_batch_packed_device(pw_batch* b);
void* pw_batch_packed_offsets_device(pw_batch* b);
int pw_batch_packed_total_async(pw_batch* b, uint64_t* host_out, void* stream);
int pw_batch_packed(pw_batch* b, uint8_t* host_out, uint64_t cap, uint64_t* offsets_out);
/* score plane of pair k (PW_FLAG_DUMP_SCORES): out[(d-dmin)*pitch + a], pitch = min(X,Y)+1, as doubles */
int pw_batch_scores(pw_batch* b, int32_t k, double* host_out, int64_t n);
/* Standard mode only: the same scores as the reference's table, host_out[x * (Y + 1) + y] for 0 <= x <= X,
 * 0 <= y <= Y (n >= (X + 1)(Y + 1)); transposed on the device.  Needs PW_FLAG_DUMP_SCORES. */
int pw_batch_table(pw_batch* b, int32_t k, double* host_out, int64_t n);

/* mean duration (ms) of the fill kernel launches of the last pw_batch_solve (PW_FLAG_PROFILE), and
 * of the dominant launch alone; < 0 if unavailable.  Synchronises on the recorded events. */
float pw_batch_fill_ms(pw_batch* b);
float pw_batch_trace_ms(pw_batch* b);

#ifdef __cplusplus
}
#endif
#endif
