"""ctypes binding of biseqt_amd/pwlib/pwlib.so (the HIP library; C ABI in include/pwlib.h and
include/pw_batch.h).

The reference binds its C library with cffi in ABI mode (``biseqt/pw.py:45-69``); cffi is not
available in this image, so the same structs are declared with ctypes -- field for field the layout of
``pwlib.h`` (checked by ``check_layout``).  There is no fallback: if the shared object is missing the
import fails with instructions, and every computing call goes to the GPU kernels.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PWLIB_SO = os.environ.get('PWLIB_SO', os.path.join(HERE, 'pwlib', 'pwlib.so'))   # override: A/B experiments only
PWLIB_H = os.path.join(HERE, 'pwlib', 'pwlib.h')

# enum values of include/pwlib.h (reference pwlib.h:30-33, 39-54, 60-65)
STD_MODE, BANDED_MODE = 0, 1
GLOBAL, LOCAL, START_ANCHORED, END_ANCHORED, OVERLAP, START_ANCHORED_OVERLAP, \
    END_ANCHORED_OVERLAP = range(7)
B_GLOBAL, B_LOCAL, B_OVERLAP = range(3)


class intpair(C.Structure):
    _fields_ = [('i', C.c_int), ('j', C.c_int)]


class alnscores(C.Structure):
    _fields_ = [('subst_scores', C.POINTER(C.POINTER(C.c_double))),
                ('gap_open_score', C.c_double),
                ('gap_extend_score', C.c_double)]


class alnframe(C.Structure):
    _fields_ = [('origin', C.POINTER(C.c_int)),
                ('mutant', C.POINTER(C.c_int)),
                ('origin_range', intpair),
                ('mutant_range', intpair)]


class std_alnparams(C.Structure):
    _fields_ = [('type', C.c_int)]


class banded_alnparams(C.Structure):
    _fields_ = [('type', C.c_int), ('dmin', C.c_int), ('dmax', C.c_int)]


class alnprob(C.Structure):
    _fields_ = [('frame', C.POINTER(alnframe)),
                ('scores', C.POINTER(alnscores)),
                ('max_new_mins', C.c_int),
                ('mode', C.c_int),
                ('params', C.c_void_p)]      # union { std_alnparams*, banded_alnparams* }


class alnchoice(C.Structure):
    pass


alnchoice._fields_ = [('op', C.c_char),
                      ('score', C.c_double),
                      ('base', C.POINTER(alnchoice)),
                      ('mins_cd', C.c_int),
                      ('cur_min', C.c_int)]


class dpcell(C.Structure):
    _fields_ = [('num_choices', C.c_int),
                ('choices', C.POINTER(alnchoice))]


class dptable(C.Structure):
    _fields_ = [('cells', C.POINTER(C.POINTER(dpcell))),
                ('num_rows', C.c_int),
                ('row_lens', C.POINTER(C.c_int)),
                ('prob', C.POINTER(alnprob))]


class alignment(C.Structure):
    _fields_ = [('origin_idx', C.c_int),
                ('mutant_idx', C.c_int),
                ('score', C.c_double),
                ('transcript', C.c_char_p)]


# ---- batch API (include/pw_batch.h) ----
class pw_scoring(C.Structure):
    _fields_ = [('mode', C.c_int), ('type', C.c_int), ('alphabet_len', C.c_int),
                ('subst', C.POINTER(C.c_double)), ('go', C.c_double), ('ge', C.c_double)]


class pw_pair(C.Structure):
    _fields_ = [('origin_off', C.c_uint64), ('mutant_off', C.c_uint64),
                ('origin_len', C.c_int32), ('mutant_len', C.c_int32),
                ('dmin', C.c_int32), ('dmax', C.c_int32)]


class pw_result(C.Structure):
    _fields_ = [('score', C.c_double), ('opt_i', C.c_int32), ('opt_j', C.c_int32),
                ('origin_idx', C.c_int32), ('mutant_idx', C.c_int32),
                ('tx_len', C.c_int32), ('status', C.c_int32)]


PW_ST_TRACED, PW_ST_EMPTY, PW_ST_PANICK, PW_ST_BADPATH = 1, 2, 4, 8
PW_FLAG_DUMP_SCORES, PW_FLAG_FORCE_F64, PW_FLAG_FORCE_GENERIC, PW_FLAG_PROFILE = 1, 2, 4, 8
PW_FLAG_NO_PACKED16 = 16
PW_FLAG_FORCE_TILED = 32
PW_FLAG_FORCE_STRIP = 64
PW_FLAG_SHARED_ARENA = 128
PW_FLAG_NO_STRIP = 256

SIZEOF = dict(intpair=8, alnscores=24, alnframe=32, std_alnparams=4, banded_alnparams=12,
              alnprob=32, alnchoice=32, dpcell=16, dptable=32, alignment=24,
              pw_scoring=40, pw_pair=32, pw_result=32)

# every symbol include/pwlib.h and include/pw_batch.h declare
EXPORTS = ['dptable_init', 'dptable_solve', 'dptable_traceback', 'dptable_free',
           'pw_last_error', 'pw_device_count', 'pw_device_memory', 'pw_pool_trim', 'pw_batch_create', 'pw_batch_destroy',
           'pw_batch_init_rc', 'pw_batch_band', 'pw_batch_pair_cells', 'pw_batch_cells',
           'pw_batch_algorithmic_bytes', 'pw_batch_score_type', 'pw_batch_kernel_name', 'pw_batch_upload_arena',
           'pw_batch_arena_device', 'pw_arena_upload', 'pw_arena_free', 'pw_batch_share_arena', 'pw_host_alloc', 'pw_host_free', 'pw_batch_upload_arena_async',
           'pw_batch_results_async', 'pw_batch_transcripts_async', 'pw_batch_solve', 'pw_batch_traceback',
           'pw_batch_traceback_from', 'pw_batch_sync', 'pw_batch_results_device',
           'pw_batch_transcripts_device', 'pw_batch_transcripts_bytes', 'pw_batch_tx_slot',
           'pw_batch_results', 'pw_batch_transcripts', 'pw_batch_scores', 'pw_batch_table', 'pw_batch_fill_ms',
           'pw_batch_trace_ms', 'pw_batch_pack_transcripts', 'pw_batch_packed_device', 'pw_batch_packed_offsets_device',
           'pw_batch_packed_total_async', 'pw_batch_packed', 'pw_plan_only']
# every symbol include/pw_seeds.h declares
SEED_EXPORTS = ['pw_seeds_create', 'pw_seeds_build', 'pw_seeds_num_rows', 'pw_seeds_is_self', 'pw_seeds_rows_device',
                'pw_seeds_rows', 'pw_seeds_count', 'pw_seeds_kmers', 'pw_seeds_band_neighbours', 'pw_seeds_graph_build', 'pw_seeds_graph_num_points', 'pw_seeds_graph_points',
                'pw_seeds_graph_counts', 'pw_seeds_graph_fetch', 'pw_seeds_graph_components', 'pw_seeds_build_ms',
                'pw_seeds_algorithmic_bytes', 'pw_seeds_destroy', 'pw_seeds_last_error']


# every symbol include/pw_overlap.h declares
OVERLAP_EXPORTS = ['pw_overlap_bands', 'pw_overlap_all_pairs', 'pw_overlap_last_ms', 'pw_overlap_last_error']


class pw_read_pair(C.Structure):
    _fields_ = [('s_off', C.c_uint64), ('t_off', C.c_uint64), ('s_len', C.c_int32), ('t_len', C.c_int32)]


def check_layout():
    for name, size in SIZEOF.items():
        assert C.sizeof(globals()[name]) == size, (name, C.sizeof(globals()[name]), size)


_lib = None


def _bind_hip_runtime():
    """One HIP runtime per process.  ``pwlib.so`` needs ``libamdhip64.so.7``; a PyTorch wheel bundles its own copy of
    that library (same SONAME) and two HIP / HSA runtimes in one process cannot both find the GPU.  If ``pwlib.so`` is
    loaded first it would pull in the system runtime and a later ``import torch`` would bring the second one, so when a
    torch installation with a bundled runtime exists and none is mapped yet, that copy is loaded first (RTLD_GLOBAL):
    ``pwlib.so`` then binds to it by SONAME, and torch, imported later or never, finds its own library already there.
    ``PWLIB_HIP_RUNTIME=system`` keeps the system runtime, ``PWLIB_HIP_RUNTIME=/path/libamdhip64.so`` picks one."""
    import sys
    choice = os.environ.get('PWLIB_HIP_RUNTIME', '')
    if choice == 'system':
        return None
    try:
        with open('/proc/self/maps') as f:
            if 'libamdhip64' in f.read():
                return None                      # a runtime is mapped already (torch imported first): bind to it
    except OSError:
        pass
    path = choice
    if not path:
        if 'torch' in sys.modules:
            return None
        try:
            import importlib.util
            spec = importlib.util.find_spec('torch')             # locates the package without importing it
        except (ImportError, ValueError):
            spec = None
        if spec is None or not spec.origin:
            return None
        path = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
    if not os.path.exists(path):
        return None
    try:
        return C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        return None


def load():
    """dlopen pwlib.so and declare its prototypes.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(PWLIB_SO):
        raise ImportError('%s is missing: build it with `python -m biseqt_amd.csrc.build` '
                          '(hipcc, gfx950); there is no CPU fallback' % PWLIB_SO)
    _bind_hip_runtime()
    lib = C.CDLL(PWLIB_SO)
    P = C.POINTER
    lib.dptable_init.argtypes = [P(dptable)]
    lib.dptable_init.restype = C.c_int
    lib.dptable_solve.argtypes = [P(dptable)]
    lib.dptable_solve.restype = intpair
    lib.dptable_traceback.argtypes = [P(dptable), intpair]
    lib.dptable_traceback.restype = P(alignment)
    lib.dptable_free.argtypes = [P(dptable)]
    lib.dptable_free.restype = None
    lib.pw_last_error.restype = C.c_char_p
    lib.pw_device_count.restype = C.c_int
    lib.pw_device_memory.argtypes = [C.c_int, P(C.c_uint64), P(C.c_uint64)]
    lib.pw_pool_trim.restype = None
    lib.pw_batch_create.argtypes = [C.c_int, P(pw_scoring), C.c_int32, P(pw_pair), C.c_uint64, C.c_uint32]
    lib.pw_batch_create.restype = C.c_void_p
    lib.pw_batch_destroy.argtypes = [C.c_void_p]
    lib.pw_batch_destroy.restype = None
    lib.pw_batch_init_rc.argtypes = [C.c_void_p, C.c_int32]
    lib.pw_batch_band.argtypes = [C.c_void_p, C.c_int32, P(C.c_int32), P(C.c_int32), P(C.c_int32)]
    lib.pw_batch_pair_cells.argtypes = [C.c_void_p, C.c_int32]
    lib.pw_batch_pair_cells.restype = C.c_int64
    lib.pw_batch_cells.argtypes = [C.c_void_p]
    lib.pw_batch_cells.restype = C.c_int64
    lib.pw_batch_algorithmic_bytes.argtypes = [C.c_void_p]
    lib.pw_batch_algorithmic_bytes.restype = C.c_int64
    lib.pw_batch_score_type.argtypes = [C.c_void_p]
    lib.pw_batch_kernel_name.argtypes = [C.c_void_p]
    lib.pw_batch_kernel_name.restype = C.c_char_p
    lib.pw_batch_upload_arena.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.pw_batch_arena_device.argtypes = [C.c_void_p]
    lib.pw_batch_arena_device.restype = C.c_void_p
    lib.pw_arena_upload.argtypes = [C.c_int, C.c_void_p, C.c_uint64]
    lib.pw_arena_upload.restype = C.c_void_p
    lib.pw_arena_free.argtypes = [C.c_int, C.c_void_p]
    lib.pw_arena_free.restype = None
    lib.pw_batch_share_arena.argtypes = [C.c_void_p, C.c_void_p]
    lib.pw_host_alloc.argtypes = [C.c_uint64]
    lib.pw_host_alloc.restype = C.c_void_p
    lib.pw_host_free.argtypes = [C.c_void_p]
    lib.pw_host_free.restype = None
    lib.pw_batch_upload_arena_async.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.pw_batch_results_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pw_batch_transcripts_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pw_batch_solve.argtypes = [C.c_void_p, C.c_void_p]
    lib.pw_batch_traceback.argtypes = [C.c_void_p, C.c_void_p]
    lib.pw_batch_traceback_from.argtypes = [C.c_void_p, P(C.c_int32), C.c_void_p]
    lib.pw_batch_sync.argtypes = [C.c_void_p, C.c_void_p]
    lib.pw_batch_results_device.argtypes = [C.c_void_p]
    lib.pw_batch_results_device.restype = C.c_void_p
    lib.pw_batch_transcripts_device.argtypes = [C.c_void_p]
    lib.pw_batch_transcripts_device.restype = C.c_void_p
    lib.pw_batch_transcripts_bytes.argtypes = [C.c_void_p]
    lib.pw_batch_transcripts_bytes.restype = C.c_uint64
    lib.pw_batch_tx_slot.argtypes = [C.c_void_p, C.c_int32, P(C.c_uint64), P(C.c_int32)]
    lib.pw_batch_results.argtypes = [C.c_void_p, C.c_void_p]
    lib.pw_batch_transcripts.argtypes = [C.c_void_p, C.c_void_p]
    lib.pw_batch_scores.argtypes = [C.c_void_p, C.c_int32, P(C.c_double), C.c_int64]
    lib.pw_batch_table.argtypes = [C.c_void_p, C.c_int32, P(C.c_double), C.c_int64]
    lib.pw_batch_fill_ms.argtypes = [C.c_void_p]
    lib.pw_batch_fill_ms.restype = C.c_float
    lib.pw_batch_trace_ms.argtypes = [C.c_void_p]
    lib.pw_batch_trace_ms.restype = C.c_float
    lib.pw_batch_pack_transcripts.argtypes = [C.c_void_p, C.c_void_p]
    lib.pw_batch_packed_device.argtypes = [C.c_void_p]
    lib.pw_batch_packed_device.restype = C.c_void_p
    lib.pw_batch_packed_offsets_device.argtypes = [C.c_void_p]
    lib.pw_batch_packed_offsets_device.restype = C.c_void_p
    lib.pw_batch_packed_total_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pw_batch_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.pw_plan_only.argtypes = [P(pw_scoring), C.c_int32, P(pw_pair), C.c_uint64, C.c_uint32, C.c_char_p, C.c_int32, P(C.c_int32)]
    # include/pw_seeds.h
    lib.pw_seeds_create.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                    P(C.c_uint64), C.c_int, C.c_int]
    lib.pw_seeds_create.restype = C.c_void_p
    lib.pw_seeds_build.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    lib.pw_seeds_num_rows.argtypes = [C.c_void_p]
    lib.pw_seeds_num_rows.restype = C.c_int64
    lib.pw_seeds_is_self.argtypes = [C.c_void_p]
    lib.pw_seeds_rows_device.argtypes = [C.c_void_p]
    lib.pw_seeds_rows_device.restype = C.c_void_p
    lib.pw_seeds_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.pw_seeds_count.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_int, C.c_int32, C.c_int32]
    lib.pw_seeds_count.restype = C.c_int64
    lib.pw_seeds_kmers.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    lib.pw_seeds_kmers.restype = C.c_int64
    lib.pw_seeds_band_neighbours.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
    lib.pw_seeds_graph_build.argtypes = [C.c_void_p, C.c_double, C.c_double]
    lib.pw_seeds_graph_build.restype = C.c_int64
    lib.pw_seeds_graph_num_points.argtypes = [C.c_void_p]
    lib.pw_seeds_graph_num_points.restype = C.c_int64
    lib.pw_seeds_graph_points.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.pw_seeds_graph_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.pw_seeds_graph_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pw_seeds_graph_components.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pw_seeds_build_ms.argtypes = [C.c_void_p]
    lib.pw_seeds_build_ms.restype = C.c_double
    lib.pw_seeds_algorithmic_bytes.argtypes = [C.c_void_p]
    lib.pw_seeds_algorithmic_bytes.restype = C.c_int64
    lib.pw_seeds_destroy.argtypes = [C.c_void_p]
    lib.pw_seeds_destroy.restype = None
    lib.pw_seeds_last_error.restype = C.c_char_p
    # include/pw_overlap.h
    lib.pw_overlap_bands.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                     C.c_double, C.c_double, C.c_double, C.c_void_p]
    lib.pw_overlap_all_pairs.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                         C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
    lib.pw_overlap_last_ms.restype = C.c_double
    lib.pw_overlap_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def last_error():
    return (load().pw_last_error() or b'').decode('utf-8', 'replace')
