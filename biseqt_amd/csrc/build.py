"""Build biseqt_amd/pwlib/pwlib.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m biseqt_amd.csrc.build [--force]

One object per (score type, diagonals-per-lane) fill translation unit plus the traceback unit and the
host API, compiled in parallel, linked into one shared object that exports the four drop-in functions
of include/pwlib.h and the batch API of include/pw_batch.h.  The header is copied next to the library
(biseqt_amd/pwlib/pwlib.h) the way the reference keeps biseqt/pwlib/pwlib.{h,so} side by side.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT_DIR = os.path.join(os.path.dirname(HERE), 'pwlib')
OBJ_DIR = os.path.join(HERE, os.environ.get('PW_OBJ_DIR', '_build'))
SO = os.path.join(OUT_DIR, os.environ.get('PW_SO_NAME', 'pwlib.so'))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'
# -ffp-contract=off: the f64 path must add exactly as the reference does, (H + ge) + go, no FMA
COMMON = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wall',
          '-Wno-unused-function'] + os.environ.get('PW_EXTRA_CXXFLAGS', '').split()
BKS = (2, 4, 8, 16, 32)
PACKED_BKS = (4, 8, 12, 16, 20, 24, 28, 32)
PACKED_RULES = (0, 1, 2, 3, 4, 5)      # pw_wave.h, WaveFill16: local, overlap, global, local x4, END_ANCHORED, START_ANCHORED
PACKED_MAT_RULES = (0, 1, 2, 3)        # ... with a substitution matrix of up to 4 x 4 letters
# Wavefronts per SIMD the packed kernels are held to, (bk, rule, matrix) -> (one pair per wavefront, lane-packed); absent / 0 =
# the compiler's default.  The max-ilp schedule spends registers freely; a bound makes it fit fewer.  Only the entries
# measured faster are listed (tests/micro/ab_occupancy.sh): BK = 8 local rules 157 -> 87 VGPRs, config 2 fill 1-2 % and
# the pipelined step 3 % faster; BK = 16 local rules 177 -> 165 VGPRs, 2.8 % on an 801-diagonal band.  Every other
# instantiation compiled without a spill at one more wavefront per SIMD too, and ran the same within 2 % -- or slower: the
# lane-packed BK = 4 body at 4 per SIMD (169 -> 109 VGPRs) lost 24-35 % on 20 000 pairs with a 21-diagonal band.
FILL16_WAVES = {      # (bk, rule, matrix)
    (8, 0, 0): (5, 0), (8, 3, 0): (5, 0),
    (16, 0, 0): (3, 0), (16, 3, 0): (3, 0),
    # the matrix form (round 3): 206 VGPRs left alone, 165 at 3 per SIMD without a spill (4 spills): config 2's shape with a
    # 4 x 4 matrix 3.65-3.71 ms against 3.78-3.80 (A/B of two builds on one box) -- level with the match / mismatch form
    (8, 0, 1): (3, 0), (8, 3, 1): (3, 0),
}
# A/B builds: PW_FILL16_WAVES_OVERRIDE="8,3,1=3,0;8,0,1=3,0" replaces / adds entries (bk,rule,matrix=one-pair,lane-packed)
for _e in filter(None, os.environ.get('PW_FILL16_WAVES_OVERRIDE', '').split(';')):
    _k, _v = _e.split('=')
    FILL16_WAVES[tuple(int(x) for x in _k.split(','))] = tuple(int(x) for x in _v.split(','))
# A/B builds: PW_FILL16_UNR_OVERRIDE="8,3,1=2" sets the unroll depth of one instantiation (default: pw_wave.h, UNR).  Measured
# for config 2's kernel (8, 3, 1): fully unrolled (4) at 3 wavefronts per SIMD, 166 VGPRs: fill 3.40 ms; 2 at 4 per SIMD (104
# VGPRs) 3.54-3.60; 1 at 5 per SIMD (94 VGPRs) 3.67-3.70; 2 at 3 per SIMD 3.61 -- the instructions the full unrolling saves
# (the letter-window shifts become register renames) are worth more than the wavefronts it costs.
FILL16_UNR = {}
for _e in filter(None, os.environ.get('PW_FILL16_UNR_OVERRIDE', '').split(';')):
    _k, _v = _e.split('=')
    FILL16_UNR[tuple(int(x) for x in _k.split(','))] = int(_v)
TYPES = (('i32', 'int32_t'), ('f64', 'double'))
# The f64 kernels never see a NaN (scores are validated finite, sums stay far from overflow): telling the compiler so lets
# v_max_f64 take values that crossed lanes as bit patterns without a canonicalising v_max_f64 x, x in front.  It licenses no
# reassociation and no contraction (-ffp-contract=off stays): every add is still the reference's add.
F64_FLAGS = {'f64': ['-fno-honor-nans'] + os.environ.get('PW_F64_EXTRA_CXXFLAGS', '').split()}     # (the extra flags: A/B builds)
# Scheduling strategy per (score type, diagonals per lane) of the wavefront kernels, where an A/B on the GPU decided: the f64
# kernel with 8 diagonals per lane (config 2's shape: 7.75 -> 7.50 ms with max-ilp; 255 VGPRs instead of 243, still two
# wavefronts per SIMD).  The narrower f64 bodies would drop a wavefront per SIMD (BK = 4: 127 -> 154 VGPRs), the 32-bit
# BK = 8 body too (159 -> 172): they keep the default.
FILL_EXTRA = {('f64', 8): ['-mllvm', '-amdgpu-sched-strategy=max-ilp']}
# objects whose kernels get a fingerprint in biseqt_amd/pwlib/kernel_hashes.json (config 2's and config 3's fill kernels)
HASHED_OBJECTS = ('pw_fill16_bk8_r3_mat.o', 'pw_fill16_bk8_r3.o', 'pw_fill16_bk8_r0.o', 'pw_fill_i32_bk8.o', 'pw_fill_f64_bk8.o', 'pw_strip.o')
HEADERS = ['pw_types.h', 'pw_wave.h', 'pw_strip.h', 'pw_plan.h', 'pw_launch.h', 'pw_device.h']


def _jobs():
    jobs = []
    for tn, t in TYPES:
        for bk in BKS:
            obj = os.path.join(OBJ_DIR, 'pw_fill_%s_bk%d.o' % (tn, bk))
            cmd = [HIPCC] + COMMON + F64_FLAGS.get(tn, []) + FILL_EXTRA.get((tn, bk), []) + ['-DPW_T=' + t, '-DPW_TNAME=' + tn, '-DPW_BK=%d' % bk, '-c',
                                      os.path.join(HERE, 'pw_fill_tu.hip'), '-o', obj]
            jobs.append((obj, cmd, [os.path.join(HERE, 'pw_fill_tu.hip')]))
    for tn, t in TYPES:
        obj = os.path.join(OBJ_DIR, 'pw_fill_mw_%s.o' % tn)
        cmd = [HIPCC] + COMMON + F64_FLAGS.get(tn, []) + ['-DPW_T=' + t, '-DPW_TNAME=' + tn, '-c', os.path.join(HERE, 'pw_fill_mw_tu.hip'), '-o', obj]
        jobs.append((obj, cmd, [os.path.join(HERE, 'pw_fill_mw_tu.hip')]))
    for tn, t in TYPES:
        obj = os.path.join(OBJ_DIR, 'pw_fill_tile_%s.o' % tn)
        cmd = [HIPCC] + COMMON + F64_FLAGS.get(tn, []) + ['-DPW_T=' + t, '-DPW_TNAME=' + tn, '-c', os.path.join(HERE, 'pw_fill_tile_tu.hip'), '-o', obj]
        jobs.append((obj, cmd, [os.path.join(HERE, 'pw_fill_tile_tu.hip')]))
    for bk, rule, mat in [(bk, r, 0) for r in PACKED_RULES for bk in PACKED_BKS] + [(bk, r, 1) for r in PACKED_MAT_RULES for bk in PACKED_BKS]:
        obj = os.path.join(OBJ_DIR, 'pw_fill16_bk%d_r%d%s.o' % (bk, rule, '_mat' if mat else ''))
        # max-ilp scheduling: dependent VOP3P ops need a wait state between them; the default (occupancy first)
        # schedule leaves ~15% of the issue slots of the packed kernel to s_nop, this one none (measured)
        occ, occ_seg = FILL16_WAVES.get((bk, rule, mat), (0, 0)) if os.environ.get('PW_FILL16_OCCUPANCY', '1') != '0' else (0, 0)
        cmd = [HIPCC] + COMMON + ['-mllvm', '-amdgpu-sched-strategy=max-ilp', '-DPW_BK=%d' % bk, '-DPW_RULE=%d' % rule, '-DPW_MAT=%d' % mat,
                                  '-DPW_FILL16_WAVES=%d' % occ, '-DPW_FILL16_WAVES_SEG=%d' % occ_seg] + \
              (['-DPW_FILL16_UNR=%d' % FILL16_UNR[(bk, rule, mat)]] if (bk, rule, mat) in FILL16_UNR else []) + ['-c',
               os.path.join(HERE, 'pw_fill16_tu.hip'), '-o', obj]
        jobs.append((obj, cmd, [os.path.join(HERE, 'pw_fill16_tu.hip')]))
    obj = os.path.join(OBJ_DIR, 'pw_trace.o')
    jobs.append((obj, [HIPCC] + COMMON + ['-c', os.path.join(HERE, 'pw_trace.hip'), '-o', obj],
                 [os.path.join(HERE, 'pw_trace.hip')]))
    obj = os.path.join(OBJ_DIR, 'pw_strip.o')
    # (PW_STRIP_CXXFLAGS: A/B builds of the strip kernel alone, e.g. -DPW_STRIP_LEAD=16)
    jobs.append((obj, [HIPCC] + COMMON + os.environ.get('PW_STRIP_CXXFLAGS', '').split() + ['-c', os.path.join(HERE, 'pw_strip.hip'), '-o', obj],
                 [os.path.join(HERE, 'pw_strip.hip')]))
    obj = os.path.join(OBJ_DIR, 'pw_seeds.o')
    jobs.append((obj, [HIPCC] + COMMON + ['-Wno-unused-parameter', '-c', os.path.join(HERE, 'pw_seeds.hip'), '-o', obj],
                 [os.path.join(HERE, 'pw_seeds.hip'), os.path.join(ROOT, 'include', 'pw_seeds.h')]))
    obj = os.path.join(OBJ_DIR, 'pw_overlap.o')
    jobs.append((obj, [HIPCC] + COMMON + ['-Wno-unused-parameter', '-c', os.path.join(HERE, 'pw_overlap.hip'), '-o', obj],
                 [os.path.join(HERE, 'pw_overlap.hip'), os.path.join(ROOT, 'include', 'pw_overlap.h')]))
    obj = os.path.join(OBJ_DIR, 'pwlib_api.o')
    jobs.append((obj, [HIPCC] + COMMON + ['-x', 'hip', '-c', os.path.join(HERE, 'pwlib_api.cpp'), '-o', obj],
                 [os.path.join(HERE, 'pwlib_api.cpp'), os.path.join(HERE, 'pw_model.h'), os.path.join(ROOT, 'include', 'pwlib.h'),
                  os.path.join(ROOT, 'include', 'pw_batch.h')]))
    return jobs


def _cmd_text(cmd):
    return ' '.join(cmd) + '\n'


def _stale(target, deps, cmd=None):
    """Out of date when a dependency is newer OR the command line differs from the one that built the object (kept beside
    it as `<object>.cmd`): the build-time knobs -- PW_FILL16_OCCUPANCY, FILL16_WAVES, PW_EXTRA_CXXFLAGS, PW_TILE_* --
    change the command but no file, and an A/B of two settings in one PW_OBJ_DIR must not compare two identical binaries."""
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    if cmd is not None:
        try:
            with open(target + '.cmd') as f:
                return f.read() != _cmd_text(cmd)
        except OSError:
            return True
    return False


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(OUT_DIR, exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    jobs = _jobs()
    todo = [(o, c) for (o, c, deps) in jobs if force or _stale(o, deps + hdrs, c)]

    def run(job):
        obj, cmd = job
        if os.path.exists(obj + '.cmd'):
            os.remove(obj + '.cmd')
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
        if r.returncode == 0:
            with open(obj + '.cmd', 'w') as f:
                f.write(_cmd_text(cmd))
        return obj, r.returncode, r.stdout

    workers = max(1, min(len(todo), (os.cpu_count() or 4)))
    if todo:
        if verbose:
            print('[build] compiling %d objects for %s with %d workers' % (len(todo), ARCH, workers))
        with concurrent.futures.ThreadPoolExecutor(workers) as ex:
            for obj, rc, out in ex.map(run, todo):
                if rc != 0:
                    raise RuntimeError('hipcc failed for %s:\n%s' % (obj, out))
                if verbose and out.strip():
                    print(out)
    objs = [o for (o, _, _) in jobs]
    # The strip kernel's hand-over loads land in accumulation registers a0..a3 behind the compiler's back (pw_strip.hip):
    # refuse a build in which the compiler allocated AGPRs of its own there, or spilled (codeobj.strip_kernel_violations)
    strip_obj = os.path.join(OBJ_DIR, 'pw_strip.o')
    if force or any(o == strip_obj for o, _ in todo) or not os.path.exists(SO):
        from . import codeobj
        bad = codeobj.strip_kernel_violations(strip_obj)
        if bad:
            os.remove(strip_obj)
            raise RuntimeError('pw_strip.o breaks the invariants its inline asm relies on:\n  ' + '\n  '.join(bad))
    if force or todo or _stale(SO, objs):
        cmd = [HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC'] + objs + ['-o', SO]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stdout)
        if verbose:
            print('[build] linked', SO)
    shutil.copyfile(os.path.join(ROOT, 'include', 'pwlib.h'), os.path.join(OUT_DIR, 'pwlib.h'))
    # (an A/B build under another PW_SO_NAME keeps its fingerprints beside its own library)
    hashes = os.path.join(OUT_DIR, 'kernel_hashes.json' if os.path.basename(SO) == 'pwlib.so' else os.path.basename(SO) + '.hashes.json')
    if force or todo or not os.path.exists(hashes):
        # Fingerprints of the kernels whose rocprofv3 counters are quoted from committed files (profiles/pmc_kernel.json):
        # bench.py reports those counters only while the kernel it runs still has the code they were collected on.
        from . import codeobj
        import json
        fp = {}
        for obj in HASHED_OBJECTS:
            fp.update(codeobj.kernel_fingerprints(os.path.join(OBJ_DIR, obj)))
        with open(hashes, 'w') as f:
            json.dump(fp, f, indent=1, sort_keys=True)
    return SO


if __name__ == '__main__':
    build(force='--force' in sys.argv)
