"""Read the gfx950 code objects inside the objects build.py produces: kernel metadata (register counts, spills,
scratch) and disassembly.  Used by build.py to refuse a build whose hand-scheduled kernels lost their invariants, and by
tests/test_codeobj.py (runs without a GPU: hipcc cross-compiles, the LLVM tools only read ELF files).

The device code of a `hipcc -c` object sits in its `.hip_fatbin` section as an offload bundle; `llvm-objdump
--offloading` unbundles it next to the input, so the object is copied into a scratch directory first.
"""
import os
import re
import shutil
import subprocess
import tempfile

LLVM_BIN = os.environ.get('PW_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
ARCH = 'gfx950'


def _tool(name):
    return os.path.join(LLVM_BIN, name)


def extract(obj_path, workdir):
    """Unbundle the gfx950 code object of a host object into `workdir`; returns its path."""
    local = os.path.join(workdir, os.path.basename(obj_path))
    shutil.copyfile(obj_path, local)
    subprocess.run([_tool('llvm-objdump'), '--offloading', local], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                   check=True, cwd=workdir)
    for f in sorted(os.listdir(workdir)):
        if f.startswith(os.path.basename(obj_path) + '.') and f.endswith(ARCH):
            return os.path.join(workdir, f)
    raise RuntimeError('no %s code object in %s' % (ARCH, obj_path))


def _demangle(names):
    filt = _tool('llvm-cxxfilt') if os.path.exists(_tool('llvm-cxxfilt')) else (shutil.which('c++filt') or '')
    if not filt or not names:
        return {n: n for n in names}
    out = subprocess.run([filt], input='\n'.join(names), stdout=subprocess.PIPE, universal_newlines=True,
                         check=True).stdout.split('\n')
    return dict(zip(names, out))


def kernel_metadata(obj_path):
    """{demangled kernel name: the kernel's metadata map without the leading dots -- agpr_count, vgpr_count, sgpr_count,
    vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size (scratch bytes per lane), group_segment_fixed_size,
    ...} for every kernel in the object (the NT_AMDGPU_METADATA note, which is YAML)."""
    import yaml
    with tempfile.TemporaryDirectory(prefix='pwco_') as wd:
        co = extract(obj_path, wd)
        notes = subprocess.run([_tool('llvm-readelf'), '--notes', co], stdout=subprocess.PIPE, universal_newlines=True,
                               check=True).stdout
    body = notes[notes.index('---'):]
    if '\n...' in body:
        body = body[:body.index('\n...')]
    doc = yaml.safe_load(body)
    kernels = [{k.lstrip('.'): v for k, v in e.items() if k != '.args'} for e in (doc.get('amdhsa.kernels') or [])]
    dm = _demangle([k['name'] for k in kernels])
    return {dm[k['name']]: k for k in kernels}


def disassembly(obj_path, symbol_substr=None):
    """Disassembly text of the object's gfx950 code; with `symbol_substr`, only the functions whose (mangled) symbol
    contains it, as {symbol: [instruction lines]}."""
    with tempfile.TemporaryDirectory(prefix='pwco_') as wd:
        co = extract(obj_path, wd)
        text = subprocess.run([_tool('llvm-objdump'), '-d', '--no-show-raw-insn', co], stdout=subprocess.PIPE,
                              universal_newlines=True, check=True).stdout
    if symbol_substr is None:
        return text
    funcs, cur = {}, None
    for line in text.split('\n'):
        m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
        if m:
            cur = m.group(1) if symbol_substr in m.group(1) else None
            if cur is not None:
                funcs[cur] = []
            continue
        if cur is not None and line.strip():
            funcs[cur].append(line.strip())
    return funcs


def kernel_fingerprints(obj_path):
    """{demangled kernel name: sha256 of its instruction stream} for every kernel of the object.  Addresses are left out
    (another kernel of the same object growing must not change the print); branch targets stay as symbol-relative text."""
    import hashlib
    md = kernel_metadata(obj_path)
    by_symbol = {k['name']: n for n, k in md.items()}
    out = {}
    for sym, lines in disassembly(obj_path, '').items():
        if sym not in by_symbol:
            continue
        h = hashlib.sha256()
        for ln in lines:
            ins = re.sub(r'\s+', ' ', ln.split('//')[0]).strip()
            ins = re.sub(r'^[0-9a-f]+:\s*', '', ins)
            h.update(ins.encode() + b'\n')
        out[by_symbol[sym]] = h.hexdigest()
    return out


# ---- the invariants of the hand-scheduled strip kernel (pw_strip.hip, DevPS::fifo_load_async / wait_vm) --------------
# The FIFO hand-over loads land in accumulation registers a0..a3 through asm statements the compiler does not track.
# That is sound only while the compiler itself never allocates AGPRs in k_fill_strip (no AGPR spills of VGPRs, no MFMA):
# exactly four AGPRs, used by exactly these instruction forms.
_AGPR_OK = (re.compile(r'^v_accvgpr_write_b32 a[0-3], 0\b'),
            re.compile(r'^global_load_dwordx2 a\[(0:1|2:3)\], v\[\d+:\d+\], off( sc1| nt)\b'),
            re.compile(r'^v_accvgpr_read_b32 v\d+, a[0-3]\b'))


def strip_kernel_violations(obj_path):
    """Empty list when the four k_fill_strip<TRACK, BROW> keep the invariants the asm relies on; otherwise what broke."""
    bad = []
    md = kernel_metadata(obj_path)
    strip = {n: k for n, k in md.items() if 'k_fill_strip' in n}
    if len(strip) != 4:
        bad.append('expected 4 k_fill_strip instantiations, found %d' % len(strip))
    for n, k in strip.items():
        if k.get('agpr_count') != 4:
            bad.append('%s: agpr_count %s != 4 (the compiler allocated accumulation registers of its own)' % (n, k.get('agpr_count')))
        if k.get('vgpr_spill_count', 0) != 0 or k.get('private_segment_fixed_size', 0) != 0:
            bad.append('%s: spills VGPRs (vgpr_spill_count %s, scratch %s B): spill code may use AGPRs'
                       % (n, k.get('vgpr_spill_count'), k.get('private_segment_fixed_size')))
    for sym, lines in disassembly(obj_path, 'k_fill_strip').items():
        ins_list = [re.sub(r'\s+', ' ', ln.split('//')[0]).strip() for ln in lines]
        for ins in ins_list:
            if re.search(r'\ba(\d+|\[)', ins) and not any(p.match(ins) for p in _AGPR_OK):
                bad.append('%s: unexpected AGPR use: %s' % (sym, ins))
        # No wait of the COMPILER's for vector memory inside the step code: on this target loads and stores share vmcnt, so such
        # a wait drains the wavefront's mask / FIFO stores -- once per block if it sits in the steady loop (round 3: a value
        # loaded at the start of a strip whose first use was in the loop put one there: config 3 17.3 -> 18.9 ms).  The waits
        # that belong there are the hand-over's (asm: s_waitcnt + v_accvgpr_read) and the polls' (asm: load + s_waitcnt).
        steps = [i for i, ins in enumerate(ins_list) if ins.startswith('v_max3_i32')]
        for i, ins in enumerate(ins_list):
            if not (ins.startswith('s_waitcnt') and 'vmcnt' in ins):
                continue
            nxt = ins_list[i + 1] if i + 1 < len(ins_list) else ''
            prv = ins_list[i - 1] if i > 0 else ''
            if nxt.startswith('v_accvgpr_read_b32') or prv.startswith('global_load') or prv.startswith('global_atomic'):
                continue
            before = sum(1 for j in steps if i - 120 <= j < i)
            after = sum(1 for j in steps if i < j <= i + 120)
            if before >= 3 and after >= 3:
                bad.append('%s: a compiler-inserted "%s" inside the step code (instruction %d): a loaded value is first used '
                           'there -- make it opaque where it is loaded (P::in_vgpr)' % (sym, ins, i))
    return bad
