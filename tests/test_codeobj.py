"""Tripwires on the compiled gfx950 code objects (no GPU needed: hipcc cross-compiles, the LLVM tools read ELF files).

Round 2's one GPU fault came from hand-scheduled code: the strip kernel's FIFO loads land in accumulation registers
a0..a3 through inline asm the compiler does not track (biseqt_amd/csrc/pw_strip.hip, DevPS::fifo_load_async / wait_vm),
which is sound only while the compiler itself never touches AGPRs in that kernel.  These tests read the kernel metadata
and the disassembly of the built objects and fail when that invariant -- or the register budgets DESIGN.md states for the
packed fill kernels (no scratch, the VGPR ceilings of their occupancy bounds) -- stops holding after a compiler, flag or
code change.  build.py runs the strip check on every build of pw_strip.o as well."""
import os
import subprocess
import tempfile

import pytest

from biseqt_amd.csrc import build as B
from biseqt_amd.csrc import codeobj

OBJ = B.OBJ_DIR


def _md(obj):
    path = os.path.join(OBJ, obj)
    if not os.path.exists(path):
        B.build()
    return {n.replace('void ', '').replace('pw::', '').split('(')[0]: k for n, k in codeobj.kernel_metadata(path).items()}


def test_strip_kernel_uses_exactly_its_four_accumulation_registers():
    path = os.path.join(OBJ, 'pw_strip.o')
    if not os.path.exists(path):
        B.build()
    assert codeobj.strip_kernel_violations(path) == []
    md = _md('pw_strip.o')
    for name in ('k_fill_strip<true, true>', 'k_fill_strip<true, false>', 'k_fill_strip<false, true>', 'k_fill_strip<false, false>'):
        k = md[name]
        assert k['agpr_count'] == 4 and k['vgpr_spill_count'] == 0 and k['private_segment_fixed_size'] == 0, (name, k)
    # the asm sites themselves: two zeroing writes per slot, one load per slot and flavour, two reads per wait
    dis = codeobj.disassembly(path, 'k_fill_strip')
    assert len(dis) == 4
    for sym, lines in dis.items():
        loads = [l for l in lines if l.startswith('global_load_dwordx2 a[')]
        assert loads and all((' sc1' in l) or (' nt' in l) for l in loads), sym
        assert any(l.startswith('v_accvgpr_read_b32') for l in lines), sym


def test_strip_checker_reports_foreign_agpr_use(monkeypatch):
    """The detector itself: an a4, an MFMA-style AGPR operand, an AGPR spill or a changed count must all be reported."""
    good_md = {'void pw::k_fill_strip<%s, %s>(pw::StripParams)' % (t, b): dict(agpr_count=4, vgpr_spill_count=0, private_segment_fixed_size=0)
               for t in ('true', 'false') for b in ('true', 'false')}
    good = ['v_accvgpr_write_b32 a0, 0', 'v_accvgpr_write_b32 a1, 0', 'global_load_dwordx2 a[0:1], v[2:3], off nt',
            'global_load_dwordx2 a[2:3], v[10:11], off sc1', 's_waitcnt vmcnt(1)', 'v_accvgpr_read_b32 v7, a3',
            'v_add_u32_e32 v1, v2, v3', 's_and_b64 s[0:1], s[2:3], exec']
    monkeypatch.setattr(codeobj, 'kernel_metadata', lambda p: good_md)
    monkeypatch.setattr(codeobj, 'disassembly', lambda p, s=None: {'k_fill_stripILb1E': list(good)})
    assert codeobj.strip_kernel_violations('x.o') == []
    for bad_line in ('v_accvgpr_write_b32 a4, v9', 'v_accvgpr_write_b32 a1, v9', 'global_load_dwordx2 a[4:5], v[2:3], off nt',
                     'global_load_dwordx2 a[0:1], v[2:3], off', 'v_accvgpr_read_b32 v7, a12',
                     'v_mfma_f32_32x32x8_f16 a[0:15], v[0:1], v[2:3], a[0:15]', 'v_accvgpr_mov_b32 a2, a3'):
        monkeypatch.setattr(codeobj, 'disassembly', lambda p, s=None, b=bad_line: {'k_fill_stripILb1E': good + [b]})
        assert len(codeobj.strip_kernel_violations('x.o')) == 1, bad_line
    monkeypatch.setattr(codeobj, 'disassembly', lambda p, s=None: {'k_fill_stripILb1E': list(good)})
    for change in (dict(agpr_count=6), dict(agpr_count=0), dict(vgpr_spill_count=3), dict(private_segment_fixed_size=16)):
        md = {n: dict(k) for n, k in good_md.items()}
        md['void pw::k_fill_strip<true, false>(pw::StripParams)'].update(change)
        monkeypatch.setattr(codeobj, 'kernel_metadata', lambda p, m=md: m)
        assert len(codeobj.strip_kernel_violations('x.o')) == 1, change


def test_strip_checker_reports_a_compiler_wait_inside_the_step_code(monkeypatch):
    """Round 3: a value loaded at the start of a strip whose first use sat in the steady loop put the compiler's wait for it
    (s_waitcnt vmcnt(0): a drain of the wavefront's own stores) into the loop -- 9 % of config 3.  The checker must tell such a
    wait from the hand-over's (asm: wait + v_accvgpr_read) and the polls' (asm: load + wait) and from waits outside the step code."""
    md = {'void pw::k_fill_strip<%s, %s>(pw::StripParams)' % (t, b): dict(agpr_count=4, vgpr_spill_count=0, private_segment_fixed_size=0)
          for t in ('true', 'false') for b in ('true', 'false')}
    monkeypatch.setattr(codeobj, 'kernel_metadata', lambda p: md)
    step = ['v_add_u32_e32 v1, v2, v3', 'v_max3_i32 v4, v5, v6, v7', 'v_cmp_eq_u32_e32 vcc, v1, v4'] * 6
    setup = ['s_load_dwordx2 s[0:1], s[2:3], 0x0', 'global_load_ubyte v9, v[10:11], off', 's_waitcnt vmcnt(0)', 'v_mov_b32_e32 v1, v9']
    handover = ['s_waitcnt vmcnt(1)', 'v_accvgpr_read_b32 v7, a3', 'v_accvgpr_read_b32 v8, a2']
    poll = ['global_load_dwordx2 v[2:3], v[4:5], off nt', 's_waitcnt vmcnt(0)']
    good = setup + ['v_mov_b32_e32 v0, 0'] * 200 + step + handover + step + poll + step
    monkeypatch.setattr(codeobj, 'disassembly', lambda p, s=None: {'k_fill_stripILb1ELb1E': list(good)})
    assert codeobj.strip_kernel_violations('x.o') == []
    bad = setup + ['v_mov_b32_e32 v0, 0'] * 200 + step + ['s_waitcnt vmcnt(0)', 'v_perm_b32 v1, v2, v2, v3'] + step
    monkeypatch.setattr(codeobj, 'disassembly', lambda p, s=None: {'k_fill_stripILb1ELb1E': list(bad)})
    out = codeobj.strip_kernel_violations('x.o')
    assert len(out) == 1 and 'compiler-inserted' in out[0], out


# (512 VGPRs per SIMD lane, allocated in granules of 8: n wavefronts per SIMD fit when each takes at most this many)
def _vgpr_ceiling(waves):
    return 512 // waves // 8 * 8


def test_packed_kernels_hold_their_occupancy_without_scratch():
    """The instantiations build.py holds to an occupancy (FILL16_WAVES: BK = 8 local rules at 5 wavefronts per SIMD, BK = 16
    at 3) must fit it WITHOUT scratch -- a spilling schedule is slower than the default one -- and use no AGPRs (on gfx950
    they come out of the same 512-register budget)."""
    for (bk, rule, mat), (occ, occ_seg) in sorted(B.FILL16_WAVES.items()):
        md = _md('pw_fill16_bk%d_r%d%s.o' % (bk, rule, '_mat' if mat else ''))
        for seg, waves in ((False, occ), (True, occ_seg)):
            if not waves:
                continue
            k = md['k_fill16<%d, %s, %d, %s>' % (bk, 'true' if seg else 'false', rule, 'true' if mat else 'false')]
            assert k['private_segment_fixed_size'] == 0 and k['vgpr_spill_count'] == 0, (bk, rule, seg, k)
            assert k['agpr_count'] == 0, (bk, rule, seg, k)
            assert k['vgpr_count'] <= _vgpr_ceiling(waves), (bk, rule, seg, waves, k['vgpr_count'])
    # config 2's kernel by name: k_fill16<8, false, 3> ("x4") at 5 wavefronts per SIMD
    assert B.FILL16_WAVES[(8, 3, 0)][0] == 5 and B.FILL16_WAVES[(8, 0, 0)][0] == 5


def test_packed_kernels_up_to_28_diagonals_per_lane_never_touch_scratch():
    """One pair per wavefront (the throughput layout): every rule, 4 .. 28 diagonals per lane, compiles without scratch."""
    for bk in B.PACKED_BKS:
        if bk > 28:
            continue
        for rule, mat in [(r, 0) for r in B.PACKED_RULES] + [(r, 1) for r in B.PACKED_MAT_RULES]:
            md = _md('pw_fill16_bk%d_r%d%s.o' % (bk, rule, '_mat' if mat else ''))
            k = md['k_fill16<%d, false, %d, %s>' % (bk, rule, 'true' if mat else 'false')]
            assert k['private_segment_fixed_size'] == 0 and k['vgpr_spill_count'] == 0, (bk, rule, k)


def test_an_occupancy_bound_that_spills_is_caught():
    """6 wavefronts per SIMD make the BK = 8 kernel spill.  Compile that instantiation with the bound bumped and
    check that the budget test above would see it (scratch or spill count > 0, or more VGPRs than the bound allows)."""
    jobs = {os.path.basename(o): c for o, c, _ in B._jobs()}
    cmd = list(jobs['pw_fill16_bk8_r3.o'])
    with tempfile.TemporaryDirectory(prefix='pwocc_') as wd:
        out = os.path.join(wd, 'pw_fill16_bk8_r3.o')
        cmd = [('-DPW_FILL16_WAVES=6' if a.startswith('-DPW_FILL16_WAVES=') else a) for a in cmd]
        cmd[cmd.index('-o') + 1] = out
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        md = {n: k for n, k in codeobj.kernel_metadata(out).items() if 'k_fill16<8, false, 3, false>' in n}
    (k,) = md.values()
    assert k['private_segment_fixed_size'] > 0 or k['vgpr_spill_count'] > 0 or k['vgpr_count'] > _vgpr_ceiling(6), k


def test_build_command_lines_are_part_of_the_staleness_check(tmp_path):
    obj = tmp_path / 'x.o'
    obj.write_bytes(b'')
    cmd = ['hipcc', '-DPW_FILL16_WAVES=5', '-c', 'x.hip']
    assert B._stale(str(obj), [], cmd)                      # no record of the command that built it
    (tmp_path / 'x.o.cmd').write_text(B._cmd_text(cmd))
    assert not B._stale(str(obj), [], cmd)
    assert B._stale(str(obj), [], ['hipcc', '-DPW_FILL16_WAVES=4', '-c', 'x.hip'])
