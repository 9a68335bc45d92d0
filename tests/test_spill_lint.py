"""The build's static check of register spills (lightweaver_amd/spill_lint.py) on small listings: the pattern that broke
lanesweep_kernel in round 5 is reported, the legitimate ones around it are not; and the listings of the library that is in the
tree (build/, written by the build) are clean."""
import glob
import os

import pytest

from lightweaver_amd import spill_lint

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lint(body):
    lines = ('\t.type\tk,@function\nk:\n' + body + '\n\ts_endpgm\n.Lfunc_end0:\n').split('\n')
    nsp, bad = spill_lint.lint_kernel('k', lines, 1)
    return nsp, sorted({slot for slot, *_ in bad})


def test_store_in_front_of_the_exec_restore_of_a_join_block_is_reported():
    # (the listing of round 5: `if (c.blk == 0) { ... }` -- three lanes -- and the allocator's spill at the top of the join)
    nsp, bad = lint('''
	v_sub_u32_e32 v246, s4, v138
	s_and_saveexec_b64 s[4:5], s[2:3]
	s_xor_b64 s[4:5], exec, s[4:5]
; %bb.174:
	s_load_dwordx2 s[6:7], s[74:75], 0x24
; %bb.175:
	s_or_saveexec_b64 s[4:5], s[4:5]
	v_mov_b32_e32 v179, s7
	s_xor_b64 exec, exec, s[4:5]
	s_cbranch_execz .LBB0_181
; %bb.176:
	ds_write2_b64 v1, v[8:9], v[10:11] offset1:1
.LBB0_181:
	scratch_store_dword off, v246, off offset:24 ; 4-byte Folded Spill
	s_or_b64 exec, exec, s[4:5]
	v_add_u32_e32 v246, -8, v1
	scratch_load_dword v246, off, off offset:24 ; 4-byte Folded Reload
''')
    assert nsp == 1 and bad == [24]


def test_store_after_the_exec_restore_is_clean():
    nsp, bad = lint('''
	s_and_saveexec_b64 s[4:5], s[2:3]
	s_cbranch_execz .LBB0_2
; %bb.1:
	ds_write2_b64 v1, v[8:9], v[10:11] offset1:1
.LBB0_2:
	s_or_b64 exec, exec, s[4:5]
	scratch_store_dword off, v246, off offset:24 ; 4-byte Folded Spill
	v_add_u32_e32 v246, -8, v1
	scratch_load_dword v246, off, off offset:24 ; 4-byte Folded Reload
''')
    assert nsp == 1 and bad == []


def test_all_lanes_are_on_between_the_two_sides_of_an_if_else():
    # s_or_saveexec ... s_xor exec: the spill in between runs with every lane of the enclosing region
    nsp, bad = lint('''
	s_and_saveexec_b64 s[4:5], s[0:1]
	s_xor_b64 s[4:5], exec, s[4:5]
	s_or_saveexec_b64 s[10:11], s[4:5]
	scratch_store_dword off, v0, off offset:16 ; 4-byte Folded Spill
	s_xor_b64 exec, exec, s[10:11]
	s_cbranch_execz .LBB0_95
; %bb.94:
	v_mov_b32_e32 v0, 1
	scratch_store_dword off, v0, off offset:16 ; 4-byte Folded Spill
.LBB0_95:
	s_or_b64 exec, exec, s[10:11]
	scratch_load_dword v3, off, off offset:16 ; 4-byte Folded Reload
''')
    # (the second store updates the slot for the lanes of the else side only -- `if (c) x = 1` --, the first one wrote all)
    assert nsp == 2 and bad == []


def test_store_and_reload_inside_the_same_region_are_clean_and_a_loop_back_edge_closes_its_region():
    nsp, bad = lint('''
	s_branch .LBB0_96
.LBB0_95:
	s_or_b64 exec, exec, s[10:11]
	s_cmp_lt_i32 s1, s14
	s_cbranch_scc0 .LBB0_98
.LBB0_96:
	s_and_saveexec_b64 s[10:11], s[12:13]
	s_cbranch_execz .LBB0_95
; %bb.97:
	scratch_store_dword off, v5, off offset:8 ; 4-byte Folded Spill
	v_mov_b32_e32 v5, 0
	scratch_load_dword v5, off, off offset:8 ; 4-byte Folded Reload
	s_branch .LBB0_95
.LBB0_98:
	scratch_store_dword off, v230, off offset:16 ; 4-byte Folded Spill
	v_mov_b32_e32 v230, 0
	scratch_load_dword v230, off, off offset:16 ; 4-byte Folded Reload
''')
    assert nsp == 2 and bad == []


def test_store_with_lanes_off_and_nothing_before_it_is_reported_across_blocks():
    nsp, bad = lint('''
	s_and_saveexec_b64 s[4:5], vcc
	s_cbranch_execz .LBB0_2
; %bb.1:
	scratch_store_dwordx2 off, v[4:5], off ; 8-byte Folded Spill
.LBB0_2:
	s_or_b64 exec, exec, s[4:5]
	s_cbranch_scc1 .LBB0_4
; %bb.3:
	scratch_load_dwordx2 v[12:13], off, off ; 8-byte Folded Reload
.LBB0_4:
''')
    assert nsp == 2 and bad == [0, 4]


def test_the_listings_of_the_library_in_the_tree_are_clean():
    """The build refuses a library with a reported kernel (lightweaver_amd.build.check_spills); the hot kernel's listing of
    the last build is checked again here (the others take a minute of CPU: the build's own report covers them)."""
    lst = glob.glob(os.path.join(ROOT, 'build', 'liblwhip.so.obj', 'lwhip_lanesweep-hip-amdgcn-amd-amdhsa-gfx950.s'))
    if not lst:
        pytest.skip('no listing: the library was built elsewhere')
    lib = os.path.join(ROOT, 'lightweaver_amd', 'liblwhip.so')
    if os.path.getmtime(lst[0]) + 3600 < os.path.getmtime(lib):
        pytest.skip('the listing is older than the library')
    report = []
    assert spill_lint.lint_file(lst[0], quiet=True, report=report) == 0
    assert any('lanesweep_kernel' in k for k, _, _ in report) or report == []
    txt = os.path.join(ROOT, 'build', 'spill_lint.txt')
    if os.path.exists(txt):
        assert all(l.rstrip().endswith(' 0 slot(s) reported') for l in open(txt) if l.strip())


def test_failed_check_is_not_bypassed_by_rebuilding(tmp_path):
    """a build that failed the check left its objects behind, the next call found them up to date, compiled
    nothing, checked nothing and linked.  The check now runs on every unit that is linked, with its verdict remembered next to
    the listing: a flagged listing fails check_spills again on the second call (from the stamp), a listing that changed is
    parsed again, and a missing listing is refused by the build (compile_one)."""
    from lightweaver_amd import build
    bad_listing = ('\t.type\tk,@function\nk:\n'
                   '\ts_and_saveexec_b64 s[4:5], vcc\n'
                   '\tscratch_store_dwordx2 off, v[12:13], off ; 8-byte Folded Spill\n'
                   '\ts_or_b64 exec, exec, s[4:5]\n'
                   '\tscratch_load_dwordx2 v[12:13], off, off ; 8-byte Folded Reload\n'
                   '\ts_endpgm\n.Lfunc_end0:\n')
    lst = tmp_path / 'unit-hip-amdgcn-amd-amdhsa-gfx950.s'
    lst.write_text(bad_listing)
    objdir = tmp_path / 'lib.obj'
    objdir.mkdir()
    first = build._lint_cached(str(lst))
    assert first[1] > 0
    with pytest.raises(RuntimeError, match='spill check'):
        build.check_spills(str(objdir), [first])
    assert os.path.exists(str(lst) + '.lint.json')
    second = build._lint_cached(str(lst))          # the "rebuild": nothing recompiled, the verdict still stands
    assert second[1] == first[1]
    with pytest.raises(RuntimeError, match='spill check'):
        build.check_spills(str(objdir), [second])
    lst.write_text('\t.type\tk,@function\nk:\n\ts_endpgm\n.Lfunc_end0:\n')   # recompiled clean: parsed again
    assert build._lint_cached(str(lst))[1] == 0
