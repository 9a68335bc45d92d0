"""The C-ABI boundary: struct layouts match include/lwhip.h, the library loads and exports every
declared symbol, and the product path fails loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from lightweaver_amd import _abi as abi
from lightweaver_amd import build, context

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'lwhip.h')

STRUCTS = [abi.lwhip_transition, abi.lwhip_atom, abi.lwhip_boundary, abi.lwhip_problem,
           abi.lwhip_options, abi.lwhip_rho_coeff, abi.lwhip_j_coeff, abi.lwhip_hprd, abi.lwhip_iter_result, abi.lwhip_prd_result, abi.lwhip_nr_args, abi.lwhip_intersection, abi.lwhip_grid2d]


def test_struct_layout_matches_header(tmp_path):
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void){']
    for st in STRUCTS:
        name = st.__name__
        lines.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for fname, _ in st._fields_:
            lines.append(f'printf("{name}.{fname} %zu\\n", offsetof({name}, {fname}));')
    lines.append('return 0;}')
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-std=c11', '-o', str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = dict(l.split() for l in out.strip().splitlines())
    for st in STRUCTS:
        assert int(got[st.__name__]) == C.sizeof(st), st.__name__
        for fname, _ in st._fields_:
            assert int(got[f'{st.__name__}.{fname}']) == getattr(st, fname).offset, (st.__name__, fname)


def test_header_enums_match_python():
    txt = open(HEADER).read()
    for name, val in [('LWHIP_ABI_VERSION', abi.ABI_VERSION)]:
        m = re.search(rf'#define {name} (\d+)', txt)
        assert m and int(m.group(1)) == val
    vals = dict((m.group(1), int(m.group(2))) for m in re.finditer(r'(LWHIP_\w+) = (\d+)[,\s]', txt))
    assert vals['LWHIP_BC_THERMALISED'] == abi.BC_THERMALISED
    assert vals['LWHIP_BC_CALLABLE'] == abi.BC_CALLABLE
    assert vals['LWHIP_FS_BEZIER3_1D'] == abi.FS_BEZIER3_1D
    assert vals['LWHIP_ERR_SINGULAR'] == abi.ERR_SINGULAR
    shifts = dict((m.group(1), 1 << int(m.group(2))) for m in re.finditer(r'(LWHIP_\w+)\s+= 1 << (\d+)', txt))
    for k in ['ATMOS', 'BACKGROUND', 'PROFILES', 'POPS', 'NSTAR', 'J', 'GAMMA', 'BC', 'RHOPRD', 'I',
              'RATES', 'DEPTHDATA', 'COLLISIONS']:
        assert shifts['LWHIP_' + k] == getattr(abi, k), k


def test_library_exports_every_declared_symbol(hip_lib):
    txt = open(HEADER).read()
    inline = set(re.findall(r'static inline [a-z0-9_ ]+?\b(lwhip_[a-z_A-Z0-9]+)\s*\(', txt))   # (header-only helpers: no symbol)
    declared = set(re.findall(r'\b(lwhip_[a-z_A-Z0-9]+)\s*\(', txt)) - inline
    bound = {name for name, _, _ in abi.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(hip_lib, name), name
    assert hip_lib.lwhip_abi_version() == abi.ABI_VERSION


def test_no_cpu_fallback_without_gpu(hip_lib):
    """Without a device the product path must fail loudly, never compute on the CPU."""
    if hip_lib.lwhip_device_count() > 0:
        pytest.skip('a GPU is present')
    from helpers import load_fixture
    prob, _ = load_fixture('falc_h_vel')
    with pytest.raises(context.LwHipError, match='no HIP device'):
        context.Context(prob)


def test_create_rejects_bad_descriptors(hip_lib):
    from helpers import load_fixture
    prob, _ = load_fixture('falc_h_vel')
    desc = prob.descriptor()
    h = C.c_void_p()
    desc.abiVersion = 99
    assert hip_lib.lwhip_create(C.byref(desc), None, C.byref(h)) == abi.ERR_INVALID
    assert b'ABI' in hip_lib.lwhip_last_error()
    desc = prob.descriptor()
    desc.formalSolver = 7
    assert hip_lib.lwhip_create(C.byref(desc), None, C.byref(h)) == abi.ERR_INVALID
    desc = prob.descriptor()
    desc.Nspace = 2
    assert hip_lib.lwhip_create(C.byref(desc), None, C.byref(h)) == abi.ERR_INVALID
    desc = prob.descriptor()
    desc.Nspace = 4096
    assert hip_lib.lwhip_create(C.byref(desc), None, C.byref(h)) == abi.ERR_UNSUPPORTED
    assert hip_lib.lwhip_create(None, None, C.byref(h)) == abi.ERR_INVALID


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(context.LwHipError, match='no CPU fallback'):
        context.load_library(str(tmp_path / 'nope.so'))
