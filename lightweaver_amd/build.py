"""Build helpers: compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m lightweaver_amd.build          # liblwhip.so (+ plugin when the Lightweaver headers exist)
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'liblwhip.so')
PLUGIN = os.path.join(PKG, 'LwHipIterScheme.so')
REF_SRC = os.environ.get('LW_REFERENCE_SOURCE', '/root/reference/Source')

HIP_SOURCES = ['lwhip_api.hip', 'lwhip_tables.hip', 'lwhip_state.hip', 'lwhip_api2d.hip', 'lwhip_batch.hip', 'lwhip_api_prd.hip',
               'lwhip_kernels.hip', 'lwhip_raymarch.hip', 'lwhip_lanesweep.hip', 'lwhip_voigt.hip', 'lwhip_prd.hip', 'lwhip_pops.hip', 'lwhip_fs2d.hip', 'lwhip_2d.hip', 'lwhip_geom2d.hip']
HIP_HEADERS = ['lwhip_internal.h', 'lwhip_device.h', 'lwhip_host.h', os.path.join('..', '..', 'include', 'lwhip.h')]


def _hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: the HIP library cannot be built')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False, defines=(), out=None, extra_flags=()):
    """hipcc --offload-arch=gfx950 -> lightweaver_amd/liblwhip.so.  Translation units are compiled
    concurrently (objects under build/, git-ignored), then linked."""
    from concurrent.futures import ThreadPoolExecutor
    lib = out or LIB
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.normpath(os.path.join(CSRC, h)) for h in HIP_HEADERS]
    if not force and not _stale(lib, deps):
        return lib
    objdir = os.path.join(ROOT, 'build', os.path.basename(lib) + '.obj')
    os.makedirs(objdir, exist_ok=True)
    # (-save-temps=obj: the compiler leaves the device listing <unit>-hip-amdgcn-amd-amdhsa-gfx950.s next to the object -- the
    # listing of exactly the code that ships, for the spill check below)
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-save-temps=obj'] + [f'-D{d}' for d in defines] + list(extra_flags)
    from concurrent.futures import ProcessPoolExecutor
    lints = []   # (the spill check of a unit starts when its listing is there, next to the other units' compilations)

    def compile_one(src):
        stem = os.path.splitext(os.path.basename(src))[0]
        obj = os.path.join(objdir, stem + '.o')
        lst = os.path.join(objdir, stem + '-hip-amdgcn-amd-amdhsa-gfx950.s')
        if force or _stale(obj, [src] + deps[len(srcs):]) or not os.path.exists(lst):
            cmd = [_hipcc()] + flags + ['-c', '-o', obj, src]
            if verbose:
                print(' '.join(cmd))
            subprocess.run(cmd, check=True, cwd=CSRC)
        # EVERY unit that goes into the link is checked, not only the ones compiled in this call: an object left behind by a
        # build that failed the check (or was interrupted before it) is up to date for make-style staleness and would
        # otherwise be linked unchecked.  A unit's verdict is remembered next to its listing (keyed by the listing's size and
        # time stamp), so an unchanged unit is not parsed again.
        if not os.path.exists(lst):
            raise RuntimeError(f'spill check: no device listing {lst} for {src}: refusing to link an unchecked object')
        lints.append(pool.submit(_lint_cached, lst))
        return obj

    with ProcessPoolExecutor(max_workers=2) as pool:
        with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as ex:
            objs = list(ex.map(compile_one, srcs))
        check_spills(objdir, [f.result() for f in lints], verbose=verbose)
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return lib


def check_spills(objdir, results, verbose=False):
    """No kernel may reload a spilled register for lanes that were switched off when it was stored (spill_lint.py: what
    ROCm 7.2's register allocator did to lanesweep_kernel in round 5).  Checked on the listings of the units just compiled
    (results: what _lint_one returned for each); the report of every kernel with spills goes to build/spill_lint.txt.  A
    reported kernel fails the build -- on every later call too, until its unit is recompiled clean: the link step is never
    reached with a flagged or unchecked object (tests/test_spill_lint.py::test_failed_check_is_not_bypassed_by_rebuilding).  Restructure the lane-divergent `if` the report points at (or relieve the register
    pressure) until it is clean."""
    if not results:
        return
    lines, nbad = [], 0
    for path, bad, report in results:
        nbad += bad
        for kernel, nsp, nslots in report:
            lines.append(f'{os.path.basename(path)}: {kernel}: {nsp} spill stores (dwords), {nslots} slot(s) reported')
    # (every unit that is linked is in `results`: the report describes THIS library, not the history of the build directory)
    with open(os.path.join(os.path.dirname(objdir), 'spill_lint.txt'), 'w') as f:
        f.write('\n'.join(lines) + '\n')
    if verbose:
        print('\n'.join(l for l in lines if not l.endswith(' 0 slot(s) reported')) or 'spill check: clean')
    if nbad:
        raise RuntimeError('spill check: %d slot(s) of spilled registers are stored with lanes switched off and reloaded for more lanes '
                           '(python -m lightweaver_amd.spill_lint <listing under %s> for the lines):\n%s'
                           % (nbad, objdir, '\n'.join(l for l in lines if not l.endswith(' 0 slot(s) reported'))))


def _lint_one(path):
    from . import spill_lint
    report = []
    bad = spill_lint.lint_file(path, quiet=True, report=report)
    return path, bad, report


def _lint_cached(path):
    """_lint_one with its result kept in <listing>.lint.json while the listing (size, mtime) and the checker are unchanged."""
    import json
    from . import spill_lint
    st = os.stat(path)
    key = [st.st_size, st.st_mtime_ns, os.stat(spill_lint.__file__).st_mtime_ns]
    stamp = path + '.lint.json'
    try:
        with open(stamp) as f:
            d = json.load(f)
        if d['key'] == key:
            return path, d['bad'], [tuple(r) for r in d['report']]
    except (OSError, ValueError, KeyError):
        pass
    _, bad, report = _lint_one(path)
    with open(stamp, 'w') as f:
        json.dump({'key': key, 'bad': bad, 'report': report}, f)
    return path, bad, report


def build_plugin(force=False, verbose=False):
    """The Lightweaver iteration-scheme plugin (fs_iteration_fns_provider).  It is compiled against
    the Lightweaver headers, so it is only (re)built where they are present; elsewhere a prebuilt
    copy (it travels with the tree) is used as is."""
    src = os.path.join(CSRC, 'plugin', 'lwhip_plugin.cpp')
    if not os.path.exists(src) or not os.path.exists(os.path.join(REF_SRC, 'Lightweaver.hpp')):
        return PLUGIN if os.path.exists(PLUGIN) else None
    deps = [src, os.path.join(ROOT, 'include', 'lwhip.h')]
    build_hip()
    if not force and not _stale(PLUGIN, deps):
        return PLUGIN
    cmd = ['g++', '-std=c++17', '-O2', '-fPIC', '-shared', '-Wno-sign-compare',
           '-DENKITS_BUILD_DLL', f'-I{REF_SRC}', f'-I{os.path.join(ROOT, "include")}',
           '-o', PLUGIN, src, os.path.join(REF_SRC, 'LightweaverAmalgamated.cpp'),
           os.path.join(REF_SRC, 'TaskScheduler.cpp'),
           f'-L{PKG}', '-llwhip', '-ldl', '-lpthread', '-Wl,-Bsymbolic', '-Wl,-rpath,$ORIGIN']
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    return PLUGIN


def build_oracle(verbose=False):
    """TEST INFRASTRUCTURE: oracle/liblworacle.so and, where /root/reference exists, oracle/_ref."""
    subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle'), 'all'], check=True,
                   stdout=None if verbose else subprocess.DEVNULL)


def build_all(force=False, verbose=False):
    build_hip(force=force, verbose=verbose)
    build_plugin(force=force, verbose=verbose)


if __name__ == '__main__':
    build_all(force='--force' in sys.argv, verbose=True)
