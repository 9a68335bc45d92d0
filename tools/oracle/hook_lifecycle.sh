#!/bin/bash
# Hook-lifecycle conformance of the plugin under the REAL Python framework, in the build container (CPU only).
#
# Builds the reference's Cython layer per SURVEY.md Appendix A under /tmp (nothing of it enters the repository),
# registers lightweaver_amd/LwHipIterScheme.so with FsIterationSchemes.load_fns_from_path, constructs
# lw.Context(fsIterScheme='mali_full_precond_hip_gfx950'), pickles it round trip, calls update_threads(), destroys
# everything, and checks (tools/oracle/hook_lifecycle.py) that alloc_global_scratch / free_global_scratch pair up and
# that the first numerical call fails with the "no HIP device" error instead of computing anything on the CPU.
# Reference: Source/LwMiddleLayer.pyx:3033-3037, 3120-3134, 4172-4184; Source/ThreadStorage.cpp:480-493, 538-566.
#
# usage: bash tools/oracle/hook_lifecycle.sh        (needs /root/reference; outputs under /tmp/lw_hooks)
set -e
REPO=$(cd "$(dirname "$0")/../.." && pwd)
REF=${LW_REFERENCE:-/root/reference}
S=$REF/Source
W=/tmp/lw_hooks
PY=python3
EXT=$($PY -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
NPINC=$($PY -c "import numpy; print(numpy.get_include())")
PYINC=$($PY -c "import sysconfig; print(sysconfig.get_paths()['include'])")
rm -rf $W && mkdir -p $W/cy $W/pkg/stubs/astropy $W/pkg/stubs/numba $W/cfg
echo "[1/4] enkiTS + Cython layer (LwCompiled)"
g++ -std=c++17 -O2 -fPIC -shared -Wno-sign-compare -DENKITS_BUILD_DLL $S/TaskScheduler.cpp -o $W/libenkiTS.so -lpthread
cp $S/LwMiddleLayer.pyx $S/CmoArray.pxd $S/CmoArrayHelper.pxd $S/CmoArrayHelper.pyx $W/cy/
(cd $W/cy && cython --cplus -3 --module-name lightweaver.LwCompiled LwMiddleLayer.pyx -o LwCompiled.cpp > cython.log 2>&1)
g++ -std=c++17 -O1 -fPIC -shared -Wno-sign-compare -w -DLW_CORE_LIB -DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION \
    -I$S -I$NPINC -I$PYINC $W/cy/LwCompiled.cpp $S/LightweaverAmalgamated.cpp -o $W/cy/LwCompiled$EXT -L$W -lenkiTS -ldl
echo "[2/4] package tree + stubs for the absent dependencies (astropy, numba, weno4, parse)"
cp -r $REF/lightweaver $W/pkg/lightweaver && chmod -R u+w $W/pkg/lightweaver
cp $W/cy/LwCompiled$EXT $W/libenkiTS.so $W/pkg/lightweaver/
mkdir -p $W/pkg/lightweaver/DefaultIterSchemes
echo "version = '0.0.0+hooks'" > $W/pkg/lightweaver/version.py
printf 'def H_6_atom(): raise NotImplementedError\ndef CaII_atom(): raise NotImplementedError\n' > $W/pkg/lightweaver/rh_atoms.py
: > $W/pkg/stubs/astropy/__init__.py
cat > $W/pkg/stubs/astropy/units.py <<'PYEOF'
import numpy as np
class _Q:
    def __init__(self, v): self.value = v
class _U:
    __array_ufunc__ = None
    def __rlshift__(self, other): return _Q(np.asarray(other) if not np.isscalar(other) else other)
    def __truediv__(self, o): return self
    def __mul__(self, o): return self
    def __pow__(self, o): return self
    def __rtruediv__(self, o): return self
def __getattr__(name): return _U()
class quantity:
    Quantity = _Q
Quantity = _Q
PYEOF
echo "def get_config_dir(name): return '$W/cfg'" > $W/pkg/stubs/astropy/config.py
cat > $W/pkg/stubs/numba/__init__.py <<'PYEOF'
def njit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k: return a[0]
    return lambda f: f
PYEOF
echo "List = list" > $W/pkg/stubs/numba/typed.py
printf 'import numpy as np\ndef weno4(xs, xp, fp, **k): return np.interp(xs, xp, fp)\n' > $W/pkg/stubs/weno4.py
echo "def parse(*a, **k): raise NotImplementedError" > $W/pkg/stubs/parse.py
echo "[3/4] the plugin"
$PY -m lightweaver_amd.build > $W/build.log 2>&1 || (cat $W/build.log; exit 1)
echo "[4/4] lifecycle"
cd $W
LWHIP_PLUGIN_TRACE=1 LD_LIBRARY_PATH=$W/pkg/lightweaver:$REPO/lightweaver_amd PYTHONPATH=$W/pkg/stubs:$W/pkg \
    $PY -W ignore $REPO/tools/oracle/hook_lifecycle.py $REPO/lightweaver_amd/LwHipIterScheme.so 2> $W/trace.log | tee $W/result.log
grep -c "lwhip-plugin-trace" $W/trace.log > /dev/null || (echo "no trace lines"; exit 1)
$PY $REPO/tools/oracle/hook_lifecycle.py --check $W/trace.log
