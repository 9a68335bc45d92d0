"""The host-side 2D geometry builder (lwhip_build_intersections) against the reference's build_intersection_list
(Source/FormalScalar2d.cpp:1188-1327): bit for bit, on the committed fixtures and -- where oracle/_ref exists --
on fresh grids straight from the reference.  No GPU needed: the builder is host code of liblwhip.so."""
import os

import numpy as np
import pytest

from lightweaver_amd import _abi as abi
from lightweaver_amd.grid2d import Grid2d, build_grid2d
from oracle import bindings

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
HAVE_REF = os.path.exists(bindings.REF_LIB)


def same_table(a: Grid2d, b: Grid2d):
    for name in ('uw', 'dw', 'substeps'):
        p, q = getattr(a, name), getattr(b, name)
        assert p.shape == q.shape, name
        for fld in ('axis', 'fracZ', 'fracX', 'distance'):
            assert np.array_equal(p[fld], q[fld]), (name, fld)
    assert np.array_equal(a.longCharIdx, b.longCharIdx)
    assert np.array_equal(a.substepOff, b.substepOff)


@pytest.mark.parametrize('fixture', ['fs2d_small.npz', 'falc2d_small.npz'])
def test_builder_matches_golden_tables(hip_lib, fixture):
    d = dict(np.load(os.path.join(GOLD, fixture)))
    prefix = 'grid/' if 'grid/x' in d else 'grid2d/'
    ref = Grid2d.from_dict(d, prefix=prefix)
    mine = build_grid2d(ref.x, ref.z, ref.mux, ref.muz, ref.temperature, ref.zLowerBc, ref.zUpperBc)
    same_table(mine, ref)


@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref not built (needs /root/reference)')
@pytest.mark.parametrize('Nx,Nz,seed', [(5, 4, 0), (33, 17, 1), (64, 82, 2)])
def test_builder_matches_reference(hip_lib, Nx, Nz, seed):
    rng = np.random.default_rng(seed)
    x = np.cumsum(rng.uniform(0.5, 1.5, Nx)) * 30e3          # irregular in x and z
    z = np.cumsum(rng.uniform(0.2, 2.0, Nz))[::-1] * 20e3
    # shallow, steep, exactly diagonal-ish and vertical rays, both signs of mux
    muz = np.array([0.1, 0.3, 0.6, 0.9, 1.0, 0.5])
    mux = np.array([1, -1, 1, -1, 0, 1]) * np.sqrt(1.0 - muz**2) * np.array([1, 1, 0.7, 1, 1, -0.3])
    T = rng.uniform(4e3, 1e4, (Nz, Nx))
    ref = bindings.Ref2d(x, z, mux, muz, T).grid()
    mine = build_grid2d(x, z, mux, muz, T)
    same_table(mine, ref)
    assert mine.substepOff.size - 1 > 0


def test_builder_uniform_grid_diagonal(hip_lib):
    # a ray through the grid's diagonal hits x face and z plane together: axis NONE everywhere, no long characteristics
    x = np.arange(6) * 10.0
    z = np.arange(5)[::-1] * 10.0
    c = np.sqrt(0.5)
    g = build_grid2d(x, z, [c], [c], np.ones((5, 6)))
    assert g.substepOff.size == 1
    assert np.all(g.uw['axis'][0, :, :, :] == abi.AXIS_NONE)


def test_builder_rejects(hip_lib):
    from lightweaver_amd.context import LwHipError
    x = np.arange(6) * 10.0
    z = np.arange(5)[::-1] * 10.0
    with pytest.raises(LwHipError):
        build_grid2d(x, z, [1.0], [0.0], np.ones((5, 6)))      # a ray along x


# ---- fixed (CALLABLE) x boundaries: no wrap-around, no long characteristics ---------------------------------------
def x_boundaries(Nrays, Nz, Nla=1, seed=0):
    """Both x boundaries CALLABLE as the reference requires (FormalScalar2d.cpp:750-759): idxs[mu, toObs] -> row of
    bcData [Nlambda, Nmu, Nz] (rows for every (mu, toObs) pair; the solver reads the side the ray enters through)."""
    from lightweaver_amd.model import Boundary
    rng = np.random.default_rng(seed)
    idxs = np.arange(2 * Nrays, dtype=np.int32).reshape(Nrays, 2)
    mk = lambda: Boundary(abi.BC_CALLABLE, idxs=idxs.copy(), bcData=rng.uniform(0.5, 2.0, (Nla, 2 * Nrays, Nz)) * 1e-8)
    return mk(), mk()


@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref not built (needs /root/reference)')
@pytest.mark.parametrize('Nx,Nz,seed', [(5, 4, 0), (33, 17, 1), (64, 82, 2)])
def test_builder_matches_reference_fixed_x_boundaries(hip_lib, Nx, Nz, seed):
    rng = np.random.default_rng(seed)
    x = np.cumsum(rng.uniform(0.5, 1.5, Nx)) * 30e3
    z = np.cumsum(rng.uniform(0.2, 2.0, Nz))[::-1] * 20e3
    muz = np.array([0.1, 0.3, 0.6, 0.9, 1.0, 0.5])
    mux = np.array([1, -1, 1, -1, 0, 1]) * np.sqrt(1.0 - muz**2) * np.array([1, 1, 0.7, 1, 1, -0.3])
    T = rng.uniform(4e3, 1e4, (Nz, Nx))
    lo, hi = x_boundaries(muz.size, Nz)
    ref = bindings.Ref2d(x, z, mux, muz, T, xLowerBc=lo, xUpperBc=hi).grid()
    mine = build_grid2d(x, z, mux, muz, T, xLowerBc=lo, xUpperBc=hi)
    same_table(mine, ref)
    assert mine.substepOff.size - 1 == 0 and not mine.periodic
