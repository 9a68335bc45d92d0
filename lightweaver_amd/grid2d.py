"""Flat description of an x-periodic 2D grid and its short-characteristic intersection table, the
input of lwhip_formal_solver_2d (include/lwhip.h: lwhip_grid2d).  The table is what the reference's
build_intersection_list produces (Source/FormalScalar2d.cpp:1188-1327); under Lightweaver the plugin reads
it from Atmosphere::intersections; the mirror builds it with build_grid2d (lwhip_build_intersections, host side)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _abi as abi

INTERSECTION_DTYPE = np.dtype([('axis', np.int32), ('_pad', np.int32), ('fracZ', np.float64),
                               ('fracX', np.float64), ('distance', np.float64)])
assert INTERSECTION_DTYPE.itemsize == C.sizeof(abi.lwhip_intersection)


@dataclass
class Grid2d:
    x: np.ndarray               # [Nx]
    z: np.ndarray               # [Nz]
    mux: np.ndarray             # [Nrays]
    muz: np.ndarray             # [Nrays]
    temperature: np.ndarray     # [Nz, Nx]
    uw: np.ndarray              # [Nrays, 2, Nz, Nx] INTERSECTION_DTYPE
    dw: np.ndarray
    longCharIdx: np.ndarray     # [Nrays, 2, Nz, Nx] int32
    substepOff: np.ndarray      # [NlongChar + 1] int32
    substeps: np.ndarray        # [substepOff[-1]] INTERSECTION_DTYPE
    zLowerBc: int = abi.BC_THERMALISED
    zUpperBc: int = abi.BC_ZERO
    # x boundaries: None = PERIODIC; else both CALLABLE -- model.Boundary(type=BC_CALLABLE, idxs[Nrays, 2],
    # bcData[Nlambda, Nmu, Nz]): the column a ray enters through is prescribed (FormalScalar2d.cpp:806-852)
    xLowerBc: object = None
    xUpperBc: object = None

    @property
    def periodic(self):
        return self.xLowerBc is None and self.xUpperBc is None

    def _xbc(self, b, keep):
        out = abi.lwhip_boundary()
        out.type = abi.BC_CALLABLE
        b.idxs = np.ascontiguousarray(b.idxs, dtype=np.int32)
        b.bcData = np.ascontiguousarray(b.bcData, dtype=np.float64)
        if b.idxs.shape != (self.Nrays, 2) or b.bcData.ndim != 3 or b.bcData.shape[2] != self.Nz:
            raise ValueError('x boundary needs idxs[Nrays, 2] and bcData[Nlambda, Nmu, Nz]')
        out.Nmu = b.bcData.shape[1]
        out.idxs = b.idxs.ctypes.data_as(abi.i32p)
        out.bcData = b.bcData.ctypes.data_as(abi.f64p)
        keep.append(out)
        return C.pointer(out)

    @property
    def Nx(self):
        return self.x.size

    @property
    def Nz(self):
        return self.z.size

    @property
    def Nrays(self):
        return self.mux.size

    def descriptor(self) -> abi.lwhip_grid2d:
        ip = C.POINTER(abi.lwhip_intersection)
        for name in ('x', 'z', 'mux', 'muz', 'temperature'):
            setattr(self, name, np.ascontiguousarray(getattr(self, name), dtype=np.float64))
        for name in ('uw', 'dw', 'substeps'):
            setattr(self, name, np.ascontiguousarray(getattr(self, name), dtype=INTERSECTION_DTYPE))
        self.longCharIdx = np.ascontiguousarray(self.longCharIdx, dtype=np.int32)
        self.substepOff = np.ascontiguousarray(self.substepOff, dtype=np.int32)
        g = abi.lwhip_grid2d()
        g.Nx, g.Nz, g.Nrays, g.periodic = self.Nx, self.Nz, self.Nrays, int(self.periodic)
        if not self.periodic:
            if self.xLowerBc is None or self.xUpperBc is None:
                raise ValueError('mixed boundary types are not supported on the x axis')
            self._keep = []
            g.xLowerBc = self._xbc(self.xLowerBc, self._keep)
            g.xUpperBc = self._xbc(self.xUpperBc, self._keep)
        g.zLowerBc, g.zUpperBc = self.zLowerBc, self.zUpperBc
        g.NlongChar = self.substepOff.size - 1
        g.x, g.z = self.x.ctypes.data_as(abi.f64p), self.z.ctypes.data_as(abi.f64p)
        g.mux, g.muz = self.mux.ctypes.data_as(abi.f64p), self.muz.ctypes.data_as(abi.f64p)
        g.temperature = self.temperature.ctypes.data_as(abi.f64p)
        g.uw, g.dw = self.uw.ctypes.data_as(ip), self.dw.ctypes.data_as(ip)
        g.longCharIdx = self.longCharIdx.ctypes.data_as(abi.i32p)
        g.substepOff = self.substepOff.ctypes.data_as(abi.i32p)
        g.substeps = self.substeps.ctypes.data_as(ip)
        return g

    def to_dict(self, prefix='grid/'):
        d = {prefix + k: getattr(self, k) for k in ('x', 'z', 'mux', 'muz', 'temperature', 'longCharIdx', 'substepOff')}
        for k in ('uw', 'dw', 'substeps'):
            a = getattr(self, k)
            d[prefix + k + '/axis'] = a['axis']
            d[prefix + k + '/vals'] = np.stack([a['fracZ'], a['fracX'], a['distance']], axis=-1)
        d[prefix + 'bc'] = np.array([self.zLowerBc, self.zUpperBc])
        if not self.periodic:
            for name, b in (('xLower', self.xLowerBc), ('xUpper', self.xUpperBc)):
                d[prefix + name + '/idxs'] = np.asarray(b.idxs)
                d[prefix + name + '/bcData'] = np.asarray(b.bcData)
        return d

    @staticmethod
    def from_dict(d, prefix='grid/'):
        def inter(k):
            ax, v = d[prefix + k + '/axis'], d[prefix + k + '/vals']
            a = np.zeros(ax.shape, dtype=INTERSECTION_DTYPE)
            a['axis'], a['fracZ'], a['fracX'], a['distance'] = ax, v[..., 0], v[..., 1], v[..., 2]
            return a
        bc = d[prefix + 'bc']
        xb = {}
        if prefix + 'xLower/idxs' in d:
            from .model import Boundary
            for name in ('xLower', 'xUpper'):
                xb[name + 'Bc'] = Boundary(abi.BC_CALLABLE, idxs=d[prefix + name + '/idxs'], bcData=d[prefix + name + '/bcData'])
        return Grid2d(**xb, x=d[prefix + 'x'], z=d[prefix + 'z'], mux=d[prefix + 'mux'], muz=d[prefix + 'muz'],
                      temperature=d[prefix + 'temperature'], uw=inter('uw'), dw=inter('dw'),
                      longCharIdx=d[prefix + 'longCharIdx'], substepOff=d[prefix + 'substepOff'],
                      substeps=inter('substeps'), zLowerBc=int(bc[0]), zUpperBc=int(bc[1]))


def build_grid2d(x, z, mux, muz, temperature, zLowerBc=abi.BC_THERMALISED, zUpperBc=abi.BC_ZERO, xLowerBc=None,
                 xUpperBc=None) -> Grid2d:
    """The counterpart of Atmosphere.make_2d's geometry step for an x-periodic grid: the intersection table of
    build_intersection_list (Source/FormalScalar2d.cpp:1188-1327) through lwhip_build_intersections.  mux, muz:
    the quadrature's direction cosines [Nrays] (toObs direction; the table holds both directions)."""
    from .context import _check, load_library
    lib = load_library()
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    x, z, mux, muz = f(x), f(z), f(mux), f(muz)
    temperature = f(temperature).reshape(z.size, x.size)
    ip = C.POINTER(abi.lwhip_intersection)
    g = abi.lwhip_grid2d()
    periodic = xLowerBc is None and xUpperBc is None
    g.Nx, g.Nz, g.Nrays, g.periodic = x.size, z.size, mux.size, int(periodic)
    g.x, g.z = x.ctypes.data_as(abi.f64p), z.ctypes.data_as(abi.f64p)
    g.mux, g.muz = mux.ctypes.data_as(abi.f64p), muz.ctypes.data_as(abi.f64p)
    nl, ns = C.c_int32(0), C.c_int64(0)
    _check(lib, lib.lwhip_build_intersections(C.byref(g), None, None, None, None, 0, None, 0, C.byref(nl),
                                              C.byref(ns)), 'lwhip_build_intersections')
    shape = (mux.size, 2, z.size, x.size)
    uw = np.zeros(shape, dtype=INTERSECTION_DTYPE)
    dw = np.zeros(shape, dtype=INTERSECTION_DTYPE)
    li = np.zeros(shape, dtype=np.int32)
    off = np.zeros(nl.value + 1, dtype=np.int32)
    sub = np.zeros(max(ns.value, 1), dtype=INTERSECTION_DTYPE)
    _check(lib, lib.lwhip_build_intersections(C.byref(g), uw.ctypes.data_as(ip), dw.ctypes.data_as(ip),
                                              li.ctypes.data_as(abi.i32p), off.ctypes.data_as(abi.i32p), nl.value,
                                              sub.ctypes.data_as(ip), ns.value, C.byref(nl), C.byref(ns)),
           'lwhip_build_intersections')
    return Grid2d(x=x, z=z, mux=mux, muz=muz, temperature=temperature, uw=uw, dw=dw, longCharIdx=li, substepOff=off,
                  substeps=sub[:ns.value], zLowerBc=zLowerBc, zUpperBc=zUpperBc, xLowerBc=xLowerBc, xUpperBc=xUpperBc)


def formal_solver_2d(grid: Grid2d, wavelength: float, rays, chi, S, device: int = 0):
    """chi, S: [nSolve, Nz, Nx]; rays[p] = 2 * mu + toObs.  Returns (I, PsiStar) of the same shape."""
    from .context import _check, load_library
    lib = load_library()
    rays = np.ascontiguousarray(rays, dtype=np.int32)
    chi = np.ascontiguousarray(chi, dtype=np.float64)
    S = np.ascontiguousarray(S, dtype=np.float64)
    assert chi.shape == S.shape == (rays.size, grid.Nz, grid.Nx)
    I = np.zeros_like(chi)
    Psi = np.zeros_like(chi)
    g = grid.descriptor()
    _check(lib, lib.lwhip_formal_solver_2d(device, C.byref(g), float(wavelength), rays.size,
                                           rays.ctypes.data_as(abi.i32p), chi.ctypes.data_as(abi.f64p),
                                           S.ctypes.data_as(abi.f64p), I.ctypes.data_as(abi.f64p),
                                           Psi.ctypes.data_as(abi.f64p)), 'lwhip_formal_solver_2d')
    return I, Psi
