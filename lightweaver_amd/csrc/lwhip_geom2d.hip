// lwhip_geom2d.hip -- the short-characteristic geometry of an x-periodic 2D grid, host side, once per atmosphere:
// build_intersection_list (Source/FormalScalar2d.cpp:1188-1327) with dw_intersection_2d (:60-105),
// uw_intersection_2d (:107-152) and the long characteristics' uw_intersection_2d_frac_x (:166-206), written straight
// into the flat tables of lwhip_grid2d.  No device code: the table is uploaded by lwhip_create /
// lwhip_formal_solver_2d like the rest of the problem.  The arithmetic follows the reference operation by operation
// (the fractional indices are compared bit for bit in tests/test_geom2d.py).
#include "../../include/lwhip.h"

#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace
{
struct Grid
{
    const double* x;
    const double* z;
    double mux, muz;
    int xStep, xStart, xEnd, zStep;
    bool periodic;
};

// approx_equal (Source/LwAtmosphere.hpp:52-74)
bool approx_equal(double a, double b, double eps)
{
    const double absA = std::fabs(a), absB = std::fabs(b), diff = std::fabs(a - b);
    const double tiny = std::numeric_limits<double>::min();
    if (a == b)
        return true;
    if (a == 0.0 || b == 0.0 || (absA + absB < tiny))
        return diff < (eps * tiny);
    return diff / std::fmin(absA + absB, std::numeric_limits<double>::max()) < eps;
}

// IntersectionResult's constructor (Source/LwAtmosphere.hpp:109-143): a hit within PlaneTol = 1e-6 (relative) of a
// grid line along its interpolation axis is snapped onto the grid point and needs no interpolation
lwhip_intersection hit(int axis, double fracZ, double fracX, double distance)
{
    lwhip_intersection r;
    std::memset(&r, 0, sizeof(r));
    r.axis = axis;
    r.fracZ = fracZ;
    r.fracX = fracX;
    r.distance = distance;
    if (axis == LWHIP_AXIS_X)
    {
        const double x = std::round(fracX);
        if (approx_equal(x, fracX, 1e-6))
        {
            r.axis = LWHIP_AXIS_NONE;
            r.fracX = x;
        }
    }
    else if (axis == LWHIP_AXIS_Z)
    {
        const double z = std::round(fracZ);
        if (approx_equal(z, fracZ, 1e-6))
        {
            r.axis = LWHIP_AXIS_NONE;
            r.fracZ = z;
        }
    }
    return r;
}

// parametrised hit with a plane of constant x / z: t = -(origin - offset) / mu (IEEE inf / nan semantics relied on)
inline double plane_t(double origin, double offset, double mu) { return -(origin - offset) / mu; }

// dw_intersection_2d :60-105 (periodic: the last column wraps onto the first; fixed boundaries: an inclined ray has no
// downwind point there -- distance 0, caught by the solver -- and a vertical one never meets an x face: the reference
// evaluates its x-face hit one column beyond the grid, where the result is infinite whatever it reads)
lwhip_intersection dw_hit(const Grid& g, int zp, int xp)
{
    bool beyond = false;
    if (xp == g.xEnd)
    {
        if (g.periodic)
            xp = g.xStart;
        else if (std::fabs(g.muz) != 1.0)
            return hit(LWHIP_AXIS_NONE, zp, xp, 0.0);
        else
            beyond = true;
    }
    const double tx = beyond ? std::numeric_limits<double>::infinity() : plane_t(g.x[xp], g.x[xp + g.xStep], g.mux);
    const double tz = plane_t(g.z[zp], g.z[zp + g.zStep], g.muz);
    if (std::fabs(tx) < std::fabs(tz))
        return hit(LWHIP_AXIS_Z, zp + g.zStep * (tx / tz), xp + g.xStep, tx);
    if (std::fabs(tz) < std::fabs(tx))
        return hit(LWHIP_AXIS_X, zp + g.zStep, xp + g.xStep * (tz / tx), tz);
    return hit(LWHIP_AXIS_NONE, zp + g.zStep, xp + g.xStep, tx);
}

// uw_intersection_2d :107-152
lwhip_intersection uw_hit(const Grid& g, int zp, int xp)
{
    bool beyond = false;
    if (xp == g.xStart)
    {
        if (g.periodic)
            xp = g.xEnd;
        else if (std::fabs(g.muz) != 1.0)
            return hit(LWHIP_AXIS_NONE, zp, xp, 0.0);
        else
            beyond = true;
    }
    const double tx = beyond ? std::numeric_limits<double>::infinity() : plane_t(g.x[xp], g.x[xp - g.xStep], g.mux);
    const double tz = plane_t(g.z[zp], g.z[zp - g.zStep], g.muz);
    if (std::fabs(tx) < std::fabs(tz))
        return hit(LWHIP_AXIS_Z, zp - g.zStep * (tx / tz), xp - g.xStep, tx);
    if (std::fabs(tz) < std::fabs(tx))
        return hit(LWHIP_AXIS_X, zp - g.zStep, xp - g.xStep * (tz / tx), tz);
    return hit(LWHIP_AXIS_NONE, zp - g.zStep, xp - g.xStep, tx);
}

// frac_idx :154-164
double frac_idx(const double* param, double fracIdx)
{
    const int xm = (int)fracIdx;
    if (xm == fracIdx)
        return param[xm];
    const double frac = fracIdx - xm;
    return (1.0 - frac) * param[xm] + frac * param[xm + 1];
}

// uw_intersection_2d_frac_x :166-206: one more step upwind from a hit on an x face
lwhip_intersection uw_hit_frac_x(const Grid& g, const lwhip_intersection& start)
{
    int xp = (int)start.fracX;
    if (xp == g.xStart)
        xp = g.xEnd;
    const double startX = g.x[xp];
    const double startZ = frac_idx(g.z, start.fracZ);
    const int zPlaneIdx = (g.zStep > 0) ? (int)start.fracZ : (int)start.fracZ - g.zStep;
    const double tx = plane_t(startX, g.x[xp - g.xStep], g.mux);
    const double tz = plane_t(startZ, g.z[zPlaneIdx], g.muz);
    if (std::fabs(tx) < std::fabs(tz))
    {
        const double fracThroughZ = std::fabs(zPlaneIdx - start.fracZ);
        return hit(LWHIP_AXIS_Z, start.fracZ - fracThroughZ * g.zStep * (tx / tz), xp - g.xStep, tx);
    }
    if (std::fabs(tz) < std::fabs(tx))
        return hit(LWHIP_AXIS_X, zPlaneIdx, xp - g.xStep * (tz / tx), tz);
    return hit(LWHIP_AXIS_NONE, zPlaneIdx, xp - g.xStep, tx);
}
}

extern "C" int lwhip_build_intersections_impl(const lwhip_grid2d* grid, lwhip_intersection* uwOut, lwhip_intersection* dwOut,
                                              int32_t* longCharIdx, int32_t* substepOff, int32_t capLongChar,
                                              lwhip_intersection* substeps, int64_t capSubsteps, int32_t* nLongChar,
                                              int64_t* nSubsteps)
{
    const int Nx = grid->Nx, Nz = grid->Nz, Nrays = grid->Nrays;
    const size_t Ns = (size_t)Nx * Nz;
    const bool fill = uwOut != nullptr;
    std::vector<int32_t> off(1, 0);
    std::vector<lwhip_intersection> steps, one;
    // a long characteristic crosses at most one x face per column and one z plane: more steps than that means the
    // walk is not converging on a z plane (a ray along x)
    const size_t maxSteps = (size_t)4 * ((size_t)Nx + Nz) + 16;
    for (int mu = 0; mu < Nrays; ++mu)
        for (int toObsI = 0; toObsI < 2; ++toObsI)
        {
            const bool toObs = toObsI != 0;
            Grid g;
            g.x = grid->x;
            g.z = grid->z;
            g.muz = toObs ? grid->muz[mu] : -grid->muz[mu];
            g.mux = toObs ? grid->mux[mu] : -grid->mux[mu];
            g.zStep = toObs ? -1 : 1;
            const int kStart = toObs ? Nz - 1 : 0;
            const int kEnd = toObs ? 0 : Nz - 1;
            g.xStep = (g.mux < 0) ? -1 : 1;
            g.xStart = (g.mux < 0) ? Nx - 1 : 0;
            g.xEnd = (g.mux < 0) ? 0 : Nx - 1;
            g.periodic = grid->periodic != 0;
            const bool edge = !g.periodic && g.mux != 0.0; // fixed boundaries: no downwind point in the last column
            const size_t base = ((size_t)mu * 2 + toObsI) * Ns;
            int k = kStart;
            // the starting plane :1252-1264
            for (int j = g.xStart; j != g.xEnd + g.xStep; j += g.xStep)
            {
                lwhip_intersection dw = dw_hit(g, k, j);
                dw.distance = std::fabs(dw.distance);
                if (edge && j == g.xEnd)
                    dw = hit(LWHIP_AXIS_NONE, k, j, 0.0);
                if (fill)
                {
                    const size_t o = base + (size_t)k * Nx + j;
                    uwOut[o] = hit(LWHIP_AXIS_NONE, k, j, 0.0);
                    dwOut[o] = dw;
                    longCharIdx[o] = -1;
                }
            }
            k += g.zStep;
            for (; k != kEnd + g.zStep; k += g.zStep)
                for (int j = g.xStart; j != g.xEnd + g.xStep; j += g.xStep)
                {
                    lwhip_intersection uw = uw_hit(g, k, j);
                    uw.distance = std::fabs(uw.distance);
                    int lc = -1;
                    if (g.periodic && j == g.xStart && uw.axis == LWHIP_AXIS_Z)
                    {
                        // periodic long characteristic :1277-1304: walk upwind until a z plane is hit, then store
                        // the steps upwind-most first with the point's own upwind hit last
                        lc = (int)off.size() - 1;
                        one.clear();
                        lwhip_intersection loc = uw;
                        while (true)
                        {
                            lwhip_intersection uuw = uw_hit_frac_x(g, loc);
                            uuw.distance = std::fabs(uuw.distance);
                            one.push_back(uuw);
                            if (uuw.axis != LWHIP_AXIS_Z)
                                break;
                            if (one.size() > maxSteps)
                                return LWHIP_ERR_UNSUPPORTED;
                            loc = uuw;
                        }
                        steps.insert(steps.end(), one.rbegin(), one.rend());
                        steps.push_back(uw);
                        off.push_back((int32_t)steps.size());
                    }
                    lwhip_intersection dw = hit(LWHIP_AXIS_NONE, k, j, 0.0);
                    if (k != kEnd && !(edge && j == g.xEnd))
                    {
                        dw = dw_hit(g, k, j);
                        dw.distance = std::fabs(dw.distance);
                    }
                    if (fill)
                    {
                        const size_t o = base + (size_t)k * Nx + j;
                        uwOut[o] = uw;
                        dwOut[o] = dw;
                        longCharIdx[o] = lc;
                    }
                }
        }
    *nLongChar = (int32_t)off.size() - 1;
    *nSubsteps = (int64_t)steps.size();
    if (fill)
    {
        if (capLongChar < *nLongChar || capSubsteps < *nSubsteps)
            return LWHIP_ERR_INVALID;
        std::memcpy(substepOff, off.data(), off.size() * sizeof(int32_t));
        if (!steps.empty())
            std::memcpy(substeps, steps.data(), steps.size() * sizeof(lwhip_intersection));
    }
    return LWHIP_OK;
}
