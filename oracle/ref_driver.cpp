// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A thin extern "C" driver around the *real* Lightweaver C++ core.  It is compiled together
// with the reference's own source files, taken from where they lie under /root/reference/Source
// (see oracle/Makefile; nothing of the reference is copied into this repository), into
// oracle/_ref/liblwref.so.  The driver builds a reference `Context` whose F64Views point at the
// host arrays of an `lwhip_problem` (include/lwhip.h) and calls the reference entry points:
//
//   lwref_formal_sol_gamma_matrices -> formal_sol_gamma_matrices()   Source/FormalScalar.cpp:678-681
//   lwref_formal_sol                -> formal_sol()                  Source/FormalScalar.cpp:691-694
//   lwref_stat_equil                -> stat_eq()                     Source/UpdatePopulations.cpp:49-53
//   lwref_compute_profiles          -> Transition::compute_phi/wphi  Source/FormalScalar.cpp:53-68,106-134
//
// The iteration scheme is the built-in "mali_full_precond_scalar" (Source/FormalInterface.cpp:83-95)
// or any plugin loaded through the reference's own FsIterationFnsManager::load_fns_from_path
// (Source/FormalInterface.cpp:62-81) -- the reference SIMD plugins built by oracle/Makefile, or this
// repository's HIP plugin, which is how the drop-in is exercised against the real core.
//
// Used by: tests/ (to pin oracle/lw_oracle.c and to generate tests/golden/*.npz), bench.py's
// cpu_baseline leg (kind "reference").  Never linked or loaded by the product path.

#include "Lightweaver.hpp"
#include "Bezier.hpp"
#include "LuSolve.hpp"
#include "Ng.hpp"

namespace PrdCores { f64 GII(f64 aDamp, f64 qEmit, f64 qAbs); }

#include "../include/lwhip.h"

#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace
{
struct RefContext
{
    Atmosphere atmos;
    Spectrum spect;
    Background background;
    DepthData depthData;
    std::vector<std::unique_ptr<Atom>> atoms;
    std::vector<std::unique_ptr<Transition>> transitions;
    std::vector<std::vector<i8>> activeMasks;
    FormalSolverManager fsManager;
    FsIterationFnsManager iterManager;
    const lwhip_problem* prob;
    int spaceStart = -1, spaceEnd = -1; // lwref_set_depth_range
    Context ctx;
    bool threadsInitialised = false;
    bool includeDetailed = false; // ExtraParams "include_detailed_atoms" of the PRD calls
    double* zDown = nullptr;      // ExtraParams "ZPlaneDecomposition" + "ZPlaneDown" / "ZPlaneUp"
    double* zUp = nullptr;

    ~RefContext()
    {
        if (threadsInitialised)
            ctx.threading.clear(&ctx);
    }
};

void set_error(char* err, int errLen, const std::string& msg)
{
    if (err && errLen > 0)
    {
        std::snprintf(err, errLen, "%s", msg.c_str());
    }
}

const char* solver_name(int solver)
{
    switch (solver)
    {
        case LWHIP_FS_LINEAR_1D: return "piecewise_linear_1d";
        case LWHIP_FS_BESSER_1D: return "piecewise_besser_1d";
        case LWHIP_FS_BEZIER3_1D: return "piecewise_bezier3_1d";
        default: return nullptr;
    }
}

void configure_boundary(AtmosphericBoundaryCondition* bc, const lwhip_boundary& b,
                        int Nwave, int Nrays)
{
    BcIdxs idxs;
    if (b.idxs)
        idxs = BcIdxs(const_cast<i32*>(b.idxs), Nrays, 2);
    *bc = AtmosphericBoundaryCondition((RadiationBc)b.type, Nwave, b.Nmu, 1, idxs);
    if (b.type == LWHIP_BC_CALLABLE && b.bcData)
    {
        F64View3D data(const_cast<f64*>(b.bcData), Nwave, b.Nmu, 1);
        bc->set_bc_data(data);
    }
}
}

extern "C"
{
void* lwref_create(const lwhip_problem* p, const char* schemeLib, int Nthreads,
                   char* err, int errLen)
{
    try
    {
        if (!p || p->abiVersion != LWHIP_ABI_VERSION)
        {
            set_error(err, errLen, "bad descriptor / ABI version");
            return nullptr;
        }
        auto rc = std::make_unique<RefContext>();
        rc->prob = p;
        const int Nspace = p->Nspace;
        const int Nrays = p->Nrays;
        const int Nla = p->Nlambda;

        auto& a = rc->atmos;
        a.Nspace = Nspace;
        a.Nrays = Nrays;
        a.Ndim = 1;
        a.Nx = 0;
        a.Ny = 0;
        a.Nz = Nspace;
        a.Noutgoing = 1;
        a.z = F64View(const_cast<f64*>(p->height), Nspace);
        a.height = a.z;
        a.temperature = F64View(const_cast<f64*>(p->temperature), Nspace);
        a.vlosMu = F64View2D(const_cast<f64*>(p->vlosMu), Nrays, Nspace);
        a.muz = F64View(const_cast<f64*>(p->muz), Nrays);
        a.wmu = F64View(const_cast<f64*>(p->wmu), Nrays);
        configure_boundary(&a.zLowerBc, p->zLowerBc, Nla, Nrays);
        configure_boundary(&a.zUpperBc, p->zUpperBc, Nla, Nrays);
        if (p->grid2d) // x-periodic 2D: the core builds its own intersection table from the grid
        {
            const lwhip_grid2d& g = *p->grid2d;
            a.Ndim = 2;
            a.Nx = g.Nx;
            a.Nz = g.Nz;
            a.Noutgoing = g.Nx;
            a.x = F64View(const_cast<f64*>(g.x), g.Nx);
            a.z = F64View(const_cast<f64*>(g.z), g.Nz);
            a.height = a.z;
            a.mux = F64View(const_cast<f64*>(g.mux), Nrays);
            if (g.periodic)
            {
                a.xLowerBc.type = PERIODIC;
                a.xUpperBc.type = PERIODIC;
            }
            else
            {
                // fixed x boundaries: CALLABLE, bcData [Nlambda, Nmu, Nz] (Source/FormalScalar2d.cpp:806-852)
                auto xbc = [&](AtmosphericBoundaryCondition* bc, const lwhip_boundary* b) {
                    if (!b || b->type != LWHIP_BC_CALLABLE || !b->idxs || !b->bcData)
                        throw std::runtime_error("2D grid without periodic x needs CALLABLE xLowerBc / xUpperBc");
                    *bc = AtmosphericBoundaryCondition(CALLABLE, Nla, b->Nmu, g.Nz, BcIdxs(const_cast<i32*>(b->idxs), Nrays, 2));
                    bc->set_bc_data(F64View3D(const_cast<f64*>(b->bcData), Nla, b->Nmu, g.Nz));
                };
                xbc(&a.xLowerBc, g.xLowerBc);
                xbc(&a.xUpperBc, g.xUpperBc);
            }
            // z boundaries of a 2D grid: ZERO / THERMALISED by type, CALLABLE with bcData [Nlambda, Nmu, Nx] (:930-938)
            auto zbc = [&](AtmosphericBoundaryCondition* bc, int type, const lwhip_boundary& b) {
                if (type == LWHIP_BC_CALLABLE)
                {
                    *bc = AtmosphericBoundaryCondition(CALLABLE, Nla, b.Nmu, g.Nx, BcIdxs(const_cast<i32*>(b.idxs), Nrays, 2));
                    bc->set_bc_data(F64View3D(const_cast<f64*>(b.bcData), Nla, b.Nmu, g.Nx));
                }
                else
                    bc->type = (type == LWHIP_BC_THERMALISED) ? THERMALISED : ZERO;
            };
            zbc(&a.zLowerBc, g.zLowerBc, p->zLowerBc);
            zbc(&a.zUpperBc, g.zUpperBc, p->zUpperBc);
            build_intersection_list(&a);
        }

        auto& s = rc->spect;
        s.wavelength = F64View(const_cast<f64*>(p->wavelength), Nla);
        s.I = F64View3D(p->I, Nla, Nrays, p->grid2d ? p->grid2d->Nx : 1);
        s.J = F64View2D(p->J, Nla, Nspace);

        auto& bg = rc->background;
        bg.chi = F64View2D(const_cast<f64*>(p->bgChi), Nla, Nspace);
        bg.eta = F64View2D(const_cast<f64*>(p->bgEta), Nla, Nspace);
        bg.sca = F64View2D(const_cast<f64*>(p->bgSca), Nla, Nspace);

        auto& dd = rc->depthData;
        dd.fill = false;
        if (p->storeDepthData && p->depthChi && p->depthEta && p->depthI)
        {
            dd.fill = true;
            dd.chi = F64View4D(p->depthChi, Nla, Nrays, 2, Nspace);
            dd.eta = F64View4D(p->depthEta, Nla, Nrays, 2, Nspace);
            dd.I = F64View4D(p->depthI, Nla, Nrays, 2, Nspace);
        }

        auto& ctx = rc->ctx;
        ctx.atmos = &rc->atmos;
        ctx.spect = &rc->spect;
        ctx.background = &rc->background;
        ctx.depthData = &rc->depthData;
        ctx.Nthreads = Nthreads;
        ctx.methodScratch = nullptr;

        // -- iteration scheme: built-in scalar or a plugin through the reference's own loader
        if (schemeLib && schemeLib[0])
        {
            if (!rc->iterManager.load_fns_from_path(schemeLib))
            {
                set_error(err, errLen, std::string("load_fns_from_path failed for ") + schemeLib);
                return nullptr;
            }
            ctx.iterFns = rc->iterManager.fns.back();
        }
        else
        {
            ctx.iterFns = rc->iterManager.fns[0];
        }

        const char* fsName = p->grid2d ? "piecewise_besser_2d" : solver_name(p->formalSolver);
        if (p->grid2d)
            ctx.interpFn = InterpFn(2, "interp_linear_2d", LwInternal::interp_linear_2d);
        bool foundFs = false;
        for (auto& fs : rc->fsManager.formalSolvers)
        {
            if (fsName && std::strcmp(fs.name, fsName) == 0)
            {
                ctx.formalSolver = fs;
                foundFs = true;
            }
        }
        if (!foundFs)
        {
            set_error(err, errLen, "unknown formal solver");
            return nullptr;
        }

        for (int ia = 0; ia < p->Natom; ++ia)
        {
            const lwhip_atom& pa = p->atoms[ia];
            auto atom = std::make_unique<Atom>();
            atom->Nlevel = pa.Nlevel;
            atom->Ntrans = pa.Ntrans;
            atom->atmos = &rc->atmos;
            atom->n = F64View2D(pa.n, pa.Nlevel, Nspace);
            atom->nStar = F64View2D(const_cast<f64*>(pa.nStar), pa.Nlevel, Nspace);
            atom->nTotal = F64View(const_cast<f64*>(pa.nTotal), Nspace);
            atom->vBroad = F64View(const_cast<f64*>(pa.vBroad), Nspace);
            atom->methodScratch = nullptr;
            if (!pa.detailed)
            {
                atom->Gamma = F64View3D(pa.Gamma, pa.Nlevel, pa.Nlevel, Nspace);
                if (pa.C)
                    atom->C = F64View3D(const_cast<f64*>(pa.C), pa.Nlevel, pa.Nlevel, Nspace);
            }
            for (int kr = 0; kr < pa.Ntrans; ++kr)
            {
                const lwhip_transition& pt = pa.trans[kr];
                auto t = std::make_unique<Transition>();
                const int Nl = pt.Nred - pt.Nblue;
                t->Nblue = pt.Nblue;
                t->Nred = pt.Nred;
                t->type = (pt.type == LWHIP_LINE) ? LINE : CONTINUUM;
                t->i = pt.i;
                t->j = pt.j;
                t->Aji = pt.Aji;
                t->Bji = pt.Bji;
                t->Bij = pt.Bij;
                t->lambda0 = pt.lambda0;
                t->dopplerWidth = pt.dopplerWidth;
                t->polarised = false;
                t->methodScratch = nullptr;
                t->wavelength = F64View(const_cast<f64*>(pt.wavelength), Nl);
                if (pt.type == LWHIP_LINE)
                {
                    t->phi = F64View4D(pt.phi, Nl, Nrays, 2, Nspace);
                    t->wphi = F64View(pt.wphi, Nspace);
                    if (pt.aDamp)
                        t->aDamp = F64View(const_cast<f64*>(pt.aDamp), Nspace);
                    if (pt.prd && pt.rhoPrd)
                        t->rhoPrd = F64View2D(pt.rhoPrd, Nl, Nspace);
                    if (pt.Qelast)
                        t->Qelast = F64View(const_cast<f64*>(pt.Qelast), Nspace);
                }
                else
                {
                    t->alpha = F64View(const_cast<f64*>(pt.alpha), Nl);
                }
                rc->activeMasks.emplace_back(Nla, (i8)0);
                auto& mask = rc->activeMasks.back();
                for (int la = pt.Nblue; la < pt.Nred; ++la)
                    mask[la] = 1;
                t->active = BoolView((bool*)mask.data(), Nla);
                t->Rij = F64View(pt.Rij, Nspace);
                t->Rji = F64View(pt.Rji, Nspace);
                atom->trans.push_back(t.get());
                rc->transitions.emplace_back(std::move(t));
            }
            atom->init_scratch(Nspace, (bool)pa.detailed,
                               ctx.iterFns.defaultWlaGijStorage,
                               ctx.iterFns.defaultPerAtomStorage);
            if (pa.detailed)
                ctx.detailedAtoms.push_back(atom.get());
            else
                ctx.activeAtoms.push_back(atom.get());
            rc->atoms.emplace_back(std::move(atom));
        }

        ctx.initialise_threads();
        rc->threadsInitialised = true;
        return rc.release();
    }
    catch (const std::exception& e)
    {
        set_error(err, errLen, e.what());
        return nullptr;
    }
}

void lwref_destroy(void* h)
{
    delete (RefContext*)h;
}

int lwref_scheme_name(void* h, char* buf, int n)
{
    auto* rc = (RefContext*)h;
    std::snprintf(buf, n, "%s", rc->ctx.iterFns.name);
    return 0;
}

// refresh CALLABLE boundary data from the descriptor (LwAtmosphere.compute_bcs,
// Source/LwMiddleLayer.pyx:762-829, copies into the core-owned bcData each iteration)
// ExtraParams of the formal solutions: the ZPlaneDecomposition outputs (SimdFullIterationTemplates.hpp:253-281)
static ExtraParams fs_params(RefContext* rc)
{
    ExtraParams params;
    if (rc->zDown || rc->zUp)
    {
        const lwhip_problem* p = rc->prob;
        params.insert("ZPlaneDecomposition", true);
        if (p->grid2d)
        {
            if (rc->zDown)
                params.insert("ZPlaneDown", F64View3D(rc->zDown, p->Nlambda, p->Nrays, p->grid2d->Nx));
            if (rc->zUp)
                params.insert("ZPlaneUp", F64View3D(rc->zUp, p->Nlambda, p->Nrays, p->grid2d->Nx));
        }
        else
        {
            if (rc->zDown)
                params.insert("ZPlaneDown", F64View2D(rc->zDown, p->Nlambda, p->Nrays));
            if (rc->zUp)
                params.insert("ZPlaneUp", F64View2D(rc->zUp, p->Nlambda, p->Nrays));
        }
    }
    return params;
}

static void refresh_bcs(RefContext* rc)
{
    const lwhip_problem* p = rc->prob;
    const int nOut = p->grid2d ? p->grid2d->Nx : 1; // third extent of a z boundary's data
    if (p->zLowerBc.type == LWHIP_BC_CALLABLE && p->zLowerBc.bcData && rc->atmos.zLowerBc.type == CALLABLE)
    {
        F64View3D data(const_cast<f64*>(p->zLowerBc.bcData), p->Nlambda, p->zLowerBc.Nmu, nOut);
        rc->atmos.zLowerBc.set_bc_data(data);
    }
    if (p->zUpperBc.type == LWHIP_BC_CALLABLE && p->zUpperBc.bcData && rc->atmos.zUpperBc.type == CALLABLE)
    {
        F64View3D data(const_cast<f64*>(p->zUpperBc.bcData), p->Nlambda, p->zUpperBc.Nmu, nOut);
        rc->atmos.zUpperBc.set_bc_data(data);
    }
    if (p->grid2d && !p->grid2d->periodic) // the x boundaries' irradiation is borrowed like the z boundaries'
    {
        const lwhip_grid2d& g = *p->grid2d;
        rc->atmos.xLowerBc.set_bc_data(F64View3D(const_cast<f64*>(g.xLowerBc->bcData), p->Nlambda, g.xLowerBc->Nmu, g.Nz));
        rc->atmos.xUpperBc.set_bc_data(F64View3D(const_cast<f64*>(g.xUpperBc->bcData), p->Nlambda, g.xUpperBc->Nmu, g.Nz));
    }
}

int lwref_compute_profiles(void* h)
{
    auto* rc = (RefContext*)h;
    try
    {
        for (auto& atom : rc->atoms)
        {
            for (auto* t : atom->trans)
            {
                if (t->type != LINE)
                    continue;
                t->compute_phi(rc->atmos, t->aDamp, atom->vBroad);
                t->compute_wphi(rc->atmos);
            }
        }
    }
    catch (const std::exception& e)
    {
        return LWHIP_ERR_INVALID;
    }
    return LWHIP_OK;
}

int lwref_formal_sol_gamma_matrices(void* h, int lambdaIterate, lwhip_iter_result* res)
{
    auto* rc = (RefContext*)h;
    try
    {
        refresh_bcs(rc);
        IterationResult r = formal_sol_gamma_matrices(rc->ctx, (bool)lambdaIterate, fs_params(rc));
        if (res)
        {
            res->updatedJ = r.updatedJ;
            res->dJMax = r.dJMax;
            res->dJMaxIdx = r.dJMaxIdx;
        }
    }
    catch (const std::exception& e)
    {
        std::fprintf(stderr, "lwref: %s\n", e.what());
        return LWHIP_ERR_INVALID;
    }
    return LWHIP_OK;
}

int lwref_formal_sol(void* h, int upOnly)
{
    auto* rc = (RefContext*)h;
    try
    {
        refresh_bcs(rc);
        formal_sol(rc->ctx, (bool)upOnly, fs_params(rc));
    }
    catch (const std::exception& e)
    {
        std::fprintf(stderr, "lwref: %s\n", e.what());
        return LWHIP_ERR_INVALID;
    }
    return LWHIP_OK;
}

// spaceStart / spaceEnd handed to the three population updates below (-1, -1: the whole atmosphere)
int lwref_set_depth_range(void* h, int spaceStart, int spaceEnd)
{
    auto* rc = (RefContext*)h;
    rc->spaceStart = spaceStart;
    rc->spaceEnd = spaceEnd;
    return LWHIP_OK;
}

int lwref_stat_equil(void* h, int atomIdx)
{
    auto* rc = (RefContext*)h;
    try
    {
        for (int ia = 0; ia < (int)rc->atoms.size(); ++ia)
        {
            if (atomIdx >= 0 && ia != atomIdx)
                continue;
            if (rc->prob->atoms[ia].detailed)
                continue;
            stat_eq(rc->ctx, rc->atoms[ia].get(), ExtraParams{}, rc->spaceStart, rc->spaceEnd);
        }
    }
    catch (const std::runtime_error& e)
    {
        return LWHIP_ERR_SINGULAR;
    }
    return LWHIP_OK;
}

// time_dependent_update (Source/UpdatePopulations.cpp:153-157 -> iterFns.time_dep_update)
int lwref_time_dep_update(void* h, int atomIdx, const double* nOld, double dt)
{
    auto* rc = (RefContext*)h;
    try
    {
        Atom* atom = rc->atoms[atomIdx].get();
        F64View2D nOldView(const_cast<double*>(nOld), atom->Nlevel, rc->prob->Nspace);
        time_dependent_update(rc->ctx, atom, nOldView, dt, ExtraParams{}, rc->spaceStart, rc->spaceEnd);
    }
    catch (const std::runtime_error& e)
    {
        return LWHIP_ERR_SINGULAR;
    }
    return LWHIP_OK;
}

// nr_post_update (Source/UpdatePopulations.cpp:473-483 -> iterFns.nr_post_update)
int lwref_nr_post_update(void* h, const lwhip_nr_args* a)
{
    auto* rc = (RefContext*)h;
    const int Ns = rc->prob->Nspace;
    try
    {
        std::vector<Atom*> atoms;
        std::vector<F64View3D> dC;
        NrTimeDependentData td;
        td.dt = a->dt;
        for (int q = 0; q < a->Natoms; ++q)
        {
            Atom* atom = rc->atoms[a->atoms[q]].get();
            atom->stages = F64View1D(const_cast<double*>(a->stages[q]), atom->Nlevel);
            atoms.push_back(atom);
            if (a->dC)
                dC.emplace_back(const_cast<double*>(a->dC[q]), atom->Nlevel, atom->Nlevel, Ns);
            if (a->nPrev)
                td.nPrev.emplace_back(const_cast<double*>(a->nPrev[q]), atom->Nlevel, Ns);
        }
        rc->atmos.ne = F64View(a->ne, Ns);
        F64View bgNe(const_cast<double*>(a->backgroundNe), Ns);
        nr_post_update(rc->ctx, &atoms, dC, bgNe, td, a->crsw, ExtraParams{}, rc->spaceStart, rc->spaceEnd);
    }
    catch (const std::runtime_error& e)
    {
        return LWHIP_ERR_SINGULAR;
    }
    return LWHIP_OK;
}

// ---- 2D: the core's own geometry (build_intersection_list) and solver (piecewise_besser_2d) ------------
struct Ref2D
{
    Atmosphere atmos;
    std::vector<double> x, z, mux, muz, temperature;
};

void* lwref_2d_create_ex(const double* x, int Nx, const double* z, int Nz, const double* mux, const double* muz,
                         int Nrays, int zLowerBc, int zUpperBc, const double* temperature, const lwhip_boundary* xLower,
                         const lwhip_boundary* xUpper);

void* lwref_2d_create(const double* x, int Nx, const double* z, int Nz, const double* mux, const double* muz,
                      int Nrays, int zLowerBc, int zUpperBc, const double* temperature)
{
    return lwref_2d_create_ex(x, Nx, z, Nz, mux, muz, Nrays, zLowerBc, zUpperBc, temperature, nullptr, nullptr);
}

// xLower / xUpper: null = periodic; else CALLABLE boundaries with bcData [1, Nmu, Nz] (the primitive has one wavelength)
void* lwref_2d_create_ex(const double* x, int Nx, const double* z, int Nz, const double* mux, const double* muz,
                         int Nrays, int zLowerBc, int zUpperBc, const double* temperature, const lwhip_boundary* xLower,
                         const lwhip_boundary* xUpper)
{
    auto* r = new Ref2D();
    r->x.assign(x, x + Nx);
    r->z.assign(z, z + Nz);
    r->mux.assign(mux, mux + Nrays);
    r->muz.assign(muz, muz + Nrays);
    r->temperature.assign(temperature, temperature + (size_t)Nx * Nz);
    Atmosphere& a = r->atmos;
    a.Ndim = 2;
    a.Nx = Nx;
    a.Ny = 0;
    a.Nz = Nz;
    a.Nspace = Nx * Nz;
    a.Nrays = Nrays;
    a.x = F64View(r->x.data(), Nx);
    a.z = F64View(r->z.data(), Nz);
    a.mux = F64View(r->mux.data(), Nrays);
    a.muz = F64View(r->muz.data(), Nrays);
    a.temperature = F64View(r->temperature.data(), Nx * Nz);
    if (xLower && xUpper)
    {
        auto xbc = [&](AtmosphericBoundaryCondition* bc, const lwhip_boundary* b) {
            *bc = AtmosphericBoundaryCondition(CALLABLE, 1, b->Nmu, Nz, BcIdxs(const_cast<i32*>(b->idxs), Nrays, 2));
            bc->set_bc_data(F64View3D(const_cast<f64*>(b->bcData), 1, b->Nmu, Nz));
        };
        xbc(&a.xLowerBc, xLower);
        xbc(&a.xUpperBc, xUpper);
    }
    else
    {
        a.xLowerBc.type = PERIODIC;
        a.xUpperBc.type = PERIODIC;
    }
    auto bc = [](int t) { return t == LWHIP_BC_THERMALISED ? THERMALISED : ZERO; };
    a.zLowerBc.type = bc(zLowerBc);
    a.zUpperBc.type = bc(zUpperBc);
    build_intersection_list(&a);
    return r;
}

void lwref_2d_destroy(void* h) { delete (Ref2D*)h; }

void lwref_2d_counts(void* h, int* nLong, int* nSub)
{
    auto* r = (Ref2D*)h;
    *nLong = (int)r->atmos.intersections.substeps.size();
    int n = 0;
    for (auto& s : r->atmos.intersections.substeps)
        n += (int)s.steps.size();
    *nSub = n;
}

static lwhip_intersection flat(const IntersectionResult& q)
{
    lwhip_intersection o{};
    o.axis = (int)q.axis;
    o.fracZ = q.fractionalZ;
    o.fracX = q.fractionalX;
    o.distance = q.distance;
    return o;
}

void lwref_2d_stencils(void* h, lwhip_intersection* uw, lwhip_intersection* dw, int32_t* longIdx, int32_t* subOff,
                       lwhip_intersection* sub)
{
    auto* r = (Ref2D*)h;
    Atmosphere& a = r->atmos;
    size_t o = 0;
    for (int mu = 0; mu < a.Nrays; ++mu)
        for (int d = 0; d < 2; ++d)
            for (int k = 0; k < a.Nz; ++k)
                for (int j = 0; j < a.Nx; ++j, ++o)
                {
                    const auto& st = a.intersections.intersections(mu, d, k, j);
                    uw[o] = flat(st.uwIntersection);
                    dw[o] = flat(st.dwIntersection);
                    longIdx[o] = st.longCharIdx;
                }
    int n = 0;
    for (size_t l = 0; l < a.intersections.substeps.size(); ++l)
    {
        subOff[l] = n;
        for (auto& st : a.intersections.substeps[l].steps)
            sub[n++] = flat(st);
    }
    subOff[a.intersections.substeps.size()] = n;
}

int lwref_2d_besser(void* h, int mu, int toObs, double wav, const double* chi, const double* S, double* I, double* Psi)
{
    auto* r = (Ref2D*)h;
    LwInternal::FormalData fd;
    fd.atmos = &r->atmos;
    const int Ns = r->atmos.Nspace;
    fd.chi = F64View(const_cast<double*>(chi), Ns);
    fd.S = F64View(const_cast<double*>(S), Ns);
    fd.I = F64View(I, Ns);
    fd.Psi = F64View(Psi, Ns);
    fd.interp = LwInternal::interp_linear_2d;
    F64View1D wave(&wav, 1);
    LwInternal::piecewise_besser_2d(&fd, 0, mu, toObs != 0, wave);
    return LWHIP_OK;
}

// Ng acceleration: the reference's own struct Ng (Source/Ng.hpp)
void* lwref_ng_new(int Norder, int Nperiod, int Ndelay, const double* sol, int len)
{
    if (sol)
        return new Ng(Norder, Nperiod, Ndelay, F64View(const_cast<double*>(sol), len));
    return new Ng();
}

void lwref_ng_free(void* h) { delete (Ng*)h; }

int lwref_ng_accelerate(void* h, double* sol, int len)
{
    try
    {
        return ((Ng*)h)->accelerate(F64View(sol, len)) ? 1 : 0;
    }
    catch (const std::runtime_error& e)
    {
        return -1;
    }
}

void lwref_ng_max_change(void* h, double* dMax, int* idx)
{
    NgChange c = ((Ng*)h)->max_change();
    *dMax = c.dMax;
    *idx = (int)c.dMaxIdx;
}

void lwref_set_zplane(void* h, double* down, double* up)
{
    ((RefContext*)h)->zDown = down;
    ((RefContext*)h)->zUp = up;
}

// what LwContext.prd_redistribute passes as extraParams (Source/LwMiddleLayer.pyx:3678-3680)
void lwref_set_include_detailed(void* h, int on)
{
    ((RefContext*)h)->includeDetailed = on != 0;
}

// configure_hprd_coeffs (Source/Prd.cpp:697-946) on the real Context, and its tables flattened into the ABI's lwhip_hprd
// (arrays owned by the returned block; JRest points at the core's own spect.JRest, which the core's iterations fill).
// The iterations of this context run in hybrid-PRD mode from here on.
struct HprdExport
{
    lwhip_hprd h{};
    std::vector<int32_t> prdIdxs, hPrdIdxs, lineAtom, lineTrans;
    std::vector<int64_t> off;
    std::vector<lwhip_j_coeff> jc;
    std::vector<std::vector<lwhip_rho_coeff>> rho;
    std::vector<const lwhip_rho_coeff*> rhoPtr;
};

lwhip_hprd* lwref_hprd_export(void* hh, int includeDetailed)
{
    auto* rc = (RefContext*)hh;
    try
    {
        configure_hprd_coeffs(rc->ctx, includeDetailed != 0);
        // (the intensity cores take their JRest view and the lines' coefficient views when they are set up: regenerate
        // them, as LwContext.update_hprd_coeffs does, Source/LwMiddleLayer.pyx:3692-3705)
        rc->ctx.update_threads();
    }
    catch (const std::exception& e)
    {
        std::fprintf(stderr, "lwref: %s\n", e.what());
        return nullptr;
    }
    auto& spect = rc->spect;
    if (!spect.JRest)
        return nullptr;
    auto* ex = new HprdExport();
    const lwhip_problem* p = rc->prob;
    const int Ns = p->Nspace, Nrays = p->Nrays;
    ex->prdIdxs.assign(spect.prdIdxs.begin(), spect.prdIdxs.end());
    ex->hPrdIdxs.assign(spect.hPrdIdxs.begin(), spect.hPrdIdxs.end());
    const int NhPrd = (int)ex->hPrdIdxs.size();
    ex->off.reserve((size_t)NhPrd * Nrays * 2 * Ns + 1);
    for (int hq = 0; hq < NhPrd; ++hq)
        for (int mu = 0; mu < Nrays; ++mu)
            for (int toObs = 0; toObs < 2; ++toObs)
                for (int k = 0; k < Ns; ++k)
                {
                    ex->off.push_back((int64_t)ex->jc.size());
                    for (const auto& c : spect.JCoeffs(hq, mu, toObs, k))
                        ex->jc.push_back(lwhip_j_coeff{ c.frac, c.idx, 0 });
                }
    ex->off.push_back((int64_t)ex->jc.size());
    // the PRD lines in the order configure_hprd_coeffs walked them: active atoms, then (on request) detailed ones
    size_t tIdx = 0;
    std::vector<std::pair<int, int>> where; // (atom, kr) of every transition object, in creation order
    for (int a = 0; a < p->Natom; ++a)
        for (int kr = 0; kr < p->atoms[a].Ntrans; ++kr)
            where.push_back({ a, kr });
    (void)tIdx;
    for (int pass = 0; pass < (includeDetailed ? 2 : 1); ++pass)
        for (size_t q = 0; q < rc->transitions.size(); ++q)
        {
            Transition* t = rc->transitions[q].get();
            const int a = where[q].first;
            if ((p->atoms[a].detailed != 0) != (pass == 1) || !t->rhoPrd || !t->hPrdCoeffs)
                continue;
            ex->lineAtom.push_back(a);
            ex->lineTrans.push_back(where[q].second);
            const int nlt = (int)t->wavelength.shape(0);
            std::vector<lwhip_rho_coeff> v((size_t)nlt * Nrays * 2 * Ns);
            for (int lt = 0; lt < nlt; ++lt)
                for (int mu = 0; mu < Nrays; ++mu)
                    for (int toObs = 0; toObs < 2; ++toObs)
                        for (int k = 0; k < Ns; ++k)
                        {
                            const auto& c = t->hPrdCoeffs(lt, mu, toObs, k);
                            v[(((size_t)lt * Nrays + mu) * 2 + toObs) * Ns + k] = lwhip_rho_coeff{ c.i0, c.i1, c.frac };
                        }
            ex->rho.push_back(std::move(v));
        }
    for (auto& v : ex->rho)
        ex->rhoPtr.push_back(v.data());
    ex->h.NprdLambda = (int32_t)ex->prdIdxs.size();
    ex->h.NhPrd = NhPrd;
    ex->h.Nlines = (int32_t)ex->lineAtom.size();
    ex->h.prdIdxs = ex->prdIdxs.data();
    ex->h.hPrdIdxs = ex->hPrdIdxs.data();
    ex->h.JRest = spect.JRest.data();
    ex->h.jCoeffOff = ex->off.data();
    ex->h.jCoeffs = ex->jc.data();
    ex->h.lineAtom = ex->lineAtom.data();
    ex->h.lineTrans = ex->lineTrans.data();
    ex->h.rhoCoeffs = ex->rhoPtr.data();
    return &ex->h; // (first member: the block is recovered from this pointer)
}

void lwref_hprd_free(lwhip_hprd* hp)
{
    delete (HprdExport*)hp;
}

// redistribute_prd_lines (Source/Prd.cpp:648-653 -> PrdTemplates.hpp:175-351)
int lwref_redistribute_prd(void* h, int maxIter, double tol, int* NprdSubIter, int* NprdOut,
                           double* dRho, int* dRhoMaxIdx, double* dJPrdMax, int* dJPrdMaxIdx)
{
    auto* rc = (RefContext*)h;
    try
    {
        refresh_bcs(rc);
        ExtraParams params;
        if (rc->includeDetailed)
            params.insert("include_detailed_atoms", true);
        IterationResult r = redistribute_prd_lines(rc->ctx, maxIter, tol, params);
        if (NprdSubIter)
            *NprdSubIter = r.NprdSubIter;
        const int Nprd = r.NprdSubIter > 0 ? (int)(r.dRho.size() / r.NprdSubIter) : 0;
        if (NprdOut)
            *NprdOut = Nprd;
        for (size_t i = 0; i < r.dRho.size(); ++i)
        {
            if (dRho)
                dRho[i] = r.dRho[i];
            if (dRhoMaxIdx)
                dRhoMaxIdx[i] = r.dRhoMaxIdx[i];
        }
        for (size_t i = 0; i < r.dJPrdMax.size(); ++i)
        {
            if (dJPrdMax)
                dJPrdMax[i] = r.dJPrdMax[i];
            if (dJPrdMaxIdx)
                dJPrdMaxIdx[i] = r.dJPrdMaxIdx[i];
        }
    }
    catch (const std::exception& e)
    {
        std::fprintf(stderr, "lwref: %s\n", e.what());
        return LWHIP_ERR_INVALID;
    }
    return LWHIP_OK;
}

double lwref_GII(double aDamp, double qEmit, double qAbs);

// ---- leaf known-answer hooks (pin the restatement function by function) ---------------------

double lwref_GII(double aDamp, double qEmit, double qAbs)
{
    return PrdCores::GII(aDamp, qEmit, qAbs);
}

void lwref_w2(double dtau, double* w)
{
    LwInternal::w2(dtau, w);
}

void lwref_bezier3_coeffs(double dt, double* out)
{
    Bezier::Bezier3_coeffs(dt, &out[0], &out[1], &out[2], &out[3], &out[4]);
}

double lwref_cent_deriv(double dsuw, double dsdw, double yuw, double y0, double ydw)
{
    return Bezier::cent_deriv(dsuw, dsdw, yuw, y0, ydw);
}

double lwref_voigt_H(double a, double v)
{
    return voigt_H(a, v);
}

void lwref_planck_nu(int n, double* T, double lambda, double* Bnu)
{
    planck_nu(n, T, lambda, Bnu);
}

int lwref_solve_lin_eq(int N, double* A, double* b, int improve)
{
    try
    {
        solve_lin_eq(F64View2D(A, N, N), F64View(b, N), (bool)improve);
    }
    catch (const std::runtime_error& e)
    {
        return LWHIP_ERR_SINGULAR;
    }
    return LWHIP_OK;
}

// single-ray formal solve with the reference's own solver on caller-provided chi, S
// (kernel-level parity for the depth sweep). Psi may be NULL.
int lwref_formal_solver_ray(void* h, int la, int mu, int toObs,
                            double* chi, double* S, double* I, double* Psi)
{
    auto* rc = (RefContext*)h;
    LwInternal::FormalData fd;
    const int Nspace = rc->atmos.Nspace;
    fd.atmos = &rc->atmos;
    fd.chi = F64View(chi, Nspace);
    fd.S = F64View(S, Nspace);
    fd.I = F64View(I, Nspace);
    if (Psi)
        fd.Psi = F64View(Psi, Nspace);
    refresh_bcs(rc);
    rc->ctx.formalSolver.solver(&fd, la, mu, (bool)toObs, rc->spect.wavelength);
    return LWHIP_OK;
}
}
