// lwhip_tables.hip -- host side of the C ABI (lwhip_host.h): validation of the problem descriptor and the per-wavelength
// activity tables the sweep kernels walk (tiles, slots, chunks, the lane sweep's flat records, PRD wavelength lists).
#include "lwhip_host.h"

namespace lwhip
{
int validate(const lwhip_problem* p, std::string& why)
{
    if (!p)
    {
        why = "null problem";
        return LWHIP_ERR_INVALID;
    }
    if (p->abiVersion != LWHIP_ABI_VERSION)
    {
        why = "ABI version mismatch";
        return LWHIP_ERR_INVALID;
    }
    if (p->Nspace < 3 || p->Nrays < 1 || p->Nlambda < 1 || p->Natom < 0)
    {
        why = "need Nspace >= 3, Nrays >= 1, Nlambda >= 1";
        return LWHIP_ERR_INVALID;
    }
    if (p->grid2d)
    {
        const lwhip_grid2d& g = *p->grid2d;
        if (g.Nx < 2 || g.Nz < 2 || g.Nx > 1024 || g.Nx * g.Nz != p->Nspace || g.Nrays != p->Nrays)
        {
            why = "2D grid: 2 <= Nx <= 1024, Nz >= 2, Nx * Nz = Nspace, Nrays matching";
            return LWHIP_ERR_UNSUPPORTED;
        }
        if ((g.zLowerBc == LWHIP_BC_CALLABLE) != (p->zLowerBc.type == LWHIP_BC_CALLABLE)
            || (g.zUpperBc == LWHIP_BC_CALLABLE) != (p->zUpperBc.type == LWHIP_BC_CALLABLE))
        {
            why = "2D grid: a CALLABLE z boundary needs lwhip_problem.zLowerBc / zUpperBc of type CALLABLE (bcData [Nlambda, Nmu, Nx])";
            return LWHIP_ERR_INVALID;
        }
        for (int bt : { g.zLowerBc, g.zUpperBc })
            if (bt != LWHIP_BC_ZERO && bt != LWHIP_BC_THERMALISED && bt != LWHIP_BC_CALLABLE)
            {
                why = "2D grid: z boundaries are ZERO, THERMALISED or CALLABLE";
                return LWHIP_ERR_UNSUPPORTED;
            }
        if (!g.periodic)
        {
            // "Mixed boundary types not supported on x-axis!" (Source/FormalScalar2d.cpp:750-759)
            auto ok = [&](const lwhip_boundary* b) {
                return b && b->type == LWHIP_BC_CALLABLE && b->idxs && b->bcData && b->Nmu > 0;
            };
            if (!ok(g.xLowerBc) || !ok(g.xUpperBc) || g.NlongChar != 0)
            {
                why = "2D grid without periodic x: both x boundaries CALLABLE (idxs, bcData [Nlambda, Nmu, Nz]), no long characteristics";
                return LWHIP_ERR_UNSUPPORTED;
            }
            for (int q = 0; q < 2 * g.Nrays; ++q)
                if (g.xLowerBc->idxs[q] < 0 || g.xLowerBc->idxs[q] >= g.xLowerBc->Nmu || g.xUpperBc->idxs[q] < 0
                    || g.xUpperBc->idxs[q] >= g.xUpperBc->Nmu)
                {
                    why = "2D grid: x boundary idxs out of range (\"Error in boundary condition indexing\")";
                    return LWHIP_ERR_INVALID;
                }
        }
        if (!g.mux || !g.uw || !g.dw || !g.longCharIdx || !g.substepOff || (g.NlongChar > 0 && !g.substeps))
        {
            why = "2D grid: incomplete intersection table";
            return LWHIP_ERR_INVALID;
        }
        if (!fs2d_long_chars_ok(&g))
        {
            why = "2D grid: a long characteristic that does not end on a z plane";
            return LWHIP_ERR_UNSUPPORTED;
        }
        if (p->storeDepthData && (!p->depthChi || !p->depthEta || !p->depthI))
        {
            why = "storeDepthData set but depth arrays are null";
            return LWHIP_ERR_INVALID;
        }
    }
    else if (p->Nspace > 1024)
    {
        why = "Nspace > 1024 is not supported by the 1D sweep kernel";
        return LWHIP_ERR_UNSUPPORTED;
    }
    if (!p->grid2d && (p->formalSolver < LWHIP_FS_LINEAR_1D || p->formalSolver > LWHIP_FS_BEZIER3_1D))
    {
        why = "unknown formal solver";
        return LWHIP_ERR_INVALID;
    }
    if (!p->height || !p->temperature || !p->muz || !p->wmu || !p->wavelength || !p->bgChi
        || !p->bgEta || !p->bgSca || !p->J || !p->I)
    {
        why = "null atmosphere / spectrum / background array";
        return LWHIP_ERR_INVALID;
    }
    if (p->storeDepthData && (!p->depthChi || !p->depthEta || !p->depthI))
    {
        why = "storeDepthData set but depth arrays are null";
        return LWHIP_ERR_INVALID;
    }
    const lwhip_boundary* bcs[2] = { &p->zLowerBc, &p->zUpperBc };
    for (auto* b : bcs)
    {
        if (b->type == LWHIP_BC_CALLABLE && (!b->idxs || !b->bcData || b->Nmu < 1))
        {
            why = "CALLABLE boundary needs idxs, bcData, Nmu";
            return LWHIP_ERR_INVALID;
        }
    }
    bool seenDetailed = false;
    for (int ia = 0; ia < p->Natom; ++ia)
    {
        const lwhip_atom& a = p->atoms[ia];
        if (a.Nlevel < 1 || a.Nlevel > 32 || a.Ntrans < 0)
        {
            why = "atom needs 1 <= Nlevel <= 32";
            return a.Nlevel > 32 ? LWHIP_ERR_UNSUPPORTED : LWHIP_ERR_INVALID;
        }
        if (a.detailed)
            seenDetailed = true;
        else if (seenDetailed)
        {
            why = "active atoms must precede detailed atoms";
            return LWHIP_ERR_INVALID;
        }
        if (!a.n || !a.nStar || !a.nTotal || !a.vBroad || (!a.detailed && !a.Gamma))
        {
            why = "null atom array";
            return LWHIP_ERR_INVALID;
        }
        for (int kr = 0; kr < a.Ntrans; ++kr)
        {
            const lwhip_transition& t = a.trans[kr];
            if (t.i < 0 || t.j < 0 || t.i >= a.Nlevel || t.j >= a.Nlevel || t.i == t.j)
            {
                why = "bad transition levels";
                return LWHIP_ERR_INVALID;
            }
            if (t.Nblue < 0 || t.Nred > p->Nlambda || t.Nred - t.Nblue < 2)
            {
                why = "bad transition wavelength range (need >= 2 points inside the grid)";
                return LWHIP_ERR_INVALID;
            }
            if (!t.wavelength || !t.Rij || !t.Rji)
            {
                why = "null transition array";
                return LWHIP_ERR_INVALID;
            }
            if (t.type == LWHIP_LINE)
            {
                if (!t.phi || !t.wphi)
                {
                    why = "line without phi/wphi";
                    return LWHIP_ERR_INVALID;
                }
                if (t.prd && !t.rhoPrd)
                {
                    why = "prd line without rhoPrd";
                    return LWHIP_ERR_INVALID;
                }
            }
            else if (t.type == LWHIP_CONTINUUM)
            {
                if (!t.alpha)
                {
                    why = "continuum without alpha";
                    return LWHIP_ERR_INVALID;
                }
            }
            else
            {
                why = "unknown transition type";
                return LWHIP_ERR_INVALID;
            }
        }
    }
    return LWHIP_OK;
}

// Everything the structure tables depend on, folded into 64 bits: sizes, solver, shard, the wavelength grid, and of every
// transition its levels, wavelength range, Einstein coefficients, own grid and cross-sections.  Two contexts with the same
// fingerprint build the same tables (populations, atmosphere, profiles, boundary data live elsewhere).
uint64_t structure_signature(const lwhip_context* c)
{
    uint64_t h = 0x9E3779B97F4A7C15ull;
    auto mix = [&](uint64_t v) {
        h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h *= 0xFF51AFD7ED558CCDull;
        h ^= h >> 33;
    };
    auto mixd = [&](double v) {
        uint64_t u;
        std::memcpy(&u, &v, 8);
        mix(u);
    };
    auto mixv = [&](const double* p, size_t n) {
        mix(n);
        for (size_t i = 0; p && i < n; ++i)
            mixd(p[i]);
    };
    const lwhip_problem& p = c->prob;
    for (int64_t v : { (int64_t)p.Nlambda, (int64_t)p.Nspace, (int64_t)p.Nrays, (int64_t)p.Natom, (int64_t)p.formalSolver,
                       (int64_t)c->laStart, (int64_t)c->laEnd, (int64_t)c->worldSize, (int64_t)c->batchHint, (int64_t)c->prdDetailed,
                       (int64_t)c->deterministic, (int64_t)(c->hprd != nullptr), (int64_t)c->is2d, (int64_t)p.storeDepthData,
                       (int64_t)c->device })
        mix((uint64_t)v);
    mixv(p.wavelength, (size_t)p.Nlambda);
    for (int ia = 0; ia < p.Natom; ++ia)
    {
        const lwhip_atom& a = c->atoms[ia];
        mix((uint64_t)a.Nlevel);
        mix((uint64_t)a.Ntrans);
        mix((uint64_t)(a.detailed != 0));
        mix((uint64_t)(a.C != nullptr));
        for (int kr = 0; kr < a.Ntrans; ++kr)
        {
            const lwhip_transition& t = a.trans[kr];
            for (int64_t v : { (int64_t)t.type, (int64_t)t.i, (int64_t)t.j, (int64_t)t.Nblue, (int64_t)t.Nred, (int64_t)(t.prd != 0),
                               (int64_t)(t.rhoPrd != nullptr) })
                mix((uint64_t)v);
            mixd(t.Aji);
            mixd(t.Bji);
            mixd(t.Bij);
            mixd(t.lambda0);
            mixd(t.dopplerWidth);
            mixv(t.wavelength, (size_t)std::max(t.Nred - t.Nblue, 0));
            if (t.type != LWHIP_LINE)
                mixv(t.alpha, (size_t)std::max(t.Nred - t.Nblue, 0));
        }
    }
    return h;
}

int build_tables(lwhip_context* c)
{
    // (lwhip_create_like: the donor of the structure tables, checked to have this context's fingerprint)
    const lwhip_context* like = c->tablesFrom;
    // LWHIP_CREATE_TIMING=1: where the host time of the tables goes
    const bool timing = std::getenv("LWHIP_CREATE_TIMING") != nullptr;
    auto tPrev = std::chrono::steady_clock::now();
    auto tick = [&](const char* what) {
        if (!timing)
            return;
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "  build_tables: %-28s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - tPrev).count());
        tPrev = t;
    };
    const int Ns = c->Ns;
    // ---- global transition list, level rows, pools ------------------------------------------
    c->levelOff.assign(c->Natom + 1, 0);
    c->atomTrOff.assign(c->Natom + 1, 0);
    c->gammaOff.assign(c->Natom, 0);
    c->gammaTot = 0;
    for (int ia = 0; ia < c->Natom; ++ia)
    {
        const lwhip_atom& a = c->atoms[ia];
        c->levelOff[ia + 1] = c->levelOff[ia] + a.Nlevel;
        c->atomTrOff[ia + 1] = c->atomTrOff[ia] + a.Ntrans;
        c->gammaOff[ia] = c->gammaTot;
        if (!a.detailed)
            c->gammaTot += (int64_t)a.Nlevel * a.Nlevel * Ns;
    }
    c->NlevTot = c->levelOff[c->Natom];
    c->Ntrans = c->atomTrOff[c->Natom];
    // ---- hybrid PRD: the tables of configure_hprd_coeffs (lwhip_options.hprd) onto the device ----------------------
    c->hLa2prdHost.clear();
    c->hLa2hHost.clear();
    c->hRhoOffHost.assign(c->Ntrans, -1);
    if (c->hprd)
    {
        const lwhip_hprd& H = *c->hprd;
        const int Nlam = c->prob.Nlambda, Nr = c->Nrays;
        // On a wavelength shard (round 4) every rank keeps rho of the WHOLE grid of each PRD line (a ray's rho is interpolated
        // between neighbouring wavelengths of the line, which may belong to another shard) and evaluates the scattering
        // integral for all of them; JRest is additive over the wavelengths (Prd.cpp:384-389), so each rank's partial sums
        // ride in the all-reduce of the PRD gather buffer (lwhip_prd_pack).
        if (c->is2d)
            return fail(LWHIP_ERR_UNSUPPORTED, "hybrid PRD: 1D contexts only");
        if (H.NprdLambda < 1 || H.Nlines < 1 || !H.prdIdxs || !H.JRest || !H.lineAtom || !H.lineTrans || !H.rhoCoeffs
            || (H.NhPrd > 0 && (!H.hPrdIdxs || !H.jCoeffOff || !H.jCoeffs)))
            return fail(LWHIP_ERR_INVALID, "hybrid PRD: incomplete lwhip_hprd");
        c->hLa2prdHost.assign(Nlam, -1);
        c->hLa2hHost.assign(Nlam, -1);
        for (int i = 0; i < H.NprdLambda; ++i)
        {
            if (H.prdIdxs[i] < 0 || H.prdIdxs[i] >= Nlam)
                return fail(LWHIP_ERR_INVALID, "hybrid PRD: prdIdxs out of range");
            c->hLa2prdHost[H.prdIdxs[i]] = i;
        }
        for (int i = 0; i < H.NhPrd; ++i)
        {
            if (H.hPrdIdxs[i] < 0 || H.hPrdIdxs[i] >= Nlam)
                return fail(LWHIP_ERR_INVALID, "hybrid PRD: hPrdIdxs out of range");
            c->hLa2hHost[H.hPrdIdxs[i]] = i;
        }
        std::vector<lwhip_rho_coeff> rho;
        for (int q = 0; q < H.Nlines; ++q)
        {
            const int ia = H.lineAtom[q], kr = H.lineTrans[q];
            if (ia < 0 || ia >= c->Natom || kr < 0 || kr >= c->atoms[ia].Ntrans || !H.rhoCoeffs[q])
                return fail(LWHIP_ERR_INVALID, "hybrid PRD: bad line list");
            const lwhip_transition& t = c->atoms[ia].trans[kr];
            if (t.type != LWHIP_LINE || !t.prd || !t.rhoPrd)
                return fail(LWHIP_ERR_INVALID, "hybrid PRD: a listed transition is not a PRD line");
            const int nlt = t.Nred - t.Nblue;
            const size_t n = (size_t)nlt * Nr * 2 * Ns;
            for (size_t i = 0; i < n; ++i)
                if (H.rhoCoeffs[q][i].i0 < 0 || H.rhoCoeffs[q][i].i1 >= nlt || H.rhoCoeffs[q][i].i0 > H.rhoCoeffs[q][i].i1)
                    return fail(LWHIP_ERR_INVALID, "hybrid PRD: rho interpolation index outside the line's grid");
            c->hRhoOffHost[c->atomTrOff[ia] + kr] = (int64_t)rho.size();
            rho.insert(rho.end(), H.rhoCoeffs[q], H.rhoCoeffs[q] + n);
        }
        rho.resize(rho.size() + 64, lwhip_rho_coeff{ 0, 0, 0.0 }); // (a ray's last block of depths reads past its row)
        HIP_TRY(c->hRho.upload(rho));
        std::vector<int32_t> l2h(c->hLa2hHost.begin() + c->laStart, c->hLa2hHost.begin() + c->laEnd); // (indexed by the shard's wavelength)
        HIP_TRY(c->hLa2h.upload(l2h));
        if (H.NhPrd > 0)
        {
            const size_t ncell = (size_t)H.NhPrd * Nr * 2 * Ns;
            std::vector<int64_t> off(H.jCoeffOff, H.jCoeffOff + ncell + 1);
            off.resize(off.size() + 8, off.back());
            const int64_t nj = off[ncell];
            for (int64_t e = 0; e < nj; ++e)
                if (H.jCoeffs[e].idx < 0 || H.jCoeffs[e].idx >= H.NprdLambda)
                    return fail(LWHIP_ERR_INVALID, "hybrid PRD: JCoeffs row outside JRest");
            std::vector<lwhip_j_coeff> jc(H.jCoeffs, H.jCoeffs + nj);
            jc.resize(jc.size() + 1, lwhip_j_coeff{ 0.0, 0, 0 });
            HIP_TRY(c->hJOff.upload(off));
            HIP_TRY(c->hJCoef.upload(jc));
        }
        HIP_TRY(c->JRest.alloc_zero((size_t)H.NprdLambda * Ns)); // (accumulated into by atomics)
    }
    c->trans.clear();
    c->Nline = c->Ncont = 0;
    c->phiTot = c->rhoTot = c->parTot = 0;
    for (int ia = 0; ia < c->Natom; ++ia)
    {
        const lwhip_atom& a = c->atoms[ia];
        for (int kr = 0; kr < a.Ntrans; ++kr)
        {
            HostTrans h;
            h.t = a.trans[kr];
            h.atom = ia;
            const int b = std::max(h.t.Nblue, c->laStart);
            const int r = std::min(h.t.Nred, c->laEnd);
            if (r > b)
            {
                h.NblueLoc = b - c->laStart;
                h.NredLoc = r - c->laStart;
                h.ltStart = b - h.t.Nblue;
            }
            else
            {
                h.NblueLoc = h.NredLoc = 0;
                h.ltStart = 0;
            }
            const int64_t nlt = h.NredLoc - h.NblueLoc;
            h.parOff = c->parTot;
            c->parTot += 4 * nlt;
            h.phiOff = h.rhoOff = -1;
            if (h.t.type == LWHIP_LINE)
            {
                h.row = c->Nline++;
                h.phiOff = c->phiTot;
                c->phiTot += nlt * c->Nrays * 2 * Ns;
                h.rhoLt0 = h.ltStart;
                h.rhoRows = (int)nlt;
                if (h.t.prd)
                {
                    if (c->hprd && c->hRhoOffHost[c->trans.size()] >= 0)
                    {
                        // hybrid PRD: rho of the line's whole grid on every shard
                        h.rhoLt0 = 0;
                        h.rhoRows = h.t.Nred - h.t.Nblue;
                    }
                    h.rhoOff = c->rhoTot;
                    c->rhoTot += (int64_t)h.rhoRows * Ns;
                }
            }
            else
            {
                h.row = c->Ncont++;
            }
            c->trans.push_back(h);
        }
    }

    tick("transition list, hybrid PRD");
    // ---- per-(transition, lt) params -----------------------------------------------------------
    std::vector<double> par((size_t)c->parTot, 0.0);
    for (auto& h : c->trans)
    {
        const lwhip_transition& t = h.t;
        for (int l = 0; l < h.NredLoc - h.NblueLoc; ++l)
        {
            const int lt = h.ltStart + l;
            double* q = &par[(size_t)h.parOff + 4 * l];
            const double wlambda = trans_wlambda(t, lt);
            if (t.type == LWHIP_LINE)
            {
                const double hnu_4pi = HC_4PI * (t.lambda0 / t.wavelength[lt]);
                q[0] = hnu_4pi * t.Bij;
                q[1] = wlambda;
                q[2] = t.Bji / t.Bij;
                q[3] = t.Aji / t.Bji;
            }
            else
            {
                const double wl = t.wavelength[lt];
                q[0] = t.alpha[lt];
                q[1] = (wlambda / wl) * PI4_H;
                q[2] = TWO_HC_NM3 / (wl * wl * wl);
                q[3] = 0.0;
            }
        }
    }

    tick("per-transition parameters");
    // ---- per-wavelength activity: [lines][mixed continua][pure continua], row programs --------------
    std::vector<DevLaHeader> hdr(c->Nla);
    std::vector<DevSlot> slots;
    std::vector<int> slotTr; // global transition of each slot record
    std::vector<DevProgram> progs;
    std::vector<DevProgRow> progRows;
    std::vector<DevProgEnt> progEnts;
    std::map<std::vector<int>, int> progIds;
    std::vector<int32_t> contLaHost;
    c->rowsTot = 0;
    c->maxL = c->maxC = c->maxM = c->maxP = 0;
    const int rowEtaA = 2, rowLevChi = 2 + c->Natom, rowLevU = 2 + c->Natom + c->NlevTot;
    for (int la = 0; la < c->Nla; ++la)
    {
        std::vector<int> lines, conts;
        for (int tr = 0; tr < c->Ntrans; ++tr)
        {
            const HostTrans& h = c->trans[tr];
            if (la < h.NblueLoc || la >= h.NredLoc)
                continue;
            (h.t.type == LWHIP_LINE ? lines : conts).push_back(tr);
        }
        auto isOp = [&](int tr) { return !c->atoms[c->trans[tr].atom].detailed; };
        // moment slots: the first two active atoms that have an active line here
        std::vector<int> momentAtoms;
        for (int tr : lines)
            if (isOp(tr) && std::find(momentAtoms.begin(), momentAtoms.end(), c->trans[tr].atom) == momentAtoms.end())
                momentAtoms.push_back(c->trans[tr].atom);
        auto msOf = [&](int atom) {
            for (int m = 0; m < (int)momentAtoms.size() && m < 2; ++m)
                if (momentAtoms[m] == atom)
                    return m;
            return -1;
        };
        auto atomHasLines = [&](int atom) {
            return std::find(momentAtoms.begin(), momentAtoms.end(), atom) != momentAtoms.end();
        };
        std::vector<int> mixed, pure;
        for (int tr : conts)
        {
            const HostTrans& h = c->trans[tr];
            bool isMixed = false;
            if (isOp(tr))
            {
                for (int lt : lines)
                {
                    const HostTrans& l = c->trans[lt];
                    if (l.atom != h.atom || !isOp(lt))
                        continue;
                    if (l.t.i == h.t.i || l.t.i == h.t.j || l.t.j == h.t.i || l.t.j == h.t.j)
                        isMixed = true;
                }
                if (atomHasLines(h.atom) && msOf(h.atom) < 0)
                    isMixed = true;
            }
            (isMixed ? mixed : pure).push_back(tr);
        }
        std::vector<int> order = lines;
        order.insert(order.end(), mixed.begin(), mixed.end());
        order.insert(order.end(), pure.begin(), pure.end());
        hdr[la].off = (int32_t)slots.size();
        hdr[la].nLine = (int16_t)lines.size();
        hdr[la].nMixed = (int16_t)mixed.size();
        hdr[la].nPure = (int16_t)pure.size();
        hdr[la].hasPrd = 0;
        if (c->hprd && c->hprd->NhPrd > 0)
            hdr[la].hasPrd = c->hLa2hHost[c->laStart + la] >= 0 ? 1 : 0; // hybrid PRD: the wavelengths that scatter into the PRD region (PrdTemplates.hpp:234-248)
        else
        for (int tr : lines)
            if (c->trans[tr].t.prd && c->trans[tr].rhoOff >= 0 && (isOp(tr) || c->prdDetailed))
                hdr[la].hasPrd = 1; // a wavelength the PRD rates pass visits (PrdTemplates.hpp:226-239)
        for (int tr : order)
        {
            const HostTrans& h = c->trans[tr];
            const int l = la - h.NblueLoc;
            DevSlot sl{};
            sl.cs = 0;
            sl.gi = c->levelOff[h.atom] + h.t.i;
            sl.gj = c->levelOff[h.atom] + h.t.j;
            sl.atom = h.atom;
            sl.flags = (isOp(tr) ? 1 : 0) | ((h.t.type == LWHIP_LINE && h.t.prd) ? 2 : 0);
            sl.row = h.row;
            sl.ms = isOp(tr) ? msOf(h.atom) : -1;
            for (int q = 0; q < 4; ++q)
                sl.p[q] = par[(size_t)h.parOff + 4 * l + q];
            sl.phiOff = (h.t.type == LWHIP_LINE) ? h.phiOff + (int64_t)l * c->Nrays * 2 * Ns : -1;
            sl.rhoOff = (h.rhoOff >= 0) ? h.rhoOff + (int64_t)(h.ltStart - h.rhoLt0 + l) * Ns : -1;
            slots.push_back(sl);
            slotTr.push_back(tr);
        }
        {
            // position q of the reference's transition order among this wavelength's continua -> ordinal in the
            // [mixed][pure] slot order (kept in the q-th continuum record's _pad)
            std::vector<int> contOrd = mixed;
            contOrd.insert(contOrd.end(), pure.begin(), pure.end());
            std::vector<int> byTrQ(contOrd.size());
            for (size_t q = 0; q < byTrQ.size(); ++q)
                byTrQ[q] = (int)q;
            std::sort(byTrQ.begin(), byTrQ.end(), [&](int x, int y) { return contOrd[x] < contOrd[y]; });
            for (size_t q = 0; q < byTrQ.size(); ++q)
                slots[hdr[la].off + lines.size() + q]._pad = byTrQ[q];
        }
        c->maxL = std::max(c->maxL, (int)lines.size());
        c->maxC = std::max(c->maxC, (int)conts.size());
        c->maxM = std::max(c->maxM, (int)mixed.size());
        c->maxP = std::max(c->maxP, (int)pure.size());

        // row program of this wavelength's continuum set (deduplicated by signature)
        std::vector<int> contOrder = mixed;
        contOrder.insert(contOrder.end(), pure.begin(), pure.end());
        std::vector<int> rowsNeeded = { 0, 1 };
        for (int tr : order)
        {
            if (!isOp(tr))
                continue;
            const HostTrans& h = c->trans[tr];
            const int gi = c->levelOff[h.atom] + h.t.i, gj = c->levelOff[h.atom] + h.t.j;
            for (int r : { rowEtaA + h.atom, rowLevChi + gi, rowLevChi + gj, rowLevU + gi, rowLevU + gj })
                if (std::find(rowsNeeded.begin(), rowsNeeded.end(), r) == rowsNeeded.end())
                    rowsNeeded.push_back(r);
        }
        std::sort(rowsNeeded.begin(), rowsNeeded.end());
        std::vector<int> key = contOrder;
        key.push_back(-1);
        key.insert(key.end(), rowsNeeded.begin(), rowsNeeded.end());
        auto it = progIds.find(key);
        if (it == progIds.end())
        {
            DevProgram pg{};
            pg.rowOff = (int32_t)progRows.size();
            pg.entOff = (int32_t)progEnts.size();
            // continua contribute in reference (transition index) order
            std::vector<int> byTr(contOrder.size());
            for (size_t q = 0; q < byTr.size(); ++q)
                byTr[q] = (int)q;
            std::sort(byTr.begin(), byTr.end(), [&](int x, int y) { return contOrder[x] < contOrder[y]; });
            for (int r : rowsNeeded)
            {
                DevProgRow pr{};
                pr.row = (int16_t)r;
                pr.entOff = (int32_t)(progEnts.size() - pg.entOff);
                for (int cq : byTr)
                {
                    const int tr = contOrder[cq];
                    const HostTrans& h = c->trans[tr];
                    const int gi = c->levelOff[h.atom] + h.t.i, gj = c->levelOff[h.atom] + h.t.j;
                    const bool op = isOp(tr);
                    if (r == 0)
                        progEnts.push_back(DevProgEnt{ (int16_t)cq, 0 });
                    else if (r == 1)
                        progEnts.push_back(DevProgEnt{ (int16_t)cq, 3 });
                    else if (op && r == rowEtaA + h.atom)
                        progEnts.push_back(DevProgEnt{ (int16_t)cq, 3 });
                    else if (op && r >= rowLevChi && r < rowLevU)
                    {
                        if (r == rowLevChi + gi)
                            progEnts.push_back(DevProgEnt{ (int16_t)cq, 0 });
                        if (r == rowLevChi + gj)
                            progEnts.push_back(DevProgEnt{ (int16_t)cq, 1 });
                    }
                    else if (op && r == rowLevU + gj)
                        progEnts.push_back(DevProgEnt{ (int16_t)cq, 2 });
                }
                pr.nEnt = (int16_t)(progEnts.size() - pg.entOff - pr.entOff);
                progRows.push_back(pr);
            }
            pg.nRows = (int32_t)(progRows.size() - pg.rowOff);
            pg.nEnt = (int32_t)(progEnts.size() - pg.entOff);
            if (pg.nRows > LWHIP_MAX_PROG_ROWS || pg.nEnt > LWHIP_MAX_PROG_ENTS || progs.size() >= 32767)
                return fail(LWHIP_ERR_UNSUPPORTED, "continuum row program exceeds the kernel's LDS tables");
            it = progIds.emplace(key, (int)progs.size()).first;
            progs.push_back(pg);
        }
        hdr[la].prog = (int16_t)it->second;
        // block of this wavelength in the continuum-row buffer (march kernel): the program's rows in
        // rowsNeeded order, then Vji of each continuum
        const int nContHere = (int)contOrder.size();
        hdr[la].nRows = (int32_t)rowsNeeded.size();
        hdr[la].rowBase = -1;
        if (nContHere > 0)
        {
            hdr[la].rowBase = (int32_t)c->rowsTot;
            c->rowsTot += (int64_t)rowsNeeded.size() + nContHere;
            contLaHost.push_back(la);
        }
        auto dense = [&](int r) -> int16_t {
            if (nContHere == 0)
                return -1;
            auto f = std::find(rowsNeeded.begin(), rowsNeeded.end(), r);
            return f == rowsNeeded.end() ? (int16_t)-1 : (int16_t)(f - rowsNeeded.begin());
        };
        for (size_t e = 0; e < order.size(); ++e)
        {
            DevSlot& sl = slots[hdr[la].off + e];
            const HostTrans& ht = c->trans[order[e]];
            const int gi = c->levelOff[ht.atom] + ht.t.i, gj = c->levelOff[ht.atom] + ht.t.j;
            sl.rEtaA = dense(rowEtaA + ht.atom);
            sl.rChiI = dense(rowLevChi + gi);
            sl.rChiJ = dense(rowLevChi + gj);
            sl.rUI = dense(rowLevU + gi);
            sl.rUJ = dense(rowLevU + gj);
            sl.rVji = (int16_t)((int)e >= (int)lines.size() ? (int)rowsNeeded.size() + ((int)e - (int)lines.size()) : -1);
            if (!isOp(order[e]))
                sl.rEtaA = sl.rChiI = sl.rChiJ = sl.rUI = sl.rUJ = -1;
        }
    }
    if (c->maxL > 8)
        return fail(LWHIP_ERR_UNSUPPORTED, "more than 8 lines overlap at one wavelength");
    if (c->is2d && c->maxM > 4)
        return fail(LWHIP_ERR_UNSUPPORTED, "2D: more than 4 continua coupled to the lines of one wavelength (rates2d_kernel's MAXM)");
    if (c->maxC < 1)
        c->maxC = 1;

    int numCU = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0)
        numCU = prop.multiProcessorCount;

    tick("wavelength activity, programs");
    // ---- tiles of the ray-column sweep (lwhip_raymarch.hip) -----------------------------------------------------
    // A tile = up to L consecutive wavelengths with the same slot records (same transitions in the same order, same
    // row program): one wavefront marches it, lane = (mu, wavelength).
    c->tiled = !c->is2d;
    if (c->tiled && (c->Nrays > 64 || Ns < 3))
        return fail(LWHIP_ERR_UNSUPPORTED, "the sweep needs Nrays <= 64 and Nspace >= 3");
    std::vector<DevTile> tiles;
    std::vector<int32_t> laneChunkSplit, laneChunkOrder; // lane sweep: the tail's shorter chunks (split factor, dispatch order)
    std::vector<DevTileSlot> tslots, tslotsPrd;
    std::vector<DevTileCopy> tcopies;
    std::vector<int32_t> tileRemap;
    std::vector<int32_t> chunkTile, chunkTilePrd, tileListPrd, tileSlotTr, tileSlotTrPrd, postChunkTile, postChunkTilePrd,
        postSlotTr, postCs(std::max<size_t>(slots.size(), 1), 0);
    std::function<bool(int, DevPostProg&, bool*)> lanePostProgram;
    std::vector<int> laneCsPure; // lane sweep: accumulator slot of every pure continuum inside its workgroup chunk
    if (c->tiled)
    {
        const int Nmu = c->Nrays;
        // The depth-across-lanes sweep serves the tiles whose line / mixed-continuum counts have a compiled kind and whose
        // pure continua fit its register program; anything else (and LWHIP_SWEEP=march) runs the ray-column march.
        c->laneSweep = false;
        // The pure continua of a wavelength as the lane sweep integrates them (DevPostProg): which level sums need a
        // register ("shared": several contributors, or a foreign one), which are the continuum's own term or zero.
        // Two encodings of the program words: the NARROW one (4 bits per field, up to LWHIP_POST_SUMS = 8 shared sums: one- and
        // two-atom problems; the finish's fast forms recognise its common words) and the WIDE one (round 5: 8 / 6 bits per field,
        // up to LWHIP_POST_SUMS_WIDE = 14 shared sums -- continua of up to four active atoms at one wavelength), chosen per
        // tile (DevLaneTile::finFast = 2).
        auto post_program = [&](int la, DevPostProg& pg, bool* wideOut) -> bool {
            pg = DevPostProg{};
            if (wideOut)
                *wideOut = false;
            const DevLaHeader& h = hdr[la];
            const int base = h.off + h.nLine, nC = h.nMixed + h.nPure;
            if (nC > LWHIP_POST_MAXC)
                return false;
            // position q of the reference's transition order -> the continuum's slot record
            auto ordOf = [&](int q) { return slots[base + q]._pad; };
            auto rec = [&](int q) -> const DevSlot& { return slots[base + ordOf(q)]; };
            struct Contrib { int q, kind; };
            const int colChi = c->Natom, colU = c->Natom + c->NlevTot;
            auto contributors = [&](int col) {
                std::vector<Contrib> v;
                for (int q = 0; q < nC; ++q)
                {
                    const DevSlot& sl = rec(q);
                    if (!(sl.flags & 1))
                        continue;
                    if (col == sl.atom)
                        v.push_back({ q, 4 });
                    if (col == colChi + sl.gi)
                        v.push_back({ q, 1 });
                    if (col == colChi + sl.gj)
                        v.push_back({ q, 2 });
                    if (col == colU + sl.gj)
                        v.push_back({ q, 3 });
                }
                return v;
            };
            // per continuum: the entry each of its four terms (+chi, -chi, U, eta) is added to (0: none, 1 .. : a shared sum), and
            // the five operands of a pure continuum (0: zero, 1 .. : a shared sum, -1 .. -4: its own +chi, -chi, U, eta)
            int target[LWHIP_POST_MAXC][4] = {};
            int operand[LWHIP_POST_MAXC][5] = {};
            std::map<int, int> sumOf;
            for (int q = 0; q < nC; ++q)
            {
                const DevSlot& sl = rec(q);
                if (ordOf(q) < h.nMixed || !(sl.flags & 1))
                    continue; // mixed continua are integrated ray by ray; detailed atoms have rates only
                const int cols[5] = { sl.atom, colChi + sl.gi, colU + sl.gj, colChi + sl.gj, colU + sl.gi };
                for (int w = 0; w < 5; ++w)
                {
                    const auto v = contributors(cols[w]);
                    int sel = 0;
                    if (v.size() == 1 && v[0].q == q)
                        sel = -v[0].kind;
                    else if (!v.empty())
                    {
                        auto it = sumOf.find(cols[w]);
                        if (it == sumOf.end())
                        {
                            if ((int)sumOf.size() >= LWHIP_POST_SUMS_WIDE)
                                return false;
                            const int m = (int)sumOf.size();
                            it = sumOf.emplace(cols[w], m).first;
                            for (const auto& cb : v)
                                target[cb.q][cb.kind - 1] = m + 1;
                        }
                        sel = it->second + 1;
                    }
                    operand[q][w] = sel;
                }
            }
            const bool wide = (int)sumOf.size() > LWHIP_POST_SUMS;
            if (wideOut)
                *wideOut = wide;
            for (int q = 0; q < LWHIP_POST_MAXC; ++q)
            {
                uint32_t code = 0, ops = 0;
                for (int f = 0; f < 4; ++f)
                {
                    const int tg = target[q][f];
                    code |= wide ? (uint32_t)(tg ? tg : 0xFF) << (8 * f) : (uint32_t)(tg ? tg : 9) << (4 * f);
                }
                for (int w = 0; w < 5; ++w)
                {
                    const int sel = operand[q][w];
                    // own terms: narrow 9 .. 12, wide 60 .. 63 (+chi, -chi, U, eta)
                    const uint32_t field = sel >= 0 ? (uint32_t)sel : (uint32_t)((wide ? 59 : 8) - sel);
                    ops |= field << ((wide ? 6 : 4) * w);
                }
                pg.code[q] = code;
                pg.ops[q] = ops;
            }
            return true;
        };
        lanePostProgram = post_program;
        {
            // Which sweep: the lane sweep wherever its compiled kinds cover the problem (measured on MI355X:
            // tools/size_sweep.sh, tools/cross_exp.sh, profiles/r03_size_sweep.md), the ray-column march otherwise.
            // LWHIP_SWEEP=lanes|march forces one.
            const char* sw = std::getenv("LWHIP_SWEEP");
            int D = 0, LR = 0, R = 0;
            const bool supported = lane_sweep_supported(Ns, Nmu, c->prob.formalSolver, &D, &LR, &R);
            const bool force = (sw && std::strcmp(sw, "lanes") == 0) || c->hprd != nullptr // (hybrid PRD lives in the lane sweep)
                               || c->deterministic;                                    // (and so does the fixed-order reduction)
            // (End of round 3 -- one tile per wavefront, rays split over two wavefronts for small launches and for the tail of
            // large ones, fixed-register finish: the lane sweep is the faster one at every size measured, 0.076 / 0.142 /
            // 0.187 / 0.217 / 0.315 / 0.426 ms per step at 1 280 / 5 120 / 8 192 / 10 240 / 15 360 / 20 480 wavelengths against
            // the march's 0.171 / 0.205 / 0.225 / 0.223-0.234 / 0.396 / 0.457, and 20 % faster on a 512-column batch.  The
            // march serves what the lane sweep's kinds do not cover, and LWHIP_SWEEP=march.)
            const bool prefer = force || !sw;
            if ((c->hprd || c->deterministic || !(sw && std::strcmp(sw, "march") == 0)) && dbg_env_int("LWHIP_TILE_GENERIC", 0) == 0 && supported && prefer)
            {
                // (round 5: the choice is per TILE -- a wavelength whose lines / mixed continua have no compiled kind runs the lane
                // sweep's generic kind, lane_rays<-1, -1>: any number of lines up to 8, walked in runs of one line and one partner,
                // any mixed continua up to 8; only hybrid PRD, whose rates are formed ray by ray, is limited to the compiled kinds)
                bool ok = true;
                for (int la = 0; la < c->Nla && ok; ++la)
                {
                    const DevLaHeader& h = hdr[la];
                    const bool fastKind = (h.nLine == 0 && h.nMixed == 0) || (h.nLine == 1 && h.nMixed <= 1) || (h.nLine == 2 && h.nMixed == 0);
                    const bool genKind = !c->hprd && h.nLine <= 8 && h.nMixed <= 8 && dbg_env_int("LWHIP_LANE_GENERIC", 1) != 0;
                    DevPostProg pg;
                    ok = (fastKind || genKind) && post_program(la, pg, nullptr);
                }
                if (ok)
                {
                    c->laneSweep = true;
                    c->laneD = D;
                    c->laneLR = LR;
                    c->laneR = R;
                }
            }
        }
        if (c->hprd && !c->laneSweep)
            return fail(LWHIP_ERR_UNSUPPORTED, "hybrid PRD needs the depth-across-lanes sweep (1D, 13 <= Nspace <= 256, at most "
                                               "two lines or a line and a mixed continuum per wavelength)");
        const int L = c->laneSweep ? c->laneR : std::max(1, std::min({ 16, 64 / Nmu, dbg_env_int("LWHIP_TILE_L", 16) }));
        const int LRD = c->laneLR * c->laneD;
        c->tileL = L;
        auto pad16 = [](int64_t v) { return (v + 15) / 16 * 16; };
        auto same_structure = [&](int la, int lb) {
            const DevLaHeader &a = hdr[la], &b = hdr[lb];
            if (a.nLine != b.nLine || a.nMixed != b.nMixed || a.nPure != b.nPure || a.prog != b.prog || a.hasPrd != b.hasPrd)
                return false;
            const int n = a.nLine + a.nMixed + a.nPure;
            for (int e = 0; e < n; ++e)
                if (slotTr[a.off + e] != slotTr[b.off + e])
                    return false;
            return true;
        };
        c->rowsTileTot = c->momTot = c->phiTTot = 0;
        c->preCols = 0;
        c->maxSlotsTile = 0;
        c->tileCap = dbg_env_int("LWHIP_TILE_GENERIC", 0); // > 0: every tile through the guarded generic march (test hook)
        // (march, one problem: the launch is one round of workgroups -- one per tile, four per CU -- and lasts as long as its
        // slowest one, march + post-pass.  The tiles with two slots march longest; capping their wavelengths shortens their
        // post-pass, as long as the extra tiles still fit the round: measured 0.196 -> 0.184 ms at 10 240 wavelengths with 10
        // instead of 12, nothing beyond ~0.9 of the slots -- profiles/r03_march_bounds.md)
        auto count_tiles = [&](int capHeavy) {
            int n = 0;
            for (int la = 0; la < c->Nla; ++n)
            {
                const DevLaHeader& h = hdr[la];
                const int Lk = std::min(L, h.nLine + h.nMixed >= 2 ? capHeavy : L);
                int nl = 1;
                while (nl < Lk && la + nl < c->Nla && same_structure(la, la + nl))
                    ++nl;
                la += nl;
            }
            return n;
        };
        int autoHeavy = L;
        if (!c->laneSweep && c->batchHint <= 1 && L >= 6)
        {
            const int slots = 4 * numCU; // workgroups of two wavefronts at two wavefronts per SIMD
            for (int cap = L - 1; cap >= L - L / 6; --cap)
                if (count_tiles(cap) * 100 <= slots * 91)
                    autoHeavy = cap;
        }
        const int tileLHeavy = std::max(1, dbg_env_int("LWHIP_TILE_LH", autoHeavy)), tileLOne = std::max(1, dbg_env_int("LWHIP_TILE_L1", L));
        for (int la = 0; la < c->Nla;)
        {
            const DevLaHeader& h = hdr[la];
            // (march: a workgroup lasts its tile's march + post-pass, and the launch as long as its slowest workgroup: tiles
            // whose march is long -- two slots -- take fewer wavelengths, so that their post-pass is shorter)
            const int Lk = c->laneSweep ? L : std::min(L, h.nLine + h.nMixed >= 2 ? tileLHeavy : h.nLine + h.nMixed == 1 ? tileLOne : L);
            int nl = 1;
            while (nl < Lk && la + nl < c->Nla && same_structure(la, la + nl))
                ++nl;
            DevTile td{};
            td.la0 = la;
            td.nl = (int16_t)nl;
            td.nLine = h.nLine;
            td.nMixed = h.nMixed;
            td.hasPrd = (int16_t)h.hasPrd;
            td.slotOff = (int32_t)tslots.size();
            td.copyOff = (int32_t)tcopies.size();
            int nRowsT = 3;
            // the five continuum rows of an active slot (eta_A, chi_i, chi_j, U_i, U_j) are consecutive tile rows,
            // zero-filled where the wavelength has no such term: the march loads them unconditionally
            // which of a slot's five level sums no continuum of this wavelength feeds (the lane sweep does not read them)
            auto zero_mask = [&](const DevSlot& sl) -> int16_t {
                bool fed[5] = { false, false, false, false, false };
                for (int e = h.nLine; e < h.nLine + h.nMixed + h.nPure; ++e)
                {
                    const DevSlot& cs = slots[h.off + e];
                    if (!(cs.flags & 1))
                        continue;
                    fed[0] |= cs.atom == sl.atom;
                    fed[1] |= cs.gi == sl.gi || cs.gj == sl.gi;
                    fed[2] |= cs.gi == sl.gj || cs.gj == sl.gj;
                    fed[3] |= cs.gj == sl.gi;
                    fed[4] |= cs.gj == sl.gj;
                }
                int16_t zm = 0;
                for (int q = 0; q < 5; ++q)
                    zm |= fed[q] ? 0 : (int16_t)(1 << q);
                return zm;
            };
            auto five_rows = [&](const DevSlot& sl) -> int16_t {
                const int16_t r0 = (int16_t)nRowsT;
                // row ids as the row programs number them (the pre-pass forms the level sums directly)
                // (lane sweep: a row that no continuum of the wavelength feeds is never read -- its zero-mask bit -- and stays
                // the zero the allocation was cleared to: the pre-pass does not write it)
                const int16_t zmS = c->laneSweep ? zero_mask(sl) : (int16_t)0;
                int qi = 0;
                for (int id : { rowEtaA + sl.atom, rowLevChi + sl.gi, rowLevChi + sl.gj, rowLevU + sl.gi, rowLevU + sl.gj })
                {
                    if (zmS & (1 << qi++))
                    {
                        ++nRowsT;
                        continue;
                    }
                    if (h.nMixed + h.nPure > 0)
                        tcopies.push_back(DevTileCopy{ 0, (int16_t)id, (int16_t)nRowsT, 0 });
                    else
                        tcopies.push_back(DevTileCopy{ 3, 0, (int16_t)nRowsT, 0 });
                    ++nRowsT;
                }
                return r0;
            };
            for (int e = 0; e < h.nLine + h.nMixed; ++e)
            {
                const DevSlot& sl = slots[h.off + e];
                DevTileSlot ts{};
                ts.tr = slotTr[h.off + e];
                ts.cs = 0;
                ts.gi = sl.gi;
                ts.gj = sl.gj;
                ts.atom = sl.atom;
                ts.flags = sl.flags;
                ts.row = sl.row;
                ts.ms = sl.ms;
                ts.r0 = (sl.flags & 1) ? five_rows(sl) : (int16_t)0;
                ts.zmask = (sl.flags & 1) ? zero_mask(sl) : (int16_t)31;
                ts.rAux = -1;
                ts.e = (int16_t)e;
                ts.phiOff = -1;
                ts.rhoOff = sl.rhoOff;
                ts.hCoefOff = -1;
                if (c->hprd && e < h.nLine && c->hRhoOffHost[slotTr[h.off + e]] >= 0)
                {
                    // hybrid PRD: the coefficient block of the tile's first wavelength, and the line's first rho row
                    const HostTrans& ht = c->trans[slotTr[h.off + e]];
                    const int64_t lt0 = ht.ltStart + la - ht.NblueLoc; // (the coefficient table covers the line's whole grid)
                    ts.hCoefOff = c->hRhoOffHost[slotTr[h.off + e]] + lt0 * c->Nrays * 2 * Ns;
                    ts.rhoOff = ht.rhoOff; // row 0 of the line
                }
                if (e < h.nLine && c->laneSweep)
                    ts.phiOff = sl.phiOff; // the profiles are read where they lie
                else if (e < h.nLine)
                {
                    ts.phiOff = c->phiTTot;
                    c->phiTTot += (int64_t)2 * Ns * 64;
                }
                if (e < h.nLine)
                {
                    if ((sl.flags & 2) && !c->laneSweep) // (the lane sweep reads rho where it lies)
                    {
                        ts.rAux = (int16_t)nRowsT;
                        tcopies.push_back(DevTileCopy{ 2, (int16_t)e, (int16_t)nRowsT++, 0 });
                    }
                }
                else
                {
                    ts.rAux = (int16_t)nRowsT;
                    tcopies.push_back(DevTileCopy{ 1, (int16_t)(e - h.nLine), (int16_t)nRowsT++, 0 });
                }
                tslots.push_back(ts);
                const bool fastKind = (h.nLine == 0 && h.nMixed == 0) || (h.nLine == 1 && h.nMixed <= 1) || (h.nLine == 2 && h.nMixed == 0);
                if (!fastKind)
                    c->tileCap = std::max(c->tileCap, std::max((int)h.nLine, (int)h.nMixed));
            }
            td.nCopy = (int32_t)tcopies.size() - td.copyOff;
            {
                // distinct level-sum columns this tile's rows are copied from (the pre-pass sizes its LDS columns by the max)
                std::vector<int16_t> srcs;
                for (int q = td.copyOff; q < (int)tcopies.size(); ++q)
                    if (tcopies[q].kind == 0 && std::find(srcs.begin(), srcs.end(), tcopies[q].src) == srcs.end())
                        srcs.push_back(tcopies[q].src);
                c->preCols = std::max(c->preCols, (int)srcs.size());
                // ... and where each dense level-sum slot lives among them (what the pre-pass's workgroups used to
                // work out for themselves, one thread walking the copy list while 127 waited)
                const int nLev = c->Natom + 2 * c->NlevTot;
                const size_t base = tileRemap.size();
                tileRemap.resize(base + nLev + 1, (int32_t)srcs.size());
                for (size_t q = 0; q < srcs.size(); ++q)
                    if (srcs[q] >= 2 && srcs[q] - 2 < nLev)
                        tileRemap[base + (srcs[q] - 2)] = (int32_t)q;
            }
            td.nRowsT = nRowsT;
            td.rowBase = c->rowsTileTot;
            td.momBase = c->momTot;
            if (c->laneSweep)
                c->rowsTileTot += (int64_t)nRowsT * L * LRD; // [row][wavelength][k]
            else
            {
                c->rowsTileTot += (int64_t)nRowsT * Ns * 16; // rows of 16 doubles whatever L (RM_ROW)
                c->momTot += pad16((int64_t)8 * Ns * L);      // [direction][J, M1, Meta0, Meta1][Ns][L]
            }
            c->maxSlotsTile = std::max(c->maxSlotsTile, h.nLine + h.nMixed);
            tiles.push_back(td);
            la += nl;
        }
        c->nTiles = (int)tiles.size();
        if (std::getenv("LWHIP_VERBOSE"))
        {
            std::map<std::pair<int, int>, std::pair<int, int>> hist;
            for (auto& td : tiles)
            {
                auto& hh = hist[{ td.nLine, td.nMixed }];
                hh.first++;
                hh.second += td.nl;
            }
            for (auto& kv : hist)
                std::fprintf(stderr, "  tiles with %d lines + %d mixed continua: %d (%d wavelengths)\n", kv.first.first, kv.first.second,
                             kv.second.first, kv.second.second);
        }
        for (auto& td : tiles)
            if (td.nLine > 8 || td.nMixed > 8)
                return fail(LWHIP_ERR_UNSUPPORTED, "more than 8 lines (or 8 continua coupled to them) overlap at one wavelength");
    }
    if (c->tiled)
    {
        // workgroups of the sweep: contiguous, cost-balanced runs of tiles; the waves of a workgroup take its tiles in turn
        c->tileWaves = std::max(1, std::min(8, c->laneSweep ? dbg_env_int("LWHIP_LWAVES", 4) : dbg_env_int("LWHIP_TWAVES", 2)));
        // (fixed-order mode: since round 5 with the default's four wavefronts per workgroup -- they add to the workgroup's
        // accumulators in turn, lwhip_lanesweep.hip ls_turn_wait; hybrid PRD adds ray by ray and keeps round 3's form, one
        // wavefront per workgroup, whose LDS sums are formed in program order)
        if (c->laneSweep && c->deterministic && c->hprd)
            c->tileWaves = 1;
        if (!c->laneSweep)
            c->deterministic = false; // (not served by the ray-column march: the flag is ignored there)
        const int W = c->tileWaves;
        int wgPerCU = std::max(1, 8 / W);
        if (c->batchHint > 1)
            wgPerCU = 1;
        struct TileChunking
        {
            std::vector<int32_t> chunkTile, slotTr;
            std::vector<int> cs;     // per tile slot (mode 0, 2) or per wavelength slot record (mode 1)
            std::vector<int> csPure; // mode 2: per wavelength slot record, the pure continua
            int maxCT = 1;
        };
        auto tile_cost = [&](const DevTile& td) { return 1.0 + 0.35 * (td.nLine + td.nMixed); };
        // mode 0: accumulator slots for the tiles' lines and mixed continua; 1: for the pure continua (post-pass chunks);
        // 2: for both (the lane sweep's workgroups finish their tiles themselves)
        // fixedCount > 0: every chunk holds that many tiles (lane sweep: one per group of wavefronts -- a workgroup's rounds
        // are what a launch of one partial round lasts)
        auto chunk_tiles = [&](const std::vector<int32_t>& list, int tgt, int mode, int fixedCount = 0,
                               const std::vector<int32_t>* bounds = nullptr) {
            const bool pure = mode == 1;
            TileChunking ck;
            const int n = (int)list.size();
            tgt = std::max(1, std::min(tgt, n));
            if (fixedCount > 0)
                tgt = (n + fixedCount - 1) / fixedCount;
            double total = 0.0;
            for (int ti : list)
                total += tile_cost(tiles[ti]);
            ck.chunkTile.push_back(0);
            double acc = 0.0;
            int made = 0;
            if (bounds) // the chunks are given ([0, ..., n])
                ck.chunkTile.assign(bounds->begin(), bounds->end() - 1);
            for (int i = 0; i < n && !bounds; ++i)
            {
                acc += tile_cost(tiles[list[i]]);
                const int remaining = n - (i + 1), remainingChunks = tgt - (made + 1);
                const bool cut = fixedCount > 0 ? (i + 1) % fixedCount == 0 : (tgt >= n || acc >= total * (made + 1) / tgt || remaining <= remainingChunks);
                if (cut && i + 1 < n && made + 1 < tgt)
                {
                    ck.chunkTile.push_back(i + 1);
                    ++made;
                }
            }
            ck.chunkTile.push_back(n);
            const int nch = (int)ck.chunkTile.size() - 1;
            std::vector<std::vector<int>> chunkTrans(nch);
            ck.cs.assign(pure ? slots.size() : tslots.size(), 0);
            if (mode == 2)
                ck.csPure.assign(slots.size(), 0);
            for (int ch = 0; ch < nch; ++ch)
            {
                std::vector<int> slotOf(c->Ntrans, -1);
                auto slot_for = [&](int tr) {
                    if (slotOf[tr] < 0)
                    {
                        slotOf[tr] = (int)chunkTrans[ch].size();
                        chunkTrans[ch].push_back(tr);
                    }
                    return slotOf[tr];
                };
                for (int i = ck.chunkTile[ch]; i < ck.chunkTile[ch + 1]; ++i)
                {
                    const DevTile& td = tiles[list[i]];
                    if (!pure)
                        for (int e = 0; e < td.nLine + td.nMixed; ++e)
                            ck.cs[td.slotOff + e] = slot_for(tslots[td.slotOff + e].tr);
                    if (mode != 0)
                        for (int q = 0; q < td.nl; ++q)
                        {
                            const DevLaHeader& h = hdr[td.la0 + q];
                            for (int e = h.nLine + h.nMixed; e < h.nLine + h.nMixed + h.nPure; ++e)
                                (mode == 2 ? ck.csPure : ck.cs)[h.off + e] = slot_for(slotTr[h.off + e]);
                        }
                }
                ck.maxCT = std::max(ck.maxCT, (int)chunkTrans[ch].size());
            }
            ck.slotTr.assign((size_t)nch * ck.maxCT, -1);
            for (int ch = 0; ch < nch; ++ch)
                for (size_t q = 0; q < chunkTrans[ch].size(); ++q)
                    ck.slotTr[(size_t)ch * ck.maxCT + q] = chunkTrans[ch][q];
            return ck;
        };
        std::vector<int32_t> all(c->nTiles);
        for (int i = 0; i < c->nTiles; ++i)
        {
            all[i] = i;
            if (tiles[i].hasPrd)
                tileListPrd.push_back(i);
        }
        // (dispatching the lane sweep's long tasks -- tiles with more slots -- first was measured and is WORSE: 0.288 against
        // 0.204 ms at 10 240 wavelengths; co-resident long tasks slow each other down)
        int tgtSweep = std::min((2 * c->nTiles + W - 1) / W, numCU * wgPerCU); // two tasks per tile (down, up)
        // fused post-pass: the workgroup is the two wavefronts of one tile (and every tile is a workgroup)
        // (column batches too: 12.3k -> 16.0k column-iterations/s at 128 columns against a separate post launch)
        c->tileFuse = W == 2 && dbg_env_int("LWHIP_TILE_FUSE", 1) != 0 && !c->laneSweep;
        if (c->batchHint > 1 && !c->tileFuse)
            tgtSweep = std::max(1, std::min(tgtSweep, numCU * 4 / c->batchHint));
        if (c->tileFuse)
            tgtSweep = c->nTiles;
        if (!c->tileFuse)
            tgtSweep = std::max(1, tgtSweep);
        // Deep columns on the march (round 6): a wavefront's march is ~1.6 us per depth point whatever the number of wavelengths,
        // so a launch of few tiles over many depth points -- the reference benchmark's 500-point FAL-C: 242 tiles -- leaves most of
        // the chip idle for 0.8 ms.  While 2 S wavefronts per tile still fit the chip's two-per-SIMD, a direction's depth points
        // are split over S wavefronts of the tile's workgroup (lwhip_raymarch.hip: raymarch_kernel, depthSplit).
        // LWHIP_DEPTH_SPLIT=1|2|4 (LWHIP_DEBUG) forces it.
        c->depthSplit = 1;
        if (c->tileFuse && c->tileCap == 0 && c->batchHint <= 1 && !c->deterministic)
        {
            const long slots = 2L * 4 * numCU; // wavefronts the chip holds at the march's two per SIMD
            // (measured at 500 points x 2 908 wavelengths, 242 tiles: S = 1 0.804 ms, S = 2 1.088, S = 4 0.652 per sweep -- every
            // later segment walks its points twice, so the split turns a latency-bound launch (a quarter of the SIMDs busy) into
            // a throughput-bound one with twice the instructions: a gain only where the unsplit launch leaves most of the chip
            // idle and four-way; S = 2 is kept for the tests)
            // (the reference benchmark's grid is 281 tiles = 2 248 wavefronts four-way: 1.10 x the slots, 0.65 against 0.80 ms per sweep)
            int S = (Ns >= 256 && 8L * c->nTiles <= slots * 3 / 2) ? 4 : 1;
            S = dbg_env_int("LWHIP_DEPTH_SPLIT", S);
            if ((S == 2 || S == 4) && Ns >= 64 * S)
                c->depthSplit = S;
        }
        c->laneSplit = 1;
        if (c->laneSweep)
        {
            // Small problems: the launch is ONE partial round of tasks and lasts as long as its longest task (a tile with two
            // slots: ~0.13 ms alone).  While S wavefronts per tile still fit the round, a tile's rays are split over the S
            // wavefronts of a group (each finishes 1/S of the tile's points): shorter tasks, same round.
            if ((!c->deterministic || !c->hprd) && c->batchHint <= 1 && W == 4)
            {
                const long slots = 2L * 4 * numCU;
                // (measured, tools/split_exp.sh: S = 2 wins while it fits.  S = 4 while IT fits: in round 3 it had the same kernel
                // time at 1 280 wavelengths and a longer step -- four times the workgroups flushing their sums --; with the
                // round-4 kernel (moments, own pre-pass) it is 0.048 against 0.059 ms there, the step 0.059 against 0.074)
                // (a launch that S = 4 fills to the brim -- 1 536 wavelengths, 2 048 wavefronts -- is no faster than with S = 2)
                c->laneSplit = 4L * c->nTiles <= slots * 9 / 10 ? 4 : 2L * c->nTiles <= slots ? 2 : 1;
                const int forced = dbg_env_int("LWHIP_LANE_SPLIT", 0);
                if (forced == 1 || forced == 2 || forced == 4)
                    c->laneSplit = forced;
            }
            tgtSweep = (c->nTiles * c->laneSplit + W - 1) / W; // one tile per group of wavefronts
            c->laneSplitPrd = c->laneSplit;
            if ((!c->deterministic || !c->hprd) && c->batchHint <= 1 && W == 4 && !tileListPrd.empty() && dbg_env_int("LWHIP_LANE_SPLIT", 0) == 0)
            {
                // (the PRD rates pass -- every ray forms its line's rates -- gains from S = 4 up to one and a half rounds of
                // wavefronts: 738 tiles, 2 952 wavefronts on 2 048 slots, 0.118 ms per sub-iteration against 0.123 with S = 2)
                const long slots = 2L * 4 * numCU, nT = (long)tileListPrd.size();
                c->laneSplitPrd = 4L * nT <= slots * 3 / 2 ? 4 : 2L * nT <= slots ? 2 : 1;
            }
        }
        // post-pass workgroups: one resident set (LDS: accumulators + the threads' level columns; 8 waves per CU by registers)
        const size_t ldsPostEst = sizeof(double) * ((size_t)7 * 4 * Ns + (size_t)(c->Natom + 2 * c->NlevTot) * 128);
        const int postPerCU = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)(150 * 1024) / ldsPostEst));
        int tgtPost = std::min(c->nTiles, numCU * postPerCU);
        if (c->batchHint > 1)
            tgtPost = std::max(1, std::min(tgtPost, numCU * 8 / c->batchHint));
        const int laneFixed = c->laneSweep ? std::max(1, W / c->laneSplit) : 0; // (one tile per group of wavefronts)
        TileChunking sw = chunk_tiles(all, tgtSweep, c->laneSweep ? 2 : 0, laneFixed);
        // Lane sweep, more workgroups than the chip holds at once (one problem, 2 workgroups of 4 wavefronts per CU): the
        // launch ends with workgroups that started late and have the chip to themselves -- whole tiles of 0.1 ms.  The
        // chunks that are dispatched last (the lightest: heaviest go first) are cut into smaller ones whose tiles' rays are
        // split over 2 or 4 wavefronts: the same work in shorter pieces, so the tail drains evenly.
        {
            const int slotsWg = 2 * numCU;
            const int nch0 = (int)sw.chunkTile.size() - 1;
            const int tailS = 2; // (the cut chunks' tiles have their rays split over two wavefronts; four measured worse, DESIGN 3.1b vi)
            // (and in a launch of ONE partial round -- more than half a round of whole tiles, so that not every tile can be
            // split -- the heaviest chunks are the ones cut, as many as still fit the round: the launch lasts as long as its
            // longest workgroup)
            const bool oneRound = nch0 <= slotsWg;
            if (c->laneSweep && c->batchHint <= 1 && W == 4 && c->laneSplit == 1 && laneFixed == W
                && (nch0 > slotsWg || (oneRound && nch0 > slotsWg / 2)) && (tailS == 2 || tailS == 4))
            {
                auto tcost = [&](int i) { return 1.0 + 0.5 * tiles[i].nLine + 0.3 * tiles[i].nMixed + (tiles[i].nLine == 0 ? -0.25 : 0.0); };
                std::vector<double> cost(nch0, 0.0);
                for (int ch = 0; ch < nch0; ++ch)
                    for (int i = sw.chunkTile[ch]; i < sw.chunkTile[ch + 1]; ++i)
                        cost[ch] += tcost(i);
                std::vector<int32_t> ord(nch0);
                for (int ch = 0; ch < nch0; ++ch)
                    ord[ch] = ch;
                std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return cost[x] > cost[y]; });
                std::vector<char> cut(nch0, 0);
                if (nch0 > slotsWg)
                {
                    // the first `keep` chunks of the cost order stay whole, the lightest numCU / 2 are cut in two: one pair of
                    // half-length workgroups per compute unit to even out the end of the launch.  (Until late in round 4 the rule
                    // was "as many as fill the last round of slots exactly", whole + 2 cut = rounds x slots; a sweep over sizes
                    // and counts -- tools/keep_rule_exp.sh, profiles/r04_tail_rule.md -- showed a spiky landscape in which that
                    // was rarely the minimum: cutting 128 chunks is 3-9 % faster at 6 656 ... 12 288 wavelengths, 3 % at the bench
                    // size, 3 % slower at 11 264.)
                    // (... and where the last round has room for fewer pairs than that but for some -- 11 264: 85 -- only as many)
                    const int rounds = (nch0 + slotsWg - 1) / slotsWg;
                    const int room = rounds * slotsWg - nch0;
                    int nCut = std::max(1, numCU / 2);
                    nCut = std::max(0, dbg_env_int("LWHIP_LANE_CUT", nCut)); // (experiments: chunks of the tail cut in two)
                    if (room >= numCU / 8 && room < nCut)
                        nCut = room;
                    const int keepAuto = std::max(0, nch0 - nCut);
                    const int keep = std::min(nch0, std::max(0, keepAuto));
                    for (int q = keep; q < nch0; ++q)
                        cut[ord[q]] = 1;
                }
                else
                {
                    // one partial round: a cut chunk becomes tailS / 1 workgroups; the heaviest first, while they fit.
                    // A round that is (nearly) full does not run as one: the dispatcher does not co-schedule every workgroup
                    // (6 144 wavelengths = 512 chunks on 512 slots took two task lengths, 0.115 ms) -- then EVERY chunk is cut
                    // and the launch is two rounds of half-length tasks
                    const int extra = tailS - 1;
                    const int fullPct = 85; // (profiles/r04_tail_rule.md)
                    int k = std::min(nch0, std::max(0, (slotsWg - nch0) / extra));
                    if (tailS == 2 && nch0 * 100 > slotsWg * fullPct)
                        k = nch0;
                    for (int q = 0; q < k; ++q)
                        cut[ord[q]] = 1;
                }
                const int piece = W / tailS; // tiles per piece
                std::vector<int32_t> bounds, split, firstNew(nch0, 0), nNew(nch0, 0);
                bounds.push_back(0);
                for (int ch = 0; ch < nch0; ++ch)
                {
                    firstNew[ch] = (int)bounds.size() - 1;
                    const int b0 = sw.chunkTile[ch], b1 = sw.chunkTile[ch + 1];
                    if (!cut[ch])
                    {
                        bounds.push_back(b1);
                        split.push_back(1);
                    }
                    else
                        for (int b = b0; b < b1; b += piece)
                        {
                            bounds.push_back(std::min(b + piece, b1));
                            split.push_back(tailS);
                        }
                    nNew[ch] = (int)bounds.size() - 1 - firstNew[ch];
                }
                sw = chunk_tiles(all, 0, 2, 0, &bounds);
                laneChunkSplit = split;
                for (int q = 0; q < nch0; ++q)
                    for (int e = 0; e < nNew[ord[q]]; ++e)
                        laneChunkOrder.push_back(firstNew[ord[q]] + e);
            }
        }
        TileChunking po = chunk_tiles(all, tgtPost, 1);
        TileChunking swPrd, poPrd;
        if (!tileListPrd.empty())
        {
            // (fused post-pass: one tile per workgroup here too)
            swPrd = chunk_tiles(tileListPrd,
                                c->laneSweep ? ((int)tileListPrd.size() * c->laneSplitPrd + W - 1) / W
                                : c->tileFuse ? (int)tileListPrd.size()
                                              : std::min((2 * (int)tileListPrd.size() + W - 1) / W, numCU * wgPerCU),
                                0, c->laneSweep ? std::max(1, W / c->laneSplitPrd) : 0);
            poPrd = chunk_tiles(tileListPrd, std::min((int)tileListPrd.size(), numCU * postPerCU), 1);
        }
        // both chunkings share one accumulator-slot capacity per kind
        c->maxCTTile = std::max(sw.maxCT, swPrd.maxCT);
        c->maxCTPost = po.maxCT;
        auto widen = [&](TileChunking& ck, int maxCT) {
            if (ck.maxCT == maxCT)
                return;
            const int nch = (int)ck.chunkTile.size() - 1;
            std::vector<int32_t> w((size_t)nch * maxCT, -1);
            for (int ch = 0; ch < nch; ++ch)
                for (int q = 0; q < ck.maxCT; ++q)
                    w[(size_t)ch * maxCT + q] = ck.slotTr[(size_t)ch * ck.maxCT + q];
            ck.slotTr.swap(w);
            ck.maxCT = maxCT;
        };
        widen(sw, c->maxCTTile);
        if (!tileListPrd.empty())
            widen(swPrd, c->maxCTTile);
        for (size_t e = 0; e < tslots.size(); ++e)
            tslots[e].cs = sw.cs[e];
        laneCsPure = sw.csPure;
        chunkTile = sw.chunkTile;
        tileSlotTr = sw.slotTr;
        c->nTileChunks = (int)chunkTile.size() - 1;
        if (c->deterministic)
        {
            // per transition: the (workgroup, slot) pairs that hold its sums, in workgroup order
            auto lists = [&](const std::vector<int32_t>& slotTrTab, int nch, std::vector<int32_t>& off, std::vector<int32_t>& ent) {
                std::vector<std::vector<int32_t>> per(c->Ntrans);
                for (int ch = 0; ch < nch; ++ch)
                    for (int q = 0; q < c->maxCTTile; ++q)
                    {
                        const int tr = slotTrTab[(size_t)ch * c->maxCTTile + q];
                        if (tr >= 0)
                            per[tr].push_back(ch * c->maxCTTile + q);
                    }
                off.assign(1, 0);
                ent.clear();
                for (int tr = 0; tr < c->Ntrans; ++tr)
                {
                    ent.insert(ent.end(), per[tr].begin(), per[tr].end());
                    off.push_back((int32_t)ent.size());
                }
                if (ent.empty())
                    ent.push_back(0);
            };
            std::vector<int32_t> o, e;
            lists(tileSlotTr, c->nTileChunks, o, e);
            HIP_TRY(c->detOff.upload_or_borrow(o, like ? &like->detOff : nullptr));
            HIP_TRY(c->detEnt.upload_or_borrow(e, like ? &like->detEnt : nullptr));
            if (!tileListPrd.empty())
            {
                lists(swPrd.slotTr, (int)swPrd.chunkTile.size() - 1, o, e);
                HIP_TRY(c->detOffPrd.upload_or_borrow(o, like ? &like->detOffPrd : nullptr));
                HIP_TRY(c->detEntPrd.upload_or_borrow(e, like ? &like->detEntPrd : nullptr));
            }
            const size_t nch = std::max<size_t>(c->nTileChunks, tileListPrd.empty() ? 0 : swPrd.chunkTile.size() - 1);
            HIP_TRY(c->detSlab.alloc(nch * c->maxCTTile * 4 * Ns));
            HIP_TRY(c->detPart.alloc(det_reduce_scratch_doubles((int)Ns, c->Ntrans)));
        }
        postChunkTile = po.chunkTile;
        postSlotTr = po.slotTr;
        for (size_t e = 0; e < slots.size(); ++e)
            postCs[e] = po.cs[e];
        c->nPostChunks = (int)postChunkTile.size() - 1;
        c->nTilesPrd = (int)tileListPrd.size();
        c->nTileChunksPrd = c->nPostChunksPrd = 0;
        if (!tileListPrd.empty())
        {
            tslotsPrd = tslots;
            for (size_t e = 0; e < tslots.size(); ++e)
                tslotsPrd[e].cs = swPrd.cs[e];
            chunkTilePrd = swPrd.chunkTile;
            tileSlotTrPrd = swPrd.slotTr;
            postChunkTilePrd = poPrd.chunkTile;
            c->nTileChunksPrd = (int)chunkTilePrd.size() - 1;
            c->nPostChunksPrd = (int)postChunkTilePrd.size() - 1;
        }
        // LDS: the workgroup's accumulators + a 2 KB exchange row per wave (sweep); accumulators + continuum columns (post)
        const size_t ldsSweep = sizeof(double) * ((size_t)c->maxCTTile * 4 * Ns + std::max((size_t)W * 5 * 16 * 9, (size_t)16 + (size_t)(c->Natom + 2 * c->NlevTot) * 128));
        const size_t ldsPost = sizeof(double) * (16 + (size_t)(c->Natom + 2 * c->NlevTot) * 128);
        if (!c->laneSweep && (ldsSweep > 150 * 1024 || ldsPost > 150 * 1024))
            return fail(LWHIP_ERR_UNSUPPORTED, "problem does not fit the 160 KB LDS budget of one workgroup");
        if (c->laneSweep && sizeof(double) * (size_t)c->maxCTTile * 4 * (c->laneLR * c->laneD) > 64 * 1024)
            return fail(LWHIP_ERR_UNSUPPORTED, "more transitions per workgroup than the lane sweep's LDS accumulators hold");
    }

    const int NR2 = 2 * c->Nrays;
    tick("tiles, chunks, work distribution");
    // ---- device tables ---------------------------------------------------------------------------------
    std::vector<DevTrans> dt(c->Ntrans);
    std::vector<int32_t> li(c->Ntrans), lj(c->Ntrans);
    for (int tr = 0; tr < c->Ntrans; ++tr)
    {
        const HostTrans& h = c->trans[tr];
        DevTrans& d = dt[tr];
        d.type = h.t.type;
        d.atom = h.atom;
        d.gi = c->levelOff[h.atom] + h.t.i;
        d.gj = c->levelOff[h.atom] + h.t.j;
        d.Nblue = h.NblueLoc;
        d.Nred = h.NredLoc;
        d.prd = (h.t.type == LWHIP_LINE && h.t.prd) ? 1 : 0;
        d.op = c->atoms[h.atom].detailed ? 0 : 1;
        d.row = h.row;
        d._pad = 0;
        d.parOff = h.parOff;
        d.phiOff = h.phiOff;
        d.rhoOff = h.rhoOff;
        li[tr] = h.t.i;
        lj[tr] = h.t.j;
    }
    std::vector<int32_t> rayAll(NR2), rayUp(c->Nrays);
    for (int r = 0; r < NR2; ++r)
        rayAll[r] = r;
    for (int m = 0; m < c->Nrays; ++m)
        rayUp[m] = 2 * m + 1;
    std::vector<int32_t> aNl(c->Natom), aDet(c->Natom);
    for (int ia = 0; ia < c->Natom; ++ia)
    {
        aNl[ia] = c->atoms[ia].Nlevel;
        aDet[ia] = c->atoms[ia].detailed ? 1 : 0;
    }
    std::vector<int32_t> atomTrOff32(c->atomTrOff.begin(), c->atomTrOff.end());

    // full own grids and trapezoid weights of the lines (device Voigt profiles)
    {
        std::vector<double> lw, lq;
        for (auto& h : c->trans)
        {
            h.waveOff = -1;
            if (h.t.type != LWHIP_LINE)
                continue;
            h.waveOff = (int64_t)lw.size();
            const int len = h.t.Nred - h.t.Nblue;
            for (int lt = 0; lt < len; ++lt)
            {
                lw.push_back(h.t.wavelength[lt]);
                lq.push_back(trans_wlambda(h.t, lt));
            }
        }
        if (lw.empty())
        {
            lw.push_back(0.0);
            lq.push_back(0.0);
        }
        HIP_TRY(c->lineWave.upload_or_borrow(lw, like ? &like->lineWave : nullptr));
        HIP_TRY(c->lineWlam.upload_or_borrow(lq, like ? &like->lineWlam : nullptr));
    }
    HIP_TRY(c->par.upload_or_borrow(par, like ? &like->par : nullptr));
    HIP_TRY(c->dtrans.upload_or_borrow(dt, like ? &like->dtrans : nullptr));
    HIP_TRY(c->laHdr.upload_or_borrow(hdr, like ? &like->laHdr : nullptr));
    HIP_TRY(c->slots.upload_or_borrow(slots, like ? &like->slots : nullptr));
    {
        std::vector<int32_t> st32(slotTr.begin(), slotTr.end());
        if (st32.empty())
            st32.push_back(0);
        HIP_TRY(c->slotTrD.upload_or_borrow(st32, like ? &like->slotTrD : nullptr));
    }
    c->hdrHost = hdr;
    c->contLaHost = contLaHost;
    HIP_TRY(c->progs.upload_or_borrow(progs, like ? &like->progs : nullptr));
    HIP_TRY(c->progRows.upload_or_borrow(progRows, like ? &like->progRows : nullptr));
    HIP_TRY(c->progEnts.upload_or_borrow(progEnts, like ? &like->progEnts : nullptr));
    if (contLaHost.empty())
        contLaHost.push_back(0);
    c->nContLa = (c->rowsTot > 0) ? (int)contLaHost.size() : 0;
    HIP_TRY(c->contLa.upload_or_borrow(contLaHost, like ? &like->contLa : nullptr));
    HIP_TRY(c->rayAll.upload_or_borrow(rayAll, like ? &like->rayAll : nullptr));
    HIP_TRY(c->rayUp.upload_or_borrow(rayUp, like ? &like->rayUp : nullptr));
    if (c->tiled)
    {
        auto up = [](auto& buf, auto& v, const auto* from) -> hipError_t {
            if (v.empty())
                v.resize(1);
            return buf.upload_or_borrow(v, from);
        };
        std::vector<DevContRec> cr(std::max<size_t>(slots.size(), 1), DevContRec{});
        std::vector<DevPostProg> pp;
        std::vector<char> ppWide; // per tile: the finish program's words are in the wide encoding
        {
            // the continua of every wavelength in transition order, ready-to-use offsets and level-column slots
            for (int la = 0; la < c->Nla; ++la)
            {
                const DevLaHeader& h = hdr[la];
                const int base = h.off + h.nLine;
                for (int q = 0; q < h.nMixed + h.nPure; ++q)
                {
                    const int ord = slots[base + q]._pad;
                    const DevSlot& sl = slots[base + ord];
                    DevContRec r{};
                    r.nI = (uint32_t)((size_t)sl.gi * Ns * sizeof(double));
                    r.nJ = (uint32_t)((size_t)sl.gj * Ns * sizeof(double));
                    r.rat = (uint32_t)((size_t)sl.row * Ns * sizeof(double));
                    r.pOff = (uint32_t)((size_t)(h.nLine + ord) * sizeof(DevSlot) + offsetof(DevSlot, p));
                    r.sEta = (uint16_t)sl.atom;
                    r.sChiI = (uint16_t)(c->Natom + sl.gi);
                    r.sChiJ = (uint16_t)(c->Natom + sl.gj);
                    r.sUi = (uint16_t)(c->Natom + c->NlevTot + sl.gi);
                    r.sUj = (uint16_t)(c->Natom + c->NlevTot + sl.gj);
                    r.flags = (uint8_t)((sl.flags & 1) | (ord >= h.nMixed ? 2 : 0));
                    r.ms = (int8_t)sl.ms;
                    r.tr = (int16_t)slotTr[base + ord];
                    r.cs = (int16_t)((c->laneSweep && ord >= h.nMixed) ? laneCsPure[base + ord] : 0);
                    cr[base + q] = r;
                }
            }
            HIP_TRY(c->contRec.upload_or_borrow(cr, like ? &like->contRec : nullptr));
        }
        if (c->laneSweep)
        {
            pp.assign(std::max<size_t>(tiles.size(), 1), DevPostProg{});
            ppWide.assign(std::max<size_t>(tiles.size(), 1), 0);
            for (size_t t = 0; t < tiles.size(); ++t)
            {
                bool wide = false;
                if (!lanePostProgram(tiles[t].la0, pp[t], &wide))
                    return fail(LWHIP_ERR_UNSUPPORTED, "lane sweep: post program");
                ppWide[t] = wide ? 1 : 0;
            }
            HIP_TRY(c->postProg.upload_or_borrow(pp, like ? &like->postProg : nullptr));
            if (std::getenv("LWHIP_VERBOSE"))
            {
                // which program words occur (the finish serves the common ones with fixed registers)
                std::map<std::pair<unsigned, unsigned>, int> hist;
                for (size_t t = 0; t < tiles.size(); ++t)
                {
                    const DevLaHeader& h = hdr[tiles[t].la0];
                    for (int q = 0; q < h.nMixed + h.nPure; ++q)
                        hist[{ pp[t].code[q], pp[t].ops[q] }]++;
                }
                for (const auto& kv : hist)
                    std::fprintf(stderr, "  finish program word code %#x ops %#x: %d continua\n", kv.first.first, kv.first.second, kv.second);
            }
        }
        // Fixed-order mode (LWHIP_OPT_DETERMINISTIC): every task's turn at each of its accumulator slots -- the value the slot's
        // turn counter holds when the task may add (lwhip_lanesweep.hip, ls_turn_wait).  Per workgroup: its rounds in order;
        // inside a round first the line / mixed-continuum adds of every task (they precede the barriers of a split tile, the
        // pure continua's follow them: a wait never points across a barrier the other way), then the pure continua's; the
        // tasks of a round in order of expected cost, so that nobody waits for a wavefront that is still marching.  A line /
        // mixed slot counts one per wavefront of the tile's group (each adds the integrals of its own rays to every point), a
        // pure continuum one per task (every point is finished by exactly one wavefront).  The turn rides in bits 8 .. 23 of
        // the slot's flags (lines, mixed continua) and in bits 16 .. 31 of DevLaneFin::feed (pure continua).
        std::vector<uint32_t> turnPure(std::max<size_t>(tiles.size() * LWHIP_POST_MAXC, 1), 0u);
        if (c->laneSweep && c->deterministic)
        {
            auto assign_turns = [&](std::vector<DevTileSlot>& ts, const std::vector<int32_t>& chTile, const std::vector<int32_t>* list,
                                    const std::vector<int32_t>* split, int launchSplit, bool pure) -> bool {
                const int nch = (int)chTile.size() - 1;
                std::vector<int> cnt(std::max(c->maxCTTile, 1));
                // (measurement only: bit 0 -- the line / mixed slots' turns are 0, nobody waits there; bit 1 -- the pure continua's;
                // bit 2 -- tasks in wavefront order instead of by expected cost.  With a bit of 0 / 1 set the sums' order is the timing's)
                const int noWaitBits = dbg_env_int("LWHIP_DET_NOWAIT", 0);
                auto tcost = [&](int t) { return 1.0 + 0.5 * tiles[t].nLine + 0.3 * tiles[t].nMixed + (tiles[t].nLine == 0 ? -0.25 : 0.0); };
                for (int ch = 0; ch < nch; ++ch)
                {
                    std::fill(cnt.begin(), cnt.end(), 0);
                    const int S = std::max(1, split ? (*split)[ch] : launchSplit);
                    const int nGroups = std::max(1, c->tileWaves / S);
                    const int n = chTile[ch + 1] - chTile[ch];
                    for (int r0 = 0; r0 < n; r0 += nGroups)
                    {
                        std::vector<int> tasks;
                        for (int i = r0; i < std::min(n, r0 + nGroups); ++i)
                            tasks.push_back(list ? (*list)[chTile[ch] + i] : chTile[ch] + i);
                        if (!(noWaitBits & 4))
                            std::stable_sort(tasks.begin(), tasks.end(), [&](int x, int y) { return tcost(x) < tcost(y); });
                        // (a tile that walks its rays more than once -- a blend, a generic tile -- adds after every run: it takes a
                        // SECOND turn for the adds of its later runs, behind the first turns of the whole round, so that it does not
                        // hold a slot while it marches again -- 28 blends of 3 429 tiles doubled the launch's time when they did)
                        for (int pass = 0; pass < 2; ++pass)
                            for (int t : tasks)
                            {
                                const int nl = tiles[t].nLine;
                                const bool fastKind = (nl == 0 && tiles[t].nMixed == 0) || (nl == 1 && tiles[t].nMixed <= 1) || (nl == 2 && tiles[t].nMixed == 0);
                                if (pass == 1 && !(fastKind && nl == 2)) // (only the compiled blend kind walks its rays twice)
                                    continue;
                                for (int e = 0; e < tiles[t].nLine + tiles[t].nMixed; ++e)
                                {
                                    DevTileSlot& u = ts[tiles[t].slotOff + e];
                                    if (u.cs < 0 || u.cs >= (int)cnt.size() || cnt[u.cs] + S > 0x7ff)
                                        return false;
                                    const int turn = (noWaitBits & 1) ? 0 : cnt[u.cs];
                                    if (pass == 0)
                                        u.flags = (u.flags & 0xff) | (turn << 8);
                                    else
                                        u.flags |= turn << 20;
                                    cnt[u.cs] += S;
                                }
                            }
                        for (int t : tasks)
                        {
                            const DevLaHeader& h = hdr[tiles[t].la0];
                            if (!pure || h.nPure == 0)
                                continue;
                            for (int q = 0; q < std::min(h.nMixed + h.nPure, (int)LWHIP_POST_MAXC); ++q)
                            {
                                const DevContRec& r = cr[h.off + h.nLine + q];
                                if (!(r.flags & 2))
                                    continue;
                                if (r.cs < 0 || r.cs >= (int)cnt.size() || cnt[r.cs] + 1 > 0xffff)
                                    return false;
                                turnPure[(size_t)t * LWHIP_POST_MAXC + q] = (noWaitBits & 2) ? 0u : (uint32_t)cnt[r.cs];
                                cnt[r.cs] += 1;
                            }
                        }
                    }
                }
                return true;
            };
            bool okT = assign_turns(tslots, chunkTile, nullptr, laneChunkSplit.empty() ? nullptr : &laneChunkSplit, c->laneSplit, true);
            if (okT && !tslotsPrd.empty() && !tileListPrd.empty())
                okT = assign_turns(tslotsPrd, chunkTilePrd, &tileListPrd, nullptr, c->laneSplitPrd, false);
            if (!okT)
                return fail(LWHIP_ERR_UNSUPPORTED, "fixed-order mode: more adds to one accumulator slot of a workgroup than a turn counter counts");
        }
        HIP_TRY(up(c->tiles, tiles, like ? &like->tiles : nullptr));
        HIP_TRY(up(c->tslots, tslots, like ? &like->tslots : nullptr));
        HIP_TRY(up(c->tslotsPrd, tslotsPrd, like ? &like->tslotsPrd : nullptr));
        HIP_TRY(up(c->tcopies, tcopies, like ? &like->tcopies : nullptr));
        HIP_TRY(up(c->tileRemap, tileRemap, like ? &like->tileRemap : nullptr));
        HIP_TRY(up(c->chunkTile, chunkTile, like ? &like->chunkTile : nullptr));
        c->chunkOrderOn = false;
        c->chunkSplitOn = false;
        if (!laneChunkOrder.empty())
        {
            HIP_TRY(up(c->chunkOrder, laneChunkOrder, like ? &like->chunkOrder : nullptr));
            HIP_TRY(up(c->chunkSplit, laneChunkSplit, like ? &like->chunkSplit : nullptr));
            c->chunkOrderOn = c->chunkSplitOn = true;
        }
        const int nchAll = (int)chunkTile.size() - 1;
        // more workgroups than the chip holds at once (2 per CU): the heavy ones go first, the light ones fill the tail
        // (measured: 8 192 wavelengths 0.188 -> 0.179 ms, 10 240: 0.203 -> 0.200; lightest first: 0.196 / 0.207)
        const int laneOrder = nchAll > 2 * numCU ? 1 : 0; // heaviest first
        if (c->laneSweep && c->batchHint <= 1 && laneOrder != 0 && laneChunkOrder.empty())
        {
            // the chunks stay what they are (neighbouring wavelengths: few accumulator slots); only the order in which the
            // workgroups take them changes: 1 = heaviest first, 2 = lightest first
            const int nch = nchAll;
            std::vector<double> cost(nch, 0.0);
            for (int ch = 0; ch < nch; ++ch)
                for (int i = chunkTile[ch]; i < chunkTile[ch + 1]; ++i)
                    cost[ch] += 1.0 + 0.5 * tiles[i].nLine + 0.3 * tiles[i].nMixed + (tiles[i].nLine == 0 ? -0.25 : 0.0);
            std::vector<int32_t> order(nch);
            for (int ch = 0; ch < nch; ++ch)
                order[ch] = ch;
            const bool heavyFirst = laneOrder == 1;
            std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return heavyFirst ? cost[x] > cost[y] : cost[x] < cost[y]; });
            HIP_TRY(up(c->chunkOrder, order, like ? &like->chunkOrder : nullptr));
            c->chunkOrderOn = true;
        }
        if (c->laneSweep)
        {
            // the flat records of the lane sweep's tasks (lwhip_internal.h: DevLaneWg, DevLaneTile, DevLaneRay, DevLaneFin)
            const int L = c->tileL;
            auto wg_records = [&](const std::vector<int32_t>& chTile, const std::vector<int32_t>* order, const std::vector<int32_t>* split,
                                  int launchSplit) {
                const int nch = (int)chTile.size() - 1;
                std::vector<DevLaneWg> w(std::max(nch, 1));
                for (int b = 0; b < nch; ++b)
                {
                    const int ch = order ? (*order)[b] : b;
                    const int S = split ? (*split)[ch] : launchSplit;
                    w[b] = DevLaneWg{ ch, S > 1 ? S : 1, chTile[ch], chTile[ch + 1] };
                }
                return w;
            };
            std::vector<int32_t> orderHost;
            if (c->chunkOrderOn)
            {
                orderHost.resize(c->chunkOrder.n);
                HIP_TRY(hipMemcpy(orderHost.data(), c->chunkOrder.p, orderHost.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            }
            auto wgs = wg_records(chunkTile, c->chunkOrderOn ? &orderHost : nullptr, c->chunkSplitOn ? &laneChunkSplit : nullptr, c->laneSplit);
            HIP_TRY(c->laneWg.upload_or_borrow(wgs, like ? &like->laneWg : nullptr));
            // which tiles run the generic kind (no compiled kind for their line / mixed-continuum counts), and for those the
            // feed bytes of every (continuum, slot) pair -- DevLaneFin::feed holds two slots' worth
            std::vector<int32_t> genOf(std::max<size_t>(tiles.size(), 1), -1);
            std::vector<uint8_t> feedG;
            c->nGenTiles = 0;
            auto feed_byte = [&](const DevSlot& cs, const DevTileSlot& us, bool isThisMixed) -> uint8_t {
                uint32_t b = 0;
                if ((cs.flags & 1) && (us.flags & 1))
                {
                    b |= cs.atom == us.atom ? 1u : 0u;
                    b |= cs.gi == us.gi ? 2u : 0u;
                    b |= cs.gj == us.gi ? 4u : 0u;
                    b |= cs.gi == us.gj ? 8u : 0u;
                    b |= cs.gj == us.gj ? 16u : 0u;
                    b |= cs.gj == us.gi ? 32u : 0u;
                    b |= cs.gj == us.gj ? 64u : 0u;
                }
                if (isThisMixed)
                    b |= 128u;
                return (uint8_t)b;
            };
            for (size_t t = 0; t < tiles.size(); ++t)
            {
                const DevTile& td = tiles[t];
                const bool fastKind = (td.nLine == 0 && td.nMixed == 0) || (td.nLine == 1 && td.nMixed <= 1) || (td.nLine == 2 && td.nMixed == 0);
                if (fastKind)
                    continue;
                genOf[t] = c->nGenTiles++;
                const DevLaHeader& h0 = hdr[td.la0];
                const size_t base = feedG.size();
                feedG.resize(base + (size_t)LWHIP_POST_MAXC * LWHIP_GEN_MAXS, 0);
                const int nC = std::min(h0.nMixed + h0.nPure, (int)LWHIP_POST_MAXC);
                for (int q = 0; q < nC; ++q)
                {
                    const int ord = slots[h0.off + h0.nLine + q]._pad;
                    const DevSlot& cs = slots[h0.off + h0.nLine + ord];
                    for (int e = 0; e < std::min(td.nLine + td.nMixed, (int)LWHIP_GEN_MAXS); ++e)
                        feedG[base + (size_t)q * LWHIP_GEN_MAXS + e] = feed_byte(cs, tslots[td.slotOff + e], e >= td.nLine && ord == e - td.nLine);
                }
            }
            // the moment scratch of the generic tiles' wavefronts: every workgroup of the largest launch, W wavefronts each
            c->momA = 0;
            for (size_t t = 0; t < tiles.size(); ++t)
                if (genOf[t] >= 0)
                {
                    const int nl = tiles[t].nLine;
                    c->momA = std::max(c->momA, 4 * std::max(nl - 1, 0) + nl * (nl - 1) / 2 + 4 * nl);
                }
            if (c->momA > 0)
            {
                const size_t nWg = std::max<size_t>(chunkTile.size(), chunkTilePrd.size());
                HIP_TRY(c->momScratch.alloc(nWg * c->tileWaves * (size_t)c->momA * c->laneD * 64)); // (written before it is read: first ray / per point)
            }
            if (feedG.empty())
                feedG.push_back(0);
            HIP_TRY(c->laneFeedG.upload_or_borrow(feedG, like ? &like->laneFeedG : nullptr));
            auto tile_records = [&](const std::vector<DevTileSlot>& ts) {
                std::vector<DevLaneTile> v(std::max<size_t>(tiles.size(), 1), DevLaneTile{});
                for (size_t t = 0; t < tiles.size(); ++t)
                {
                    const DevLaHeader& h = hdr[tiles[t].la0];
                    v[t].td = tiles[t];
                    v[t].nCont = h.nPure > 0 ? h.nMixed + h.nPure : 0;
                    v[t].nContAll = std::min(h.nMixed + h.nPure, (int)LWHIP_POST_MAXC);
                    {
                        // the finish's named-register form serves the tiles whose continua all carry the common words
                        bool fast = !ppWide[t];
                        const int nC = h.nMixed + h.nPure;
                        for (int q = 0; q < nC && q < LWHIP_POST_MAXC; ++q)
                        {
                            const DevContRec& r = cr[h.off + h.nLine + q];
                            const unsigned code = (r.flags & 1) ? pp[t].code[q] : 0x9999u, ops = pp[t].ops[q];
                            if (code != 0x9999u && code != 0x1239u && code != 0x4569u)
                                fast = false;
                            if ((r.flags & 2) && (r.flags & 1) && ops != 0x03291u && ops != 0x06594u && ops != 0x0AB9Cu)
                                fast = false;
                        }
                        v[t].finFast = ppWide[t] ? 2 : (fast && dbg_env_int("LWHIP_FIN_FAST", 1) != 0) ? 1 : 0; // (2: the wide encoding, general form)
                    }
                    for (int e = 0; e < std::min(2, tiles[t].nLine + tiles[t].nMixed); ++e)
                        v[t].slot[e] = ts[tiles[t].slotOff + e];
                    v[t].gen = genOf[t];
                }
                return v;
            };
            {
                auto v = tile_records(tslots);
                HIP_TRY(c->laneTiles.upload_or_borrow(v, like ? &like->laneTiles : nullptr));
                if (!tslotsPrd.empty() && tslotsPrd.size() == tslots.size() && !tileListPrd.empty())
                {
                    v = tile_records(tslotsPrd);
                    HIP_TRY(c->laneTilesPrd.upload_or_borrow(v, like ? &like->laneTilesPrd : nullptr));
                    auto wp = wg_records(chunkTilePrd, nullptr, nullptr, c->laneSplitPrd);
                    HIP_TRY(c->laneWgPrd.upload_or_borrow(wp, like ? &like->laneWgPrd : nullptr));
                }
            }
            std::vector<DevLaneRay> rays(std::max<size_t>(tiles.size() * L, 1), DevLaneRay{});
            std::vector<DevLaneFin> fin(std::max<size_t>(tiles.size() * LWHIP_POST_MAXC, 1), DevLaneFin{});
            std::vector<double> finPar(std::max<size_t>(tiles.size() * LWHIP_POST_MAXC * L * 3, 1), 0.0);
            for (size_t t = 0; t < tiles.size(); ++t)
            {
                const DevTile& td = tiles[t];
                const DevLaHeader& h0 = hdr[td.la0];
                const int nSl = std::min(2, td.nLine + td.nMixed);
                for (int r = 0; r < L; ++r)
                {
                    const int la = td.la0 + std::min(r, td.nl - 1);
                    DevLaneRay& lr = rays[t * L + r];
                    lr.wav = c->prob.wavelength[c->laStart + la];
                    for (int e = 0; e < nSl; ++e)
                        for (int q = 0; q < 4; ++q)
                            lr.p[e][q] = slots[hdr[la].off + tslots[td.slotOff + e].e].p[q];
                }
                const int nC = h0.nMixed + h0.nPure;
                for (int q = 0; q < nC && q < LWHIP_POST_MAXC; ++q)
                {
                    const DevContRec& r = cr[h0.off + h0.nLine + q];
                    uint32_t w[8];
                    std::memcpy(w, &r, sizeof(w));
                    // the rows of the tile's slots this continuum feeds (zero_mask's matching, per continuum)
                    uint32_t feed = 0;
                    {
                        const int ord = slots[h0.off + h0.nLine + q]._pad; // transition-order position q -> slot ordinal
                        const DevSlot& cs = slots[h0.off + h0.nLine + ord];
                        for (int e = 0; e < nSl; ++e)
                        {
                            const DevTileSlot& us = tslots[td.slotOff + e];
                            uint32_t b = 0;
                            if ((cs.flags & 1) && (us.flags & 1))
                            {
                                b |= cs.atom == us.atom ? 1u : 0u;
                                b |= cs.gi == us.gi ? 2u : 0u;
                                b |= cs.gj == us.gi ? 4u : 0u;
                                b |= cs.gi == us.gj ? 8u : 0u;
                                b |= cs.gj == us.gj ? 16u : 0u;
                                b |= cs.gj == us.gi ? 32u : 0u;
                                b |= cs.gj == us.gj ? 64u : 0u;
                            }
                            if (e >= td.nLine && ord == e - td.nLine)
                                b |= 128u;
                            feed |= b << (8 * e);
                        }
                    }
                    feed |= (turnPure[t * LWHIP_POST_MAXC + q] & 0xffffu) << 16; // (fixed-order mode: the task's turn at this continuum's slot)
                    fin[t * LWHIP_POST_MAXC + q] = DevLaneFin{ w[0], w[1], w[2], w[6], w[7], pp[t].code[q], pp[t].ops[q], feed };
                    for (int rr = 0; rr < L; ++rr)
                    {
                        const int la = td.la0 + std::min(rr, td.nl - 1);
                        const char* lsB = (const char*)&slots[hdr[la].off];
                        for (int k = 0; k < 3; ++k)
                        {
                            double v;
                            std::memcpy(&v, lsB + r.pOff + 8 * k, sizeof(double));
                            finPar[((t * LWHIP_POST_MAXC + q) * L + rr) * 3 + k] = v;
                        }
                    }
                }
            }
            HIP_TRY(c->laneRays.upload_or_borrow(rays, like ? &like->laneRays : nullptr));
            HIP_TRY(c->laneFin.upload_or_borrow(fin, like ? &like->laneFin : nullptr));
            HIP_TRY(c->laneFinPar.upload_or_borrow(finPar, like ? &like->laneFinPar : nullptr));
        }
        HIP_TRY(up(c->chunkTilePrd, chunkTilePrd, like ? &like->chunkTilePrd : nullptr));
        HIP_TRY(up(c->tileListPrd, tileListPrd, like ? &like->tileListPrd : nullptr));
        HIP_TRY(up(c->tileSlotTr, tileSlotTr, like ? &like->tileSlotTr : nullptr));
        HIP_TRY(up(c->tileSlotTrPrd, tileSlotTrPrd, like ? &like->tileSlotTrPrd : nullptr));
        HIP_TRY(up(c->postChunkTile, postChunkTile, like ? &like->postChunkTile : nullptr));
        HIP_TRY(up(c->postChunkTilePrd, postChunkTilePrd, like ? &like->postChunkTilePrd : nullptr));
        HIP_TRY(up(c->postSlotTr, postSlotTr, like ? &like->postSlotTr : nullptr));
        HIP_TRY(up(c->postCs, postCs, like ? &like->postCs : nullptr));
    }
    HIP_TRY(c->transLi.upload_or_borrow(li, like ? &like->transLi : nullptr));
    HIP_TRY(c->transLj.upload_or_borrow(lj, like ? &like->transLj : nullptr));
    HIP_TRY(c->atomNlevel.upload_or_borrow(aNl, like ? &like->atomNlevel : nullptr));
    HIP_TRY(c->atomDetailed.upload_or_borrow(aDet, like ? &like->atomDetailed : nullptr));
    HIP_TRY(c->atomTrOffD.upload_or_borrow(atomTrOff32, like ? &like->atomTrOffD : nullptr));
    HIP_TRY(c->atomGammaOff.upload(c->gammaOff));
    tick("device tables, lane records");
    return LWHIP_OK;
}
}
