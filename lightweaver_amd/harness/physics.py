"""Host-side input producers for the harness (NOT on the hot path, NOT parity-checked).

The reference computes these quantities in its pure-Python layer (atmosphere.py, atomic_model.py,
atomic_set.py, broadening.py, collisional_rates.py, Background.cpp), none of which is in scope
 and whose model-atom file (rh_atoms.py) is absent from the mount.  "Identical
inputs" for parity are defined at the Context array boundary, so these producers only have to be
physically sensible and deterministic.  Constants are the reference's (Source/Constants.hpp:6-47).
"""
import numpy as np

CLight = 2.99792458E+08
HPlanck = 6.6260755E-34
HC = HPlanck * CLight
KBoltzmann = 1.380658E-23
Amu = 1.6605402E-27
MElectron = 9.1093897E-31
QElectron = 1.60217733E-19
Epsilon0 = 8.854187817E-12
ERydberg = 2.1798741E-18
EV = 1.60217733E-19
NM_TO_M = 1.0E-09
CM_TO_M = 1.0E-02
KM_TO_M = 1.0E+03
VMICRO_CHAR = 3.0e3
SIGMA_THOMSON = 6.6524587e-29  # m^2


def planck_nu(T, lambda_nm):
    """B_nu(T) at wavelength lambda [nm]; same form and cutoff as planck_nu
    (Source/LwMisc.hpp:29-46)."""
    hc_kla = HC / (KBoltzmann * NM_TO_M) / lambda_nm
    twohnu3_c2 = (2.0 * HC) / NM_TO_M**3 / lambda_nm**3
    x = hc_kla / np.asarray(T, dtype=np.float64)
    out = np.zeros_like(x)
    ok = x <= 150.0
    out[ok] = twohnu3_c2 / (np.exp(x[ok]) - 1.0)
    return out


def planck_grid(T, wavelengths_nm):
    """planck_nu for every wavelength of a grid at once, [Nlambda, Nspace]: the same elementwise operations (the
    per-wavelength factors are formed one by one exactly as planck_nu forms them), so the same bits as stacking its rows --
    5 800 calls per column were most of the time a 512-column batch spent building its problems."""
    T = np.asarray(T, dtype=np.float64)
    wl = np.asarray(wavelengths_nm)
    hc_kla = np.array([HC / (KBoltzmann * NM_TO_M) / w for w in wl])
    twohnu3_c2 = np.array([(2.0 * HC) / NM_TO_M**3 / w**3 for w in wl])
    x = hc_kla[:, None] / T[None, :]
    out = np.zeros_like(x)
    ok = x <= 150.0
    out[ok] = np.broadcast_to(twohnu3_c2[:, None], x.shape)[ok] / (np.exp(x[ok]) - 1.0)
    return out


def gauss_legendre_mu(Nrays):
    """Gauss-Legendre quadrature on mu in [0, 1] (lightweaver/atmosphere.py:1399-1408)."""
    x, w = np.polynomial.legendre.leggauss(Nrays)
    return 0.5 + 0.5 * x, 0.5 * w


def line_quadrature(lambda0, qCore, qWing, Nlambda):
    """RH-style linear-core/exponential-wing wavelength grid around lambda0 [nm], in units of
    VMICRO_CHAR Doppler widths (LinearCoreExpWings, lightweaver/atomic_model.py:279-340)."""
    beta = 1.0 if qWing <= 2.0 * qCore else qWing / (2.0 * qCore)
    Nl = Nlambda // 2 if Nlambda % 2 == 1 else (Nlambda - 1) // 2
    Nl += 1
    y = beta + np.sqrt(beta**2 + (beta - 1.0) * Nl + 2.0 - 3.0 * beta)
    b = 2.0 * np.log(y) / (Nl - 1)
    a = qWing / (Nl - 2.0 + y**2)
    nl = np.arange(Nl)
    q = a * (nl + (np.exp(b * nl) - 1.0))
    full = np.zeros(2 * Nl - 1)
    mid = Nl - 1
    full[:mid][::-1] = -q[1:]
    full[mid + 1:] = q[1:]
    return lambda0 + full * lambda0 * (VMICRO_CHAR / CLight)


def einstein_coefficients(f, gi, gj, lambda0_nm):
    """Aji, Bji, Bij from the oscillator strength (lightweaver/atomic_model.py:536-557)."""
    lam = lambda0_nm * NM_TO_M
    Cc = 2 * np.pi * (QElectron / Epsilon0) * (QElectron / MElectron) / CLight
    Aji = Cc / lam**2 * (gi / gj) * f
    Bji = lam**3 / (2.0 * HC) * Aji
    Bij = gj / gi * Bji
    return Aji, Bji, Bij


def lte_populations(E_J, g, stage, nTotal, T, ne):
    """Saha-Boltzmann populations n*[Nlevel, Nspace] normalised to nTotal.
    E_J: level energies [J] above the ground level of the lowest stage."""
    E_J = np.asarray(E_J)[:, None]
    g = np.asarray(g, dtype=np.float64)[:, None]
    stage = np.asarray(stage)
    kT = KBoltzmann * T[None, :]
    saha = (2.0 * np.pi * MElectron * KBoltzmann * T / HPlanck**2)**1.5 * 2.0 / ne  # [Nspace]
    s0 = stage.min()
    logn = np.log(g) - E_J / kT + (stage - s0)[:, None] * np.log(saha)[None, :]
    logn -= logn.max(axis=0, keepdims=True)
    n = np.exp(logn)
    n *= (nTotal / n.sum(axis=0))[None, :]
    return n


def h_minus_background(wavelength_nm, T, ne, nH_ground, with_planck=False):
    """Continuous background: H- bound-free + free-free (Gray 2005 polynomial fits), Thomson and
    Rayleigh-H scattering.  Returns chi, eta, sca [Nlambda, Nspace] in SI (m^-1, W m^-3 Hz^-1 sr^-1).
    Stands in for Source/Background.cpp (out of scope); only has to be smooth and positive."""
    lamA = np.asarray(wavelength_nm)[:, None] * 10.0
    theta = (5040.0 / T)[None, :]
    Pe = (ne * KBoltzmann * T)[None, :] * 10.0  # dyn cm^-2
    a = [1.99654, -1.18267e-5, 2.64243e-6, -4.40524e-10, 3.23992e-14, -1.39568e-18, 2.78701e-23]
    lam_bf = np.minimum(lamA, 16300.0)
    abf = sum(c * lam_bf**i for i, c in enumerate(a)) * 1e-18
    abf = np.where(lamA < 16300.0, np.maximum(abf, 0.0), 0.0)
    stim = 1.0 - np.exp(-HC / (lamA * 1e-10) / (KBoltzmann * T[None, :]))
    k_bf = 4.158e-10 * abf * Pe * theta**2.5 * 10.0**(0.754 * theta) * stim
    ll = np.log10(np.clip(lamA, 2600.0, 1.0e5))
    f0 = -2.2763 - 1.6850 * ll + 0.76661 * ll**2 - 0.053346 * ll**3
    f1 = 15.2827 - 9.2846 * ll + 1.99381 * ll**2 - 0.142631 * ll**3
    f2 = -197.789 + 190.266 * ll - 67.9775 * ll**2 + 10.6913 * ll**3 - 0.625151 * ll**4
    lt = np.log10(theta)
    k_ff = 1e-26 * Pe * 10.0**(f0 + f1 * lt + f2 * lt**2)
    chi_abs = (k_bf + k_ff) * 1e-4 * nH_ground[None, :]  # cm^2 per H -> m^2, times n_H [m^-3]
    lamR = np.maximum(lamA, 1300.0)
    sig_R = (5.799e-13 / lamR**4 + 1.422e-6 / lamR**6 + 2.784 / lamR**8) * 1e-4
    sca = SIGMA_THOMSON * ne[None, :] + sig_R * nH_ground[None, :]
    B = planck_grid(T, wavelength_nm)
    chi = chi_abs + sca
    eta = chi_abs * B
    if with_planck:
        return chi, eta, sca, B # (B_nu(T) of the grid: the caller's LTE start for J)
    return chi, eta, sca
