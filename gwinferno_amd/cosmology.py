"""Flat-LCDM distance tables used at SETUP time (never inside the per-step hot path).

Mirrors the behaviour of the reference's ``gwinferno/cosmology.py`` (``Cosmology`` :27-138,
``PLANCK_2015_LVK_Cosmology`` :150-155): a comoving-distance table on ``z = arange(0, max_z, dz)``
built by the same trapezoid recurrence (``update`` :48-63 -- note it evaluates the integrand at
``z[i] + dz`` rather than ``z[i+1]``), linear interpolation into it (``z2Dc`` :111-120) and
``dVc/dz = 4 pi Dc^2 (c/H0)/E(z)`` (``dVcdz`` :95-101).  The recurrence is written here as one
vectorised cumulative sum instead of a 10 000-step loop.
"""
import numpy as np

C_SI = 299792458.0  # m/s (reference cosmology.py:13)

# Planck-2015 "LVK" parameters (reference cosmology.py:19-22); H0 in m/s/Mpc
PLANCK15_LVK_H0 = 67.90 / 1e-3
PLANCK15_LVK_OMEGA_M = 0.3065
# Planck-2015 (reference cosmology.py:14-17)
PLANCK15_H0 = 67.74 / 1e-3
PLANCK15_OMEGA_M = 0.3089

DEFAULT_DZ = 1e-3


class FlatLambdaCDM:
    """Tabulated flat LCDM cosmology (distances in Mpc, volumes in Mpc^3)."""

    def __init__(self, H0, omega_matter, omega_radiation=0.0, max_z=10.0, dz=DEFAULT_DZ):
        self.H0 = H0
        self.c_over_H0 = C_SI / H0
        self.omega_matter = omega_matter
        self.omega_radiation = omega_radiation
        self.omega_lambda = 1.0 - omega_matter
        self.omega_kappa = 1.0 - (self.omega_matter + self.omega_radiation + self.omega_lambda)
        if self.omega_kappa != 0:
            raise ValueError("only flat cosmologies are implemented (reference cosmology.py:40)")
        self._tabulate(max_z, dz)

    # -- integrand pieces (reference cosmology.py:79-93) -------------------------------------
    def efunc(self, z):
        opz = 1.0 + np.asarray(z, dtype=np.float64)
        return (self.omega_lambda + self.omega_kappa * opz**2 + self.omega_matter * opz**3 + self.omega_radiation * opz**4) ** 0.5

    def dDc_dz(self, z):
        return self.c_over_H0 / self.efunc(z)

    # -- table (reference cosmology.py:48-77) ------------------------------------------------
    def _tabulate(self, max_z, dz):
        z = np.arange(0, max_z, dz)
        step = z[1] - z[0]
        left = self.dDc_dz(z[:-1])
        right = self.dDc_dz(z[:-1] + step)
        dc = np.concatenate([[0.0], np.cumsum(0.5 * (left + right) * step)])
        dv_left = 4 * np.pi * dc[:-1] ** 2 * left
        dv_right = 4 * np.pi * dc[1:] ** 2 * right
        vc = np.concatenate([[0.0], np.cumsum(0.5 * (dv_left + dv_right) * step)])
        self.z, self.Dc, self.Vc = z, dc, vc

    def z_to_Dc(self, z):
        z = np.asarray(z, dtype=np.float64)
        if z.size and np.max(z) > self.z[-1]:
            self._tabulate(float(np.max(z)), self.z[1] - self.z[0])
        return np.interp(z, self.z, self.Dc)

    def dVc_dz(self, z):
        z = np.asarray(z, dtype=np.float64)
        return 4 * np.pi * self.z_to_Dc(z) ** 2 * self.dDc_dz(z)

    def dVc_dz_expr(self, z, max_z=None):
        """``dVc_dz`` of a per-sample setup expression ``z`` (gwinferno_amd.expr): the same operation sequence as
        :meth:`dVc_dz` -- linear interpolation into the comoving-distance table, ``4 pi Dc^2 (c/H0)/E(z)`` -- with the
        integer powers of ``1 + z`` written as products, evaluated on the device when a catalog is ingested.  ``max_z``:
        the largest redshift the expression will see (the table is extended to it first, as :meth:`z_to_Dc` does)."""
        from . import expr as E

        if max_z is not None and max_z > self.z[-1]:
            self._tabulate(float(max_z), self.z[1] - self.z[0])
        opz = 1.0 + z
        opz2 = opz * opz
        e2 = self.omega_lambda + self.omega_kappa * opz2 + self.omega_matter * (opz2 * opz) + self.omega_radiation * (opz2 * opz2)
        dc = E.interp(z, self.z, self.Dc)
        return 4 * np.pi * (dc * dc) * (self.c_over_H0 / E.sqrt(e2))

    def log_dVc_dz(self, z):
        z = np.asarray(z, dtype=np.float64)
        return np.log(4 * np.pi) + 2 * np.log(self.z_to_Dc(z)) + np.log(self.dDc_dz(z))

    def z_to_DL(self, z):
        """cosmology.py:131-138: linear interpolation into the tabulated D_L = D_c (1+z) (:122-124), not
        D_c(z) (1+z) -- the two differ by up to ~5e-5 between the dz = 1e-3 grid points."""
        z = np.asarray(z, dtype=np.float64)
        self.z_to_Dc(z)  # extends the table when needed
        return np.interp(z, self.z, self.Dc * (1 + self.z))


_PLANCK15_LVK = None


def planck15_lvk():
    """Shared instance of the cosmology every reference model uses (``Planck15`` in
    models/parametric/parametric.py:4 and models/bsplines/single.py:8)."""
    global _PLANCK15_LVK
    if _PLANCK15_LVK is None:
        _PLANCK15_LVK = FlatLambdaCDM(PLANCK15_LVK_H0, PLANCK15_LVK_OMEGA_M)
    return _PLANCK15_LVK
