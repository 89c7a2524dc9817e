"""
qingdai_amd/forcing.py -- mirror of pygcm/orbital.py:10-77 and pygcm/forcing.py:12-165.

The orbital geometry is a handful of host scalars per step (two stars: flux,
declination, right ascension; planet rotation angle theta); the per-cell cos-zenith
insolation and Teq = (I (1-albedo) / sigma)^(1/4) run on the device (qd_forcing).
"""
from __future__ import annotations

import numpy as np

from .params import PLANET_OMEGA, SIGMA

G = 6.67430e-11
M_SUN = 1.989e30
L_SUN = 3.828e26
AU = 1.496e11
M_A = 0.914 * M_SUN
L_A = 0.7 * L_SUN
M_B = 0.8 * M_SUN
L_B = 0.410 * L_SUN
M_TOTAL_STARS = M_A + M_B
A_BINARY = 0.5 * AU
A_PLANET = 1.32 * AU
PLANET_AXIAL_TILT = 27.0


class OrbitalSystem:
    def __init__(self):
        self.T_binary = 2 * np.pi * np.sqrt(A_BINARY ** 3 / (G * M_TOTAL_STARS))
        self.T_planet = 2 * np.pi * np.sqrt(A_PLANET ** 3 / (G * M_TOTAL_STARS))
        self.omega_binary = 2 * np.pi / self.T_binary
        self.omega_planet = 2 * np.pi / self.T_planet
        self.r_A = A_BINARY * (M_B / M_TOTAL_STARS)
        self.r_B = A_BINARY * (M_A / M_TOTAL_STARS)

    def calculate_stellar_positions(self, t):
        c, s = np.cos(self.omega_binary * t), np.sin(self.omega_binary * t)
        return self.r_A * c, self.r_A * s, -self.r_B * c, -self.r_B * s

    def calculate_total_flux(self, t):
        x_A, y_A, x_B, y_B = self.calculate_stellar_positions(t)
        x_p = A_PLANET * np.cos(self.omega_planet * t)
        y_p = A_PLANET * np.sin(self.omega_planet * t)
        d_A = np.sqrt((x_p - x_A) ** 2 + (y_p - y_A) ** 2)
        d_B = np.sqrt((x_p - x_B) ** 2 + (y_p - y_B) ** 2)
        return L_A / (4 * np.pi * d_A ** 2) + L_B / (4 * np.pi * d_B ** 2)


class ThermalForcing:
    def __init__(self, grid, orbital_system):
        self.grid = grid
        self.orbital_system = orbital_system
        tilt = np.deg2rad(PLANET_AXIAL_TILT)
        self.n_hat = np.array([np.sin(tilt), 0.0, np.cos(tilt)])
        x_in = np.array([1.0, 0.0, 0.0])
        self.x_eq = x_in - np.dot(x_in, self.n_hat) * self.n_hat
        self.x_eq /= np.linalg.norm(self.x_eq)
        self.y_eq = np.cross(self.n_hat, self.x_eq)

    def star_scalars(self, t):
        """((flux, delta, alpha) for star A, same for B, theta) -- forcing.py:85-98,112-125."""
        o = self.orbital_system
        ang = o.omega_planet * t
        xA, yA, xB, yB = o.calculate_stellar_positions(t)
        xp, yp = A_PLANET * np.cos(ang), A_PLANET * np.sin(ang)
        out = []
        for xs, ys, L in ((xA, yA, L_A), (xB, yB, L_B)):
            vec = np.array([xs - xp, ys - yp, 0.0])
            dist = np.linalg.norm(vec)
            flux = L / (4 * np.pi * (dist ** 2))
            s_hat = vec / (np.linalg.norm(vec) + 1e-15)
            delta = np.arcsin(np.clip(np.dot(s_hat, self.n_hat), -1.0, 1.0))
            alpha = np.arctan2(np.dot(s_hat, self.y_eq), np.dot(s_hat, self.x_eq))
            out.append((float(flux), float(delta), float(alpha)))
        theta = float((t * PLANET_OMEGA) % (2 * np.pi))
        return out[0], out[1], theta

    def star_table(self, times):
        """[n][7] rows (flux_A, decl_A, ra_A, flux_B, decl_B, ra_B, theta) for qd_step_n."""
        rows = []
        for t in times:
            a, b, th = self.star_scalars(float(t))
            rows.append([*a, *b, th])
        return np.asarray(rows, dtype=np.float64)

    def _dev(self):
        return self.grid._ops()

    def update_device(self, t, with_teq=True):
        """isr_A / isr_B / isr (and Teq from the resident albedo) computed in place on the device."""
        a, b, th = self.star_scalars(t)
        self._dev().forcing(a, b, th, with_teq)

    def calculate_insolation_components(self, t):
        self.update_device(t, with_teq=False)
        d = self._dev()
        return d.get("ISR_A").copy(), d.get("ISR_B").copy()

    def calculate_insolation(self, t):
        self.update_device(t, with_teq=False)
        return self._dev().get("ISR").copy()

    def calculate_equilibrium_temp(self, t, albedo):
        d = self._dev()
        d.set("ALBEDO", albedo)
        self.update_device(t, with_teq=True)
        return d.get("TEQ").copy()
