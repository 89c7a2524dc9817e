"""
oracle/qd_oracle/forcing.py -- TEST INFRASTRUCTURE ONLY (CPU oracle).

Two-star orbital geometry and per-cell insolation / equilibrium temperature
(pygcm/orbital.py:10-77, pygcm/forcing.py:12-165, constants.py:9-35).
"""
from __future__ import annotations

import numpy as np

from .params import SIGMA_SB, PLANET_OMEGA

G = 6.67430e-11
M_SUN = 1.989e30
L_SUN = 3.828e26
AU = 1.496e11
M_A = 0.914 * M_SUN
L_A = 0.7 * L_SUN
M_B = 0.8 * M_SUN
L_B = 0.410 * L_SUN
M_TOTAL = M_A + M_B
A_BINARY = 0.5 * AU
A_PLANET = 1.32 * AU
AXIAL_TILT = 27.0


class Orbit:
    """orbital.py:15-48"""

    def __init__(self):
        self.T_binary = 2 * np.pi * np.sqrt(A_BINARY ** 3 / (G * M_TOTAL))
        self.T_planet = 2 * np.pi * np.sqrt(A_PLANET ** 3 / (G * M_TOTAL))
        self.omega_binary = 2 * np.pi / self.T_binary
        self.omega_planet = 2 * np.pi / self.T_planet
        self.r_A = A_BINARY * (M_B / M_TOTAL)
        self.r_B = A_BINARY * (M_A / M_TOTAL)

    def stellar_positions(self, t):
        c = np.cos(self.omega_binary * t)
        s = np.sin(self.omega_binary * t)
        return self.r_A * c, self.r_A * s, -self.r_B * c, -self.r_B * s


class Forcing:
    """forcing.py:16-165"""

    def __init__(self, grid, orbit=None):
        self.grid = grid
        self.orbit = orbit or Orbit()
        tilt = np.deg2rad(AXIAL_TILT)
        self.n_hat = np.array([np.sin(tilt), 0.0, np.cos(tilt)])
        x_in = np.array([1.0, 0.0, 0.0])
        self.x_eq = x_in - np.dot(x_in, self.n_hat) * self.n_hat
        self.x_eq /= np.linalg.norm(self.x_eq)
        self.y_eq = np.cross(self.n_hat, self.x_eq)

    def star_scalars(self, t):
        """Per-star (flux, sin(delta), cos(delta), alpha) + theta: the handful of
        host scalars the per-cell cos_z formula needs (forcing.py:85-98,112-125)."""
        ang = self.orbit.omega_planet * t
        xA, yA, xB, yB = self.orbit.stellar_positions(t)
        xp = A_PLANET * np.cos(ang)
        yp = A_PLANET * np.sin(ang)
        out = []
        for (xs, ys, L) in ((xA, yA, L_A), (xB, yB, L_B)):
            vec = np.array([xs - xp, ys - yp, 0.0])
            dist = np.linalg.norm(vec)
            flux = L / (4 * np.pi * (dist ** 2))
            s_hat = vec / (np.linalg.norm(vec) + 1e-15)
            delta = np.arcsin(np.clip(np.dot(s_hat, self.n_hat), -1.0, 1.0))
            alpha = np.arctan2(np.dot(s_hat, self.y_eq), np.dot(s_hat, self.x_eq))
            out.append((float(flux), float(delta), float(alpha)))
        theta = (t * PLANET_OMEGA) % (2 * np.pi)
        return out, float(theta)

    def _single(self, flux, delta, alpha, theta):
        lon_rad = np.deg2rad(self.grid.lon_mesh)
        lat_rad = np.deg2rad(self.grid.lat_mesh)
        h = theta + lon_rad - alpha
        cos_z = np.sin(lat_rad) * np.sin(delta) + np.cos(lat_rad) * np.cos(delta) * np.cos(h)
        return flux * np.maximum(0.0, cos_z)

    def insolation_components(self, t):
        """forcing.py:78-103"""
        stars, theta = self.star_scalars(t)
        return (self._single(*stars[0], theta), self._single(*stars[1], theta))

    def insolation(self, t):
        a, b = self.insolation_components(t)
        return a + b

    def equilibrium_temp(self, t, albedo):
        """forcing.py:138-165"""
        num = self.insolation(t) * (1 - albedo)
        num[num < 0] = 0
        return (num / SIGMA_SB) ** 0.25
