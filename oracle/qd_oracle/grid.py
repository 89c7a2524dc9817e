"""
oracle/qd_oracle/grid.py -- TEST INFRASTRUCTURE ONLY (CPU oracle).

Regular lat-lon grid of the reference (pygcm/grid.py:10-96): lat=linspace(-90,90),
lon=linspace(0,360) with the duplicated end meridian, plus the roll-periodic
divergence / vorticity operators (grid.py:41-88).
"""
from __future__ import annotations

import numpy as np

from .params import PLANET_RADIUS, PLANET_OMEGA


class Grid:
    def __init__(self, n_lat, n_lon):
        self.n_lat = int(n_lat)
        self.n_lon = int(n_lon)
        self.lat = np.linspace(-90, 90, self.n_lat)
        self.lon = np.linspace(0, 360, self.n_lon)
        self.lon_mesh, self.lat_mesh = np.meshgrid(self.lon, self.lat)
        self.coriolis_param = 2 * PLANET_OMEGA * np.sin(np.deg2rad(self.lat_mesh))
        self.dlat_rad = np.deg2rad(self.lat[1] - self.lat[0])
        self.dlon_rad = np.deg2rad(self.lon[1] - self.lon[0])

    def _curl_like(self, p, q_cos, sign):
        a = PLANET_RADIUS
        cos_lat = np.cos(np.deg2rad(self.lat_mesh))
        capped = np.maximum(cos_lat, 1e-6)
        dp_dlon = (np.roll(p, -1, axis=1) - np.roll(p, 1, axis=1)) / (2 * self.dlon_rad)
        qc = q_cos * cos_lat
        dq_dlat = (np.roll(qc, -1, axis=0) - np.roll(qc, 1, axis=0)) / (2 * self.dlat_rad)
        dq_dlat[0, :] = 0
        dq_dlat[-1, :] = 0
        if sign > 0:
            return (1 / (a * capped)) * (dp_dlon + dq_dlat)
        return (1 / (a * capped)) * (dp_dlon - dq_dlat)

    def divergence(self, u, v):
        """grid.py:41-68"""
        return self._curl_like(u, v, +1)

    def vorticity(self, u, v):
        """grid.py:70-88"""
        return self._curl_like(v, u, -1)
