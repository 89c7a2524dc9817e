"""
qingdai_amd/grid.py -- host mirror of pygcm/grid.py:10-96 (SphericalGrid): the regular
lat-lon grid with its duplicated end meridian and Coriolis map.  Pure host data; the
divergence / vorticity operators run on the device through the operator seam.
"""
from __future__ import annotations

import numpy as np

from .params import PLANET_OMEGA


class SphericalGrid:
    def __init__(self, n_lat, n_lon):
        self.n_lat = int(n_lat)
        self.n_lon = int(n_lon)
        self.lat = np.linspace(-90, 90, self.n_lat)
        self.lon = np.linspace(0, 360, self.n_lon)
        self.lon_mesh, self.lat_mesh = np.meshgrid(self.lon, self.lat)
        self.coriolis_param = 2 * PLANET_OMEGA * np.sin(np.deg2rad(self.lat_mesh))
        self.dlat_rad = np.deg2rad(self.lat[1] - self.lat[0])
        self.dlon_rad = np.deg2rad(self.lon[1] - self.lon[0])
        self._device = None          # set by the first model built on this grid

    def _ops(self):
        if self._device is None:
            from .device import Device
            self._device = Device(self)
        return self._device

    def divergence(self, u, v):
        """grid.py:41-68 on the device."""
        return self._ops().op_divvort(u, v, vort=False)

    def vorticity(self, u, v):
        """grid.py:70-88 on the device."""
        return self._ops().op_divvort(u, v, vort=True)
