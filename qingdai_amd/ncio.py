"""
qingdai_amd/ncio.py -- the on-disk formats either side of the hot path.

One small writer / reader over netCDF4 when it is importable and NetCDF-3 (64-bit offset) through
scipy.io otherwise, so a run on a machine without netCDF4 (the GPU box) checkpoints with the same
variable names, dimensions and dtypes as the reference's files:
  restart / data/atmosphere.nc   run_simulation.py:63-124,161-183,248-270
  data/topography.nc             run_simulation.py:126-159; pygcm/topography.py:349-426 (export), 428-575 (load)
  data/ocean.nc                  run_simulation.py:185-246
Classic NetCDF has no unsigned byte: `u1` variables are written as `i1` there (values 0/1 survive).
"""
from __future__ import annotations

import os

import numpy as np


def backend():
    try:
        import netCDF4  # noqa: F401
        return "netCDF4"
    except Exception:
        return "scipy"


def write_nc(path, dims, variables, attrs=None):
    """dims: {name: size}; variables: {name: (dtype code, dim names tuple, array-or-scalar)}; attrs: {name: str|float}."""
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    attrs = attrs or {}
    if backend() == "netCDF4":
        from netCDF4 import Dataset
        with Dataset(path, "w") as ds:
            for d, n in dims.items():
                ds.createDimension(d, n)
            for name, (code, vdims, data) in variables.items():
                v = ds.createVariable(name, code, tuple(vdims))
                if vdims:
                    v[:] = np.asarray(data)
                else:
                    v[...] = data
            for k, val in attrs.items():
                ds.setncattr(k, val)
        return
    from scipy.io import netcdf_file
    with netcdf_file(path, "w", version=2) as ds:
        for d, n in dims.items():
            ds.createDimension(d, n)
        for name, (code, vdims, data) in variables.items():
            c = "i1" if code == "u1" else code
            v = ds.createVariable(name, c, tuple(vdims))
            if vdims:
                v[:] = np.asarray(data).astype(np.dtype(c))
            else:
                v[()] = data
        for k, val in attrs.items():
            setattr(ds, k, val.encode() if isinstance(val, str) else val)


def read_nc(path, names=None):
    """-> ({variable: native-endian ndarray}, {global attribute: value}); `names` restricts the variables."""
    out, attrs = {}, {}
    if backend() == "netCDF4":
        from netCDF4 import Dataset
        with Dataset(path, "r") as ds:
            for name in (names or list(ds.variables)):
                if name in ds.variables:
                    out[name] = np.array(ds.variables[name][...])
            for k in ds.ncattrs():
                attrs[k] = ds.getncattr(k)
        return out, attrs
    from scipy.io import netcdf_file
    with netcdf_file(path, "r", mmap=False) as ds:
        for name in (names or list(ds.variables)):
            if name in ds.variables:
                var = ds.variables[name]
                a = np.array(var[:]) if var.shape else np.array(var.getValue())
                out[name] = a.astype(a.dtype.newbyteorder("="))      # classic NetCDF is big-endian on disk
        for k, val in ds._attributes.items():
            attrs[k] = val.decode() if isinstance(val, bytes) else (float(val) if np.ndim(val) == 0 else val)
    return out, attrs


def read_nc_full(path):
    """-> (dims {name: size}, variables {name: (dtype code, dim names, native-endian ndarray)}, attrs): everything a file holds, in
    the shape write_nc takes -- used to carry the variables this build does not own through a rewrite of a shared file."""
    dims, out, attrs = {}, {}, {}
    if backend() == "netCDF4":
        from netCDF4 import Dataset
        with Dataset(path, "r") as ds:
            for d, dim in ds.dimensions.items():
                dims[d] = len(dim)
            for name, var in ds.variables.items():
                a = np.array(var[...])
                out[name] = (a.dtype.str.lstrip("<>=|"), tuple(var.dimensions), a)
            for k in ds.ncattrs():
                attrs[k] = ds.getncattr(k)
        return dims, out, attrs
    from scipy.io import netcdf_file
    with netcdf_file(path, "r", mmap=False) as ds:
        for d, n in ds.dimensions.items():
            dims[d] = int(n) if n is not None else 0
        for name, var in ds.variables.items():
            a = np.array(var[:]) if var.shape else np.array(var.getValue())
            a = a.astype(a.dtype.newbyteorder("="))
            out[name] = (a.dtype.str.lstrip("<>=|"), tuple(var.dimensions), a)
        for k, val in ds._attributes.items():
            attrs[k] = val.decode() if isinstance(val, bytes) else (float(val) if np.ndim(val) == 0 else val)
    return dims, out, attrs
