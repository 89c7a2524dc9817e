"""
qingdai_amd -- MI355X-native per-timestep grid update of PyGCM-for-Qingdai.

Python host code (this package) calls hand-written HIP kernels for gfx950 through the
C-ABI of include/qingdai_hip.h (ctypes; no PyTorch, no Triton).  The classes mirror the
reference's own interfaces for this path:

    SphericalGrid            pygcm/grid.py
    SpectralModel            pygcm/dynamics.py
    WindDrivenSlabOcean      pygcm/ocean.py
    OrbitalSystem, ThermalForcing   pygcm/orbital.py, pygcm/forcing.py
    hip_compat (is_enabled, backend, to_numpy, laplacian_sphere, hyperdiffuse, advect_semilag)   pygcm/jax_compat.py
    DoubleBufferingArray     pygcm/numerics/double_buffer.py
    energy (compute_energy_diagnostics, autotune_greenhouse_params)   pygcm/energy.py:494-579

There is no CPU fallback: importing is cheap, but building a model without
libqingdai_hip.so or without a GPU raises.
"""
from .params import QdParams                           # noqa: F401
from .grid import SphericalGrid                        # noqa: F401
from .dynamics import SpectralModel                    # noqa: F401
from .ocean import WindDrivenSlabOcean                 # noqa: F401
from .forcing import OrbitalSystem, ThermalForcing     # noqa: F401
from . import topography                               # noqa: F401
from .double_buffer import DoubleBufferingArray       # noqa: F401
from . import hip_compat, energy, ncio, phyto, spectral  # noqa: F401

__version__ = "0.1.0"
