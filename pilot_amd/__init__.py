"""pilot_amd -- MI355X-native pairwise-Wasserstein engine behind PILOT's ``tl.wasserstein_distance``.

``import pilot_amd as pl; pl.tl.wasserstein_distance(adata, ...)`` mirrors
``pilotpy.tl.wasserstein_distance`` (pilotpy/tools/Trajectory.py:36-115).  The pair grid and the
centroid distance matrix run on the GPU through ``libpilot_ot.so`` (ctypes, C ABI in
``include/pilot_ot.h``); there is no CPU fallback.
"""
__version__ = "0.1.0"

from . import tl  # noqa: E402,F401
