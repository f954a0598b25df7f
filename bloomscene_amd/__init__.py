"""MI355X-native (gfx950) drop-in for BloomScene's depth-diff Gaussian rasterizer -- hot path only.

Public surface = the reference module's: ``GaussianRasterizationSettings``, ``GaussianRasterizer``
(``forward`` / ``visible_filter`` / ``markVisible``) and ``rasterize_gaussians``.  Native code lives
in ``libbloomscene_rast.so`` (``csrc/``, C ABI in ``include/bloomscene_rast.h``); importing this
package does not need a GPU, calling it does, and there is no CPU fallback.
"""
from .rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
)
from .numerics import numerics  # noqa: F401  (thread-local default of the per-call numerics flags)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "numerics"]
