"""Import shim: BloomScene does
``from depth_diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer``
(reference gaussian_renderer/__init__.py:16).  With this repository on ``sys.path`` that import
resolves to the MI355X-native implementation, so the reference's renderer and training loop run
unmodified."""
from bloomscene_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
    _RasterizeGaussians,
    cpu_deep_copy_tuple,
)
