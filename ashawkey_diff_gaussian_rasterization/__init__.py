"""Drop-in import name used by the reference (`from ashawkey_diff_gaussian_rasterization import
GaussianRasterizationSettings, GaussianRasterizer`, /root/reference/gaussian_renderer/__init__.py:15,
utils/sam_refinement_utils.py:21).  Everything lives in opengaussian_amd.rasterizer (HIP, gfx950)."""
from opengaussian_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
)
