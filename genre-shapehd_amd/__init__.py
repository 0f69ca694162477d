"""genre-shapehd_amd -- MI355X-native geometric hot path of GenRe / ShapeHD.

Hand-written gfx950 HIP kernels (csrc/, C ABI in include/genre_hip.h) behind the
reference's own ``torch.autograd.Function`` boundary:

    CameraBackProjection, SphericalBackProjection, get_surface_mask,
    Camera_back_projection_layer           (toolbox/cam_bp)
    CalcStopProb                           (toolbox/calc_prob)
    render_spherical, sph_pad, gen_sph_grid (toolbox/spherical_proj.py)
    NNDFunction, nndistance, nndistance_w_idx, nndistance_score, NNDModule
                                           (toolbox/nndistance)

The directory name carries a hyphen, so import it through the top-level alias
module ``genre_shapehd_amd`` (repo root), or put this directory (and
``toolbox/`` for ``nndistance``) on ``sys.path`` and keep the reference's own
import lines (``from toolbox.cam_bp.cam_bp.functions import ...``) unchanged.
"""
from .toolbox.cam_bp.cam_bp.functions import CameraBackProjection, SphericalBackProjection, get_surface_mask
from .toolbox.cam_bp.cam_bp.modules.camera_backprojection_module import Camera_back_projection_layer
from .toolbox.calc_prob.calc_prob.functions.calc_prob import CalcStopProb
from .toolbox.spherical_proj import render_spherical, sph_pad, gen_sph_grid
from .toolbox.nndistance.functions.nnd import (NNDFunction, nndistance, nndistance_w_idx, nndistance_score)
from .toolbox.nndistance.modules.nnd import NNDModule

__all__ = ["CameraBackProjection", "SphericalBackProjection", "get_surface_mask",
           "Camera_back_projection_layer", "CalcStopProb", "render_spherical", "sph_pad",
           "gen_sph_grid", "NNDFunction", "nndistance", "nndistance_w_idx", "nndistance_score",
           "NNDModule"]
