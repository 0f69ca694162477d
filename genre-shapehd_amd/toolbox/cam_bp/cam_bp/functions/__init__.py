"""public names of the back-projection ops (the reference exports the same three, functions/__init__.py:1-3)"""
from . import cam_back_projection as _cam, get_surface_mask as _mask, sperical_to_tdf as _sph

CameraBackProjection = _cam.CameraBackProjection
SphericalBackProjection = _sph.SphericalBackProjection
get_surface_mask = _mask.get_surface_mask

__all__ = ["CameraBackProjection", "SphericalBackProjection", "get_surface_mask"]
