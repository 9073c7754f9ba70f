"""MI355X-native per-pixel hot path of ins2cvt/renderer: host-side mirror of the C ABI in include/invert.h."""
from . import api  # noqa: F401
from .api import (BLOCKS, BRUTE_FORCE, INVERT, LAMBERT, LANE_TRAVERSAL, RGBA, ROOT_IDLE, SPARSE, TILE_TRAVERSAL, Camera, Context, InvertError, Mesh,
                  Outputs, Params, Scene, camera_reference, camera_turntable, make_params, sparse_layout, tile_layout, version)

__all__ = ["api", "BLOCKS", "BRUTE_FORCE", "INVERT", "LAMBERT", "LANE_TRAVERSAL", "RGBA", "ROOT_IDLE", "SPARSE", "TILE_TRAVERSAL", "Camera", "Context",
           "InvertError", "Mesh", "Outputs", "Params", "Scene", "camera_reference", "camera_turntable", "make_params", "sparse_layout",
           "tile_layout", "version"]
