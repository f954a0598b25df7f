"""Synthetic cameras in the conventions the rasterizer consumes.

The rasterizer takes its matrices from the caller (``GaussianRasterizationSettings.viewmatrix / projmatrix /
campos``); this module only builds such matrices for the synthetic scenes of the benchmarks and tests, derived
here in closed form from what the kernels do with them:

* a point is transformed as ``p' = m[0] x + m[4] y + m[8] z + m[12]`` etc. (reference
  ``cuda_rasterizer/auxiliary.h:58-77``), i.e. the 16 floats are the column-vector matrix stored column by
  column -- as a row-major ``[4, 4]`` tensor that is the TRANSPOSE of the usual matrix.  BloomScene's cameras hold
  exactly these transposed tensors (``world_view_transform``, ``full_proj_transform``; reference
  ``scene/cameras.py:59-62,75-78``);
* view space looks down +z, x right, y down; ``p_hom.w`` = view depth (reference ``utils/graphics.py:57-77``
  with ``z_sign = 1``), so for a symmetric frustum the projection is
  ``diag(1 / tan(fovx / 2), 1 / tan(fovy / 2))`` on x, y and ``(zfar z - zfar znear) / (zfar - znear)`` on z;
* ``campos`` is the camera position in world space.

The rotate360 sweep (reference ``utils/trajectory.py:16-24,110-121`` -> ``scene/dataset_readers.py:105-131``) is
N cameras at the origin, view i yawed by ``360 i / N`` degrees about +Y, ``FoVx = 0.95 * camera_angle_x``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class MiniCam:
    """The camera attributes BloomScene's render() reads (reference scene/cameras.py:67-78)."""
    image_width: int
    image_height: int
    FoVy: float
    FoVx: float
    znear: float
    zfar: float
    world_view_transform: torch.Tensor   # [4, 4], transposed world-to-view matrix
    full_proj_transform: torch.Tensor    # [4, 4], world_view_transform @ projection^T
    camera_center: torch.Tensor          # [3]

    def to(self, device):
        return MiniCam(self.image_width, self.image_height, self.FoVy, self.FoVx, self.znear, self.zfar,
                       self.world_view_transform.to(device), self.full_proj_transform.to(device),
                       self.camera_center.to(device))


def vertical_fov(fovx, width, height):
    """FoVy of square pixels: tan(fovy / 2) = tan(fovx / 2) * height / width."""
    return 2.0 * math.atan(math.tan(0.5 * fovx) * height / width)


def projection_transposed(znear, zfar, fovx, fovy):
    """Transposed projection of a symmetric frustum (see the module docstring): float32 [4, 4]."""
    Pt = torch.zeros(4, 4)
    Pt[0, 0] = 1.0 / math.tan(0.5 * fovx)
    Pt[1, 1] = 1.0 / math.tan(0.5 * fovy)
    Pt[2, 2] = zfar / (zfar - znear)
    Pt[3, 2] = -(zfar * znear) / (zfar - znear)
    Pt[2, 3] = 1.0                               # w = view z
    return Pt


def make_minicam(cam_to_world_rot, world_to_cam_trans, fovx, fovy, width, height, znear=0.01, zfar=100.0, device="cpu"):
    """Camera with axes ``cam_to_world_rot`` (3x3, columns = the camera's x / y / z axes in world space) and
    world-to-view translation ``world_to_cam_trans``:  p_view = R^T p_world + t."""
    R = np.asarray(cam_to_world_rot, dtype=np.float64)
    t = np.asarray(world_to_cam_trans, dtype=np.float64).reshape(3)
    Vt = np.zeros((4, 4), dtype=np.float64)      # transposed [R^T | t; 0 0 0 1]
    Vt[:3, :3] = R                               # (R^T)^T
    Vt[3, :3] = t
    Vt[3, 3] = 1.0
    view_t = torch.from_numpy(Vt.astype(np.float32))
    full_t = view_t @ projection_transposed(znear, zfar, fovx, fovy)
    centre = torch.from_numpy((-(R @ t)).astype(np.float32))     # p_view = 0  <=>  p_world = -R t
    return MiniCam(int(width), int(height), float(fovy), float(fovx), znear, zfar,
                   view_t.contiguous().to(device), full_t.contiguous().to(device), centre.contiguous().to(device))


def yaw_rotation(yaw_deg):
    """Camera-to-world rotation of a camera turned by ``yaw_deg`` about the world's +Y axis (its view-space
    rotation is [[c, 0, s], [0, 1, 0], [-s, 0, c]], the pose matrix of the rotate360 preset)."""
    th = math.radians(yaw_deg)
    c, s = math.cos(th), math.sin(th)
    return np.array([[c, 0.0, -s], [0.0, 1.0, 0.0], [s, 0.0, c]])


def rotate360_cameras(n_views, width, height, camera_angle_x, device="cpu"):
    """The rotate360 view list: n_views cameras at the origin, yaw 360 i / n_views degrees, FoVx = 0.95 * angle."""
    fovx = 0.95 * camera_angle_x
    fovy = vertical_fov(fovx, width, height)
    return [make_minicam(yaw_rotation(360.0 / n_views * i), np.zeros(3), fovx, fovy, width, height, device=device)
            for i in range(n_views)]


def identity_camera(width, height, fovx, device="cpu"):
    """Camera at the origin looking down +z (synthetic scene A)."""
    return make_minicam(np.eye(3), np.zeros(3), fovx, vertical_fov(fovx, width, height), width, height, device=device)
