"""Camera conventions consumed by the rasterizer.

Restates, for the host side of the hot path only, what BloomScene's callers hand to
``GaussianRasterizationSettings``:

* ``getWorld2View2`` / ``getProjectionMatrix``   -- reference ``utils/graphics.py:43-77``
* ``MiniCam``                                    -- reference ``scene/cameras.py:67-78``
* the rotate360 preset (yaw about +Y, zero translation)
                                                 -- reference ``utils/trajectory.py:16-24,102-126``
  turned into cameras as ``loadCameraPreset`` does -- reference ``scene/dataset_readers.py:101-131``

Matrices are stored transposed (row-major tensors whose flat index ``m[4*j+i]`` is the
column-vector matrix element (i, j)), which is what the kernels index (reference
``cuda_rasterizer/auxiliary.h:58-77``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch


def get_world2view2(R, t, translate=np.array([0.0, 0.0, 0.0]), scale=1.0):
    """reference utils/graphics.py:43-54 (R is stored transposed, 'glm' convention)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    cam_center = C2W[:3, 3]
    cam_center = (cam_center + translate) * scale
    C2W[:3, 3] = cam_center
    Rt = np.linalg.inv(C2W)
    return np.float32(Rt)


def get_projection_matrix(znear, zfar, fovX, fovY):
    """reference utils/graphics.py:57-77."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = torch.zeros(4, 4)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


@dataclass
class MiniCam:
    """Same attributes as reference scene/cameras.py:67-78 (tensors live on ``device``)."""
    image_width: int
    image_height: int
    FoVy: float
    FoVx: float
    znear: float
    zfar: float
    world_view_transform: torch.Tensor
    full_proj_transform: torch.Tensor
    camera_center: torch.Tensor

    def to(self, device):
        return MiniCam(self.image_width, self.image_height, self.FoVy, self.FoVx, self.znear, self.zfar,
                       self.world_view_transform.to(device), self.full_proj_transform.to(device),
                       self.camera_center.to(device))


def make_minicam(R, T, fovx, fovy, width, height, znear=0.01, zfar=100.0, device="cpu"):
    """Build the three tensors exactly as reference scene/dataset_readers.py:124-131 does."""
    world_view_transform = torch.tensor(get_world2view2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
    projection_matrix = get_projection_matrix(znear=znear, zfar=zfar, fovX=fovx, fovY=fovy).transpose(0, 1)
    full_proj_transform = (world_view_transform.unsqueeze(0).bmm(projection_matrix.unsqueeze(0))).squeeze(0)
    camera_center = torch.inverse(world_view_transform)[3][:3]
    return MiniCam(int(width), int(height), float(fovy), float(fovx), znear, zfar,
                   world_view_transform.contiguous().to(device), full_proj_transform.contiguous().to(device),
                   camera_center.contiguous().to(device))


def generate_seed_360(viewangle, n_views):
    """reference utils/trajectory.py:16-24 (in-place yaw, zero translation)."""
    N = n_views
    render_poses = np.zeros((N, 3, 4))
    for i in range(N):
        th = (viewangle / N) * i / 180 * np.pi
        render_poses[i, :3, :3] = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    return render_poses


def rotate360_cameras(n_views, width, height, camera_angle_x, device="cpu"):
    """The rotate360 view list: ``get_camerapaths`` (reference utils/trajectory.py:102-126)
    followed by ``loadCameraPreset`` (reference scene/dataset_readers.py:101-131)."""
    yz_reverse = np.array([[1, 0, 0], [0, -1, 0], [0, 0, -1]])
    fovx = camera_angle_x * 0.95  # dataset_readers.py:105
    fovy = focal2fov(fov2focal(fovx, width), height)
    cams = []
    for pose in generate_seed_360(360, n_views):
        Rw2i = pose[:3, :3]
        Tw2i = pose[:3, 3:4]
        Ri2w = np.matmul(yz_reverse, Rw2i).T
        Ti2w = -np.matmul(Ri2w, np.matmul(yz_reverse, Tw2i))
        c2w = np.concatenate((Ri2w, Ti2w), axis=1)
        c2w = np.concatenate((c2w, np.array([0, 0, 0, 1]).reshape((1, 4))), axis=0)
        c2w[:3, 1:3] *= -1  # OpenGL/Blender -> COLMAP axes, dataset_readers.py:115
        w2c = np.linalg.inv(c2w)
        R = np.transpose(w2c[:3, :3])
        T = w2c[:3, 3]
        cams.append(make_minicam(R, T, fovx, fovy, width, height, device=device))
    return cams


def identity_camera(width, height, fovx, device="cpu"):
    """Camera at the origin looking down +z (synthetic scene A)."""
    fovy = 2.0 * math.atan(math.tan(fovx / 2.0) * height / width)
    return make_minicam(np.eye(3), np.zeros(3), fovx, fovy, width, height, device=device)
