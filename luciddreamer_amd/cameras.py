"""Camera matrices in the conventions the rasterizer consumes.

Restates the matrix conventions of the reference (nothing here is on the hot path):
  - world->view / projection construction: /root/reference/utils/graphics.py:41-75
  - row-vector ("transposed") storage and full_proj = view @ proj, camera_center =
    inverse(view)[3, :3]: /root/reference/scene/cameras.py:58-61, 64-75 (MiniCam)
  - FoVy from FoVx and the aspect ratio: /root/reference/utils/camera.py:27-29
  - camera paths "rotate360" (pure rotations about the world y axis at the origin) and
    "lookaround": the poses stored in /root/reference/cameras/rotate360.json are, after the reference's
    OpenGL -> COLMAP axis flip, R_y(-theta_i), theta_i = i * 0.5 deg, i = 0..719, no translation; they are
    regenerated analytically here so that nothing reads /root/reference at run time.
"""
import math
from typing import List, NamedTuple

import numpy as np
import torch

FOVX_DEFAULT = 0.8279103882874479  # camera_angle_x of every preset in /root/reference/cameras/*.json


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def fovy_from_fovx(fovx, width, height):
    return focal2fov(fov2focal(fovx, width), height)


def world2view(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    """4x4 world->camera matrix for rotation R (stored transposed, as the reference does) and
    translation t (graphics.py:41-48 with translate=0, scale=1)."""
    Rt = np.zeros((4, 4), dtype=np.float64)
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    return Rt.astype(np.float32)


def projection_matrix(znear, zfar, fovx, fovy) -> torch.Tensor:
    """graphics.py:55-75."""
    tan_y = math.tan(fovy / 2)
    tan_x = math.tan(fovx / 2)
    top, right = tan_y * znear, tan_x * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


class MiniCam(NamedTuple):
    """Field names follow /root/reference/scene/cameras.py:64-75 so render() can consume it."""
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
        return self._replace(world_view_transform=self.world_view_transform.to(device),
                             full_proj_transform=self.full_proj_transform.to(device),
                             camera_center=self.camera_center.to(device))


def make_camera(c2w: np.ndarray, width: int, height: int, fovx: float = FOVX_DEFAULT,
                znear: float = 0.01, zfar: float = 100.0) -> MiniCam:
    """c2w: 4x4 (or 3x4) camera-to-world in the COLMAP axis convention (y down, z forward)."""
    c2w = np.asarray(c2w, dtype=np.float64)
    if c2w.shape[0] == 3:
        c2w = np.concatenate([c2w, np.array([[0, 0, 0, 1.0]])], axis=0)
    w2c = np.linalg.inv(c2w)
    R = np.transpose(w2c[:3, :3])
    T = w2c[:3, 3]
    fovy = fovy_from_fovx(fovx, width, height)
    view = torch.as_tensor(world2view(R, T)).T.contiguous()
    proj = projection_matrix(znear, zfar, fovx, fovy).T.contiguous()
    full = (view @ proj).contiguous()
    center = torch.inverse(view)[3][:3].contiguous()
    return MiniCam(width, height, fovy, fovx, znear, zfar, view, full, center)


def identity_camera(width, height, fovx=FOVX_DEFAULT) -> MiniCam:
    """Camera at the origin looking down +z (the 'box cloud' view of SURVEY.md section 8d)."""
    return make_camera(np.eye(4), width, height, fovx)


def _rot_y(theta):
    c, s = math.cos(theta), math.sin(theta)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def _rot_x(phi):
    c, s = math.cos(phi), math.sin(phi)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def rotate360_path(width, height, n_views=30, n_frames=720, fovx=FOVX_DEFAULT) -> List[MiniCam]:
    """Every (n_frames // n_views)-th pose of the 720-frame rotate360 preset.  Frame i of
    /root/reference/cameras/rotate360.json is the OpenGL-convention c2w [[c,0,s],[0,-1,0],[s,0,-c]], theta = i * 0.5 deg;
    after the reference's axis flip `c2w[:3, 1:3] *= -1` (scene/dataset_readers.py:267) that is R_y(-theta) in the COLMAP
    convention make_camera expects (tests/test_oracle_golden.py checks the match against the JSON)."""
    stride = max(1, n_frames // n_views)
    cams = []
    for i in list(range(0, n_frames, stride))[:n_views]:
        c2w = np.eye(4)
        c2w[:3, :3] = _rot_y(-2.0 * math.pi * i / n_frames)
        cams.append(make_camera(c2w, width, height, fovx))
    return cams


def lookaround_path(width, height, n_views=30, max_yaw_deg=30.0, max_pitch_deg=15.0,
                    fovx=FOVX_DEFAULT) -> List[MiniCam]:
    """A smooth look-around sweep about the origin (yaw/pitch Lissajous), same convention."""
    cams = []
    for i in range(n_views):
        a = 2.0 * math.pi * i / max(1, n_views)
        yaw = math.radians(max_yaw_deg) * math.sin(a)
        pitch = math.radians(max_pitch_deg) * math.sin(2 * a)
        c2w = np.eye(4)
        c2w[:3, :3] = _rot_y(yaw) @ _rot_x(pitch)
        cams.append(make_camera(c2w, width, height, fovx))
    return cams
