"""Box geometry around the hot path: host-side numpy for the handful of boxes of a frame, HIP kernels
(through the C ABI) for everything that is per point.

Mirrors, under the same names, the functions of mmdet/core/bbox3d/geometry.py that the dataset, the augmentor and the data
preparation call:
    corners_nd :289, rotation_2d :323, rotation_3d_in_axis :338, center_to_corner_box2d :358, center_to_corner_box3d :380,
    rotation_points_single_angle :485, minmax_to_corner_2d :540, filter_gt_box_outside_range :546,
    corner_to_surfaces_3d(_jit) :76,561, surface_equ_3d_jit :176, box_camera_to_lidar / camera_to_lidar :36-48,
    projection_matrix_to_CRT_kitti :23, get_frustum :4                                       -- numpy, same arithmetic
    points_in_convex_polygon_3d_jit :189, points_in_rbbox :63, remove_outside_points :50    -- sassd_points_in_polytopes (GPU)
    box_collision_test :593                                                                  -- sassd_box_collision_test (host C++)
The per-point functions take either a numpy array (uploaded, result downloaded -- the reference's calling convention) or
a torch tensor that already lives on the GPU (result stays there)."""
import ctypes

import numpy as np
import torch

from . import _C
from .kitti_common import limit_period  # noqa: F401  (geometry.py:404)

_FACES = np.array([[0, 1, 2, 3], [7, 6, 5, 4], [0, 3, 7, 4], [1, 5, 6, 2], [0, 4, 5, 1], [3, 2, 6, 7]])


def corners_nd(dims, origin=0.5):
    """[N,ndim] side lengths -> [N,2**ndim,ndim] corner offsets around `origin` (2-D: clockwise from the minimum corner;
    3-D: x0y0z0, x0y0z1, x0y1z1, x0y1z0, x1y0z0, x1y0z1, x1y1z1, x1y1z0)."""
    ndim = int(dims.shape[1])
    unit = np.stack(np.unravel_index(np.arange(2 ** ndim), [2] * ndim), axis=1).astype(dims.dtype)
    unit = unit[[0, 1, 3, 2]] if ndim == 2 else unit[[0, 1, 3, 2, 4, 5, 7, 6]] if ndim == 3 else unit
    unit = unit - np.array(origin, dtype=dims.dtype)
    return dims.reshape([-1, 1, ndim]) * unit.reshape([1, 2 ** ndim, ndim])


def rotation_2d(points, angles):
    s, c = np.sin(angles), np.cos(angles)
    return np.einsum('aij,jka->aik', points, np.stack([[c, -s], [s, c]]))


def rotation_3d_in_axis(points, angles, axis=0):
    s, c = np.sin(angles), np.cos(angles)
    o, z = np.ones_like(c), np.zeros_like(c)
    if axis == 1:
        rot_t = np.stack([[c, z, -s], [z, o, z], [s, z, c]])
    elif axis in (2, -1):
        rot_t = np.stack([[c, -s, z], [s, c, z], [z, z, o]])
    elif axis == 0:
        rot_t = np.stack([[z, c, -s], [z, s, c], [o, z, z]])
    else:
        raise ValueError("axis should in range")
    return np.einsum('aij,jka->aik', points, rot_t)


def center_to_corner_box2d(centers, dims, angles=None, origin=0.5):
    corners = corners_nd(dims, origin=origin)
    if angles is not None:
        corners = rotation_2d(corners, angles)
    corners += centers.reshape([-1, 1, 2])
    return corners


def center_to_corner_box3d(centers, origin=(0.5, 0.5, 0), axis=2):
    """[N,7] boxes (x, y, z, dims[3], angle) -> [N,8,3]; origin (0.5, 0.5, 0) / axis 2 for lidar boxes, (0.5, 1.0, 0.5) /
    axis 1 for camera boxes."""
    corners = corners_nd(centers[:, 3:6], origin=list(origin))
    corners = rotation_3d_in_axis(corners, centers[:, -1], axis=axis)
    corners += centers[:, :3].reshape([-1, 1, 3])
    return corners


def rotation_points_single_angle(points, angle, axis=0):
    s, c = np.sin(angle), np.cos(angle)
    if axis == 1:
        rot_t = [[c, 0, -s], [0, 1, 0], [s, 0, c]]
    elif axis in (2, -1):
        rot_t = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
    elif axis == 0:
        rot_t = [[1, 0, 0], [0, c, -s], [0, s, c]]
    else:
        raise ValueError("axis should in range")
    return points @ np.array(rot_t, dtype=points.dtype)


def minmax_to_corner_2d(minmax_box):
    ndim = minmax_box.shape[-1] // 2
    lo = minmax_box[..., :ndim]
    return center_to_corner_box2d(lo, minmax_box[..., ndim:] - lo, origin=0.0)


def points_in_convex_polygon_jit(points, polygon, clockwise=True):
    """[P,2] points x [M,K,2] convex polygons -> [P,M] bool (strictly inside)."""
    prev = np.roll(polygon, 1, axis=1)
    edge = (polygon - prev) if clockwise else (prev - polygon)
    cross = edge[None, :, :, 1] * (polygon[None, :, :, 0] - points[:, None, None, 0])
    cross = cross - edge[None, :, :, 0] * (polygon[None, :, :, 1] - points[:, None, None, 1])
    return ~(cross >= 0).any(-1)


def filter_gt_box_outside_range(gt_boxes, limit_range):
    """keep a box when at least one of its BEV corners lies inside (xmin, ymin, xmax, ymax)."""
    corners = center_to_corner_box2d(gt_boxes[:, [0, 1]], gt_boxes[:, [3, 4]], gt_boxes[:, 6])
    frame = minmax_to_corner_2d(np.asarray(limit_range)[np.newaxis, ...])
    return points_in_convex_polygon_jit(corners.reshape(-1, 2), frame).reshape(-1, 4).any(axis=1)


def corner_to_surfaces_3d(corners):
    """[N,8,3] corners -> [N,6,4,3] faces, normals pointing inwards."""
    return corners[:, _FACES]


corner_to_surfaces_3d_jit = corner_to_surfaces_3d


def surface_equ_3d_jit(polygon_surfaces):
    """[N,S,>=3,3] -> (normal [N,S,3], d [N,S]) of n.x + d = 0."""
    edge = polygon_surfaces[:, :, :2, :] - polygon_surfaces[:, :, 1:3, :]
    normal = np.cross(edge[:, :, 0, :], edge[:, :, 1, :])
    return normal, -np.einsum('aij, aij->ai', normal, polygon_surfaces[:, :, 0, :])


def camera_to_lidar(points, r_rect, velo2cam):
    if points.shape[-1] == 3:
        points = np.concatenate([points, np.ones(list(points.shape[:-1]) + [1])], axis=-1)
    return (points @ np.linalg.inv((r_rect @ velo2cam).T))[..., :3]


def box_camera_to_lidar(data, r_rect, velo2cam):
    """[N,7] camera boxes (x, y, z, l, h, w, ry) -> lidar boxes (x, y, z, w, l, h, ry); 4x4 calibration matrices."""
    return np.concatenate([camera_to_lidar(data[:, 0:3], r_rect, velo2cam), data[:, 5:6], data[:, 3:4], data[:, 4:5],
                           data[:, 6:7]], axis=1)


def projection_matrix_to_CRT_kitti(proj):
    """P = C @ [R|T] with C upper triangular (QR of the inverse)."""
    cr, ct = proj[0:3, 0:3], proj[0:3, 3]
    rinv, cinv = np.linalg.qr(np.linalg.inv(cr))
    return np.linalg.inv(cinv), np.linalg.inv(rinv), cinv @ ct


def get_frustum(bbox_image, C, near_clip=0.001, far_clip=100):
    """the 8 camera-frame corners of the viewing frustum through an image box (x1, y1, x2, y2)."""
    fku, fkv, u0v0 = C[0, 0], -C[1, 1], C[0:2, 2]
    b = bbox_image
    rect = np.array([[b[0], b[1]], [b[0], b[3]], [b[2], b[3]], [b[2], b[1]]], dtype=C.dtype)
    near = (rect - u0v0) / np.array([fku / near_clip, -fkv / near_clip], dtype=C.dtype)
    far = (rect - u0v0) / np.array([fku / far_clip, -fkv / far_clip], dtype=C.dtype)
    z = np.array([near_clip] * 4 + [far_clip] * 4, dtype=C.dtype)[:, np.newaxis]
    return np.concatenate([np.concatenate([near, far], axis=0), z], axis=1)


def frustum_in_lidar(rect, Trv2c, P2, image_shape):
    """corners [8,3] (lidar frame) of the camera-2 viewing frustum of an image of shape (h, w)."""
    C, R, T = projection_matrix_to_CRT_kitti(P2)
    frustum = get_frustum([0, 0, image_shape[1], image_shape[0]], C)
    frustum -= T
    frustum = np.linalg.inv(R) @ frustum.T
    return camera_to_lidar(frustum.T, rect, Trv2c)


# ---- per-point work: GPU ------------------------------------------------------------------------------------------------

def _as_device_points(points, device):
    if torch.is_tensor(points):
        if not points.is_cuda:
            raise RuntimeError("sassd.geometry: tensors must live on the GPU (there is no CPU path)")
        return points, False
    if not torch.cuda.is_available():
        raise RuntimeError("sassd.geometry needs an MI355X for per-point work (no CPU fallback; the CPU checker lives "
                           "in oracle/ and tests/)")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    return torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(dev), True


def planes_of_surfaces(polygon_surfaces):
    """[M,6,>=3,3] faces -> ([M,6,4] float64 (nx, ny, nz, d), f32_math) for sassd_points_in_polytopes."""
    normal, d = surface_equ_3d_jit(polygon_surfaces[:, :, :3, :])
    f32 = normal.dtype == np.float32
    return np.ascontiguousarray(np.concatenate([normal, d[..., None]], axis=-1), dtype=np.float64), f32


def points_in_polytopes(points, planes, f32_math, device=None):
    """points [N,>=3] (numpy or GPU tensor) x planes [M,6,4] -> [N,M] bool (numpy for numpy input, GPU tensor otherwise)."""
    pts, was_numpy = _as_device_points(points, device)
    if pts.dtype != torch.float32 or pts.stride(-1) != 1 or pts.stride(0) < 3:
        pts = pts.float().contiguous()
    n, m = pts.shape[0], planes.shape[0]
    mask = torch.zeros((n, m), dtype=torch.uint8, device=pts.device)
    if n and m:
        pl = torch.from_numpy(planes).to(pts.device)
        with torch.cuda.device(pts.device):
            _C.check(_C.lib().sassd_points_in_polytopes(pts.data_ptr(), n, pts.stride(0), pl.data_ptr(), m, int(f32_math),
                                                        mask.data_ptr(), _C.stream()), "sassd_points_in_polytopes")
    mask = mask.bool()
    return mask.cpu().numpy() if was_numpy else mask


def points_in_convex_polygon_3d_jit(points, polygon_surfaces, num_surfaces=None, device=None):
    if num_surfaces is not None:
        raise NotImplementedError("every polytope of this path has 6 faces")
    planes, f32 = planes_of_surfaces(np.asarray(polygon_surfaces))
    return points_in_polytopes(points[:, :3] if not torch.is_tensor(points) else points, planes, f32, device)


def points_in_rbbox(points, rbbox, lidar=True, device=None):
    """points [N,>=3] x boxes [M,7] -> [N,M] bool.  The float type of `rbbox` decides the arithmetic, as in the reference:
    float32 boxes give float32 planes and a float32 sign test, float64 boxes a float64 one."""
    rbbox = np.asarray(rbbox)
    origin, axis = ((0.5, 0.5, 0), 2) if lidar else ((0.5, 1.0, 0.5), 1)
    return points_in_convex_polygon_3d_jit(points, corner_to_surfaces_3d(center_to_corner_box3d(rbbox, origin, axis)),
                                           device=device)


def remove_outside_points(points, rect, Trv2c, P2, image_shape, device=None):
    """keep the points inside the camera-2 viewing frustum (velodyne -> velodyne_reduced)."""
    frustum = frustum_in_lidar(rect, Trv2c, P2, image_shape)
    keep = points_in_convex_polygon_3d_jit(points, corner_to_surfaces_3d(frustum[np.newaxis, ...]), device=device)
    return points[keep.reshape([-1])]


# ---- boxes against boxes: native host code ---------------------------------------------------------------------------------

def box_collision_test(boxes, qboxes, clockwise=True):
    """[N,4,2] x [K,4,2] rotated-rectangle corners -> [N,K] bool (crossing edges or one inside the other)."""
    if not clockwise:
        raise NotImplementedError("the reference only uses clockwise corner order")
    f64 = boxes.dtype != np.float32 or qboxes.dtype != np.float32
    t = np.float64 if f64 else np.float32
    a, b = np.ascontiguousarray(boxes, dtype=t), np.ascontiguousarray(qboxes, dtype=t)
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.uint8)
    _C.check(_C.lib().sassd_box_collision_test(a.ctypes.data_as(ctypes.c_void_p), a.shape[0],
                                               b.ctypes.data_as(ctypes.c_void_p), b.shape[0], int(f64),
                                               out.ctypes.data_as(ctypes.c_void_p)), "sassd_box_collision_test")
    return out.astype(np.bool_)
