"""Synthetic KITTI-range / Waymo-scale LiDAR frames (measurement contract).

These generators define the bench/parity inputs; RNG draw order matters for the quoted voxel counts
(seed 0: K21 -> 16111 voxels, K17 -> 13435, waymo_synth -> 79302).  Pure numpy, no reference code.
"""
import numpy as np

KITTI_RANGE = (0.0, -40.0, -3.0, 70.4, 40.0, 1.0)     # configs/car_cfg.py generator.point_cloud_range
KITTI_VOXEL = (0.05, 0.05, 0.1)
WAYMO_RANGE = (-75.2, -75.2, -2.0, 75.2, 75.2, 4.0)
WAYMO_VOXEL = (0.1, 0.1, 0.15)


def _ray_cloud(seed, elev_deg, az_rad, r_ob, r_max, crop):
    rng = np.random.default_rng(seed)
    e = np.deg2rad(np.linspace(elev_deg[0], elev_deg[1], 64))
    E, A = np.meshgrid(e, az_rad, indexing="ij")
    E = E.ravel()
    A = A.ravel()
    n = E.size
    rg = np.where(E < 0, 1.73 / np.maximum(np.sin(-E), 1e-9), np.inf)   # ground plane z = -1.73
    u = rng.random(n)
    rob = rng.uniform(r_ob[0], r_ob[1], n)
    r = np.where(E < 0, np.where((u < 0.25) & (rob < rg), rob, rg), np.where(u < 0.35, rob, np.inf))
    r = r + rng.normal(0.0, 0.02, n)
    keep = np.isfinite(r) & (r < r_max)
    r, E, A = r[keep], E[keep], A[keep]
    x = r * np.cos(E) * np.cos(A)
    y = r * np.cos(E) * np.sin(A)
    z = r * np.sin(E)
    inten = rng.random(r.size)
    pts = np.stack([x, y, z, inten], 1).astype(np.float32)
    lo = np.asarray(crop[:3], np.float32)
    hi = np.asarray(crop[3:], np.float32)
    m = np.all((pts[:, :3] >= lo) & (pts[:, :3] < hi), axis=1)
    pts = pts[m]
    rng.shuffle(pts)
    return np.ascontiguousarray(pts)


def lidar64(seed=0, n_az=469, fov=40.5, elev=(2.0, -24.8), r_ob=(5.0, 70.0), r_max=80.0,
            crop=KITTI_RANGE):
    """KITTI-like 64-beam frontal cloud. seed 0 -> 27124 rows; K21 = [:21500], K17 = [:17000]."""
    a = np.deg2rad(np.linspace(-fov, fov, n_az))
    return _ray_cloud(seed, elev, a, r_ob, r_max, crop)


def waymo_synth(seed=0):
    """360-degree 64-beam cloud, 3300 azimuths. seed 0 -> 184569 rows; use [:180000]."""
    a = np.deg2rad(np.linspace(-180.0, 180.0, 3300, endpoint=False))
    crop = (-75.2, -75.2, -2.0, 75.2, 75.2, 4.0)
    return _ray_cloud(seed, (2.4, -17.6), a, (5.0, 75.0), 110.0, crop)


def k21(seed=0):
    return lidar64(seed)[:21500]


def k17(seed=0):
    return lidar64(seed)[:17000]
