"""Mirror of mmdet/core/point_cloud/voxel_generator.py:4-41 (the boundary object the dataset holds)."""
import numpy as np

from .points_ops import points_to_voxel


class VoxelGenerator:
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        point_cloud_range = np.array(point_cloud_range, dtype=np.float32)
        voxel_size = np.array(voxel_size, dtype=np.float32)
        grid_size = (point_cloud_range[3:] - point_cloud_range[:3]) / voxel_size
        self._grid_size = np.round(grid_size).astype(np.int64)
        self._voxel_size = voxel_size
        self._point_cloud_range = point_cloud_range
        self._max_num_points = max_num_points
        self._max_voxels = max_voxels

    def generate(self, points):
        return points_to_voxel(points, self._voxel_size, self._point_cloud_range, self._max_num_points, True,
                               self._max_voxels)

    voxel_size = property(lambda self: self._voxel_size)
    max_num_points_per_voxel = property(lambda self: self._max_num_points)
    point_cloud_range = property(lambda self: self._point_cloud_range)
    grid_size = property(lambda self: self._grid_size)
