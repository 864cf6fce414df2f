// augment_core.h -- per-point arithmetic of the training-side augmentation kernels, written as
// __host__ __device__ functions so that the very same code is (a) called by the HIP kernels in augment.hip and
// (b) looped over on the CPU by the test harness oracle/harness/augment_harness.hip, which checks it against vectors
// produced by the reference's numba functions without needing a GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sassd_aug {

// points_in_convex_polygon_3d_jit (mmdet/core/bbox3d/geometry.py:189-227): a point is inside polytope j when none of
// its 6 planes gives n.p + d >= 0.  `planes` holds (nx, ny, nz, d) per surface as float64; with f32_math the values
// came from float32 arithmetic (float32 boxes) and the reference evaluates the sign in float32 too -- same operation
// order, products rounded separately (no fused multiply-add).
__host__ __device__ inline bool inside_polytope(float x, float y, float z, const double *planes, bool f32_math)
{
#pragma clang fp contract(off)
    for (int k = 0; k < 6; ++k) {
        const double *p = planes + 4 * k;
        if (f32_math) {
            const float s = x * (float)p[0] + y * (float)p[1] + z * (float)p[2] + (float)p[3];
            if (s >= 0.f) return false;
        } else {
            const double s = (double)x * p[0] + (double)y * p[1] + (double)z * p[2] + p[3];
            if (s >= 0.0) return false;
        }
    }
    return true;
}

// points_transform_ (mmdet/core/point_cloud/point_augmentor.py:44-62): the FIRST valid box whose mask holds the point
// moves it: rotate about the box centre by the box's accepted yaw noise (float32 rotation matrix built from the
// float64 sine / cosine), then translate by the accepted centre noise (float64 add, rounded back to float32).
__host__ __device__ inline void transform_point(float *p, const uint8_t *mask_row, int m, const uint8_t *valid,
                                                const float *centers, const float *rot_sin, const float *rot_cos,
                                                const double *loc)
{
#pragma clang fp contract(off)
    for (int j = 0; j < m; ++j) {
        if (!valid[j] || !mask_row[j]) continue;
        const float cx = centers[3 * j], cy = centers[3 * j + 1], cz = centers[3 * j + 2];
        const float x = p[0] - cx, y = p[1] - cy, z = p[2] - cz;
        const float s = rot_sin[j], c = rot_cos[j];
        const float xr = x * c + y * s, yr = x * (-s) + y * c;          // [x y z] @ [[c,-s,0],[s,c,0],[0,0,1]]
        p[0] = (float)((double)(xr + cx) + loc[3 * j]);
        p[1] = (float)((double)(yr + cy) + loc[3 * j + 1]);
        p[2] = (float)((double)(z + cz) + loc[3 * j + 2]);
        return;
    }
}

// random_flip + global_rotation + global_scaling of the point cloud in one pass (point_augmentor.py:279-303):
// y -> -y when flipped; [x y z] @ [[c,-s,0],[s,c,0],[0,0,1]] with a float32 matrix; xyz *= scale (float32 array times a
// float64 scalar keeps float32 in NumPy: the scalar is rounded to float32 first).
__host__ __device__ inline void global_point(float *p, int flip, float s, float c, float scale)
{
#pragma clang fp contract(off)
    const float x = p[0], y = flip ? -p[1] : p[1];
    p[0] = (x * c + y * s) * scale;
    p[1] = (x * (-s) + y * c) * scale;
    p[2] = p[2] * scale;
}

// Output row r of the pasted ground-truth objects (PointAugmentor.sample_all, point_augmentor.py:232-242): object k owns
// rows [out_start[k], out_start[k+1]) and copies its database points shifted by the object's box centre (and the
// road-plane height correction) -- a float64 add rounded to float32, as `s_points[:, :3] += box3d_lidar[:3]` does.
__host__ __device__ inline int object_of_row(const int64_t *out_start, int n_obj, int64_t r)
{
    int lo = 0, hi = n_obj;                     // largest k with out_start[k] <= r
    while (hi - lo > 1) {
        const int mid = (lo + hi) / 2;
        if (out_start[mid] <= r) lo = mid; else hi = mid;
    }
    return lo;
}

__host__ __device__ inline void paste_point(const float *src, float *dst, const double *shift, const double *lower)
{
    dst[0] = (float)((double)src[0] + shift[0]);
    dst[1] = (float)((double)src[1] + shift[1]);
    float z = (float)((double)src[2] + shift[2]);
    if (lower) z = (float)((double)z - *lower);          // `s_points[:, 2] -= mv_height[i]`: a second rounding
    dst[2] = z;
    dst[3] = src[3];
}

}  // namespace sassd_aug
