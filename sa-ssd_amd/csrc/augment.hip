// augment.hip -- device half of the training-side augmentation and of the offline data preparation.
// The reference does all of it in DataLoader workers with numba CPU loops (mmdet/core/point_cloud/point_augmentor.py,
// mmdet/core/bbox3d/geometry.py, tools/create_data.py); here the raw sweep is uploaded once and every O(points) or
// O(points x boxes) step runs on the GPU, one thread per point, next to the voxelizer that consumes the result:
//   points_in_polytopes   points_in_rbbox / remove_outside_points    geometry.py:63-74,50-61,189-227
//   points_transform      points_transform_                          point_augmentor.py:44-62
//   global_transform      random_flip + global_rotation + global_scaling (points)   point_augmentor.py:279-303
//   paste_objects         the sampled ground-truth objects' points   point_augmentor.py:232-242
// The O(boxes^2 x tries) sequential choices (collision tests, noise selection) are host code in augment_host.hip.
// All four kernels are HBM streams of 16 B per point (+ m mask bytes); the arithmetic lives in augment_core.h.
#include "augment_core.h"
#include "common.h"

namespace {
using namespace sassd_aug;

constexpr int kMaxPlanesLds = 256;            // polytopes staged in LDS per pass (256 x 6 x 4 doubles = 48 KB)

__global__ void __launch_bounds__(256) points_in_polytopes_kernel(const float *__restrict__ pts, int n, int stride,
                                                                  const double *__restrict__ planes, int m,
                                                                  int f32_math, uint8_t *__restrict__ mask)
{
    __shared__ double lds[kMaxPlanesLds * 24];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float x = 0.f, y = 0.f, z = 0.f;
    if (i < n) { x = pts[(size_t)i * stride]; y = pts[(size_t)i * stride + 1]; z = pts[(size_t)i * stride + 2]; }
    for (int j0 = 0; j0 < m; j0 += kMaxPlanesLds) {
        const int cnt = min(kMaxPlanesLds, m - j0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt * 24; t += blockDim.x) lds[t] = planes[(size_t)j0 * 24 + t];
        __syncthreads();
        if (i < n)
            for (int j = 0; j < cnt; ++j)
                mask[(size_t)i * m + j0 + j] = inside_polytope(x, y, z, lds + j * 24, f32_math != 0) ? 1 : 0;
    }
}

__global__ void __launch_bounds__(256) points_transform_kernel(float *__restrict__ pts, int n, int stride,
                                                               const uint8_t *__restrict__ mask, int m,
                                                               const uint8_t *__restrict__ valid,
                                                               const float *__restrict__ centers,
                                                               const float *__restrict__ rot_sin,
                                                               const float *__restrict__ rot_cos,
                                                               const double *__restrict__ loc)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    transform_point(pts + (size_t)i * stride, mask + (size_t)i * m, m, valid, centers, rot_sin, rot_cos, loc);
}

__global__ void __launch_bounds__(256) global_transform_kernel(float *__restrict__ pts, int n, int stride, int flip,
                                                               float s, float c, float scale)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    global_point(pts + (size_t)i * stride, flip, s, c, scale);
}

__global__ void __launch_bounds__(256) paste_objects_kernel(const float *__restrict__ db_points,
                                                            const int64_t *__restrict__ src_start,
                                                            const int64_t *__restrict__ out_start, int n_obj,
                                                            int64_t n_out, const double *__restrict__ shift,
                                                            const double *__restrict__ lower, float *__restrict__ out)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_out) return;
    const int k = object_of_row(out_start, n_obj, r);
    const float *src = db_points + (src_start[k] + (r - out_start[k])) * 4;
    paste_point(src, out + r * 4, shift + 3 * k, lower ? lower + k : nullptr);
}
}  // namespace

extern "C" int sassd_points_in_polytopes(const float *points, int n, int stride, const double *planes, int m,
                                         int f32_math, uint8_t *mask, void *stream_)
{
    if (n < 0 || m < 0 || stride < 3) return SASSD_EINVAL;
    if (n == 0 || m == 0) return SASSD_OK;
    if (!points || !planes || !mask) return SASSD_EINVAL;
    hipLaunchKernelGGL(points_in_polytopes_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream_, points, n,
                       stride, planes, m, f32_math, mask);
    return sassd_launch_status();
}

extern "C" int sassd_points_transform(float *points, int n, int stride, const uint8_t *mask, int m,
                                      const uint8_t *valid, const float *centers, const float *rot_sin,
                                      const float *rot_cos, const double *loc, void *stream_)
{
    if (n < 0 || m < 0 || stride < 3) return SASSD_EINVAL;
    if (n == 0 || m == 0) return SASSD_OK;
    if (!points || !mask || !valid || !centers || !rot_sin || !rot_cos || !loc) return SASSD_EINVAL;
    hipLaunchKernelGGL(points_transform_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream_, points, n,
                       stride, mask, m, valid, centers, rot_sin, rot_cos, loc);
    return sassd_launch_status();
}

extern "C" int sassd_points_global_transform(float *points, int n, int stride, int flip, float rot_sin, float rot_cos,
                                             float scale, void *stream_)
{
    if (n < 0 || stride < 3) return SASSD_EINVAL;
    if (n == 0) return SASSD_OK;
    if (!points) return SASSD_EINVAL;
    hipLaunchKernelGGL(global_transform_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream_, points, n,
                       stride, flip, rot_sin, rot_cos, scale);
    return sassd_launch_status();
}

extern "C" int sassd_paste_objects(const float *db_points, const int64_t *src_start, const int64_t *out_start,
                                   int n_obj, int64_t n_out, const double *shift, const double *lower, float *out,
                                   void *stream_)
{
    if (n_obj < 0 || n_out < 0) return SASSD_EINVAL;
    if (n_obj == 0 || n_out == 0) return SASSD_OK;
    if (!db_points || !src_start || !out_start || !shift || !out) return SASSD_EINVAL;
    hipLaunchKernelGGL(paste_objects_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                       db_points, src_start, out_start, n_obj, n_out, shift, lower, out);
    return sassd_launch_status();
}
