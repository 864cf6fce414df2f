// TEST INFRASTRUCTURE (never linked into libsassd.so, never imported by the product path).
// Loops the __host__ __device__ per-point functions of sa-ssd_amd/csrc/augment_core.h -- the exact code the HIP kernels
// of augment.hip call per thread -- over arrays on the CPU, so the arithmetic of the device path can be checked against
// the reference-generated vectors (tests/golden/augment_ref.npz) in the GPU-less test suite.  Built host-only by
// oracle/build.py::build_harness (hipcc --cuda-host-only).
#include "../../sa-ssd_amd/csrc/augment_core.h"

using namespace sassd_aug;

extern "C" void hst_points_in_polytopes(const float *pts, int n, int stride, const double *planes, int m, int f32_math,
                                        uint8_t *mask)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j)
            mask[(size_t)i * m + j] = inside_polytope(pts[(size_t)i * stride], pts[(size_t)i * stride + 1],
                                                      pts[(size_t)i * stride + 2], planes + (size_t)j * 24,
                                                      f32_math != 0) ? 1 : 0;
}

extern "C" void hst_points_transform(float *pts, int n, int stride, const uint8_t *mask, int m, const uint8_t *valid,
                                     const float *centers, const float *rot_sin, const float *rot_cos, const double *loc)
{
    for (int i = 0; i < n; ++i)
        transform_point(pts + (size_t)i * stride, mask + (size_t)i * m, m, valid, centers, rot_sin, rot_cos, loc);
}

extern "C" void hst_global_transform(float *pts, int n, int stride, int flip, float s, float c, float scale)
{
    for (int i = 0; i < n; ++i) global_point(pts + (size_t)i * stride, flip, s, c, scale);
}

extern "C" void hst_paste_objects(const float *db_points, const int64_t *src_start, const int64_t *out_start, int n_obj,
                                  int64_t n_out, const double *shift, const double *lower, float *out)
{
    for (int64_t r = 0; r < n_out; ++r) {
        const int k = object_of_row(out_start, n_obj, r);
        paste_point(db_points + (src_start[k] + (r - out_start[k])) * 4, out + r * 4, shift + 3 * k,
                    lower ? lower + k : nullptr);
    }
}
