// pointops.hip -- training-side point operators of the auxiliary network:
//   three_nn / three_interpolate (+grad)   mmdet/ops/pointnet2/src/interpolate_gpu.cu:9-146 (wrappers
//                                          interpolate.cpp:13,25,40, used by necks/cmn.py:175-189)
//   pts_in_boxes3d                         mmdet/ops/points_op/src/points_op.cpp:92-144 (a serial CPU loop after a
//                                          .cpu() sync in the reference, cmn.py:48-54)
#include "common.h"

namespace {

constexpr int kNnTile = 1024;     // known points staged per LDS tile (16 KB)

// One thread per unknown point; known points stream through LDS in ascending order, so the strict '<' updates give
// exactly the reference's first-index-wins top-3 (interpolate_gpu.cu:29-50).
__global__ void __launch_bounds__(256) three_nn_kernel(int n, int m, const float *__restrict__ unknown,
                                                       const float *__restrict__ known, float *__restrict__ dist2,
                                                       int *__restrict__ idx)
{
#pragma clang fp contract(off)      // same IEEE fp32 operation sequence as the CPU oracle (squared distances bit-exact)
    __shared__ float4 tile[kNnTile];
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < n) u = ((const float4 *)unknown)[p];
    double b1 = 1e40, b2 = 1e40, b3 = 1e40;
    int i1 = 0, i2 = 0, i3 = 0;
    for (int t0 = 0; t0 < m; t0 += kNnTile) {
        const int cnt = min(kNnTile, m - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) tile[i] = ((const float4 *)known)[t0 + i];
        __syncthreads();
        if (p < n) {
            for (int k = 0; k < cnt; ++k) {
                const float4 q = tile[k];
                if (q.x != u.x) continue;
                const float d = (u.y - q.y) * (u.y - q.y) + (u.z - q.z) * (u.z - q.z) + (u.w - q.w) * (u.w - q.w);
                if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = t0 + k; }
                else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = t0 + k; }
                else if (d < b3) { b3 = d; i3 = t0 + k; }
            }
        }
    }
    if (p < n) {
        dist2[p * 3 + 0] = (float)b1; dist2[p * 3 + 1] = (float)b2; dist2[p * 3 + 2] = (float)b3;
        idx[p * 3 + 0] = i1; idx[p * 3 + 1] = i2; idx[p * 3 + 2] = i3;
    }
}

// one thread per (point, channel); consecutive threads -> consecutive channels (coalesced rows)
__global__ void three_interpolate_kernel(int c, int m, int n, const float *__restrict__ points,
                                         const int *__restrict__ idx, const float *__restrict__ weight,
                                         float *__restrict__ out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * c) return;
    const int p = t / c, ch = t - (size_t)p * c;
    const int *id = idx + p * 3;
    const float *w = weight + p * 3;
    out[t] = w[0] * points[(size_t)id[0] * c + ch] + w[1] * points[(size_t)id[1] * c + ch] +
             w[2] * points[(size_t)id[2] * c + ch];
}

__global__ void three_interpolate_grad_kernel(int c, int n, int m, const float *__restrict__ grad_out,
                                              const int *__restrict__ idx, const float *__restrict__ weight,
                                              float *__restrict__ grad_points)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * c) return;
    const int p = t / c, ch = t - (size_t)p * c;
    const int *id = idx + p * 3;
    const float *w = weight + p * 3;
    const float g = grad_out[t];
    atomicAdd(grad_points + (size_t)id[0] * c + ch, g * w[0]);
    atomicAdd(grad_points + (size_t)id[1] * c + ch, g * w[1]);
    atomicAdd(grad_points + (size_t)id[2] * c + ch, g * w[2]);
}

// points_op.cpp:92-105 -- note the literal argument order of the call site (:130-133): w = box[3], l = box[4],
// h = box[5]; double-precision halves exactly as the C++ (x / 2.0 promotes to double).
__device__ __forceinline__ int pt_in_box3d(float x, float y, float z, float cx, float cy, float bottom_z, float w,
                                           float l, float h, float angle)
{
    const float max_dis = 10.0f;
    const float cz = (float)((double)bottom_z + (double)h / 2.0);
    if ((fabsf(x - cx) > max_dis) || ((double)fabsf(z - cz) > (double)h / 2.0) || (fabsf(y - cy) > max_dis)) return 0;
    const float cosa = cosf(angle), sina = sinf(angle);
    const float x_rot = (x - cx) * cosa + (y - cy) * (-sina);
    const float y_rot = (x - cx) * sina + (y - cy) * cosa;
    return ((double)x_rot >= -(double)w / 2.0) & ((double)x_rot <= (double)w / 2.0) &
           ((double)y_rot >= -(double)l / 2.0) & ((double)y_rot <= (double)l / 2.0);
}

// one thread per point, boxes in ascending order: the LAST containing box defines the centre offset (:135-139)
__global__ void pts_in_boxes3d_kernel(const float *__restrict__ pts, int n, const float *__restrict__ boxes, int m,
                                      int32_t *__restrict__ flag, float *__restrict__ reg)
{
#pragma clang fp contract(off)
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const float x = pts[j * 3], y = pts[j * 3 + 1], z = pts[j * 3 + 2];
    for (int i = 0; i < m; ++i) {
        const float *b = boxes + i * 7;
        const int in = pt_in_box3d(x, y, z, b[0], b[1], b[2], b[3], b[4], b[5], b[6]);
        flag[(size_t)i * n + j] = in;
        if (in == 1) {
            reg[j * 3] = x - b[0];
            reg[j * 3 + 1] = y - b[1];
            reg[j * 3 + 2] = (float)((double)z - ((double)b[2] + (double)b[3] / 2.0));
        }
    }
}

}  // namespace

extern "C" int sassd_three_nn(int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx,
                              void *stream_)
{
    if (n < 0 || m < 0 || !dist2 || !idx) return SASSD_EINVAL;
    if (n == 0) return SASSD_OK;
    if (!unknown || (m > 0 && !known)) return SASSD_EINVAL;
    hipLaunchKernelGGL(three_nn_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream_, n, m, unknown, known,
                       dist2, idx);
    return sassd_launch_status();
}

extern "C" int sassd_three_interpolate(int c, int m, int n, const float *points, const int32_t *idx,
                                       const float *weight, float *out, void *stream_)
{
    if (c < 1 || m < 0 || n < 0 || !idx || !weight || !out) return SASSD_EINVAL;
    if (n == 0) return SASSD_OK;
    if (!points) return SASSD_EINVAL;
    const size_t tot = (size_t)n * c;
    hipLaunchKernelGGL(three_interpolate_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream_, c, m, n, points, idx, weight, out);
    return sassd_launch_status();
}

extern "C" int sassd_three_interpolate_grad(int c, int n, int m, const float *grad_out, const int32_t *idx,
                                            const float *weight, float *grad_points, void *stream_)
{
    if (c < 1 || m < 0 || n < 0 || !idx || !weight || !grad_points) return SASSD_EINVAL;
    if (n == 0) return SASSD_OK;
    if (!grad_out) return SASSD_EINVAL;
    const size_t tot = (size_t)n * c;
    hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream_, c, n, m, grad_out, idx, weight, grad_points);
    return sassd_launch_status();
}

extern "C" int sassd_pts_in_boxes3d(const float *pts, int n, const float *boxes3d, int m, int32_t *pts_flag,
                                    float *reg_target, void *stream_)
{
    if (n < 0 || m < 0 || !pts_flag || !reg_target) return SASSD_EINVAL;
    if (n == 0 || m == 0) return SASSD_OK;
    if (!pts || !boxes3d) return SASSD_EINVAL;
    hipLaunchKernelGGL(pts_in_boxes3d_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream_, pts, n, boxes3d,
                       m, pts_flag, reg_target);
    return sassd_launch_status();
}
