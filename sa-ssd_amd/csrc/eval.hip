// eval.hip -- KITTI-evaluation rotated IoU: replaces the numba.cuda kernel
// `rotate_iou_gpu_eval` of mmdet/core/post_processing/rotate_nms_gpu.py:536-627 (device functions :153-388), the O(N*K)
// part of mmdet/core/evaluation/kitti_eval.py.  Boxes are (cx, cy, x_dim, y_dim, angle).  The intersection polygon is
// collected from the corners lying inside the other box plus the 16 edge-edge crossings, ordered by angle around the
// centroid and summed as a triangle fan -- the same procedure as the reference (including its behaviour on exactly
// coincident boxes, where duplicated vertices make the fan under-count), so results agree to fp32 rounding.
// One thread per (box, query) pair; everything lives in registers / a 24-float private array.
#include "common.h"

namespace {
struct Quad { float c[8]; };

__device__ __forceinline__ Quad eval_corners(const float *rb)
{
    Quad q;
    const double a_cos = cos((double)rb[4]), a_sin = sin((double)rb[4]);
    const float hx = (float)(rb[2] / 2.0), hy = (float)(rb[3] / 2.0);
    const float cx[4] = {-hx, -hx, hx, hx}, cy[4] = {-hy, hy, hy, -hy};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        q.c[2 * i] = (float)(a_cos * cx[i] + a_sin * cy[i] + rb[0]);
        q.c[2 * i + 1] = (float)(-a_sin * cx[i] + a_cos * cy[i] + rb[1]);
    }
    return q;
}

__device__ __forceinline__ bool eval_pt_in_quad(float px, float py, const float *c)
{
#pragma clang fp contract(off)
    const float ab0 = c[2] - c[0], ab1 = c[3] - c[1], ad0 = c[6] - c[0], ad1 = c[7] - c[1];
    const float ap0 = px - c[0], ap1 = py - c[1];
    const float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
    const float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
    return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}

__device__ __forceinline__ bool eval_seg_isect(const float *p1, const float *p2, int i, int j, float *out)
{
#pragma clang fp contract(off)
    const float A0 = p1[2 * i], A1 = p1[2 * i + 1], B0 = p1[2 * ((i + 1) & 3)], B1 = p1[2 * ((i + 1) & 3) + 1];
    const float C0 = p2[2 * j], C1 = p2[2 * j + 1], D0 = p2[2 * ((j + 1) & 3)], D1 = p2[2 * ((j + 1) & 3) + 1];
    const float BA0 = B0 - A0, BA1 = B1 - A1, DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
    const bool acd = DA1 * CA0 > CA1 * DA0;
    const bool bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
    if (acd == bcd) return false;
    const bool abc = CA1 * BA0 > BA1 * CA0, abd = DA1 * BA0 > BA1 * DA0;
    if (abc == abd) return false;
    const float DC0 = D0 - C0, DC1 = D1 - C1;
    const float ABBA = A0 * B1 - B0 * A1, CDDC = C0 * D1 - D0 * C1;
    const float DH = BA1 * DC0 - BA0 * DC1, Dx = ABBA * DC0 - BA0 * CDDC, Dy = ABBA * DC1 - BA1 * CDDC;
    out[0] = Dx / DH;
    out[1] = Dy / DH;
    return true;
}

__device__ double eval_inter(const float *rb1, const float *rb2)
{
#pragma clang fp contract(off)
    const Quad q1 = eval_corners(rb1), q2 = eval_corners(rb2);
    float ip[48];
    int n = 0;
    for (int i = 0; i < 4; ++i) {
        if (eval_pt_in_quad(q1.c[2 * i], q1.c[2 * i + 1], q2.c)) { ip[2 * n] = q1.c[2 * i]; ip[2 * n + 1] = q1.c[2 * i + 1]; ++n; }
        if (eval_pt_in_quad(q2.c[2 * i], q2.c[2 * i + 1], q1.c)) { ip[2 * n] = q2.c[2 * i]; ip[2 * n + 1] = q2.c[2 * i + 1]; ++n; }
    }
    float tp[2];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (eval_seg_isect(q1.c, q2.c, i, j, tp)) { ip[2 * n] = tp[0]; ip[2 * n + 1] = tp[1]; ++n; }
    if (n > 0) {
        float cen0 = 0.f, cen1 = 0.f, vs[24];
        for (int i = 0; i < n; ++i) { cen0 += ip[2 * i]; cen1 += ip[2 * i + 1]; }
        cen0 /= n; cen1 /= n;
        for (int i = 0; i < n; ++i) {
            float v0 = ip[2 * i] - cen0, v1 = ip[2 * i + 1] - cen1;
            const double d = sqrt((double)(v0 * v0 + v1 * v1));
            v0 = (float)(v0 / d); v1 = (float)(v1 / d);
            if (v1 < 0) v0 = -2 - v0;
            vs[i] = v0;
        }
        for (int i = 1; i < n; ++i) {
            if (vs[i - 1] > vs[i]) {
                const float temp = vs[i], tx = ip[2 * i], ty = ip[2 * i + 1];
                int j = i;
                while (j > 0 && vs[j - 1] > temp) {
                    vs[j] = vs[j - 1]; ip[2 * j] = ip[2 * j - 2]; ip[2 * j + 1] = ip[2 * j - 1]; --j;
                }
                vs[j] = temp; ip[2 * j] = tx; ip[2 * j + 1] = ty;
            }
        }
    }
    double area = 0.0;
    for (int i = 0; i < n - 2; ++i) {
        const float *a = ip, *b = ip + 2 * i + 2, *c = ip + 2 * i + 4;
        area += fabs(((a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0])) / 2.0);
    }
    return area;
}

__global__ void rotate_iou_eval_kernel(const float *__restrict__ boxes, int n, const float *__restrict__ qboxes, int k,
                                       int criterion, float *__restrict__ iou)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * k) return;
    const int i = t / k, j = t - (size_t)i * k;
    float r1[5], r2[5];
#pragma unroll
    for (int e = 0; e < 5; ++e) { r1[e] = qboxes[5 * j + e]; r2[e] = boxes[5 * i + e]; }
    const float a1 = r1[2] * r1[3], a2 = r2[2] * r2[3];
    const double in = eval_inter(r1, r2);
    double v;
    if (criterion == -1) v = in / (a1 + a2 - in);
    else if (criterion == 0) v = in / a1;
    else if (criterion == 1) v = in / a2;
    else v = in;
    iou[t] = (float)v;
}
}  // namespace

extern "C" int sassd_rotate_iou_eval(const float *boxes, int n, const float *query_boxes, int k, int criterion,
                                     float *iou, void *stream_)
{
    if (n < 0 || k < 0 || !iou) return SASSD_EINVAL;
    if (n == 0 || k == 0) return SASSD_OK;
    if (!boxes || !query_boxes) return SASSD_EINVAL;
    const size_t tot = (size_t)n * k;
    hipLaunchKernelGGL(rotate_iou_eval_kernel, dim3((unsigned)((tot + 127) / 128)), dim3(128), 0, (hipStream_t)stream_,
                       boxes, n, query_boxes, k, criterion, iou);
    return sassd_launch_status();
}
