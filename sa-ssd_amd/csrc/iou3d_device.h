// iou3d_device.h -- rotated BEV rectangle overlap / IoU, device functions shared by the NMS and IoU kernels.
//
// Same geometry as mmdet/ops/iou3d/src/iou3d_kernel.cu:34-221 (rotate 4 corners about the centre, 16 edge-edge
// intersections with EPS 1e-8, corner-in-box tests with MARGIN 1e-5, centroid, angular sort, shoelace fan), written
// for CDNA4: corners kept in registers as SoA, the angular sort is a fixed 16-slot insertion sort on precomputed
// atan2 keys (the reference bubble-sorts and recomputes atan2f 2x per comparison), FP contraction is disabled so the
// arithmetic is the same sequence of IEEE fp32 operations as the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace iou3d {

constexpr float kEps = 1e-8f;
constexpr float kMargin = 1e-5f;

struct P2 { float x, y; };

__device__ __forceinline__ float cross3(P2 p1, P2 p2, P2 p0)
{
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

__device__ __forceinline__ bool rect_cross(P2 p1, P2 p2, P2 q1, P2 q2)
{
    return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
           fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

__device__ __forceinline__ bool seg_x(P2 p1, P2 p0, P2 q1, P2 q0, P2 &ans)
{
    if (!rect_cross(p0, p1, q0, q1)) return false;
    const float s1 = cross3(q0, p1, p0);
    const float s2 = cross3(p1, q1, p0);
    const float s3 = cross3(p0, q1, q0);
    const float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0.f && s3 * s4 > 0.f)) return false;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > kEps) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

struct Box {              // (x1,y1,x2,y2,ry) + derived
    float x1, y1, x2, y2, r;
};

__device__ __forceinline__ Box load_box(const float *p)
{
    Box b;
    b.x1 = p[0]; b.y1 = p[1]; b.x2 = p[2]; b.y2 = p[3]; b.r = p[4];
    return b;
}

__device__ __forceinline__ bool in_box(const Box &b, P2 p)
{
    const float cx = (b.x1 + b.x2) / 2, cy = (b.y1 + b.y2) / 2;
    const float ac = cosf(-b.r), as = sinf(-b.r);
    const float rx = (p.x - cx) * ac + (p.y - cy) * as + cx;
    const float ry = -(p.x - cx) * as + (p.y - cy) * ac + cy;
    return rx > b.x1 - kMargin && rx < b.x2 + kMargin && ry > b.y1 - kMargin && ry < b.y2 + kMargin;
}

__device__ __forceinline__ void corners(const Box &b, P2 (&c)[5])
{
    const float cx = (b.x1 + b.x2) / 2, cy = (b.y1 + b.y2) / 2;
    const float ac = cosf(b.r), as = sinf(b.r);
    const float xs[4] = {b.x1, b.x2, b.x2, b.x1};
    const float ys[4] = {b.y1, b.y1, b.y2, b.y2};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c[k].x = (xs[k] - cx) * ac + (ys[k] - cy) * as + cx;
        c[k].y = -(xs[k] - cx) * as + (ys[k] - cy) * ac + cy;
    }
    c[4] = c[0];
}

__device__ inline float box_overlap(const Box &a, const Box &b)
{
    P2 A[5], B[5];
    corners(a, A);
    corners(b, B);
    P2 cp[16];
    float key[16];
    float cxs = 0.f, cys = 0.f;
    int cnt = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            P2 t;
            if (seg_x(A[i + 1], A[i], B[j + 1], B[j], t)) { cxs = cxs + t.x; cys = cys + t.y; cp[cnt++] = t; }
        }
    for (int k = 0; k < 4; ++k) {
        if (in_box(a, B[k])) { cxs = cxs + B[k].x; cys = cys + B[k].y; cp[cnt++] = B[k]; }
        if (in_box(b, A[k])) { cxs = cxs + A[k].x; cys = cys + A[k].y; cp[cnt++] = A[k]; }
    }
    if (cnt < 3) return 0.f;          // the reference's fan sum is empty/zero for < 3 points (and 0/0 centre unused)
    cxs /= cnt; cys /= cnt;
    for (int i = 0; i < cnt; ++i) key[i] = atan2f(cp[i].y - cys, cp[i].x - cxs);
    // stable insertion sort ascending by angle == the reference's bubble sort (strict '>' swaps, stable)
    for (int i = 1; i < cnt; ++i) {
        const P2 t = cp[i];
        const float kt = key[i];
        int j = i - 1;
        while (j >= 0 && key[j] > kt) { cp[j + 1] = cp[j]; key[j + 1] = key[j]; --j; }
        cp[j + 1] = t; key[j + 1] = kt;
    }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; ++k) {
        const float ux = cp[k].x - cp[0].x, uy = cp[k].y - cp[0].y;
        const float vx = cp[k + 1].x - cp[0].x, vy = cp[k + 1].y - cp[0].y;
        area += ux * vy - uy * vx;
    }
    return fabsf(area) / 2.0f;
}

__device__ __forceinline__ float iou_bev(const Box &a, const Box &b)
{
    const float sa = (a.x2 - a.x1) * (a.y2 - a.y1);
    const float sb = (b.x2 - b.x1) * (b.y2 - b.y1);
    const float s = box_overlap(a, b);
    return s / fmaxf(sa + sb - s, kEps);
}

}  // namespace iou3d
