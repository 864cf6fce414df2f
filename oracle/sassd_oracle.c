/*
 * oracle/sassd_oracle.c -- CPU restatement of the SA-SSD hot path's integer / geometry kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sa-ssd_amd/ may link, import or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker / timed CPU port.
 *
 * Each function cites the reference lines (relative to /root/reference) whose behaviour it restates.
 * Parity pinning: voxelizer checked against the reference's own points_ops.py run under an identity-jit
 * numba stub (tests/golden/make_golden.py); rotated IoU checked against oracle/_ref (the reference's
 * iou3d_kernel.cu device functions compiled for the host, see oracle/build.py).
 *
 * Build: gcc -O3 -fno-fast-math -ffp-contract=off -shared -fPIC (see oracle/build.py)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * Voxelizer -- mmdet/ops/points_op/points_ops.py:5-50 (_points_to_voxel_reverse_kernel) and :104-164.
 * Serial first-touch order, <= max_points points per voxel kept in arrival order, `break` (not continue)
 * when a NEW voxel would exceed max_voxels (:41-42).  f32 subtract + true f32 divide + floor (:31).
 * grid_size = round((hi-lo)/vs) computed in f32 (:24).  coords written reversed (z,y,x) (:35).
 * Like the reference it allocates and fills the dense coor_to_voxelidx grid on every call (:145).
 * Returns voxel_num.  voxels [max_voxels,max_points,ndim] and num_points must be zeroed by the caller
 * exactly as points_ops.py:143-148 does (we do it here).
 * ---------------------------------------------------------------------------------------------- */
int orc_points_to_voxel(const float *points, int n, int ndim, const float *voxel_size /*3*/,
                        const float *coors_range /*6*/, int max_points, int max_voxels,
                        float *voxels, int32_t *coors /*[max_voxels,3] zyx*/, int32_t *num_points)
{
    int32_t grid[3];
    for (int j = 0; j < 3; ++j) {
        float g = (coors_range[3 + j] - coors_range[j]) / voxel_size[j];
        grid[j] = (int32_t)rintf(g);               /* np.round: half-to-even, same as rintf */
    }
    const size_t vol = (size_t)grid[0] * grid[1] * grid[2];
    int32_t *map = (int32_t *)malloc(vol * sizeof(int32_t));
    if (!map) return -1;
    memset(map, 0xFF, vol * sizeof(int32_t));       /* -1 everywhere */
    memset(voxels, 0, (size_t)max_voxels * max_points * ndim * sizeof(float));
    memset(coors, 0, (size_t)max_voxels * 3 * sizeof(int32_t));
    memset(num_points, 0, (size_t)max_voxels * sizeof(int32_t));
    int voxel_num = 0;
    for (int i = 0; i < n; ++i) {
        int32_t c[3];
        int failed = 0;
        for (int j = 0; j < 3; ++j) {
            volatile float d = points[(size_t)i * ndim + j] - coors_range[j];
            volatile float q = d / voxel_size[j];
            float f = floorf(q);
            if (f < 0 || f >= (float)grid[j]) { failed = 1; break; }
            c[2 - j] = (int32_t)f;
        }
        if (failed) continue;
        /* map is indexed [z][y][x] with shape grid reversed (:139-141) */
        size_t lin = ((size_t)c[0] * grid[1] + c[1]) * grid[0] + c[2];
        int32_t v = map[lin];
        if (v == -1) {
            v = voxel_num;
            if (voxel_num >= max_voxels) break;
            voxel_num++;
            map[lin] = v;
            coors[v * 3 + 0] = c[0]; coors[v * 3 + 1] = c[1]; coors[v * 3 + 2] = c[2];
        }
        int32_t num = num_points[v];
        if (num < max_points) {
            memcpy(voxels + ((size_t)v * max_points + num) * ndim, points + (size_t)i * ndim,
                   ndim * sizeof(float));
            num_points[v] = num + 1;
        }
    }
    free(map);
    return voxel_num;
}

/* SimpleVoxel.forward -- mmdet/models/backbones/vxnet.py:110-116: sum over the (zero padded) point slots
 * divided by num_points.  Sum order: slot 0..T-1 sequential fp32. */
void orc_voxel_mean(const float *voxels, const int32_t *num_points, int m, int max_points, int ndim,
                    int nfeat, float *mean /*[m,nfeat]*/)
{
    for (int v = 0; v < m; ++v)
        for (int f = 0; f < nfeat; ++f) {
            float s = 0.f;
            for (int t = 0; t < max_points; ++t) s += voxels[((size_t)v * max_points + t) * ndim + f];
            mean[(size_t)v * nfeat + f] = s / (float)num_points[v];
        }
}

/* ------------------------------------------------------------------------------------------------
 * Rotated BEV overlap / IoU -- mmdet/ops/iou3d/src/iou3d_kernel.cu:14-221.
 * Same arithmetic, same order of operations, fp32 throughout (cosf/sinf/atan2f of libm instead of the
 * CUDA device math library).  Boxes are (x1,y1,x2,y2,ry).
 * ---------------------------------------------------------------------------------------------- */
#define ORC_EPS 1e-8f
typedef struct { float x, y; } pt_t;

static inline float cross3(pt_t p1, pt_t p2, pt_t p0)   /* iou3d_kernel.cu:38-40 */
{ return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }

static inline float cross2(pt_t a, pt_t b) { return a.x * b.y - a.y * b.x; }   /* :34-36 */

static inline int rect_cross(pt_t p1, pt_t p2, pt_t q1, pt_t q2)               /* :42-48 */
{
    return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
           fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

static inline int in_box2d(const float *box, pt_t p)                           /* :50-65 */
{
    const float MARGIN = 1e-5f;
    float cx = (box[0] + box[2]) / 2, cy = (box[1] + box[3]) / 2;
    float ac = cosf(-box[4]), as = sinf(-box[4]);
    float rx = (p.x - cx) * ac + (p.y - cy) * as + cx;
    float ry = -(p.x - cx) * as + (p.y - cy) * ac + cy;
    return (rx > box[0] - MARGIN && rx < box[2] + MARGIN && ry > box[1] - MARGIN && ry < box[3] + MARGIN);
}

static inline int seg_intersection(pt_t p1, pt_t p0, pt_t q1, pt_t q0, pt_t *ans)   /* :67-96 */
{
    if (!rect_cross(p0, p1, q0, q1)) return 0;
    float s1 = cross3(q0, p1, p0);
    float s2 = cross3(p1, q1, p0);
    float s3 = cross3(p0, q1, q0);
    float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > ORC_EPS) {
        ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        float D = a0 * b1 - a1 * b0;
        ans->x = (b0 * c1 - b1 * c0) / D;
        ans->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

static inline void rot_center(pt_t c, float ac, float as, pt_t *p)              /* :98-102 */
{
    float nx = (p->x - c.x) * ac + (p->y - c.y) * as + c.x;
    float ny = -(p->x - c.x) * as + (p->y - c.y) * ac + c.y;
    p->x = nx; p->y = ny;
}

float orc_box_overlap(const float *a, const float *b)                           /* :108-212 */
{
    pt_t ca = { (a[0] + a[2]) / 2, (a[1] + a[3]) / 2 };
    pt_t cb = { (b[0] + b[2]) / 2, (b[1] + b[3]) / 2 };
    pt_t A[5] = { {a[0], a[1]}, {a[2], a[1]}, {a[2], a[3]}, {a[0], a[3]} };
    pt_t B[5] = { {b[0], b[1]}, {b[2], b[1]}, {b[2], b[3]}, {b[0], b[3]} };
    float aca = cosf(a[4]), asa = sinf(a[4]), acb = cosf(b[4]), asb = sinf(b[4]);
    for (int k = 0; k < 4; ++k) { rot_center(ca, aca, asa, &A[k]); rot_center(cb, acb, asb, &B[k]); }
    A[4] = A[0]; B[4] = B[0];
    pt_t cp[16], pc = {0.f, 0.f};
    int cnt = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (seg_intersection(A[i + 1], A[i], B[j + 1], B[j], &cp[cnt])) {
                pc.x += cp[cnt].x; pc.y += cp[cnt].y; cnt++;
            }
    for (int k = 0; k < 4; ++k) {
        if (in_box2d(a, B[k])) { pc.x += B[k].x; pc.y += B[k].y; cp[cnt++] = B[k]; }
        if (in_box2d(b, A[k])) { pc.x += A[k].x; pc.y += A[k].y; cp[cnt++] = A[k]; }
    }
    pc.x /= cnt; pc.y /= cnt;
    for (int j = 0; j < cnt - 1; ++j)                          /* bubble sort by atan2 (:188-196) */
        for (int i = 0; i < cnt - j - 1; ++i)
            if (atan2f(cp[i].y - pc.y, cp[i].x - pc.x) > atan2f(cp[i + 1].y - pc.y, cp[i + 1].x - pc.x)) {
                pt_t t = cp[i]; cp[i] = cp[i + 1]; cp[i + 1] = t;
            }
    float area = 0;
    for (int k = 0; k < cnt - 1; ++k) {
        pt_t u = { cp[k].x - cp[0].x, cp[k].y - cp[0].y };
        pt_t v = { cp[k + 1].x - cp[0].x, cp[k + 1].y - cp[0].y };
        area += cross2(u, v);
    }
    return fabsf(area) / 2.0f;
}

float orc_iou_bev(const float *a, const float *b)                               /* :214-221 */
{
    float sa = (a[2] - a[0]) * (a[3] - a[1]);
    float sb = (b[2] - b[0]) * (b[3] - b[1]);
    float s = orc_box_overlap(a, b);
    return s / fmaxf(sa + sb - s, ORC_EPS);
}

void orc_boxes_overlap_bev(const float *a, int na, const float *b, int nb, float *out)   /* :223-234 */
{
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) out[(size_t)i * nb + j] = orc_box_overlap(a + i * 5, b + j * 5);
}

void orc_boxes_iou_bev(const float *a, int na, const float *b, int nb, float *out)       /* :236-248 */
{
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) out[(size_t)i * nb + j] = orc_iou_bev(a + i * 5, b + j * 5);
}

/* nms_kernel (:250-292) + host greedy reduce (iou3d.cpp:84-119).  boxes must already be sorted by
 * descending score (iou3d_utils.py:121-123).  Writes kept indices (into the sorted array), returns count.
 * mask_out (optional, [n, ceil(n/64)] u64) receives the suppression bitmask exactly as the kernel builds it. */
int orc_nms_rotated(const float *boxes, int n, float thr, int64_t *keep, uint64_t *mask_out)
{
    const int cb = (n + 63) / 64;
    uint64_t *mask = mask_out ? mask_out : (uint64_t *)malloc((size_t)n * (cb ? cb : 1) * sizeof(uint64_t));
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < cb; ++c) {
            uint64_t t = 0;
            int cs = n - c * 64 < 64 ? n - c * 64 : 64;
            int start = (i / 64 == c) ? (i % 64) + 1 : 0;
            for (int j = start; j < cs; ++j)
                if (orc_iou_bev(boxes + i * 5, boxes + (c * 64 + j) * 5) > thr) t |= 1ULL << j;
            mask[(size_t)i * cb + c] = t;
        }
    uint64_t *remv = (uint64_t *)calloc(cb ? cb : 1, sizeof(uint64_t));
    int nk = 0;
    for (int i = 0; i < n; ++i) {
        int nb = i / 64, ib = i % 64;
        if (!(remv[nb] & (1ULL << ib))) {
            keep[nk++] = i;
            for (int j = nb; j < cb; ++j) remv[j] |= mask[(size_t)i * cb + j];
        }
    }
    free(remv);
    if (!mask_out) free(mask);
    return nk;
}

/* ------------------------------------------------------------------------------------------------
 * Training-side point operators.
 * three_nn / three_interpolate (+grad): mmdet/ops/pointnet2/src/interpolate_gpu.cu:9-56, :80-102, :124-146.
 * pts_in_boxes3d: mmdet/ops/points_op/src/points_op.cpp:92-144 (literal argument order of the call site).
 * ---------------------------------------------------------------------------------------------- */
void orc_three_nn(int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx)
{
    for (int p = 0; p < n; ++p) {
        const float ub = unknown[p * 4], ux = unknown[p * 4 + 1], uy = unknown[p * 4 + 2], uz = unknown[p * 4 + 3];
        double b1 = 1e40, b2 = 1e40, b3 = 1e40;
        int i1 = 0, i2 = 0, i3 = 0;
        for (int k = 0; k < m; ++k) {
            if (known[k * 4] != ub) continue;
            const float x = known[k * 4 + 1], y = known[k * 4 + 2], z = known[k * 4 + 3];
            const float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
            if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
            else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
            else if (d < b3) { b3 = d; i3 = k; }
        }
        dist2[p * 3] = (float)b1; dist2[p * 3 + 1] = (float)b2; dist2[p * 3 + 2] = (float)b3;
        idx[p * 3] = i1; idx[p * 3 + 1] = i2; idx[p * 3 + 2] = i3;
    }
}

void orc_three_interpolate(int c, int m, int n, const float *points, const int32_t *idx, const float *weight,
                           float *out)
{
    (void)m;
    for (int p = 0; p < n; ++p)
        for (int ch = 0; ch < c; ++ch)
            out[(size_t)p * c + ch] = weight[p * 3] * points[(size_t)idx[p * 3] * c + ch] +
                                      weight[p * 3 + 1] * points[(size_t)idx[p * 3 + 1] * c + ch] +
                                      weight[p * 3 + 2] * points[(size_t)idx[p * 3 + 2] * c + ch];
}

void orc_three_interpolate_grad(int c, int n, int m, const float *grad_out, const int32_t *idx, const float *weight,
                                float *grad_points /* zeroed by the caller */)
{
    (void)m;
    for (int p = 0; p < n; ++p)
        for (int ch = 0; ch < c; ++ch)
            for (int j = 0; j < 3; ++j)
                grad_points[(size_t)idx[p * 3 + j] * c + ch] += grad_out[(size_t)p * c + ch] * weight[p * 3 + j];
}

static int orc_pt_in_box3d(float x, float y, float z, float cx, float cy, float bottom_z, float w, float l, float h,
                           float angle)
{
    float max_dis = 10.0, x_rot, y_rot, cosa, sina, cz;
    cz = bottom_z + h / 2.0;
    if ((fabsf(x - cx) > max_dis) || (fabsf(z - cz) > h / 2.0) || (fabsf(y - cy) > max_dis)) return 0;
    cosa = cosf(angle); sina = sinf(angle);
    x_rot = (x - cx) * cosa + (y - cy) * (-sina);
    y_rot = (x - cx) * sina + (y - cy) * cosa;
    return (x_rot >= -w / 2.0) & (x_rot <= w / 2.0) & (y_rot >= -l / 2.0) & (y_rot <= l / 2.0);
}

void orc_pts_in_boxes3d(const float *pts, int n, const float *boxes, int m, int32_t *flag /*[m,n]*/,
                        float *reg /*[n,3], zeroed by the caller*/)
{
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            const float *b = boxes + i * 7;
            const int in = orc_pt_in_box3d(pts[j * 3], pts[j * 3 + 1], pts[j * 3 + 2], b[0], b[1], b[2], b[3], b[4],
                                           b[5], b[6]);
            flag[(size_t)i * n + j] = in;
            if (in == 1) {
                reg[j * 3] = pts[j * 3] - b[0];
                reg[j * 3 + 1] = pts[j * 3 + 1] - b[1];
                reg[j * 3 + 2] = pts[j * 3 + 2] - (b[2] + b[3] / 2.0);
            }
        }
}

/* ----------------------------------------------------------------------------------------------
 * KITTI-evaluation rotated IoU: rotate_iou_gpu_eval,
 * mmdet/core/post_processing/rotate_nms_gpu.py:153-388 (device functions) and :536-627 (eval kernel + wrapper).
 * Boxes are (cx, cy, x_dim, y_dim, angle); the intersection polygon is collected from corners-inside-the-other-box
 * and the 16 edge-edge crossings, sorted by angle around the centroid (insertion sort on a cos-like key) and summed
 * as a triangle fan.  Arithmetic follows numba's typing of the reference: float32 storage, float64 where a Python
 * float literal / true division / accumulator promotes (x/2, /2.0, the centroid sums, the area accumulator).
 * ---------------------------------------------------------------------------------------------- */
static void orc_eval_corners(const float *rb, float *c /*8*/)
{
    const double a_cos = cos((double)rb[4]), a_sin = sin((double)rb[4]);   /* math.cos on float32 -> float64 in numba */
    const float cx[4] = {(float)(-rb[2] / 2.0), (float)(-rb[2] / 2.0), (float)(rb[2] / 2.0), (float)(rb[2] / 2.0)};
    const float cy[4] = {(float)(-rb[3] / 2.0), (float)(rb[3] / 2.0), (float)(rb[3] / 2.0), (float)(-rb[3] / 2.0)};
    for (int i = 0; i < 4; ++i) {
        c[2 * i] = (float)(a_cos * cx[i] + a_sin * cy[i] + rb[0]);
        c[2 * i + 1] = (float)(-a_sin * cx[i] + a_cos * cy[i] + rb[1]);
    }
}

static int orc_eval_pt_in_quad(float px, float py, const float *c)
{
    const float ab0 = c[2] - c[0], ab1 = c[3] - c[1], ad0 = c[6] - c[0], ad1 = c[7] - c[1];
    const float ap0 = px - c[0], ap1 = py - c[1];
    const float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
    const float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
    return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}

static int orc_eval_seg_isect(const float *p1, const float *p2, int i, int j, float *out)
{
    const float A0 = p1[2 * i], A1 = p1[2 * i + 1], B0 = p1[2 * ((i + 1) % 4)], B1 = p1[2 * ((i + 1) % 4) + 1];
    const float C0 = p2[2 * j], C1 = p2[2 * j + 1], D0 = p2[2 * ((j + 1) % 4)], D1 = p2[2 * ((j + 1) % 4) + 1];
    const float BA0 = B0 - A0, BA1 = B1 - A1, DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
    const int acd = DA1 * CA0 > CA1 * DA0;
    const int bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
    if (acd != bcd) {
        const int abc = CA1 * BA0 > BA1 * CA0, abd = DA1 * BA0 > BA1 * DA0;
        if (abc != abd) {
            const float DC0 = D0 - C0, DC1 = D1 - C1;
            const float ABBA = A0 * B1 - B0 * A1, CDDC = C0 * D1 - D0 * C1;
            const float DH = BA1 * DC0 - BA0 * DC1, Dx = ABBA * DC0 - BA0 * CDDC, Dy = ABBA * DC1 - BA1 * CDDC;
            out[0] = Dx / DH;
            out[1] = Dy / DH;
            return 1;
        }
    }
    return 0;
}

static double orc_eval_inter(const float *rb1, const float *rb2)
{
    float c1[8], c2[8], ip[16 + 32];
    int n = 0;
    orc_eval_corners(rb1, c1);
    orc_eval_corners(rb2, c2);
    for (int i = 0; i < 4; ++i) {
        if (orc_eval_pt_in_quad(c1[2 * i], c1[2 * i + 1], c2)) { ip[2 * n] = c1[2 * i]; ip[2 * n + 1] = c1[2 * i + 1]; ++n; }
        if (orc_eval_pt_in_quad(c2[2 * i], c2[2 * i + 1], c1)) { ip[2 * n] = c2[2 * i]; ip[2 * n + 1] = c2[2 * i + 1]; ++n; }
    }
    float tp[2];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (orc_eval_seg_isect(c1, c2, i, j, tp)) { ip[2 * n] = tp[0]; ip[2 * n + 1] = tp[1]; ++n; }
    if (n > 0) {                                         /* sort_vertex_in_convex_polygon */
        float cen[2] = {0.f, 0.f}, vs[24];
        for (int i = 0; i < n; ++i) { cen[0] += ip[2 * i]; cen[1] += ip[2 * i + 1]; }
        cen[0] /= n; cen[1] /= n;
        for (int i = 0; i < n; ++i) {
            float v0 = ip[2 * i] - cen[0], v1 = ip[2 * i + 1] - cen[1];
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

/* iou[n][k] = f(query k, box n): criterion -1 IoU, 0 inter / area(query), 1 inter / area(box), else inter */
void orc_rotate_iou_eval(const float *boxes, int n, const float *qboxes, int k, int criterion, float *iou)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < k; ++j) {
            const float *r1 = qboxes + 5 * j, *r2 = boxes + 5 * i;
            const float a1 = r1[2] * r1[3], a2 = r2[2] * r2[3];
            const double in = orc_eval_inter(r1, r2);
            double v;
            if (criterion == -1) v = in / (a1 + a2 - in);
            else if (criterion == 0) v = in / a1;
            else if (criterion == 1) v = in / a2;
            else v = in;
            iou[(size_t)i * k + j] = (float)v;
        }
}
