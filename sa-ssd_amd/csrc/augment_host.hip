// augment_host.hip -- host half of the training-side augmentation: the decisions that are sequential
// over a few dozen boxes -- which sampled database objects collide with the scene, which of the 100 noise draws of
// each ground-truth box is the first one that keeps it clear of the others.  numba CPU code in the reference
// (`box_collision_test` mmdet/core/bbox3d/geometry.py:593-672, `noise_per_box` mmdet/core/point_cloud/
// point_augmentor.py:73-105); O(boxes^2 x tries) with early exits, well under a millisecond in C++, and its result
// steers the per-point kernels in augment.hip.  Plain C++ behind the C ABI; touches no device memory.
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/sassd.h"

#pragma clang fp contract(off)      // the reference's numba loops round every product (no fused multiply-add)

namespace {

template <typename T> struct Quad { T x[4], y[4]; };

// geometry.py:593-672.  Two rotated rectangles (corners in the reference's clockwise order) collide when their
// axis-aligned hulls overlap and either two edges cross or one rectangle lies completely inside the other.
template <typename T> bool collide(const Quad<T> &a, const Quad<T> &b)
{
    T a_lo_x = a.x[0], a_hi_x = a.x[0], a_lo_y = a.y[0], a_hi_y = a.y[0];
    T b_lo_x = b.x[0], b_hi_x = b.x[0], b_lo_y = b.y[0], b_hi_y = b.y[0];
    for (int i = 1; i < 4; ++i) {
        a_lo_x = std::fmin(a_lo_x, a.x[i]); a_hi_x = std::fmax(a_hi_x, a.x[i]);
        a_lo_y = std::fmin(a_lo_y, a.y[i]); a_hi_y = std::fmax(a_hi_y, a.y[i]);
        b_lo_x = std::fmin(b_lo_x, b.x[i]); b_hi_x = std::fmax(b_hi_x, b.x[i]);
        b_lo_y = std::fmin(b_lo_y, b.y[i]); b_hi_y = std::fmax(b_hi_y, b.y[i]);
    }
    if (!(std::fmin(a_hi_x, b_hi_x) - std::fmax(a_lo_x, b_lo_x) > 0)) return false;
    if (!(std::fmin(a_hi_y, b_hi_y) - std::fmax(a_lo_y, b_lo_y) > 0)) return false;

    for (int k = 0; k < 4; ++k) {                                  // edge A->B of a against edge C->D of b
        const T Ax = a.x[k], Ay = a.y[k], Bx = a.x[(k + 1) & 3], By = a.y[(k + 1) & 3];
        for (int l = 0; l < 4; ++l) {
            const T Cx = b.x[l], Cy = b.y[l], Dx = b.x[(l + 1) & 3], Dy = b.y[(l + 1) & 3];
            const bool acd = (Dy - Ay) * (Cx - Ax) > (Cy - Ay) * (Dx - Ax);
            const bool bcd = (Dy - By) * (Cx - Bx) > (Cy - By) * (Dx - Bx);
            if (acd == bcd) continue;
            const bool abc = (Cy - Ay) * (Bx - Ax) > (By - Ay) * (Cx - Ax);
            const bool abd = (Dy - Ay) * (Bx - Ax) > (By - Ay) * (Dx - Ax);
            if (abc != abd) return true;
        }
    }
    // no crossing edges: all corners of `in` strictly inside `out`?
    auto encloses = [](const Quad<T> &out, const Quad<T> &in) {
        for (int l = 0; l < 4; ++l)
            for (int k = 0; k < 4; ++k) {
                const T vx = -(out.x[k] - out.x[(k + 1) & 3]), vy = -(out.y[k] - out.y[(k + 1) & 3]);
                T cross = vy * (out.x[k] - in.x[l]);
                cross -= vx * (out.y[k] - in.y[l]);
                if (cross >= 0) return false;
            }
        return true;
    };
    return encloses(a, b) || encloses(b, a);
}

template <typename T> void collision_matrix(const T *boxes, int n, const T *qboxes, int k, uint8_t *out)
{
    std::vector<Quad<T>> q((size_t)k);
    for (int j = 0; j < k; ++j)
        for (int c = 0; c < 4; ++c) { q[j].x[c] = qboxes[j * 8 + 2 * c]; q[j].y[c] = qboxes[j * 8 + 2 * c + 1]; }
    for (int i = 0; i < n; ++i) {
        Quad<T> a;
        for (int c = 0; c < 4; ++c) { a.x[c] = boxes[i * 8 + 2 * c]; a.y[c] = boxes[i * 8 + 2 * c + 1]; }
        for (int j = 0; j < k; ++j) out[(size_t)i * k + j] = collide(a, q[j]) ? 1 : 0;
    }
}

// corners of (x, y, w, l, yaw) in float32, the order (-,-) (-,+) (+,+) (+,-) rotated by [[c,-s],[s,c]] (geometry.py:519-538)
Quad<float> corners_of(const float *b)
{
    const float s = (float)std::sin((double)b[4]), c = (float)std::cos((double)b[4]);
    const float ux[4] = {-0.5f, -0.5f, 0.5f, 0.5f}, uy[4] = {-0.5f, 0.5f, 0.5f, -0.5f};
    Quad<float> q;
    for (int i = 0; i < 4; ++i) {
        const float px = b[2] * ux[i], py = b[3] * uy[i];
        q.x[i] = (px * c + py * s) + b[0];
        q.y[i] = (px * (-s) + py * c) + b[1];
    }
    return q;
}
}  // namespace

extern "C" int sassd_box_collision_test(const void *boxes, int n, const void *qboxes, int k, int is_f64, uint8_t *out)
{
    if (n < 0 || k < 0) return SASSD_EINVAL;
    if (n == 0 || k == 0) return SASSD_OK;
    if (!boxes || !qboxes || !out) return SASSD_EINVAL;
    if (is_f64) collision_matrix((const double *)boxes, n, (const double *)qboxes, k, out);
    else collision_matrix((const float *)boxes, n, (const float *)qboxes, k, out);
    return SASSD_OK;
}

extern "C" int sassd_noise_per_box(const float *boxes, const uint8_t *valid, const double *loc_noises,
                                   const double *rot_noises, int n, int num_try, int64_t *success)
{
    if (n < 0 || num_try < 0) return SASSD_EINVAL;
    if (n == 0) return SASSD_OK;
    if (!boxes || !valid || !success || (num_try > 0 && (!loc_noises || !rot_noises))) return SASSD_EINVAL;
    std::vector<Quad<float>> corner((size_t)n);
    for (int i = 0; i < n; ++i) corner[i] = corners_of(boxes + 5 * i);
    for (int i = 0; i < n; ++i) {
        success[i] = -1;
        if (!valid[i]) continue;
        const float cx = boxes[5 * i], cy = boxes[5 * i + 1];
        for (int j = 0; j < num_try; ++j) {
            const double a = rot_noises[(size_t)i * num_try + j];
            const float s = (float)std::sin(a), c = (float)std::cos(a);
            const double *loc = loc_noises + ((size_t)i * num_try + j) * 3;
            Quad<float> cur;
            for (int p = 0; p < 4; ++p) {                          // rotate about the centre, then centre + noise
                const float rx = corner[i].x[p] - cx, ry = corner[i].y[p] - cy;
                cur.x[p] = (float)((double)(rx * c + ry * s) + ((double)cx + loc[0]));
                cur.y[p] = (float)((double)(rx * (-s) + ry * c) + ((double)cy + loc[1]));
            }
            bool hit = false;
            for (int o = 0; o < n && !hit; ++o)
                if (o != i) hit = collide(cur, corner[o]);
            if (!hit) { success[i] = j; corner[i] = cur; break; }
        }
    }
    return SASSD_OK;
}
