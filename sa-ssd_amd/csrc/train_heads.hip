// train_heads.hip -- the guided-anchor / rescoring tail of the TRAINING step, fused (padded, device counts, no host sync):
//
//   sassd_guided_decode_fwd / _bwd   ssd_rotate_head.py:316-388 get_guided_anchors (train mode): the anchors chosen by
//                                    sassd_guided_select are decoded (second_box_decode, :53-91), direction-flipped
//                                    (:352-356) and written behind the sample's ground-truth boxes (:357-359) into one
//                                    padded [B, Gmax + cap, 7] tensor; backward maps d(guided) to d(box_preds).  Replaces
//                                    ~50 gather / split / elementwise / cat launches each way.
//   sassd_boxes_iou3d_batch          iou3d_utils.py:79-111 boxes_iou3d_gpu (RotateIou3dSimilarity) of every sample's guided
//                                    boxes against its ground truth in ONE launch (rotated BEV overlap x height overlap),
//                                    laid out for sassd_assign_targets' `overlaps` argument; ~30 launches per sample before.
//   sassd_focal_loss                 ssd_rotate_head.py:456-490 PSWarpHead.loss: sigmoid focal loss of the rescoring logits
//                                    against labels (-1 ignore / 0 / positive) normalised by the positives of the batch:
//                                    the sum AND its gradient in one pass (losses.py:35-62).
#include "common.h"
#include "iou3d_device.h"

namespace {

struct GuidedDecodeArgs {
    const float *box;          // [B, A, 7]
    const float *dir;          // [B, A, 2] or null
    const float *anchors;      // [A, 7] or [B, A, 7]
    size_t anchor_stride;      // 0 or A * 7
    const int64_t *sel;        // [B, cap] ascending anchor indices (padding beyond cnt)
    const int *cnt;            // [B] selected per sample
    const float *gt;           // [T, 7]
    const int *gt_off;         // [B + 1]
    int A, B, cap, rows;       // rows = Gmax + cap
    float *guided;             // [B, rows, 7]
    int *counts;               // [B] = G_b + cnt[b]
    const float *dguided;      // backward
    float *dbox;               // [B, A, 7], zeroed by the caller
};

__global__ void __launch_bounds__(256) guided_decode_fwd_kernel(GuidedDecodeArgs P)
{
#pragma clang fp contract(off)
    const int b = blockIdx.y;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= P.rows) return;
    const int g0 = P.gt_off[b], G = P.gt_off[b + 1] - g0;
    const int K = min(P.cnt[b], P.cap);
    if (r == 0) P.counts[b] = G + K;
    float *o = P.guided + ((size_t)b * P.rows + r) * 7;
    if (r < G) {
        const float *g = P.gt + (size_t)(g0 + r) * 7;
#pragma unroll
        for (int j = 0; j < 7; ++j) o[j] = g[j];
        return;
    }
    const int k = r - G;
    if (k >= K) {
#pragma unroll
        for (int j = 0; j < 7; ++j) o[j] = 0.f;
        return;
    }
    const int64_t a = P.sel[(size_t)b * P.cap + k];
    const float *t = P.box + ((size_t)b * P.A + a) * 7;
    const float *an = P.anchors + (size_t)b * P.anchor_stride + (size_t)a * 7;
    const float xa = an[0], ya = an[1], wa = an[3], la = an[4], ha = an[5], ra = an[6];
    const float za = an[2] + ha / 2.f;
    const float diag = sqrtf(la * la + wa * wa);
    const float xg = t[0] * diag + xa, yg = t[1] * diag + ya, zg = t[2] * ha + za;
    const float lg = expf(t[4]) * la, wg = expf(t[3]) * wa, hg = expf(t[5]) * ha;
    float rg = t[6] + ra;
    if (P.dir) {
        const float *d = P.dir + ((size_t)b * P.A + a) * 2;
        const bool opp = (rg > 0.f) != (d[1] > d[0]);                    // (rot > 0) ^ argmax(dir)
        rg = rg + (opp ? 1.f : 0.f) * 3.14159274101257324f;               // fp32(np.pi)
    }
    o[0] = xg; o[1] = yg; o[2] = zg - hg / 2.f; o[3] = wg; o[4] = lg; o[5] = hg; o[6] = rg;
}

__global__ void __launch_bounds__(256) guided_decode_bwd_kernel(GuidedDecodeArgs P)
{
    const int b = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int g0 = P.gt_off[b], G = P.gt_off[b + 1] - g0;
    const int K = min(P.cnt[b], P.cap);
    if (k >= K) return;
    const int64_t a = P.sel[(size_t)b * P.cap + k];
    const float *t = P.box + ((size_t)b * P.A + a) * 7;
    const float *an = P.anchors + (size_t)b * P.anchor_stride + (size_t)a * 7;
    const float *d = P.dguided + ((size_t)b * P.rows + G + k) * 7;
    const float wa = an[3], la = an[4], ha = an[5];
    const float diag = sqrtf(la * la + wa * wa);
    const float lg = expf(t[4]) * la, wg = expf(t[3]) * wa, hg = expf(t[5]) * ha;
    float *o = P.dbox + ((size_t)b * P.A + a) * 7;                        // selected anchors are distinct: plain stores
    o[0] = d[0] * diag; o[1] = d[1] * diag; o[2] = d[2] * ha;
    o[3] = d[3] * wg; o[4] = d[4] * lg; o[5] = d[5] * hg - d[2] * (hg / 2.f);
    o[6] = d[6];
}

// ---- 3-D IoU of padded guided boxes against the sample's ground truth -------------------------------------------------
struct Iou3dBatchArgs {
    const float *boxes;        // [B, rows, 7]
    const int *counts;         // [B] valid rows
    const float *gt;           // [T, 7]
    const int *gt_off;         // [B + 1]
    const int64_t *ov_off;     // [B + 1] element offsets of the samples' [rows, G_b] matrices
    int B, rows;
    float *ov;
};

__device__ __forceinline__ iou3d::Box bev_of(const float *b)
{
#pragma clang fp contract(off)
    iou3d::Box o;                                                         // iou3d_utils.py:47-60 (cols 0, 1, 3, 4, 6)
    const float hx = b[3] / 2.f, hy = b[4] / 2.f;
    o.x1 = b[0] - hx; o.y1 = b[1] - hy; o.x2 = b[0] + hx; o.y2 = b[1] + hy; o.r = b[6];
    return o;
}

__global__ void __launch_bounds__(256) iou3d_batch_kernel(Iou3dBatchArgs P, int gmax)
{
#pragma clang fp contract(off)
    const int b = blockIdx.y;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int r = (int)(t / gmax), g = (int)(t - (long)r * gmax);
    const int g0 = P.gt_off[b], G = P.gt_off[b + 1] - g0;
    if (r >= P.rows || g >= G) return;
    float *dst = P.ov + P.ov_off[b] + (size_t)r * G + g;
    if (r >= P.counts[b]) { *dst = 0.f; return; }
    const float *a = P.boxes + ((size_t)b * P.rows + r) * 7;
    const float *q = P.gt + (size_t)(g0 + g) * 7;
    const float ov = iou3d::box_overlap(bev_of(a), bev_of(q));
    const float top = fminf(a[2] + a[5], q[2] + q[5]), bot = fmaxf(a[2], q[2]);
    const float oh = fmaxf(top - bot, 0.f);
    const float o3 = ov * oh;
    const float va = a[3] * a[4] * a[5], vb = q[3] * q[4] * q[5];
    *dst = o3 / fmaxf(va + vb - o3, 1e-7f);
}

// ---- sigmoid focal loss (gamma 2, alpha 0.25) of [n] logits, weight cared / max(sum npos, 1) ----------------------------
struct FocalArgs {
    const float *x;            // [n]
    const int64_t *labels;     // [n]: -1 ignore, 0 negative, > 0 positive
    const int *npos;           // [nb]
    int n, nb;
    float *grad;               // [n]
    float *part;               // [nblocks]
};

__global__ void __launch_bounds__(256) focal_loss_kernel(FocalArgs P)
{
    __shared__ float red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float l = 0.f;
    if (i < P.n) {
        int np = 0;
        for (int k = 0; k < P.nb; ++k) np += P.npos[k];
        const float norm = fmaxf((float)np, 1.f);
        const int64_t lab = P.labels[i];
        const float cw = (lab >= 0 ? 1.f : 0.f) / norm;
        const float t = lab > 0 ? 1.f : 0.f;
        const float x = P.x[i];
        const float p = 1.f / (1.f + expf(-x));
        const float pt = (1.f - p) * t + p * (1.f - t);
        const float aw = (0.25f * t + 0.75f * (1.f - t)) * cw;
        const float w = aw * (pt * pt);
        const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
        l = bce * w;
        const float dpt = p * (1.f - p) * (1.f - 2.f * t);
        P.grad[i] = (p - t) * w + bce * aw * 2.f * pt * dpt;
    }
    for (int o = 32; o > 0; o >>= 1) l += __shfl_down(l, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) P.part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256) focal_sum_kernel(const float *__restrict__ part, int n, float *__restrict__ out)
{
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)part[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}
}  // namespace

extern "C" int sassd_guided_decode_fwd(const float *box_preds, const float *dir_preds, const float *anchors,
                                       int anchors_per_sample, const int64_t *sel, const int32_t *sel_count,
                                       const float *gt_boxes, const int32_t *gt_off, int A, int B, int cap, int gmax,
                                       float *guided, int32_t *counts, void *stream_)
{
    if (!box_preds || !anchors || !sel || !sel_count || !gt_off || A < 1 || B < 1 || cap < 1 || gmax < 0 || !guided ||
        !counts || (gmax > 0 && !gt_boxes))
        return SASSD_EINVAL;
    GuidedDecodeArgs P = {};
    P.box = box_preds; P.dir = dir_preds; P.anchors = anchors; P.anchor_stride = anchors_per_sample ? (size_t)A * 7 : 0;
    P.sel = sel; P.cnt = sel_count; P.gt = gt_boxes; P.gt_off = gt_off; P.A = A; P.B = B; P.cap = cap; P.rows = gmax + cap;
    P.guided = guided; P.counts = counts;
    hipLaunchKernelGGL(guided_decode_fwd_kernel, dim3(cdiv(P.rows, 256), B), dim3(256), 0, (hipStream_t)stream_, P);
    return sassd_launch_status();
}

extern "C" int sassd_guided_decode_bwd(const float *box_preds, const float *anchors, int anchors_per_sample,
                                       const int64_t *sel, const int32_t *sel_count, const int32_t *gt_off, int A, int B,
                                       int cap, int gmax, const float *dguided, float *dbox, void *stream_)
{
    if (!box_preds || !anchors || !sel || !sel_count || !gt_off || A < 1 || B < 1 || cap < 1 || gmax < 0 || !dguided ||
        !dbox)
        return SASSD_EINVAL;
    hipStream_t s = (hipStream_t)stream_;
    GuidedDecodeArgs P = {};
    P.box = box_preds; P.anchors = anchors; P.anchor_stride = anchors_per_sample ? (size_t)A * 7 : 0;
    P.sel = sel; P.cnt = sel_count; P.gt_off = gt_off; P.A = A; P.B = B; P.cap = cap; P.rows = gmax + cap;
    P.dguided = dguided; P.dbox = dbox;
    if (hipMemsetAsync(dbox, 0, (size_t)B * A * 7 * sizeof(float), s) != hipSuccess) return sassd_launch_status();
    hipLaunchKernelGGL(guided_decode_bwd_kernel, dim3(cdiv(cap, 256), B), dim3(256), 0, s, P);
    return sassd_launch_status();
}

extern "C" int sassd_boxes_iou3d_batch(const float *boxes, const int32_t *counts, int B, int rows, const float *gt_boxes,
                                       const int32_t *gt_off, int gmax, const int64_t *ov_off, float *overlaps,
                                       void *stream_)
{
    if (!boxes || !counts || B < 1 || rows < 1 || !gt_off || gmax < 0 || !ov_off || (gmax > 0 && (!gt_boxes || !overlaps)))
        return SASSD_EINVAL;
    if (gmax == 0) return SASSD_OK;
    Iou3dBatchArgs P = {};
    P.boxes = boxes; P.counts = counts; P.gt = gt_boxes; P.gt_off = gt_off; P.ov_off = ov_off; P.B = B; P.rows = rows;
    P.ov = overlaps;
    const long tot = (long)rows * gmax;
    hipLaunchKernelGGL(iou3d_batch_kernel, dim3((unsigned)((tot + 255) / 256), B), dim3(256), 0, (hipStream_t)stream_, P,
                       gmax);
    return sassd_launch_status();
}

extern "C" size_t sassd_focal_loss_workspace_bytes(int n) { return align_up((size_t)(n > 0 ? cdiv(n, 256) : 1) * 4, 256); }

extern "C" int sassd_focal_loss(const float *logits, const int64_t *labels, int n, const int32_t *num_pos, int nb,
                                float *loss_sum, float *grad, void *workspace, size_t workspace_bytes, void *stream_)
{
    if (!logits || !labels || n < 1 || !num_pos || nb < 1 || !loss_sum || !grad || !workspace) return SASSD_EINVAL;
    if (workspace_bytes < sassd_focal_loss_workspace_bytes(n)) return SASSD_ENOSPC;
    hipStream_t s = (hipStream_t)stream_;
    FocalArgs P = {};
    P.x = logits; P.labels = labels; P.npos = num_pos; P.n = n; P.nb = nb; P.grad = grad; P.part = (float *)workspace;
    const int nblk = cdiv(n, 256);
    hipLaunchKernelGGL(focal_loss_kernel, dim3(nblk), dim3(256), 0, s, P);
    hipLaunchKernelGGL(focal_sum_kernel, dim3(1), dim3(256), 0, s, (const float *)workspace, nblk, loss_sum);
    return sassd_launch_status();
}
