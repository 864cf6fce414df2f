// train_loss.hip -- fused anchor-target assignment and RPN loss of the training step.
//
//   sassd_assign_targets   mmdet/core/bbox3d/target_ops.py:139-277 (create_target_np / create_target_torch) with
//                          NearestIouSimilarity (mmdet/ops/iou3d/iou3d_utils.py:163-183) and second_box_encode
//                          (ssd_rotate_head.py:15-50) for a whole batch: labels, regression targets, best overlap and
//                          the positives per sample, in two launches instead of ~60 elementwise launches per sample.
//   sassd_rpn_loss         ssd_rotate_head.py:128-314 loss(): NormByNumPositives weights, sigmoid focal loss
//                          (losses.py:35-62), sin-difference smooth-L1 (losses.py:72-96, beta 1/9), direction
//                          cross-entropy -- the three loss sums AND their gradients with respect to the head outputs in
//                          one pass (the backward of the autograd node is a scale by the upstream gradient).
//
// The arithmetic is the reference's fp32 sequence operation by operation (contraction off), so thresholds, ties and
// the "anchors tying a ground truth's best overlap" rule select the same anchors as the elementwise formulation.
#include "common.h"

namespace {
#pragma clang fp contract(off)

constexpr float kPiF = 3.14159274101257324f;         // float(math.pi)
constexpr float kQuarterPiF = 0.785398185253143311f; // float(math.pi / 4)
constexpr int kMaxGtLds = 128;

struct NearBox { float x0, y0, x1, y1; };

// boxes3d_to_near_torch: rotated (x, y, w, l, r) -> nearest axis-aligned box
__device__ __forceinline__ NearBox near_box(const float *b)
{
    const float r = b[6];
    const float lim = r - floorf(r / kPiF + 0.5f) * kPiF;
    const bool swap = fabsf(lim) > kQuarterPiF;
    const float sx = swap ? b[4] : b[3], sy = swap ? b[3] : b[4];
    NearBox n;
    n.x0 = b[0] - sx / 2.f; n.y0 = b[1] - sy / 2.f; n.x1 = b[0] + sx / 2.f; n.y1 = b[1] + sy / 2.f;
    return n;
}

__device__ __forceinline__ float near_iou(const NearBox &a, const NearBox &g)
{
    const float w = fmaxf(fminf(a.x1, g.x1) - fmaxf(a.x0, g.x0), 0.f);
    const float h = fmaxf(fminf(a.y1, g.y1) - fmaxf(a.y0, g.y0), 0.f);
    const float ov = w * h;
    const float a1 = (a.x1 - a.x0) * (a.y1 - a.y0), a2 = (g.x1 - g.x0) * (g.y1 - g.y0);
    return ov / (a1 + a2 - ov);
}

struct AssignArgs {
    const float *anchors;          // [A,7] (anchor_stride = 0) or [B,A,7]
    size_t anchor_stride;
    const uint8_t *anchor_mask;    // [B,A] or null
    const float *gt;               // [sum G, 7]
    const int64_t *gt_cls;         // [sum G] or null (all 1)
    const uint8_t *gt_ok;          // [sum G] or null
    const int32_t *gt_off;         // [B+1] (device)
    const float *ov;               // optional precomputed overlaps, row-major [A, G] per sample at ov_off[b]
    const int64_t *ov_off;
    int A, B;
    size_t out_stride;             // anchors between consecutive samples in labels / targets / best_out (>= A)
    float matched, unmatched;
    int *g2a;                      // [sum G] best overlap per ground truth (float bits), pre-set to bits(-1.0f)
    float *a_best;                 // [B,A] best overlap per anchor (pass 1 -> pass 2)
    int *a_arg;                    // [B,A]
    int64_t *labels;               // [B,A]
    float *targets;                // [B,A,7]
    float *best_out;               // [B,A] or null
    int *npos;                     // [B]
};

__device__ __forceinline__ float overlap_of(const AssignArgs &P, int b, int a, int g, int G, const NearBox &an,
                                            const NearBox &gn)
{
    if (P.ov) return P.ov[P.ov_off[b] + (size_t)a * G + g];
    return near_iou(an, gn);
}

// pass 1: per anchor the best overlap / argmax over the sample's ground truths; per ground truth the best overlap
__global__ void __launch_bounds__(256) assign_pass1_kernel(AssignArgs P)
{
    __shared__ NearBox gbox[kMaxGtLds];
    __shared__ int gok[kMaxGtLds];
    __shared__ int gmax[kMaxGtLds];
    const int b = blockIdx.y;
    const int g0 = P.gt_off[b], G = P.gt_off[b + 1] - g0;
    const int a = blockIdx.x * 256 + threadIdx.x;
    const bool in = a < P.A;
    const bool row_ok = in && (!P.anchor_mask || P.anchor_mask[(size_t)b * P.A + a]);
    NearBox an = {0, 0, 0, 0};
    if (in && !P.ov) an = near_box(P.anchors + b * P.anchor_stride + (size_t)a * 7);
    float best = -1.f;
    int arg = 0;
    for (int c0 = 0; c0 < G; c0 += kMaxGtLds) {
        const int n = min(kMaxGtLds, G - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) {
            if (!P.ov) gbox[i] = near_box(P.gt + (size_t)(g0 + c0 + i) * 7);
            gok[i] = !P.gt_ok || P.gt_ok[g0 + c0 + i];
            gmax[i] = __float_as_int(-1.f);
        }
        __syncthreads();
        if (in) {
            for (int i = 0; i < n; ++i) {
                const float v = (row_ok && gok[i]) ? overlap_of(P, b, a, c0 + i, G, an, gbox[i]) : -1.f;
                if (v > best) { best = v; arg = c0 + i; }                      // first maximum wins (torch.max)
                if (v >= 0.f) atomicMax(&gmax[i], __float_as_int(v));
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256)
            if (gmax[i] >= 0) atomicMax(&P.g2a[g0 + c0 + i], gmax[i]);
    }
    if (in) {
        P.a_best[(size_t)b * P.A + a] = best;
        P.a_arg[(size_t)b * P.A + a] = arg;
    }
}

// pass 2: labels (match / no match / forced by a ground truth's best overlap), encoded targets, positives per sample
__global__ void __launch_bounds__(256) assign_pass2_kernel(AssignArgs P)
{
    __shared__ NearBox gbox[kMaxGtLds];
    __shared__ float gbest[kMaxGtLds];
    __shared__ int gok[kMaxGtLds];
    __shared__ int wcnt[4];
    const int b = blockIdx.y;
    const int g0 = P.gt_off[b], G = P.gt_off[b + 1] - g0;
    const int a = blockIdx.x * 256 + threadIdx.x;
    const bool in = a < P.A;
    const size_t ia = (size_t)b * P.A + a;                       // workspace / mask index
    const size_t io = (size_t)b * P.out_stride + a;              // output index
    const bool row_ok = in && (!P.anchor_mask || P.anchor_mask[ia]);
    const float *anc = P.anchors + b * P.anchor_stride + (size_t)(in ? a : 0) * 7;
    NearBox an = {0, 0, 0, 0};
    if (in && !P.ov) an = near_box(anc);
    bool forced = false;
    for (int c0 = 0; c0 < G; c0 += kMaxGtLds) {
        const int n = min(kMaxGtLds, G - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) {
            if (!P.ov) gbox[i] = near_box(P.gt + (size_t)(g0 + c0 + i) * 7);
            gok[i] = !P.gt_ok || P.gt_ok[g0 + c0 + i];
            gbest[i] = __int_as_float(P.g2a[g0 + c0 + i]);
        }
        __syncthreads();
        if (row_ok)
            for (int i = 0; i < n; ++i)
                if (gok[i] && gbest[i] > 0.f && overlap_of(P, b, a, c0 + i, G, an, gbox[i]) == gbest[i]) forced = true;
    }
    int pos = 0;
    if (in) {
        const float best = G > 0 ? P.a_best[ia] : -1.f;
        int64_t label = -1;
        float t[7] = {0, 0, 0, 0, 0, 0, 0};
        if (G == 0) {
            label = row_ok ? 0 : -1;
        } else {
            const int arg = P.a_arg[ia];
            const int64_t cls = P.gt_cls ? P.gt_cls[g0 + arg] : 1;
            if (best >= P.matched) label = cls;
            if (best < P.unmatched) label = 0;
            if (forced) label = cls;
            if (!row_ok) label = -1;
            if (label > 0) {
                const float *g = P.gt + (size_t)(g0 + arg) * 7;                   // second_box_encode
                const float zg = g[2] + g[5] / 2.f, za = anc[2] + anc[5] / 2.f;
                const float diag = sqrtf(anc[4] * anc[4] + anc[3] * anc[3]);
                t[0] = (g[0] - anc[0]) / diag; t[1] = (g[1] - anc[1]) / diag; t[2] = (zg - za) / anc[5];
                t[3] = logf(g[3] / anc[3]); t[4] = logf(g[4] / anc[4]); t[5] = logf(g[5] / anc[5]);
                t[6] = g[6] - anc[6];
            }
        }
        P.labels[io] = label;
#pragma unroll
        for (int j = 0; j < 7; ++j) P.targets[io * 7 + j] = t[j];
        if (P.best_out) P.best_out[io] = row_ok ? (G > 0 ? best : 0.f) : -1.f;
        pos = label > 0;
    }
    const unsigned long long m = __ballot(pos);
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int s = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        if (s) atomicAdd(&P.npos[b], s);
    }
}

__global__ void fill_int_kernel(int *p, int n, int v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------------------------------------------------
struct RpnLossArgs {
    const float *box, *cls, *dir;      // [B,A,7], [B,A,NC], [B,A,2] (dir may be null)
    const int64_t *labels;             // [B,A]
    const float *targets;              // [B,A,7]
    const float *anchors;              // [A,7] / [B,A,7]
    size_t anchor_stride;
    const int *npos;                   // [B]
    int A, B, NC;
    float *gbox, *gcls, *gdir;         // gradients of the UNSCALED sums (loc, cls, dir)
    float *part;                       // [nblocks][3]
};

__global__ void __launch_bounds__(256) rpn_loss_kernel(RpnLossArgs P)
{
    __shared__ float red[3][4];
    const int b = blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    float l_loc = 0.f, l_cls = 0.f, l_dir = 0.f;
    if (a < P.A) {
        const size_t ia = (size_t)b * P.A + a;
        const int64_t label = P.labels[ia];
        const float norm = fmaxf((float)P.npos[b], 1.f);
        const bool pos = label > 0, neg = label == 0;
        const float cls_w = ((neg ? 1.f : 0.f) * 1.f + 1.f * (pos ? 1.f : 0.f)) / norm;
        const float reg_w = (pos ? 1.f : 0.f) / norm;
        // ---- sigmoid focal loss (gamma 2, alpha 0.25), gradient through the modulating factor as autograd has it
        for (int c = 0; c < P.NC; ++c) {
            const float x = P.cls[ia * P.NC + c];
            const float t = (label == c + 1) ? 1.f : 0.f;
            const float p = 1.f / (1.f + expf(-x));
            const float pt = (1.f - p) * t + p * (1.f - t);
            const float aw = (0.25f * t + 0.75f * (1.f - t)) * cls_w;
            const float w = aw * (pt * pt);
            // binary_cross_entropy_with_logits: max(x,0) - x*t + log(1 + exp(-|x|))
            const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
            l_cls += bce * w;
            const float dpt = p * (1.f - p) * (1.f - 2.f * t);
            P.gcls[ia * P.NC + c] = (p - t) * w + bce * aw * 2.f * pt * dpt;
        }
        // ---- smooth L1 (beta 1/9) with the sin-difference encoding of the angle
        const float beta = 1.f / 9.f;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const float pr = P.box[ia * 7 + j], tg = P.targets[ia * 7 + j];
            float diff, dd = 1.f;
            if (j < 6) {
                diff = pr - tg;
            } else {
                const float sp = sinf(pr), cp = cosf(pr), st = sinf(tg), ct = cosf(tg);
                diff = sp * ct - cp * st;
                dd = cp * ct + sp * st;
            }
            const float d = fabsf(diff);
            l_loc += (d < beta ? 0.5f * d * d / beta : d - 0.5f * beta) * reg_w;
            const float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
            P.gbox[ia * 7 + j] = (d < beta ? d / beta : 1.f) * sg * dd * reg_w;
        }
        // ---- direction classifier: 2-way cross-entropy on (target angle + anchor angle > 0), positives only
        if (P.dir) {
            const float rot = P.targets[ia * 7 + 6] + P.anchors[b * P.anchor_stride + (size_t)a * 7 + 6];
            const int dl = rot > 0.f ? 1 : 0;
            const float x0 = P.dir[ia * 2], x1 = P.dir[ia * 2 + 1];
            const float mx = fmaxf(x0, x1);
            const float e0 = expf(x0 - mx), e1 = expf(x1 - mx);
            const float lse = mx + logf(e0 + e1);
            l_dir = (lse - (dl ? x1 : x0)) * reg_w;
            const float s0 = e0 / (e0 + e1), s1 = e1 / (e0 + e1);
            P.gdir[ia * 2] = (s0 - (dl ? 0.f : 1.f)) * reg_w;
            P.gdir[ia * 2 + 1] = (s1 - (dl ? 1.f : 0.f)) * reg_w;
        }
    }
    // fixed-order block reduction -> one partial triple per block (summed in order by rpn_loss_sum_kernel)
    float v[3] = {l_loc, l_cls, l_dir};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = v[k];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        P.part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 + k] = (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]);
    }
}

__global__ void __launch_bounds__(256) rpn_loss_sum_kernel(const float *__restrict__ part, int n, float *__restrict__ out)
{
    __shared__ double red[3][4];
    double s[3] = {0, 0, 0};
    for (int i = threadIdx.x; i < n; i += 256)
        for (int k = 0; k < 3; ++k) s[k] += (double)part[(size_t)i * 3 + k];
    for (int k = 0; k < 3; ++k) {
        double v = s[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3) out[threadIdx.x] = (float)((red[threadIdx.x][0] + red[threadIdx.x][1]) +
                                                    (red[threadIdx.x][2] + red[threadIdx.x][3]));
}

// ---------------------------------------------------------------------------------------------------------------------
// Guided-anchor selection of the training step (ssd_rotate_head.py:316-388: anchors inside the anchor mask whose best
// class score exceeds train_cfg.rpn.anchor_thr), without the host round trip of a boolean compaction: ascending
// anchor indices of the selected anchors land in sel[b][0 .. cnt[b]) of a FIXED-capacity buffer (the rest stays 0) and
// the count stays on the device, so the decode / PSWarp / loss that follow run on padded tensors with a device count.
struct GuidedArgs {
    const float *cls;          // [B,A,NC] logits
    const uint8_t *mask;       // [B,A] or null
    int A, B, NC, cap, nblk;
    float thr;
    int64_t *sel;              // [B,cap]
    int *cnt;                  // [B]
    int *overflow;             // [1]: set when a sample selected more than cap anchors (the surplus is dropped)
    int *blk;                  // [B,nblk] workspace
};

__device__ __forceinline__ bool guided_flag(const GuidedArgs &P, int b, int a)
{
    if (a >= P.A) return false;
    const size_t ia = (size_t)b * P.A + a;
    if (P.mask && !P.mask[ia]) return false;
    float best = -INFINITY;
    for (int c = 0; c < P.NC; ++c) best = fmaxf(best, P.cls[ia * P.NC + c]);
    return 1.f / (1.f + expf(-best)) > P.thr;              // sigmoid is monotone: max of sigmoids = sigmoid of max
}

__global__ void guided_iota_kernel(int64_t *sel, int cap, int n_anchors)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < cap) sel[(size_t)blockIdx.y * cap + p] = p < n_anchors ? p : 0;
}

__global__ void __launch_bounds__(256) guided_count_kernel(GuidedArgs P)
{
    __shared__ int wc[4];
    const int b = blockIdx.y;
    const bool f = guided_flag(P, b, blockIdx.x * 256 + threadIdx.x);
    const unsigned long long m = __ballot(f);
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) P.blk[b * P.nblk + blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

__global__ void __launch_bounds__(256) guided_emit_kernel(GuidedArgs P)
{
    __shared__ int red[4];
    __shared__ int wc[4];
    const int b = blockIdx.y;
    int part = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) part += P.blk[b * P.nblk + i];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    const int a = blockIdx.x * 256 + threadIdx.x;
    const bool f = guided_flag(P, b, a);
    const unsigned long long m = __ballot(f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wc[wave] = __popcll(m);
    __syncthreads();
    const int base = red[0] + red[1] + red[2] + red[3];
    int off = base;
    for (int w = 0; w < wave; ++w) off += wc[w];
    const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
    if (f && pos < P.cap) P.sel[(size_t)b * P.cap + pos] = a;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const int total = base + wc[0] + wc[1] + wc[2] + wc[3];
        P.cnt[b] = min(total, P.cap);
        if (total > P.cap) atomicExch(P.overflow, 1);
    }
}
}  // namespace

extern "C" size_t sassd_assign_targets_workspace_bytes(int batch, int n_anchors, int total_gt)
{
    if (batch < 1 || n_anchors < 0 || total_gt < 0) return 0;
    return align_up((size_t)batch * n_anchors * 4, 256) * 2 + align_up((size_t)(total_gt + 1) * 4, 256);
}

extern "C" int sassd_assign_targets(const float *anchors, int anchors_per_sample, const uint8_t *anchor_mask,
                                    int n_anchors, int batch, const float *gt_boxes, const int64_t *gt_classes,
                                    const uint8_t *gt_ok, const int32_t *gt_offsets, int total_gt,
                                    const float *overlaps, const int64_t *overlap_offsets, float matched_threshold,
                                    float unmatched_threshold, int64_t *labels, float *targets, float *best_overlap,
                                    size_t out_sample_stride, int32_t *num_pos, int zero_num_pos, void *workspace,
                                    size_t workspace_bytes, void *stream_)
{
    if (!anchors || !gt_offsets || !labels || !targets || !num_pos || !workspace || batch < 1 || batch > 65535 ||
        n_anchors < 1 || total_gt < 0 || (total_gt > 0 && !gt_boxes) || (overlaps && !overlap_offsets) ||
        out_sample_stride < (size_t)n_anchors)
        return SASSD_EINVAL;
    if (workspace_bytes < sassd_assign_targets_workspace_bytes(batch, n_anchors, total_gt)) return SASSD_ENOSPC;
    hipStream_t s = (hipStream_t)stream_;
    const size_t per = align_up((size_t)batch * n_anchors * 4, 256);
    AssignArgs P;
    P.anchors = anchors; P.anchor_stride = anchors_per_sample ? (size_t)n_anchors * 7 : 0;
    P.anchor_mask = anchor_mask; P.gt = gt_boxes; P.gt_cls = gt_classes; P.gt_ok = gt_ok; P.gt_off = gt_offsets;
    P.ov = overlaps; P.ov_off = overlap_offsets;
    P.A = n_anchors; P.B = batch; P.out_stride = out_sample_stride;
    P.matched = matched_threshold; P.unmatched = unmatched_threshold;
    P.a_best = (float *)workspace; P.a_arg = (int *)((char *)workspace + per); P.g2a = (int *)((char *)workspace + 2 * per);
    P.labels = labels; P.targets = targets; P.best_out = best_overlap; P.npos = num_pos;
    int rc;
    if (zero_num_pos && (rc = sassd_hip(hipMemsetAsync(num_pos, 0, (size_t)batch * 4, s)))) return rc;
    if (total_gt > 0)
        hipLaunchKernelGGL(fill_int_kernel, dim3(cdiv(total_gt, 256)), dim3(256), 0, s, P.g2a, total_gt,
                           0xBF800000 /* bits of -1.0f */);
    const dim3 grid(cdiv(n_anchors, 256), batch);
    if (total_gt > 0) hipLaunchKernelGGL(assign_pass1_kernel, grid, dim3(256), 0, s, P);
    hipLaunchKernelGGL(assign_pass2_kernel, grid, dim3(256), 0, s, P);
    return sassd_launch_status();
}

extern "C" size_t sassd_rpn_loss_workspace_bytes(int batch, int n_anchors)
{
    return batch < 1 || n_anchors < 1 ? 0 : (size_t)batch * cdiv(n_anchors, 256) * 3 * sizeof(float);
}

extern "C" int sassd_rpn_loss(const float *box_preds, const float *cls_preds, const float *dir_preds, int num_class,
                              const int64_t *labels, const float *targets, const float *anchors,
                              int anchors_per_sample, const int32_t *num_pos, int n_anchors, int batch,
                              float *grad_box, float *grad_cls, float *grad_dir, float *loss_sums, void *workspace,
                              size_t workspace_bytes, void *stream_)
{
    if (!box_preds || !cls_preds || !labels || !targets || !anchors || !num_pos || !grad_box || !grad_cls ||
        !loss_sums || !workspace || (dir_preds && !grad_dir) || num_class < 1 || batch < 1 || batch > 65535 ||
        n_anchors < 1)
        return SASSD_EINVAL;
    if (workspace_bytes < sassd_rpn_loss_workspace_bytes(batch, n_anchors)) return SASSD_ENOSPC;
    RpnLossArgs P;
    P.box = box_preds; P.cls = cls_preds; P.dir = dir_preds; P.labels = labels; P.targets = targets;
    P.anchors = anchors; P.anchor_stride = anchors_per_sample ? (size_t)n_anchors * 7 : 0; P.npos = num_pos;
    P.A = n_anchors; P.B = batch; P.NC = num_class;
    P.gbox = grad_box; P.gcls = grad_cls; P.gdir = grad_dir; P.part = (float *)workspace;
    hipStream_t s = (hipStream_t)stream_;
    const dim3 grid(cdiv(n_anchors, 256), batch);
    hipLaunchKernelGGL(rpn_loss_kernel, grid, dim3(256), 0, s, P);
    hipLaunchKernelGGL(rpn_loss_sum_kernel, dim3(1), dim3(256), 0, s, (const float *)workspace,
                       (int)(grid.x * grid.y), loss_sums);
    return sassd_launch_status();
}

extern "C" size_t sassd_guided_select_workspace_bytes(int batch, int n_anchors)
{
    return batch < 1 || n_anchors < 1 ? 0 : (size_t)batch * cdiv(n_anchors, 256) * sizeof(int);
}

extern "C" int sassd_guided_select(const float *cls_preds, const uint8_t *anchor_mask, int n_anchors, int batch,
                                   int num_class, float score_thr, int cap, int64_t *sel, int32_t *counts,
                                   int32_t *overflow, void *workspace, size_t workspace_bytes, void *stream_)
{
    if (!cls_preds || !sel || !counts || !overflow || !workspace || n_anchors < 1 || batch < 1 || batch > 65535 ||
        num_class < 1 || cap < 1)
        return SASSD_EINVAL;
    if (workspace_bytes < sassd_guided_select_workspace_bytes(batch, n_anchors)) return SASSD_ENOSPC;
    hipStream_t s = (hipStream_t)stream_;
    GuidedArgs P;
    P.cls = cls_preds; P.mask = anchor_mask; P.A = n_anchors; P.B = batch; P.NC = num_class; P.cap = cap;
    P.nblk = cdiv(n_anchors, 256); P.thr = score_thr; P.sel = sel; P.cnt = counts; P.overflow = overflow;
    P.blk = (int *)workspace;
    // padding entries: sel[b][p] = p (valid, DISTINCT anchor indices -- the backward of the gather that follows is a
    // scatter-add, and thousands of padding rows aimed at one index would serialise on its atomics)
    hipLaunchKernelGGL(guided_iota_kernel, dim3(cdiv(cap, 256), batch), dim3(256), 0, s, sel, cap, n_anchors);
    const dim3 grid(P.nblk, batch);
    hipLaunchKernelGGL(guided_count_kernel, grid, dim3(256), 0, s, P);
    hipLaunchKernelGGL(guided_emit_kernel, grid, dim3(256), 0, s, P);
    return sassd_launch_status();
}

