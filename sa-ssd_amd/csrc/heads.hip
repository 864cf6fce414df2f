// heads.hip -- detection-head post-processing without host round trips:
//   decode + score filter + ordered compaction   (ssd_rotate_head.py:307-372, :53-91)
//   part-sensitive warping sample                (ssd_rotate_head.py:374-414, 438-442)
//   rescoring + rotated NMS                      (ssd_rotate_head.py:487-533, iou3d_utils.py:47-60,114-128,
//                                                 iou3d_kernel.cu:250-292, iou3d.cpp:100-116)
//   iou3d_cuda operator surface                  (iou3d.cpp:31,52,73)
// The reference runs ~30 tiny torch kernels with >= 6 host syncs per sample (boolean indexing, .numel(), D2H of
// the NMS mask and a serial CPU loop).  Here every data-dependent count stays in device memory; ordering of the
// surviving boxes (ascending anchor index, then stable descending score) is reproduced with block scans and a
// rank sort; the NMS suppression word of a (row, 64-column block) is one wave64 ballot.
#include "common.h"
#include "iou3d_device.h"

namespace {

constexpr float kPi = 3.14159265358979323846f;      // float(np.pi)

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------------------
// decode + filter
// ------------------------------------------------------------------------------------------------
struct DfParams {
    const float *box, *cls, *dir, *anchors;
    const uint8_t *mask;
    size_t bs;                  // batch stride (floats) of box / cls / dir
    int B, ncls, A, HW, Atot, capK;
    float thr;
};

constexpr int kDfPerThread = 4;
constexpr int kDfPerBlock = 256 * kDfPerThread;

// returns selected?; *score = max-class sigmoid, *label = argmax class
__device__ __forceinline__ bool df_select(const DfParams &P, int b, int r, float *score, int *label)
{
    if (r >= P.Atot) return false;
    if (!P.mask[(size_t)b * P.Atot + r]) return false;
    const int per_cls = P.HW * P.A;
    const int ci = r / per_cls, rem = r - ci * per_cls;
    const int pix = rem / P.A, a = rem - pix * P.A;
    const float *c = P.cls + (size_t)b * P.bs + (size_t)(ci * P.A * P.ncls + a * P.ncls) * P.HW + pix;
    float best = sigmoidf(c[0]);
    int bl = 0;
    for (int j = 1; j < P.ncls; ++j) {
        const float s = sigmoidf(c[(size_t)j * P.HW]);
        if (s > best) { best = s; bl = j; }
    }
    *score = best;
    *label = bl;
    return best > P.thr;
}

__global__ void __launch_bounds__(256) df_count_kernel(DfParams P, int nblk, int *__restrict__ bsum)
{
    __shared__ int wsum[17];
    const int b = blockIdx.y;
    const int base = blockIdx.x * kDfPerBlock + threadIdx.x * kDfPerThread;
    int s = 0;
    float sc; int lb;
#pragma unroll
    for (int k = 0; k < kDfPerThread; ++k) s += df_select(P, b, base + k, &sc, &lb) ? 1 : 0;
    int tot;
    block_exclusive_scan(s, wsum, &tot);
    if (threadIdx.x == 0) bsum[b * nblk + blockIdx.x] = tot;
}

// every block sums the counts of the blocks before it (a few hundred ints: no separate single-workgroup scan launch);
// the last block publishes the sample's candidate count
__global__ void __launch_bounds__(256) df_emit_kernel(DfParams P, int nblk, const int *__restrict__ bsum,
                                                      float *__restrict__ guided, int32_t *__restrict__ labels,
                                                      float *__restrict__ scores, int32_t *__restrict__ counts,
                                                      int32_t *status)
{
    __shared__ int wsum[17];
    const int b = blockIdx.y;
    int part = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) part += bsum[b * nblk + j];
    int before;
    block_exclusive_scan(part, wsum, &before);
    const int base = blockIdx.x * kDfPerBlock + threadIdx.x * kDfPerThread;
    bool sel[kDfPerThread];
    float sc[kDfPerThread];
    int lb[kDfPerThread];
    int s = 0;
#pragma unroll
    for (int k = 0; k < kDfPerThread; ++k) { sel[k] = df_select(P, b, base + k, &sc[k], &lb[k]); s += sel[k] ? 1 : 0; }
    int tot;
    int o = before + block_exclusive_scan(s, wsum, &tot);
    if ((int)blockIdx.x == nblk - 1 && threadIdx.x == 0) {
        int total = before + tot;
        if (total > P.capK) { if (status) atomicOr(status, SASSD_ST_BOX_OVERFLOW); total = P.capK; }
        counts[b] = total;
    }
#pragma unroll
    for (int k = 0; k < kDfPerThread; ++k) {
        if (!sel[k]) continue;
        const int slot = o++;
        if (slot >= P.capK) continue;
        const int r = base + k;
        const int per_cls = P.HW * P.A;
        const int ci = r / per_cls, rem = r - ci * per_cls;
        const int pix = rem / P.A, a = rem - pix * P.A;
        const float *bx = P.box + (size_t)b * P.bs + (size_t)(ci * P.A * 7 + a * 7) * P.HW + pix;
        const float *dr = P.dir + (size_t)b * P.bs + (size_t)(ci * P.A * 2 + a * 2) * P.HW + pix;
        const float *an = P.anchors + (size_t)r * 7;
        const float xt = bx[0], yt = bx[(size_t)P.HW], zt = bx[(size_t)2 * P.HW], wt = bx[(size_t)3 * P.HW],
                    lt = bx[(size_t)4 * P.HW], ht = bx[(size_t)5 * P.HW], rt = bx[(size_t)6 * P.HW];
        const float xa = an[0], ya = an[1], wa = an[3], la = an[4], ha = an[5], ra = an[6];
        const float za = an[2] + ha / 2;
        const float diag = sqrtf(la * la + wa * wa);
        const float xg = xt * diag + xa;
        const float yg = yt * diag + ya;
        float zg = zt * ha + za;
        const float lg = expf(lt) * la;
        const float wg = expf(wt) * wa;
        const float hg = expf(ht) * ha;
        float rg = rt + ra;
        zg = zg - hg / 2;
        const int dl = dr[(size_t)P.HW] > dr[0] ? 1 : 0;          // torch.max(...)[1]: first max wins ties
        if ((rg > 0.f) != (dl != 0)) rg += kPi;
        float *g = guided + ((size_t)b * P.capK + slot) * 7;
        g[0] = xg; g[1] = yg; g[2] = zg; g[3] = wg; g[4] = lg; g[5] = hg; g[6] = rg;
        labels[(size_t)b * P.capK + slot] = lb[k];
        if (scores) scores[(size_t)b * P.capK + slot] = sc[k];
    }
}

// ------------------------------------------------------------------------------------------------
// PSWarp sampling: 32 lanes per box (28 grid points), channel k sampled at grid point k
// ------------------------------------------------------------------------------------------------
struct WarpParams {
    const float *feat, *guided;
    const int32_t *counts;
    float *logits;
    int B, H, W, capK;
    float offx, offy, scale;
    float lin4[4], lin7[7];
};

__global__ void __launch_bounds__(256) pswarp_kernel(WarpParams P)
{
    const int b = blockIdx.y;
    const int K = min(P.counts[b], P.capK);
    const int box = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int k = threadIdx.x & 31;
    float val = 0.f;
    if (box < K && k < 28) {
        const float *g = P.guided + ((size_t)b * P.capK + box) * 7;
        const float xg = g[0], yg = g[1], wg = g[3], lg = g[4], rg = g[6];
        const float ct = cosf(rg), st = sinf(rg);
        const int ix = k / 7, iy = k - ix * 7;
        const float xx = P.lin4[ix] * wg, yy = P.lin7[iy] * lg;
        const float x = xx * ct + yy * st + xg;
        const float y = yy * ct - xx * st + yg;
        const float u = (x + P.offx) * P.scale, v = (y + P.offy) * P.scale;
        // normalise exactly like bilinear_interpolate_torch_gridsample, then grid_sample(align_corners=True)
        const float gx = u / (float)(P.W - 1) * 2.f - 1.f, gy = v / (float)(P.H - 1) * 2.f - 1.f;
        float fx = (gx + 1.f) / 2.f * (float)(P.W - 1), fy = (gy + 1.f) / 2.f * (float)(P.H - 1);
        // keep the float -> int conversion and the +1 in range (int overflow is UB and hipcc exploits it in the
        // bounds tests); samples outside [-1, size] contribute zero either way (zero padding), NaN -> outside
        fx = fminf(fmaxf(fx, -2.f), (float)P.W + 1.f);
        fy = fminf(fmaxf(fy, -2.f), (float)P.H + 1.f);
        const float x0f = floorf(fx), y0f = floorf(fy);
        const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
        const float wx1 = fx - x0f, wx0 = (x0f + 1.f) - fx, wy1 = fy - y0f, wy0 = (y0f + 1.f) - fy;
        const float *im = P.feat + ((size_t)b * 28 + k) * P.H * P.W;
        const bool xin0 = x0 >= 0 && x0 < P.W, xin1 = x1 >= 0 && x1 < P.W;
        const bool yin0 = y0 >= 0 && y0 < P.H, yin1 = y1 >= 0 && y1 < P.H;
        if (yin0 && xin0) val += im[(size_t)y0 * P.W + x0] * (wx0 * wy0);
        if (yin0 && xin1) val += im[(size_t)y0 * P.W + x1] * (wx1 * wy0);
        if (yin1 && xin0) val += im[(size_t)y1 * P.W + x0] * (wx0 * wy1);
        if (yin1 && xin1) val += im[(size_t)y1 * P.W + x1] * (wx1 * wy1);
    }
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) val += __shfl_xor(val, o, 32);
    if (box < K && k == 0) P.logits[(size_t)b * P.capK + box] = val / 28.f;
}

// backward of pswarp_kernel (training): d logits -> d feat (atomicAdd into [B,28,H,W]) and d guided [B,capK,7]
// (the reference differentiates grid_sample w.r.t. input AND grid, so the rescoring loss also steers x,y,w,l,r)
__global__ void __launch_bounds__(256) pswarp_bwd_kernel(WarpParams P, const float *__restrict__ dlogits,
                                                         float *__restrict__ dfeat, float *__restrict__ dguided)
{
    const int b = blockIdx.y;
    const int K = min(P.counts[b], P.capK);
    const int box = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int k = threadIdx.x & 31;
    float gxg = 0.f, gyg = 0.f, gw = 0.f, gl = 0.f, gr = 0.f;
    if (box < K && k < 28) {
        const float *g = P.guided + ((size_t)b * P.capK + box) * 7;
        const float xg = g[0], yg = g[1], wg = g[3], lg = g[4], rg = g[6];
        const float ct = cosf(rg), st = sinf(rg);
        const int ix = k / 7, iy = k - ix * 7;
        const float xx = P.lin4[ix] * wg, yy = P.lin7[iy] * lg;
        const float x = xx * ct + yy * st + xg;
        const float y = yy * ct - xx * st + yg;
        const float u = (x + P.offx) * P.scale, v = (y + P.offy) * P.scale;
        const float gx = u / (float)(P.W - 1) * 2.f - 1.f, gy = v / (float)(P.H - 1) * 2.f - 1.f;
        float fx = (gx + 1.f) / 2.f * (float)(P.W - 1), fy = (gy + 1.f) / 2.f * (float)(P.H - 1);
        const bool inr = fx > -1.f && fx < (float)P.W && fy > -1.f && fy < (float)P.H;   // else: all taps padded
        fx = fminf(fmaxf(fx, -2.f), (float)P.W + 1.f);
        fy = fminf(fmaxf(fy, -2.f), (float)P.H + 1.f);
        const float x0f = floorf(fx), y0f = floorf(fy);
        const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
        const float wx1 = fx - x0f, wx0 = (x0f + 1.f) - fx, wy1 = fy - y0f, wy0 = (y0f + 1.f) - fy;
        const float go = dlogits[(size_t)b * P.capK + box] / 28.f;
        const size_t plane = ((size_t)b * 28 + k) * P.H * P.W;
        const float *im = P.feat + plane;
        float *di = dfeat + plane;
        const bool xin0 = x0 >= 0 && x0 < P.W, xin1 = x1 >= 0 && x1 < P.W;
        const bool yin0 = y0 >= 0 && y0 < P.H, yin1 = y1 >= 0 && y1 < P.H;
        float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
        if (inr) {
            if (yin0 && xin0) { v00 = im[(size_t)y0 * P.W + x0]; atomicAdd(&di[(size_t)y0 * P.W + x0], go * wx0 * wy0); }
            if (yin0 && xin1) { v01 = im[(size_t)y0 * P.W + x1]; atomicAdd(&di[(size_t)y0 * P.W + x1], go * wx1 * wy0); }
            if (yin1 && xin0) { v10 = im[(size_t)y1 * P.W + x0]; atomicAdd(&di[(size_t)y1 * P.W + x0], go * wx0 * wy1); }
            if (yin1 && xin1) { v11 = im[(size_t)y1 * P.W + x1]; atomicAdd(&di[(size_t)y1 * P.W + x1], go * wx1 * wy1); }
        }
        // d sample / d fx, d fy ; fx = u, fy = v (the normalise / un-normalise pair is the identity)
        const float dfx = ((v01 - v00) * wy0 + (v11 - v10) * wy1) * go;
        const float dfy = ((v10 - v00) * wx0 + (v11 - v01) * wx1) * go;
        const float dx = dfx * P.scale, dy = dfy * P.scale;
        gxg = dx; gyg = dy;
        gw = (dx * ct - dy * st) * P.lin4[ix];
        gl = (dx * st + dy * ct) * P.lin7[iy];
        gr = dx * (-xx * st + yy * ct) + dy * (-yy * st - xx * ct);
    }
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) {
        gxg += __shfl_xor(gxg, o, 32); gyg += __shfl_xor(gyg, o, 32); gw += __shfl_xor(gw, o, 32);
        gl += __shfl_xor(gl, o, 32); gr += __shfl_xor(gr, o, 32);
    }
    if (box < K && k == 0 && dguided) {
        float *d = dguided + ((size_t)b * P.capK + box) * 7;
        d[0] = gxg; d[1] = gyg; d[2] = 0.f; d[3] = gw; d[4] = gl; d[5] = 0.f; d[6] = gr;
    }
}

// ------------------------------------------------------------------------------------------------
// rotated NMS building blocks
// ------------------------------------------------------------------------------------------------
// one wave computes the suppression word of (row, 64-column block cb): bit j = iou(row, cb*64+j) > thr, col > row
__device__ __forceinline__ unsigned long long nms_word(const float *__restrict__ boxes, int n, int row, int cb,
                                                       float thr, int lane)
{
    const int col = cb * 64 + lane;
    bool hit = false;
    if (col < n && col > row) {
        const iou3d::Box a = iou3d::load_box(boxes + (size_t)row * 5);
        const iou3d::Box b = iou3d::load_box(boxes + (size_t)col * 5);
        // boxes whose circumscribed circles are disjoint have overlap exactly 0 (no intersection points, no
        // contained corners) -> iou 0 > thr is false for any thr >= 0: skip the polygon clipping
        const float dx = (a.x1 + a.x2) * 0.5f - (b.x1 + b.x2) * 0.5f, dy = (a.y1 + a.y2) * 0.5f - (b.y1 + b.y2) * 0.5f;
        const float wa = a.x2 - a.x1, la = a.y2 - a.y1, wb = b.x2 - b.x1, lb = b.y2 - b.y1;
        const float rr = 0.5f * (sqrtf(wa * wa + la * la) + sqrtf(wb * wb + lb * lb)) + 1e-3f;
        if (thr < 0.f || dx * dx + dy * dy <= rr * rr) hit = iou3d::iou_bev(a, b) > thr;
    }
    return __ballot(hit);
}

// greedy reduce by ONE wave: lane l keeps remv words l, l+64, ... (n <= 64*64*WPL boxes)
template <int WPL>
__device__ __forceinline__ int nms_greedy(const unsigned long long *mask, int n, int ncb, int lane, int64_t *keep64,
                                          int32_t *keep32, int cap_keep)
{
    unsigned long long remv[WPL];
#pragma unroll
    for (int w = 0; w < WPL; ++w) remv[w] = 0ull;
    int nk = 0;
    for (int i = 0; i < n; ++i) {
        const int nb = i >> 6;
        unsigned long long word = 0ull;
#pragma unroll
        for (int w = 0; w < WPL; ++w)
            if ((nb >> 6) == w) word = __shfl(remv[w], nb & 63, 64);
        if (!((word >> (i & 63)) & 1ull)) {
            if (nk < cap_keep && lane == 0) { if (keep64) keep64[nk] = i; if (keep32) keep32[nk] = i; }
            ++nk;
#pragma unroll
            for (int w = 0; w < WPL; ++w) {
                const int cb = w * 64 + lane;
                if (cb >= nb && cb < ncb) remv[w] |= mask[(size_t)i * ncb + cb];
            }
        }
    }
    return nk;
}

// Greedy reduce, 64 rows at a time, by ONE wave (n <= 4096 boxes: lane l owns the removal word of column block l).
//   * the 64x64 diagonal block is resolved on the scalar unit: lane i holds row i's diagonal word, the running
//     removal word lives in SGPRs and is updated through v_readlane -- no memory access in the serial chain;
//   * the kept rows' remaining words are then OR-ed into the later column blocks with independent, pipelined
//     loads (one 8-byte word per lane per kept row).
// Identical result to the serial host loop of iou3d.cpp:100-116.
__device__ __forceinline__ int nms_greedy_chunked(const unsigned long long *__restrict__ mask, int n, int ncb, int lane,
                                                  int *__restrict__ keep, int cap_keep)
{
    unsigned long long remv = 0ull;                  // lane l: removal bits of boxes 64l .. 64l+63
    int nk = 0;
    for (int cb0 = 0; cb0 < ncb; ++cb0) {
        const int r = cb0 * 64 + lane;
        const unsigned long long dw = (r < n) ? mask[(size_t)r * ncb + cb0] : 0ull;
        const unsigned dlo = (unsigned)dw, dhi = (unsigned)(dw >> 32);
        unsigned long long rem = __shfl(remv, cb0, 64);
        rem = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(rem >> 32)) << 32) |
              (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rem);
        const int rows = min(64, n - cb0 * 64);
        unsigned long long kept = 0ull;
        for (int i = 0; i < rows; ++i) {
            if (!((rem >> i) & 1ull)) {
                kept |= 1ull << i;
                rem |= ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)dhi, i) << 32) |
                       (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)dlo, i);
            }
        }
        // record kept rows in order
        if ((kept >> lane) & 1ull) {
            const int pos = nk + __popcll(kept & ((1ull << lane) - 1ull));
            if (pos < cap_keep) keep[pos] = r;
        }
        nk += __popcll(kept);
        // propagate to later column blocks
        unsigned long long acc = 0ull;
        unsigned long long km = kept;
        while (km) {
            const int i = __ffsll((long long)km) - 1;
            km &= km - 1ull;
            if (lane > cb0 && lane < ncb) acc |= mask[(size_t)(cb0 * 64 + i) * ncb + lane];
        }
        remv |= acc;
    }
    return nk;
}

// axis-aligned IoU of (x1,y1,x2,y2,ry) boxes, the rotation ignored (iou3d_kernel.cu:295-303 `iou_normal`)
__device__ __forceinline__ unsigned long long nms_word_normal(const float *__restrict__ boxes, int n, int row, int cb,
                                                              float thr, int lane)
{
    const int col = cb * 64 + lane;
    bool hit = false;
    if (col < n && col > row) {
        const float *a = boxes + (size_t)row * 5, *b = boxes + (size_t)col * 5;
        const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
        const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
        const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
        const float inter = width * height;
        const float sa = (a[2] - a[0]) * (a[3] - a[1]), sb = (b[2] - b[0]) * (b[3] - b[1]);
        hit = inter / fmaxf(sa + sb - inter, iou3d::kEps) > thr;
    }
    return __ballot(hit);
}

__global__ void __launch_bounds__(256) nms_mask_kernel(const float *__restrict__ boxes, int n, int ncb, float thr,
                                                       unsigned long long *__restrict__ mask, int normal)
{
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n * ncb) return;
    const int row = item / ncb, cb = item - row * ncb;
    unsigned long long w = 0ull;
    if (cb >= (row >> 6)) w = normal ? nms_word_normal(boxes, n, row, cb, thr, lane) : nms_word(boxes, n, row, cb, thr, lane);
    if (lane == 0) mask[item] = w;
}

__global__ void __launch_bounds__(64) nms_greedy_kernel(const unsigned long long *__restrict__ mask, int n, int ncb,
                                                        int64_t *__restrict__ keep, int32_t *__restrict__ num_keep)
{
    const int lane = threadIdx.x;
    int nk;
    if (ncb <= 64) nk = nms_greedy<1>(mask, n, ncb, lane, keep, nullptr, n);
    else nk = nms_greedy<4>(mask, n, ncb, lane, keep, nullptr, n);
    if (lane == 0) *num_keep = nk;
}

// ------------------------------------------------------------------------------------------------
// fused rescore + NMS: one workgroup (16 waves) per sample
// ------------------------------------------------------------------------------------------------
struct RnParams {
    const float *guided, *logits;
    const int32_t *labels, *counts;
    float *out_boxes, *out_scores;
    int32_t *out_labels, *out_counts, *status;
    int capK, capD;
    float score_thr, iou_thr;
    // workspace (per sample strides)
    int *cand;                    // [B][capK]   original index of candidate
    float *cscore;                // [B][capK]
    int *order;                   // [B][capK]   order[rank] = candidate
    float *sboxes;                // [B][capK][5] BEV boxes in sorted order
    unsigned long long *mask;     // [B][capK][ncb_cap]
    int *keep;                    // [B][capK]
    int ncb_cap;
};

constexpr int kRnLdsScores = 4096;

// stage A (one workgroup per sample): sigmoid + threshold (ordered compaction), stable rank sort, sorted BEV boxes
__global__ void __launch_bounds__(1024) rescore_prep_kernel(RnParams P, int *__restrict__ mcount)
{
    __shared__ int wsum[17];
    __shared__ float s_score[kRnLdsScores];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int K = min(P.counts[b], P.capK);
    int *cand = P.cand + (size_t)b * P.capK;
    float *cscore = P.cscore + (size_t)b * P.capK;
    int *order = P.order + (size_t)b * P.capK;
    float *sb = P.sboxes + (size_t)b * P.capK * 5;
    const float *guided = P.guided + (size_t)b * P.capK * 7;
    int running = 0;
    for (int i0 = 0; i0 < K; i0 += 1024) {
        const int i = i0 + tid;
        float s = 0.f;
        bool sel = false;
        if (i < K) { s = sigmoidf(P.logits[(size_t)b * P.capK + i]); sel = s > P.score_thr; }
        int tot;
        const int ex = block_exclusive_scan(sel ? 1 : 0, wsum, &tot);
        if (sel) { cand[running + ex] = i; cscore[running + ex] = s; if (running + ex < kRnLdsScores) s_score[running + ex] = s; }
        running += tot;
    }
    const int M = running;
    if (tid == 0) mcount[b] = M;
    __syncthreads();
    if (M == 0) return;
    // stable descending rank sort: rank_i = #{j : s_j > s_i or (s_j == s_i and j < i)}
    for (int i = tid; i < M; i += 1024) {
        const float si = cscore[i];
        int rank = 0;
        if (M <= kRnLdsScores) {
#pragma unroll 8
            for (int j = 0; j < M; ++j) {
                const float sj = s_score[j];
                rank += (sj > si || (sj == si && j < i)) ? 1 : 0;
            }
        } else {
            for (int j = 0; j < M; ++j) {
                const float sj = cscore[j];
                rank += (sj > si || (sj == si && j < i)) ? 1 : 0;
            }
        }
        order[rank] = i;
    }
    __syncthreads();
    // BEV boxes (x - w/2, y - l/2, x + w/2, y + l/2, r) in sorted order  (iou3d_utils.py:47-60)
    for (int r = tid; r < M; r += 1024) {
        const float *g = guided + (size_t)cand[order[r]] * 7;
        const float hx = g[3] / 2, hy = g[4] / 2;
        sb[r * 5 + 0] = g[0] - hx; sb[r * 5 + 1] = g[1] - hy;
        sb[r * 5 + 2] = g[0] + hx; sb[r * 5 + 3] = g[1] + hy;
        sb[r * 5 + 4] = g[6];
    }
}

// stage B (many workgroups, grid-stride over (row, column block) items of the upper triangle)
__global__ void __launch_bounds__(256) rescore_mask_kernel(RnParams P, const int *__restrict__ mcount)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int M = mcount[b];
    const int ncb = (M + 63) >> 6;
    const float *sb = P.sboxes + (size_t)b * P.capK * 5;
    unsigned long long *mask = P.mask + (size_t)b * P.capK * P.ncb_cap;
    const int nwaves = gridDim.x * 4;
    for (int item = blockIdx.x * 4 + (threadIdx.x >> 6); item < M * ncb; item += nwaves) {
        const int row = item / ncb, cb = item - row * ncb;
        if (cb < (row >> 6)) continue;
        const unsigned long long w = nms_word(sb, M, row, cb, P.iou_thr, lane);
        if (lane == 0) mask[(size_t)row * ncb + cb] = w;
    }
}

// stage C (one workgroup per sample): greedy reduce by wave 0, then gather
__global__ void __launch_bounds__(256) rescore_final_kernel(RnParams P, const int *__restrict__ mcount)
{
    __shared__ int s_nk;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = mcount[b];
    if (M == 0) { if (tid == 0) P.out_counts[b] = 0; return; }
    const int ncb = (M + 63) >> 6;
    const int *cand = P.cand + (size_t)b * P.capK;
    const float *cscore = P.cscore + (size_t)b * P.capK;
    const int *order = P.order + (size_t)b * P.capK;
    const unsigned long long *mask = P.mask + (size_t)b * P.capK * P.ncb_cap;
    int *keep = P.keep + (size_t)b * P.capK;
    const float *guided = P.guided + (size_t)b * P.capK * 7;
    if (wave == 0) {
        const int nk = nms_greedy_chunked(mask, M, ncb, lane, keep, P.capK);
        if (lane == 0) s_nk = nk;
    }
    __syncthreads();
    int nk = s_nk;
    if (nk > P.capD) { if (tid == 0 && P.status) atomicOr(P.status, SASSD_ST_BOX_OVERFLOW); nk = P.capD; }
    for (int t = tid; t < nk; t += 256) {
        const int c = order[keep[t]];
        const int src = cand[c];
        float *ob = P.out_boxes + ((size_t)b * P.capD + t) * 7;
#pragma unroll
        for (int q = 0; q < 7; ++q) ob[q] = guided[(size_t)src * 7 + q];
        P.out_scores[(size_t)b * P.capD + t] = cscore[c];
        P.out_labels[(size_t)b * P.capD + t] = P.labels[(size_t)b * P.capK + src];
    }
    if (tid == 0) P.out_counts[b] = nk;
}

// ------------------------------------------------------------------------------------------------
// pairwise matrices
// ------------------------------------------------------------------------------------------------
template <bool IOU>
__global__ void __launch_bounds__(256) pair_kernel(const float *__restrict__ a, int na, const float *__restrict__ b,
                                                   int nb, float *__restrict__ out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)na * nb) return;
    const int i = t / nb, j = t - (size_t)i * nb;
    const iou3d::Box A = iou3d::load_box(a + (size_t)i * 5), Bx = iou3d::load_box(b + (size_t)j * 5);
    out[t] = IOU ? iou3d::iou_bev(A, Bx) : iou3d::box_overlap(A, Bx);
}

struct RnLayout { size_t cand, cscore, order, sboxes, mask, keep, mcount, total; int ncb_cap; };
RnLayout rn_layout(int B, int capK)
{
    RnLayout L;
    L.ncb_cap = (capK + 63) / 64;
    size_t o = 0;
    L.cand = o;   o += align_up((size_t)B * capK * 4, 256);
    L.cscore = o; o += align_up((size_t)B * capK * 4, 256);
    L.order = o;  o += align_up((size_t)B * capK * 4, 256);
    L.sboxes = o; o += align_up((size_t)B * capK * 5 * 4, 256);
    L.mask = o;   o += align_up((size_t)B * capK * L.ncb_cap * 8, 256);
    L.keep = o;   o += align_up((size_t)B * capK * 4, 256);
    L.mcount = o; o += align_up((size_t)B * 4, 256);
    L.total = o;
    return L;
}

// torch.linspace(start, end, steps) fp32 values (symmetric evaluation, ATen RangeFactories)
void linspace_f32(float start, float end, int steps, float *out)
{
    const float step = (end - start) / (float)(steps - 1);
    const int half = steps / 2;
    for (int i = 0; i < steps; ++i) out[i] = (i < half) ? start + step * (float)i : end - step * (float)(steps - i - 1);
}

}  // namespace

extern "C" size_t sassd_decode_filter_workspace_bytes(int batch, int n_anchors_total)
{
    const int nblk = cdiv(n_anchors_total, kDfPerBlock);
    return 2 * align_up((size_t)batch * nblk * 4, 256);
}

extern "C" int sassd_decode_filter(const float *box, const float *cls, const float *dir, size_t batch_stride,
                                   int batch, int num_class, int anchors_per_loc, int H, int W,
                                   const float *anchors, const uint8_t *mask, float thr, float *guided,
                                   int32_t *labels, float *scores, int32_t *counts, int capK, int32_t *status,
                                   void *workspace, size_t workspace_bytes, void *stream_)
{
    if (!box || !cls || !dir || !anchors || !mask || !guided || !labels || !counts || !workspace) return SASSD_EINVAL;
    if (batch < 1 || num_class < 1 || anchors_per_loc < 1 || capK < 1) return SASSD_EINVAL;
    DfParams P;
    P.box = box; P.cls = cls; P.dir = dir; P.anchors = anchors; P.mask = mask; P.bs = batch_stride;
    P.B = batch; P.ncls = num_class; P.A = anchors_per_loc; P.HW = H * W; P.Atot = num_class * H * W * anchors_per_loc;
    P.capK = capK; P.thr = thr;
    const int nblk = cdiv(P.Atot, kDfPerBlock);
    if (workspace_bytes < sassd_decode_filter_workspace_bytes(batch, P.Atot)) return SASSD_ENOSPC;
    int *bsum = (int *)workspace;
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(df_count_kernel, dim3(nblk, batch), dim3(256), 0, stream, P, nblk, bsum);
    hipLaunchKernelGGL(df_emit_kernel, dim3(nblk, batch), dim3(256), 0, stream, P, nblk, (const int *)bsum, guided,
                       labels, scores, counts, status);
    return sassd_launch_status();
}

extern "C" int sassd_pswarp_sample(const float *feat, int batch, int H, int W, const float *guided,
                                   const int32_t *counts, int capK, float grid_off_x, float grid_off_y,
                                   float spatial_scale, float *logits, void *stream_)
{
    if (!feat || !guided || !counts || !logits || batch < 1 || capK < 1) return SASSD_EINVAL;
    WarpParams P;
    P.feat = feat; P.guided = guided; P.counts = counts; P.logits = logits;
    P.B = batch; P.H = H; P.W = W; P.capK = capK;
    P.offx = grid_off_x; P.offy = grid_off_y; P.scale = spatial_scale;
    linspace_f32(-0.5f, 0.5f, 4, P.lin4);
    linspace_f32(-0.5f, 0.5f, 7, P.lin7);
    hipLaunchKernelGGL(pswarp_kernel, dim3(cdiv(capK, 8), batch), dim3(256), 0, (hipStream_t)stream_, P);
    return sassd_launch_status();
}

extern "C" int sassd_pswarp_sample_bwd(const float *feat, int batch, int H, int W, const float *guided,
                                       const int32_t *counts, int capK, float grid_off_x, float grid_off_y,
                                       float spatial_scale, const float *dlogits, float *dfeat, float *dguided,
                                       void *stream_)
{
    if (!feat || !guided || !counts || !dlogits || !dfeat || batch < 1 || capK < 1) return SASSD_EINVAL;
    WarpParams P;
    P.feat = feat; P.guided = guided; P.counts = counts; P.logits = nullptr;
    P.B = batch; P.H = H; P.W = W; P.capK = capK;
    P.offx = grid_off_x; P.offy = grid_off_y; P.scale = spatial_scale;
    linspace_f32(-0.5f, 0.5f, 4, P.lin4);
    linspace_f32(-0.5f, 0.5f, 7, P.lin7);
    hipLaunchKernelGGL(pswarp_bwd_kernel, dim3(cdiv(capK, 8), batch), dim3(256), 0, (hipStream_t)stream_, P, dlogits,
                       dfeat, dguided);
    return sassd_launch_status();
}

extern "C" size_t sassd_rescore_nms_workspace_bytes(int batch, int capK) { return rn_layout(batch, capK).total; }

extern "C" int sassd_rescore_nms(const float *guided, const float *logits, const int32_t *labels,
                                 const int32_t *counts, int batch, int capK, float score_thr, float iou_thr,
                                 float *out_boxes, float *out_scores, int32_t *out_labels, int32_t *out_counts,
                                 int capD, int32_t *status, void *workspace, size_t workspace_bytes, void *stream_)
{
    if (!guided || !logits || !labels || !counts || !out_boxes || !out_scores || !out_labels || !out_counts ||
        !workspace)
        return SASSD_EINVAL;
    if (batch < 1 || capK < 1 || capK > 4096 || capD < 1) return SASSD_EINVAL;
    const RnLayout L = rn_layout(batch, capK);
    if (workspace_bytes < L.total) return SASSD_ENOSPC;
    char *w = (char *)workspace;
    RnParams P;
    P.guided = guided; P.logits = logits; P.labels = labels; P.counts = counts;
    P.out_boxes = out_boxes; P.out_scores = out_scores; P.out_labels = out_labels; P.out_counts = out_counts;
    P.status = status; P.capK = capK; P.capD = capD; P.score_thr = score_thr; P.iou_thr = iou_thr;
    P.cand = (int *)(w + L.cand); P.cscore = (float *)(w + L.cscore); P.order = (int *)(w + L.order);
    P.sboxes = (float *)(w + L.sboxes); P.mask = (unsigned long long *)(w + L.mask); P.keep = (int *)(w + L.keep);
    P.ncb_cap = L.ncb_cap;
    int *mcount = (int *)(w + L.mcount);
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(rescore_prep_kernel, dim3(batch), dim3(1024), 0, stream, P, mcount);
    hipLaunchKernelGGL(rescore_mask_kernel, dim3(512, batch), dim3(256), 0, stream, P, (const int *)mcount);
    hipLaunchKernelGGL(rescore_final_kernel, dim3(batch), dim3(256), 0, stream, P, (const int *)mcount);
    return sassd_launch_status();
}

extern "C" int sassd_boxes_overlap_bev(const float *boxes_a, int num_a, const float *boxes_b, int num_b,
                                       float *ans_overlap, void *stream_)
{
    if (!boxes_a || !boxes_b || !ans_overlap || num_a < 0 || num_b < 0) return SASSD_EINVAL;
    const size_t tot = (size_t)num_a * num_b;
    if (tot == 0) return SASSD_OK;
    hipLaunchKernelGGL(pair_kernel<false>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                       boxes_a, num_a, boxes_b, num_b, ans_overlap);
    return sassd_launch_status();
}

extern "C" int sassd_boxes_iou_bev(const float *boxes_a, int num_a, const float *boxes_b, int num_b, float *ans_iou,
                                   void *stream_)
{
    if (!boxes_a || !boxes_b || !ans_iou || num_a < 0 || num_b < 0) return SASSD_EINVAL;
    const size_t tot = (size_t)num_a * num_b;
    if (tot == 0) return SASSD_OK;
    hipLaunchKernelGGL(pair_kernel<true>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                       boxes_a, num_a, boxes_b, num_b, ans_iou);
    return sassd_launch_status();
}

extern "C" size_t sassd_nms_workspace_bytes(int n)
{
    const size_t ncb = (size_t)(n + 63) / 64;
    return align_up((size_t)(n > 0 ? n : 1) * (ncb ? ncb : 1) * 8, 256);
}

static int nms_impl(const float *boxes, int n, float thresh, int64_t *keep, int32_t *num_keep, void *workspace,
                    size_t workspace_bytes, void *stream_, int normal)
{
    if (!keep || !num_keep || n < 0 || n > 16384) return SASSD_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    if (n == 0) return sassd_hip(hipMemsetAsync(num_keep, 0, 4, stream));
    if (!boxes || !workspace) return SASSD_EINVAL;
    if (workspace_bytes < sassd_nms_workspace_bytes(n)) return SASSD_ENOSPC;
    const int ncb = (n + 63) / 64;
    unsigned long long *mask = (unsigned long long *)workspace;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(cdiv(n * ncb, 4)), dim3(256), 0, stream, boxes, n, ncb, thresh, mask, normal);
    hipLaunchKernelGGL(nms_greedy_kernel, dim3(1), dim3(64), 0, stream, (const unsigned long long *)mask, n, ncb, keep,
                       num_keep);
    return sassd_launch_status();
}

extern "C" int sassd_nms_gpu(const float *boxes, int n, float thresh, int64_t *keep, int32_t *num_keep,
                             void *workspace, size_t workspace_bytes, void *stream)
{
    return nms_impl(boxes, n, thresh, keep, num_keep, workspace, workspace_bytes, stream, 0);
}

extern "C" int sassd_nms_normal_gpu(const float *boxes, int n, float thresh, int64_t *keep, int32_t *num_keep,
                                    void *workspace, size_t workspace_bytes, void *stream)
{
    return nms_impl(boxes, n, thresh, keep, num_keep, workspace, workspace_bytes, stream, 1);
}
