// aux_head.hip -- the auxiliary (point-wise supervision) head of SpMiddleFHD in TRAINING, fused
// (mmdet/models/necks/cmn.py:27-29 point_fc / point_cls / point_reg, :45-72 build_aux_target, :74-104 aux_loss,
// :121-135 + :175-189 nearest_neighbor_interpolate, mmdet/core/bbox/transforms.py:218-223 tensor2points).
//
// The reference (and sassd's module path) runs it as ~150 small launches per step: three tensor2points, three
// (sqrt, add, reciprocal, sum, div, interpolate) chains, a concat, three Linear layers on a library GEMM (0.7 GFLOP at
// 4 TF/s: skinny shapes), the point-in-box labels per sample, focal / smooth-L1 arithmetic and the autograd mirror of all of
// it.  Here:
//   aux_prepare_kernel      voxel centres of the three scales (same fp32 operation order as the torch expression),
//                           points_mean, point labels / centre offsets (the reference's point-in-box test, last
//                           containing box of the point's own sample wins) and the positive count
//   [sassd_three_nn_binned x 3, unchanged]
//   aux_fwd_kernel          ONE THREAD PER POINT: inverse-distance weights, the three 3-neighbour interpolations straight
//                           into the 160 -> 64 layer (h accumulates in 64 registers; W1 broadcast from LDS), the 64 -> 1 / 3
//                           heads, focal + smooth-L1 terms AND their gradients with respect to the four outputs
//   aux_bwd_points_kernel   dh = d(out) W2, df = dh W1 (per point, same shape as the forward) -> df [N, 160]
//   aux_scatter_kernel      df scattered to the gradients of the three scales' features by float atomics (as the
//                           reference's three_interpolate_grad), one thread per (point, channel): the 64 lanes of a wave
//                           add to CONSECUTIVE channels of one row (a first version issued the atomics from the
//                           row-per-thread kernel, 64 different rows per instruction: 920 us instead of 60)
//   aux_wgrad_kernel        dW1 = sum_p dh f^T, dW2 = sum_p d(out) h^T: 64-point chunks staged in LDS, each thread owns a
//                           5 x 8 patch of dW1 and one entry of dW2; per-workgroup partials, fixed-order reduction
// fp32 VALU throughout: on gfx950 the fp32 vector rate equals the fp32 MFMA rate, and a row-per-thread formulation needs
// no operand shuffles.
#include "common.h"

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kH = 64, kF = 160, kOut = 4;      // hidden width, interpolated features (32 + 64 + 64), outputs (cls, reg x3)

__device__ __forceinline__ int aux_pt_in_box3d(float x, float y, float z, float cx, float cy, float bottom_z, float w,
                                               float l, float h, float angle)
{
#pragma clang fp contract(off)
    // points_op.cpp:92-105, literal argument order of the call site (:130-133): see pointops.hip::pt_in_box3d
    const float max_dis = 10.0f;
    const float cz = (float)((double)bottom_z + (double)h / 2.0);
    if ((fabsf(x - cx) > max_dis) || ((double)fabsf(z - cz) > (double)h / 2.0) || (fabsf(y - cy) > max_dis)) return 0;
    const float cosa = cosf(angle), sina = sinf(angle);
    const float x_rot = (x - cx) * cosa + (y - cy) * (-sina);
    const float y_rot = (x - cx) * sina + (y - cy) * cosa;
    return ((double)x_rot >= -(double)w / 2.0) & ((double)x_rot <= (double)w / 2.0) &
           ((double)y_rot >= -(double)l / 2.0) & ((double)y_rot <= (double)l / 2.0);
}

struct AuxPrepArgs {
    const float *vfeat;            // [N, vstride] voxel means (x, y, z, ...)
    const int32_t *coors;          // [N, 4] (b, z, y, x)
    int N, vstride;
    const int32_t *idx[3];         // [M_s, 4] (b, z, y, x) of the three middle tensors
    int M[3];
    float vs[3][3], off[3], half_vs[3][3];     // voxel size per scale (x, y, z), offset (x, y, z), 0.5 * voxel size
    const float *gt;               // [T, 7]
    const int32_t *gt_off;         // [B + 1]
    int B;
    float *points;                 // [N, 4] (b, x, y, z)
    float *known[3];               // [M_s, 4] (b, x, y, z)
    uint8_t *label;                // [N]
    float *target;                 // [N, 3]
    int *npos;                     // [1], zeroed by the caller
};

// blockIdx.y: 0 = points (labels, targets), 1..3 = voxel centres of scale y - 1
__global__ void __launch_bounds__(256) aux_prepare_kernel(AuxPrepArgs P)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int job = blockIdx.y;
    if (job > 0) {
        const int s = job - 1;
        if (i >= P.M[s]) return;
        const int4 c = ((const int4 *)P.idx[s])[i];
        // ind[:, 1:].flip(1) * vs + off + .5 * vs  (left to right, fp32, no contraction)
        const float x = ((float)c.w * P.vs[s][0] + P.off[0]) + P.half_vs[s][0];
        const float y = ((float)c.z * P.vs[s][1] + P.off[1]) + P.half_vs[s][1];
        const float z = ((float)c.y * P.vs[s][2] + P.off[2]) + P.half_vs[s][2];
        ((float4 *)P.known[s])[i] = make_float4((float)c.x, x, y, z);
        return;
    }
    bool pos = false;
    if (i < P.N) {
        const int b = P.coors[(size_t)i * 4];
        const float x = P.vfeat[(size_t)i * P.vstride], y = P.vfeat[(size_t)i * P.vstride + 1],
                    z = P.vfeat[(size_t)i * P.vstride + 2];
        ((float4 *)P.points)[i] = make_float4((float)b, x, y, z);
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        if (b >= 0 && b < P.B) {
            for (int g = P.gt_off[b]; g < P.gt_off[b + 1]; ++g) {
                const float *bx = P.gt + (size_t)g * 7;
                if (aux_pt_in_box3d(x, y, z, bx[0], bx[1], bx[2], bx[3], bx[4], bx[5], bx[6]) == 1) {
                    pos = true;                                  // the LAST containing box defines the offsets
                    t0 = x - bx[0];
                    t1 = y - bx[1];
                    t2 = (float)((double)z - ((double)bx[2] + (double)bx[3] / 2.0));
                }
            }
        }
        P.label[i] = pos ? 1 : 0;
        P.target[(size_t)i * 3] = t0; P.target[(size_t)i * 3 + 1] = t1; P.target[(size_t)i * 3 + 2] = t2;
    }
    __shared__ int wcnt[4];
    const unsigned long long m = __ballot(pos);
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int s = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        if (s) atomicAdd(P.npos, s);
    }
}

struct AuxArgs {
    int N;
    const float *feat[3];          // [M_s, C_s], C = 32, 64, 64
    const int32_t *nn_idx[3];      // [N, 3]
    const float *nn_d2[3];         // [N, 3] squared distances
    const float *w1;               // [64, 160]  (torch Linear layout [out, in])
    const float *w2;               // [4, 64]    (point_cls row, then the three point_reg rows)
    const uint8_t *label;          // [N]
    const float *target;           // [N, 3]
    const int *npos;               // [1]
    float *wgt;                    // [N, 9] interpolation weights (forward writes, backward reads)
    float *h;                      // [N, 64]
    float *out;                    // [N, 4] (cls, reg)
    float *gout;                   // [N, 4] d(loss sums) / d(out), of the UNSCALED sums (cls term, reg term)
    float *part;                   // [nblocks][2] loss partials
    const float *gup;              // [2] upstream gradients of (cls sum, reg sum)      (backward)
    float *gfeat[3];               // [M_s, C_s] gradient accumulators, zeroed by the caller (backward)
    float *dfbuf;                  // [N, 160] gradient of the interpolated features (backward)
    float *wpart;                  // [nwg][64 * 160 + 4 * 64] weight-gradient partials  (backward)
    int chunks_per_wg;
};

template <int C>
__device__ __forceinline__ void interp_chunk(const float *feat, const int (&id)[3], const float (&w)[3], int c4, f32x4 &f)
{
    const f32x4 a = *(const f32x4 *)(feat + (size_t)id[0] * C + c4 * 4);
    const f32x4 b = *(const f32x4 *)(feat + (size_t)id[1] * C + c4 * 4);
    const f32x4 c = *(const f32x4 *)(feat + (size_t)id[2] * C + c4 * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = w[0] * a[j] + w[1] * b[j] + w[2] * c[j];
}

// h[0..63] += f[0..3] (channels k0 .. k0+3) x W1T[k][0..63]   (W1T in LDS, [160][64]; every lane reads the same address)
__device__ __forceinline__ void accum_h(float (&h)[kH], const f32x4 &f, const float *w1t, int k0)
{
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const float *row = w1t + (k0 + cc) * kH;
#pragma unroll
        for (int o4 = 0; o4 < kH / 4; ++o4) {
            const f32x4 w = *(const f32x4 *)(row + o4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) h[o4 * 4 + j] = fmaf(f[cc], w[j], h[o4 * 4 + j]);
        }
    }
}

template <int C>
__device__ __forceinline__ void fwd_scale(const AuxArgs &P, int s, int p, int koff, float (&h)[kH], const float *w1t)
{
    int id[3];
    float w[3];
    {
        const float d0 = sqrtf(P.nn_d2[s][p * 3]), d1 = sqrtf(P.nn_d2[s][p * 3 + 1]), d2 = sqrtf(P.nn_d2[s][p * 3 + 2]);
        const float r0 = 1.0f / (d0 + 1e-8f), r1 = 1.0f / (d1 + 1e-8f), r2 = 1.0f / (d2 + 1e-8f);
        const float nrm = (r0 + r1) + r2;
        w[0] = r0 / nrm; w[1] = r1 / nrm; w[2] = r2 / nrm;
        P.wgt[(size_t)p * 9 + s * 3] = w[0]; P.wgt[(size_t)p * 9 + s * 3 + 1] = w[1]; P.wgt[(size_t)p * 9 + s * 3 + 2] = w[2];
        id[0] = P.nn_idx[s][p * 3]; id[1] = P.nn_idx[s][p * 3 + 1]; id[2] = P.nn_idx[s][p * 3 + 2];
    }
    for (int c4 = 0; c4 < C / 4; ++c4) {
        f32x4 f;
        interp_chunk<C>(P.feat[s], id, w, c4, f);
        accum_h(h, f, w1t, koff + c4 * 4);
    }
}

__global__ void __launch_bounds__(256) aux_fwd_kernel(AuxArgs P)
{
    __shared__ __attribute__((aligned(16))) float w1t[kF * kH];        // [k][o]  40 KB
    __shared__ __attribute__((aligned(16))) float w2s[kOut * kH];
    __shared__ float red[2][4];
    for (int i = threadIdx.x; i < kF * kH; i += 256) {                  // transpose [o][k] -> [k][o]
        const int o = i / kF, k = i - o * kF;
        w1t[k * kH + o] = P.w1[i];
    }
    for (int i = threadIdx.x; i < kOut * kH; i += 256) w2s[i] = P.w2[i];
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    float l_cls = 0.f, l_reg = 0.f;
    if (p < P.N) {
        float h[kH];
#pragma unroll
        for (int o = 0; o < kH; ++o) h[o] = 0.f;
        fwd_scale<32>(P, 0, p, 0, h, w1t);
        fwd_scale<64>(P, 1, p, 32, h, w1t);
        fwd_scale<64>(P, 2, p, 96, h, w1t);
        float o4[kOut] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < kH; ++o) {
#pragma unroll
            for (int j = 0; j < kOut; ++j) o4[j] = fmaf(h[o], w2s[j * kH + o], o4[j]);
        }
#pragma unroll
        for (int q = 0; q < kH / 4; ++q)
            *(f32x4 *)(P.h + (size_t)p * kH + q * 4) = (f32x4){h[q * 4], h[q * 4 + 1], h[q * 4 + 2], h[q * 4 + 3]};
        *(f32x4 *)(P.out + (size_t)p * 4) = (f32x4){o4[0], o4[1], o4[2], o4[3]};
        // ---- losses (cmn.py:74-104): focal over every point with weight 1 / max(#pos, 1); smooth-L1 (beta 1/9) on the
        // centre offsets of the positive points with weight pos / max(#pos, 1); gradients of the two sums
        const float norm = fmaxf((float)*P.npos, 1.f);
        const float t = P.label[p] ? 1.f : 0.f;
        f32x4 g;
        {
            const float x = o4[0];
            const float pr = 1.f / (1.f + expf(-x));
            const float pt = (1.f - pr) * t + pr * (1.f - t);
            const float aw = (0.25f * t + 0.75f * (1.f - t)) * (1.f / norm);
            const float w = aw * (pt * pt);
            const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
            l_cls = bce * w;
            const float dpt = pr * (1.f - pr) * (1.f - 2.f * t);
            g[0] = (pr - t) * w + bce * aw * 2.f * pt * dpt;
        }
        const float reg_w = t / norm, beta = 1.f / 9.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float diff = o4[1 + j] - P.target[(size_t)p * 3 + j];
            const float d = fabsf(diff);
            l_reg += (d < beta ? 0.5f * d * d / beta : d - 0.5f * beta) * reg_w;
            const float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
            g[1 + j] = (d < beta ? d / beta : 1.f) * sg * reg_w;
        }
        *(f32x4 *)(P.gout + (size_t)p * 4) = g;
    }
    float v[2] = {l_cls, l_reg};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float s = v[k];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < 2)
        P.part[(size_t)blockIdx.x * 2 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) +
                                                       (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ void __launch_bounds__(256) aux_loss_sum_kernel(const float *__restrict__ part, int n, float *__restrict__ out)
{
    __shared__ double red[2][4];
    double s[2] = {0, 0};
    for (int i = threadIdx.x; i < n; i += 256)
        for (int k = 0; k < 2; ++k) s[k] += (double)part[(size_t)i * 2 + k];
    for (int k = 0; k < 2; ++k) {
        double v = s[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 2) out[threadIdx.x] = (float)((red[threadIdx.x][0] + red[threadIdx.x][1]) +
                                                    (red[threadIdx.x][2] + red[threadIdx.x][3]));
}

// ---- backward, per point: dh = d W2, df = dh W1, scattered to the neighbours ------------------------------------------
template <int C>
__device__ __forceinline__ void bwd_scale(const AuxArgs &P, int p, int koff, const float (&dh)[kH], const float *w1s)
{
    for (int c4 = 0; c4 < C / 4; ++c4) {
        f32x4 df = {0.f, 0.f, 0.f, 0.f};
        const float *col = w1s + koff + c4 * 4;                         // W1[o][k .. k+3], row stride 160
#pragma unroll
        for (int o = 0; o < kH; ++o) {
            const f32x4 wv = *(const f32x4 *)(col + o * kF);
#pragma unroll
            for (int j = 0; j < 4; ++j) df[j] = fmaf(dh[o], wv[j], df[j]);
        }
        *(f32x4 *)(P.dfbuf + (size_t)p * kF + koff + c4 * 4) = df;
    }
}

// one thread per (point, interpolated channel): three atomics, coalesced over the channels of a neighbour row
__global__ void __launch_bounds__(256) aux_scatter_kernel(AuxArgs P)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)P.N * kF) return;
    const int p = (int)(t / kF), k = (int)(t - (size_t)p * kF);
    const int s = k < 32 ? 0 : (k < 96 ? 1 : 2);
    const int C = s == 0 ? 32 : 64, c = k - (s == 0 ? 0 : (s == 1 ? 32 : 96));
    const float g = P.dfbuf[t];
    const int *id = P.nn_idx[s] + (size_t)p * 3;
    const float *w = P.wgt + (size_t)p * 9 + s * 3;
    float *dst = P.gfeat[s];
    unsafeAtomicAdd(dst + (size_t)id[0] * C + c, g * w[0]);
    unsafeAtomicAdd(dst + (size_t)id[1] * C + c, g * w[1]);
    unsafeAtomicAdd(dst + (size_t)id[2] * C + c, g * w[2]);
}

__global__ void __launch_bounds__(256) aux_bwd_points_kernel(AuxArgs P)
{
    __shared__ __attribute__((aligned(16))) float w1s[kH * kF];        // [o][k]  40 KB
    __shared__ __attribute__((aligned(16))) float w2s[kOut * kH];
    for (int i = threadIdx.x; i < kH * kF / 4; i += 256) ((f32x4 *)w1s)[i] = ((const f32x4 *)P.w1)[i];
    for (int i = threadIdx.x; i < kOut * kH; i += 256) w2s[i] = P.w2[i];
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P.N) return;
    const f32x4 g = *(const f32x4 *)(P.gout + (size_t)p * 4);
    const float gc = P.gup[0], gr = P.gup[1];
    const float d[kOut] = {g[0] * gc, g[1] * gr, g[2] * gr, g[3] * gr};
    float dh[kH];
#pragma unroll
    for (int o = 0; o < kH; ++o)
        dh[o] = (d[0] * w2s[o] + d[1] * w2s[kH + o]) + (d[2] * w2s[2 * kH + o] + d[3] * w2s[3 * kH + o]);
    bwd_scale<32>(P, p, 0, dh, w1s);
    bwd_scale<64>(P, p, 32, dh, w1s);
    bwd_scale<64>(P, p, 96, dh, w1s);
}

// ---- backward, weights: dW1[o][k] = sum_p dh[p][o] f[p][k], dW2[j][o] = sum_p d[p][j] h[p][o] -------------------------
constexpr int kChunk = 64;                                              // points per staged chunk
constexpr int kFP = kF + 1;                                             // LDS row pitch of f (odd: conflict-free columns)

__global__ void __launch_bounds__(256) aux_wgrad_kernel(AuxArgs P)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *fs = lds;                                   // [64][161]
    float *dhs = fs + kChunk * kFP;                    // [64][64]
    float *hs = dhs + kChunk * kH;                     // [64][64]
    float *ds = hs + kChunk * kH;                      // [64][4]
    float *w2s = ds + kChunk * kOut;                   // [4][64]
    for (int i = threadIdx.x; i < kOut * kH; i += 256) w2s[i] = P.w2[i];
    const int t = threadIdx.x;
    const int og = t & 7, kg = t >> 3;                 // dW1 patch: o = og*8 .. +7, k = kg*5 .. +4
    const int j2 = t >> 6, o2 = t & 63;                // dW2 entry
    float acc[5][8];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = 0.f;
    float acc2 = 0.f;
    const float gc = P.gup[0], gr = P.gup[1];
    const int pt = t >> 2, part = t & 3;               // staging: 4 threads per point
    for (int ch = 0; ch < P.chunks_per_wg; ++ch) {
        const int p0 = (blockIdx.x * P.chunks_per_wg + ch) * kChunk;
        if (p0 >= P.N) break;
        __syncthreads();                               // previous chunk fully consumed (and w2s visible the first time)
        const int p = p0 + pt;
        const bool ok = p < P.N;
        // d (scaled output gradients), h
        if (part == 0) {
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            if (ok) g = *(const f32x4 *)(P.gout + (size_t)p * 4);
            ds[pt * 4] = g[0] * gc; ds[pt * 4 + 1] = g[1] * gr; ds[pt * 4 + 2] = g[2] * gr; ds[pt * 4 + 3] = g[3] * gr;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {                  // 16 floats of h per thread
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *(const f32x4 *)(P.h + (size_t)p * kH + part * 16 + q * 4);
            *(f32x4 *)(hs + pt * kH + part * 16 + q * 4) = v;
        }
        // f: the three interpolations, this thread's quarter of every scale's channels
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int C = s == 0 ? 32 : 64, koff = s == 0 ? 0 : (s == 1 ? 32 : 96);
            const int nq = C / 16;                     // float4 chunks per thread
            int id[3] = {0, 0, 0};
            float w[3] = {0.f, 0.f, 0.f};
            if (ok) {
#pragma unroll
                for (int j = 0; j < 3; ++j) { id[j] = P.nn_idx[s][p * 3 + j]; w[j] = P.wgt[(size_t)p * 9 + s * 3 + j]; }
            }
            for (int q = 0; q < nq; ++q) {
                const int c4 = part * nq + q;
                f32x4 f = {0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    if (s == 0) interp_chunk<32>(P.feat[0], id, w, c4, f);
                    else interp_chunk<64>(P.feat[s], id, w, c4, f);
                }
                float *dst = fs + pt * kFP + koff + c4 * 4;
                dst[0] = f[0]; dst[1] = f[1]; dst[2] = f[2]; dst[3] = f[3];
            }
        }
        __syncthreads();
        // dh of the chunk: thread (pt, part) computes 16 hidden units
        {
            const float d0 = ds[pt * 4], d1 = ds[pt * 4 + 1], d2 = ds[pt * 4 + 2], d3 = ds[pt * 4 + 3];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int o = part * 16 + q;
                dhs[pt * kH + o] = (d0 * w2s[o] + d1 * w2s[kH + o]) + (d2 * w2s[2 * kH + o] + d3 * w2s[3 * kH + o]);
            }
        }
        __syncthreads();
        for (int r = 0; r < kChunk; ++r) {
            const f32x4 a0 = *(const f32x4 *)(dhs + r * kH + og * 8), a1 = *(const f32x4 *)(dhs + r * kH + og * 8 + 4);
            const float dv[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            float fv[5];
#pragma unroll
            for (int a = 0; a < 5; ++a) fv[a] = fs[r * kFP + kg * 5 + a];
#pragma unroll
            for (int a = 0; a < 5; ++a)
#pragma unroll
                for (int b = 0; b < 8; ++b) acc[a][b] = fmaf(fv[a], dv[b], acc[a][b]);
            acc2 = fmaf(ds[r * 4 + j2], hs[r * kH + o2], acc2);
        }
    }
    float *wp = P.wpart + (size_t)blockIdx.x * (kH * kF + kOut * kH);
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) wp[(og * 8 + b) * kF + kg * 5 + a] = acc[a][b];
    wp[kH * kF + j2 * kH + o2] = acc2;
}

// 8 threads per element, each summing every 8th workgroup's partial (two chains), then a fixed-order sum through LDS: the
// chain of dependent memory round trips is what this kernel costs (see wgrad_reduce_kernel in spconv.hip)
__global__ void __launch_bounds__(256) aux_wgrad_reduce_kernel(const float *__restrict__ wpart, int nwg,
                                                               float *__restrict__ dw1, float *__restrict__ dw2)
{
    __shared__ float red[8][32];
    constexpr int tot = kH * kF + kOut * kH;
    const int e = threadIdx.x & 31, pt = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + e;
    float s0 = 0.f, s1 = 0.f;
    if (i < tot) {
        int w = pt;
        for (; w + 8 < nwg; w += 16) {
            s0 += wpart[(size_t)w * tot + i];
            s1 += wpart[(size_t)(w + 8) * tot + i];
        }
        if (w < nwg) s0 += wpart[(size_t)w * tot + i];
    }
    red[pt][e] = s0 + s1;
    __syncthreads();
    if (pt == 0 && i < tot) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += red[j][e];
        if (i < kH * kF) dw1[i] = s;
        else dw2[i - kH * kF] = s;
    }
}

constexpr size_t kWgradLds = (size_t)(kChunk * kFP + 2 * kChunk * kH + kChunk * kOut + kOut * kH) * 4;
int aux_wgrad_wgs(int N, int *chunks_per_wg)
{
    const int chunks = cdiv(N, kChunk);
    int nwg = chunks < 256 ? chunks : 256;
    if (nwg < 1) nwg = 1;
    *chunks_per_wg = cdiv(chunks, nwg);
    return cdiv(chunks, *chunks_per_wg);
}
}  // namespace

extern "C" size_t sassd_aux_head_workspace_bytes(int N)
{
    if (N < 1) return 0;
    int cpw;
    const int nwg = aux_wgrad_wgs(N, &cpw);
    const size_t a = align_up((size_t)cdiv(N, 256) * 2 * sizeof(float), 256);
    const size_t b = align_up((size_t)nwg * (kH * kF + kOut * kH) * sizeof(float), 256);
    return (a > b ? a : b) + align_up((size_t)N * kF * sizeof(float), 256);      // + df [N, 160] of the backward
}

extern "C" int sassd_aux_prepare(const float *voxel_feats, int vstride, const int32_t *coors, int N,
                                 const int32_t *const *indices, const int *M, const float *voxel_size,
                                 const float *offset, const float *gt_boxes, const int32_t *gt_off, int B, float *points,
                                 float *const *known, uint8_t *label, float *target, int *npos, void *stream_)
{
    if (!voxel_feats || !coors || N < 1 || vstride < 3 || !indices || !M || !voxel_size || !offset || !gt_off ||
        B < 1 || !points || !known || !label || !target || !npos)
        return SASSD_EINVAL;
    hipStream_t s = (hipStream_t)stream_;
    AuxPrepArgs P = {};
    P.vfeat = voxel_feats; P.vstride = vstride; P.coors = coors; P.N = N;
    int mmax = N;
    for (int k = 0; k < 3; ++k) {
        if (M[k] < 0 || (M[k] > 0 && (!indices[k] || !known[k]))) return SASSD_EINVAL;
        P.idx[k] = indices[k]; P.M[k] = M[k]; P.known[k] = known[k];
        mmax = M[k] > mmax ? M[k] : mmax;
        const float mult = (float)(2 << k);                  // the three scales: 2, 4, 8 x the input voxel
        for (int a = 0; a < 3; ++a) {
            P.vs[k][a] = voxel_size[a] * mult;
            P.half_vs[k][a] = 0.5f * P.vs[k][a];
        }
    }
    for (int a = 0; a < 3; ++a) P.off[a] = offset[a];
    P.gt = gt_boxes; P.gt_off = gt_off; P.B = B;
    P.points = points; P.label = label; P.target = target; P.npos = npos;
    if (hipMemsetAsync(npos, 0, sizeof(int), s) != hipSuccess) return sassd_launch_status();
    hipLaunchKernelGGL(aux_prepare_kernel, dim3(cdiv(mmax, 256), 4), dim3(256), 0, s, P);
    return sassd_launch_status();
}

namespace {
int fill_common(AuxArgs &P, int N, const float *const *feats, const int32_t *const *nn_idx, const float *const *nn_d2,
                const float *w1, const float *w2)
{
    if (N < 1 || !feats || !nn_idx || !w1 || !w2) return SASSD_EINVAL;
    P.N = N;
    for (int k = 0; k < 3; ++k) {
        if (!feats[k] || !nn_idx[k]) return SASSD_EINVAL;
        P.feat[k] = feats[k]; P.nn_idx[k] = nn_idx[k];
        P.nn_d2[k] = nn_d2 ? nn_d2[k] : nullptr;
    }
    P.w1 = w1; P.w2 = w2;
    return SASSD_OK;
}
}  // namespace

extern "C" int sassd_aux_head_fwd(int N, const float *const *feats, const int32_t *const *nn_idx,
                                  const float *const *nn_d2, const float *w1, const float *w2, const uint8_t *label,
                                  const float *target, const int *npos, float *wgt, float *h, float *out, float *gout,
                                  float *loss_sums, void *workspace, size_t workspace_bytes, void *stream_)
{
    AuxArgs P = {};
    int rc = fill_common(P, N, feats, nn_idx, nn_d2, w1, w2);
    if (rc) return rc;
    if (!nn_d2 || !nn_d2[0] || !nn_d2[1] || !nn_d2[2] || !label || !target || !npos || !wgt || !h || !out || !gout ||
        !loss_sums || !workspace)
        return SASSD_EINVAL;
    if (workspace_bytes < sassd_aux_head_workspace_bytes(N)) return SASSD_ENOSPC;
    hipStream_t s = (hipStream_t)stream_;
    P.label = label; P.target = target; P.npos = npos; P.wgt = wgt; P.h = h; P.out = out; P.gout = gout;
    P.part = (float *)workspace;
    const int nb = cdiv(N, 256);
    hipLaunchKernelGGL(aux_fwd_kernel, dim3(nb), dim3(256), 0, s, P);
    hipLaunchKernelGGL(aux_loss_sum_kernel, dim3(1), dim3(256), 0, s, (const float *)workspace, nb, loss_sums);
    return sassd_launch_status();
}

extern "C" int sassd_aux_head_bwd(int N, const float *const *feats, const int *M, const int32_t *const *nn_idx,
                                  const float *w1, const float *w2, const float *wgt, const float *h, const float *gout,
                                  const float *grad_sums, float *const *grad_feats, float *dw1, float *dw2,
                                  void *workspace, size_t workspace_bytes, void *stream_)
{
    AuxArgs P = {};
    int rc = fill_common(P, N, feats, nn_idx, nullptr, w1, w2);
    if (rc) return rc;
    if (!M || !wgt || !h || !gout || !grad_sums || !grad_feats || !dw1 || !dw2 || !workspace) return SASSD_EINVAL;
    if (workspace_bytes < sassd_aux_head_workspace_bytes(N)) return SASSD_ENOSPC;
    hipStream_t s = (hipStream_t)stream_;
    P.wgt = (float *)wgt; P.h = (float *)h; P.gout = (float *)gout; P.gup = grad_sums;
    const int C[3] = {32, 64, 64};
    for (int k = 0; k < 3; ++k) {
        if (!grad_feats[k] || M[k] < 1) return SASSD_EINVAL;
        P.gfeat[k] = grad_feats[k];
        if (hipMemsetAsync(grad_feats[k], 0, (size_t)M[k] * C[k] * sizeof(float), s) != hipSuccess)
            return sassd_launch_status();
    }
    int cpw0;
    const size_t wpart_bytes = align_up((size_t)aux_wgrad_wgs(N, &cpw0) * (kH * kF + kOut * kH) * sizeof(float), 256);
    const size_t part_bytes = align_up((size_t)cdiv(N, 256) * 2 * sizeof(float), 256);
    P.dfbuf = (float *)((char *)workspace + (wpart_bytes > part_bytes ? wpart_bytes : part_bytes));
    hipLaunchKernelGGL(aux_bwd_points_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, P);
    hipLaunchKernelGGL(aux_scatter_kernel, dim3((unsigned)(((size_t)N * kF + 255) / 256)), dim3(256), 0, s, P);
    static std::atomic<unsigned long long> attr_done{0};
    rc = sassd_dyn_lds((const void *)aux_wgrad_kernel, kWgradLds, attr_done);
    if (rc) return rc;
    const int nwg = aux_wgrad_wgs(N, &P.chunks_per_wg);
    P.wpart = (float *)workspace;
    hipLaunchKernelGGL(aux_wgrad_kernel, dim3(nwg), dim3(256), kWgradLds, s, P);
    hipLaunchKernelGGL(aux_wgrad_reduce_kernel, dim3(cdiv(kH * kF + kOut * kH, 32)), dim3(256), 0, s,
                       (const float *)workspace, nwg, dw1, dw2);
    return sassd_launch_status();
}
