// bn2d.hip -- training-mode BatchNorm2d + ReLU over NCHW fp32 maps, forward and backward (the BEV stack and the
// part-sensitive head: Conv2d -> BatchNorm2d(eps 1e-3, momentum 0.01) -> ReLU, mmdet/models/necks/cmn.py:233-282,
// mmdet/models/single_stage_heads/ssd_rotate_head.py:424-429).
//
// torch runs the pair as MIOpen BatchNorm + clamp (forward) and threshold_backward + MIOpen BatchNorm backward: four
// passes over a 72 MB tensor at batch 2 where two each way are enough, and the ReLU mask is recomputed from x instead
// of being read from y.  A channel's B x HW elements are cut into S contiguous-per-image chunks (S * C >= ~1024
// workgroups):
//   forward   bn2d_stats_kernel   (c, s): float4 loads, <= 64 elements per thread accumulated in fp32, then the block's
//                                 sum / sum of squares in double -> part[c][s][2]
//             bn2d_apply_kernel   (c, s): adds the S partials of its channel in a fixed order (every block computes the
//                                 same mean / invstd; no inter-workgroup hand-off inside a launch), split 0 stores them
//                                 and updates the running statistics (unbiased variance, like torch), then
//                                 y = max(0, (x - mean) * (invstd * gamma) + beta)
//   backward  bn2d_bwd_reduce     dz = dy * (z > 0) with z recomputed from x; partials of sum dz, sum dz * xhat
//             bn2d_bwd_apply      dx = gamma * invstd * (dz - sum dz / N - xhat * sum dz xhat / N); split 0 stores
//                                 dbeta / dgamma
#include "common.h"

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxSplit = 16;

struct Bn2dArgs {
    const float *x, *dy;
    float *y, *dx;
    int B, C, HW, S, chunk;          // chunk = elements of one image plane per split (multiple of 4)
    double *part;                    // [C][S][2]
    const float *gamma, *beta;
    float *mean, *invstd, *rmean, *rvar, *dgamma, *dbeta;
    float momentum, eps;
};

// block-wide sum of two doubles (256 threads); result valid in every thread
__device__ __forceinline__ void block_sum2(double &a, double &b, double (*red)[4])
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = a; red[1][w] = b; }
    __syncthreads();
    a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    __syncthreads();
}

__global__ void __launch_bounds__(256) bn2d_stats_kernel(Bn2dArgs P)
{
    __shared__ double red[2][4];
    const int c = blockIdx.x / P.S, s = blockIdx.x - c * P.S;
    const int i0 = s * P.chunk, i1 = min(i0 + P.chunk, P.HW);
    double ds = 0.0, dq = 0.0;
    // raw sums as in rounds 3-5, and next to them fp32 sums of the RESIDUALS to the first value a thread loads, re-based to zero
    // in double (sum x^2 = sum d^2 + 2 s sum d + k s^2): the variance is taken from those only where the raw form cancels
    // (mean^2 > 1024 var) -- see bn.hip for why the normal regime keeps its bits
    double d1 = 0.0, d2 = 0.0;
    for (int b = 0; b < P.B; ++b) {
        const float *plane = P.x + ((size_t)b * P.C + c) * P.HW;
        float fs = 0.f, fq = 0.f, rs = 0.f, rq = 0.f, sh = 0.f;
        bool have = false;
        int cnt = 0;
        for (int i = i0 + threadIdx.x * 4; i < i1; i += 1024) {
            const f32x4 v = *(const f32x4 *)(plane + i);
            if (!have) { sh = v[0]; have = true; }
            const f32x4 d = v - sh;
            fs += (v[0] + v[1]) + (v[2] + v[3]);
            fq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            rs += (d[0] + d[1]) + (d[2] + d[3]);
            rq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            if (++cnt == 16) {
                ds += (double)fs; dq += (double)fq;
                d1 += (double)rs + 64.0 * (double)sh;
                d2 += (double)rq + 2.0 * (double)sh * (double)rs + 64.0 * (double)sh * (double)sh;
                fs = fq = rs = rq = 0.f; cnt = 0;
            }
        }
        ds += (double)fs;
        dq += (double)fq;
        d1 += (double)rs + 4.0 * cnt * (double)sh;
        d2 += (double)rq + 2.0 * (double)sh * (double)rs + 4.0 * cnt * (double)sh * (double)sh;
    }
    {
        block_sum2(d1, d2, red);                             // the re-based pair: a second [C][S][2] block behind the first
        if (threadIdx.x == 0) {
            P.part[(size_t)P.C * kMaxSplit * 2 + ((size_t)c * P.S + s) * 2 + 0] = d1;
            P.part[(size_t)P.C * kMaxSplit * 2 + ((size_t)c * P.S + s) * 2 + 1] = d2;
        }
    }
    block_sum2(ds, dq, red);
    if (threadIdx.x == 0) {
        P.part[((size_t)c * P.S + s) * 2 + 0] = ds;
        P.part[((size_t)c * P.S + s) * 2 + 1] = dq;
    }
}

__global__ void __launch_bounds__(256) bn2d_apply_kernel(Bn2dArgs P)
{
    const int c = blockIdx.x / P.S, s = blockIdx.x - c * P.S;
    double ss = 0.0, qq = 0.0;
    for (int k = 0; k < P.S; ++k) {                          // fixed order, identical in every block of the channel
        ss += P.part[((size_t)c * P.S + k) * 2 + 0];
        qq += P.part[((size_t)c * P.S + k) * 2 + 1];
    }
    const double n = (double)P.B * (double)P.HW;
    double mean = ss / n;
    double var = qq / n - mean * mean;
    if (mean * mean > 1024.0 * var) {                        // the raw form cancels: the residual-based sum of squares
        double s2 = 0.0, q2 = 0.0;
        for (int k = 0; k < P.S; ++k) {
            s2 += P.part[(size_t)P.C * kMaxSplit * 2 + ((size_t)c * P.S + k) * 2 + 0];
            q2 += P.part[(size_t)P.C * kMaxSplit * 2 + ((size_t)c * P.S + k) * 2 + 1];
        }
        mean = s2 / n;
        var = q2 / n - mean * mean;
    }
    if (var < 0.0) var = 0.0;
    const float m = (float)mean, is = (float)(1.0 / sqrt(var + (double)P.eps));
    const float sc = is * P.gamma[c], sh = P.beta[c];
    if (s == 0 && threadIdx.x == 0) {
        P.mean[c] = m;
        P.invstd[c] = is;
        if (P.rmean) {
            const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
            P.rmean[c] = (float)((1.0 - P.momentum) * P.rmean[c] + P.momentum * mean);
            P.rvar[c] = (float)((1.0 - P.momentum) * P.rvar[c] + P.momentum * unb);
        }
    }
    const int i0 = s * P.chunk, i1 = min(i0 + P.chunk, P.HW);
    for (int b = 0; b < P.B; ++b) {
        const size_t base = ((size_t)b * P.C + c) * P.HW;
        for (int i = i0 + threadIdx.x * 4; i < i1; i += 1024) {
            const f32x4 v = *(const f32x4 *)(P.x + base + i);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = fmaf(v[j] - m, sc, sh);       // (the backward recomputes z with this very expression)
                o[j] = z > 0.f ? z : 0.f;
            }
            *(f32x4 *)(P.y + base + i) = o;
        }
    }
}

__global__ void __launch_bounds__(256) bn2d_bwd_reduce_kernel(Bn2dArgs P)
{
    __shared__ double red[2][4];
    const int c = blockIdx.x / P.S, s = blockIdx.x - c * P.S;
    const int i0 = s * P.chunk, i1 = min(i0 + P.chunk, P.HW);
    const float m = P.mean[c], is = P.invstd[c], g = P.gamma[c], bt = P.beta[c];
    const float sc = is * g;                                 // z exactly as the forward computed it: the mask is y > 0
    double db = 0.0, dg = 0.0;
    for (int b = 0; b < P.B; ++b) {
        const size_t base = ((size_t)b * P.C + c) * P.HW;
        float fb = 0.f, fg = 0.f;
        int cnt = 0;
        for (int i = i0 + threadIdx.x * 4; i < i1; i += 1024) {
            const f32x4 v = *(const f32x4 *)(P.x + base + i), d = *(const f32x4 *)(P.dy + base + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (v[j] - m) * is;
                const float z = fmaf(v[j] - m, sc, bt);
                const float dz = z > 0.f ? d[j] : 0.f;
                fb += dz;
                fg += dz * xh;
            }
            if (++cnt == 16) { db += (double)fb; dg += (double)fg; fb = fg = 0.f; cnt = 0; }
        }
        db += (double)fb;
        dg += (double)fg;
    }
    block_sum2(db, dg, red);
    if (threadIdx.x == 0) {
        P.part[((size_t)c * P.S + s) * 2 + 0] = db;
        P.part[((size_t)c * P.S + s) * 2 + 1] = dg;
    }
}

__global__ void __launch_bounds__(256) bn2d_bwd_apply_kernel(Bn2dArgs P)
{
    const int c = blockIdx.x / P.S, s = blockIdx.x - c * P.S;
    double a = 0.0, b2 = 0.0;
    for (int k = 0; k < P.S; ++k) {
        a += P.part[((size_t)c * P.S + k) * 2 + 0];
        b2 += P.part[((size_t)c * P.S + k) * 2 + 1];
    }
    const float m = P.mean[c], is = P.invstd[c], g = P.gamma[c], bt = P.beta[c];
    const float inv_n = (float)(1.0 / ((double)P.B * (double)P.HW));
    const float sdb = (float)a, sdg = (float)b2, sc = is * g;
    if (s == 0 && threadIdx.x == 0) {
        P.dbeta[c] = sdb;
        P.dgamma[c] = sdg;
    }
    const int i0 = s * P.chunk, i1 = min(i0 + P.chunk, P.HW);
    for (int b = 0; b < P.B; ++b) {
        const size_t base = ((size_t)b * P.C + c) * P.HW;
        for (int i = i0 + threadIdx.x * 4; i < i1; i += 1024) {
            const f32x4 v = *(const f32x4 *)(P.x + base + i), d = *(const f32x4 *)(P.dy + base + i);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (v[j] - m) * is;
                const float z = fmaf(v[j] - m, sc, bt);
                const float dz = z > 0.f ? d[j] : 0.f;
                o[j] = g * is * (dz - sdb * inv_n - xh * sdg * inv_n);
            }
            *(f32x4 *)(P.dx + base + i) = o;
        }
    }
}

// statistics only (round 6): what bn2d_apply_kernel computes per channel before it touches the map, for consumers that apply
// the normalisation themselves (sassd_conv2d_bf16_bnrelu_fwd / sassd_conv2d_bwd_weight_bf16_bnrelu): mean, invstd, running
// statistics, and the affine triple [3][C] = mean | invstd * gamma | beta those kernels read.  Same sums, same order.
__global__ void __launch_bounds__(256) bn2d_finalize_kernel(Bn2dArgs P, float *__restrict__ aff)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= P.C) return;
    double ss = 0.0, qq = 0.0;
    for (int k = 0; k < P.S; ++k) {
        ss += P.part[((size_t)c * P.S + k) * 2 + 0];
        qq += P.part[((size_t)c * P.S + k) * 2 + 1];
    }
    const double n = (double)P.B * (double)P.HW;
    double mean = ss / n;
    double var = qq / n - mean * mean;
    if (mean * mean > 1024.0 * var) {
        double s2 = 0.0, q2 = 0.0;
        for (int k = 0; k < P.S; ++k) {
            s2 += P.part[(size_t)P.C * kMaxSplit * 2 + ((size_t)c * P.S + k) * 2 + 0];
            q2 += P.part[(size_t)P.C * kMaxSplit * 2 + ((size_t)c * P.S + k) * 2 + 1];
        }
        mean = s2 / n;
        var = q2 / n - mean * mean;
    }
    if (var < 0.0) var = 0.0;
    const float m = (float)mean, is = (float)(1.0 / sqrt(var + (double)P.eps));
    P.mean[c] = m;
    P.invstd[c] = is;
    if (P.rmean) {
        const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
        P.rmean[c] = (float)((1.0 - P.momentum) * P.rmean[c] + P.momentum * mean);
        P.rvar[c] = (float)((1.0 - P.momentum) * P.rvar[c] + P.momentum * unb);
    }
    aff[c] = m;
    aff[P.C + c] = is * P.gamma[c];
    aff[2 * P.C + c] = P.beta[c];
}

bool bn2d_shape_ok(int B, int C, int HW) { return B >= 1 && C >= 1 && HW >= 4 && HW % 4 == 0; }

void bn2d_geometry(Bn2dArgs &P)
{
    int S = cdiv(1024, P.C);
    S = S < 1 ? 1 : (S > kMaxSplit ? kMaxSplit : S);
    const int quads = P.HW / 4;
    if (S > quads) S = quads;
    P.chunk = cdiv(quads, S) * 4;
    P.S = cdiv(P.HW, P.chunk);
}
}  // namespace

extern "C" size_t sassd_bn2d_relu_workspace_bytes(int C)
{
    return C < 1 ? 0 : align_up((size_t)C * kMaxSplit * 4 * sizeof(double), 256);
}

extern "C" int sassd_bn2d_relu_fwd(const float *x, int B, int C, int HW, const float *gamma, const float *beta,
                                   float *running_mean, float *running_var, float momentum, float eps, float *y,
                                   float *save_mean, float *save_invstd, void *workspace, size_t workspace_bytes,
                                   void *stream_)
{
    if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || !workspace || !bn2d_shape_ok(B, C, HW) ||
        (!running_mean) != (!running_var) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15))
        return SASSD_EINVAL;
    if (workspace_bytes < sassd_bn2d_relu_workspace_bytes(C)) return SASSD_ENOSPC;
    hipStream_t s = (hipStream_t)stream_;
    Bn2dArgs P = {};
    P.x = x; P.y = y; P.B = B; P.C = C; P.HW = HW;
    bn2d_geometry(P);
    P.part = (double *)workspace;
    P.gamma = gamma; P.beta = beta; P.mean = save_mean; P.invstd = save_invstd; P.rmean = running_mean; P.rvar = running_var;
    P.momentum = momentum; P.eps = eps;
    hipLaunchKernelGGL(bn2d_stats_kernel, dim3(C * P.S), dim3(256), 0, s, P);
    hipLaunchKernelGGL(bn2d_apply_kernel, dim3(C * P.S), dim3(256), 0, s, P);
    return sassd_launch_status();
}

extern "C" int sassd_bn2d_relu_bwd(const float *x, const float *dy, int B, int C, int HW, const float *gamma,
                                   const float *beta, const float *save_mean, const float *save_invstd, float *dx,
                                   float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes, void *stream_)
{
    if (!x || !dy || !gamma || !beta || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || !workspace ||
        !bn2d_shape_ok(B, C, HW) || ((uintptr_t)x & 15) || ((uintptr_t)dy & 15) || ((uintptr_t)dx & 15))
        return SASSD_EINVAL;
    if (workspace_bytes < sassd_bn2d_relu_workspace_bytes(C)) return SASSD_ENOSPC;
    hipStream_t s = (hipStream_t)stream_;
    Bn2dArgs P = {};
    P.x = x; P.dy = dy; P.dx = dx; P.B = B; P.C = C; P.HW = HW;
    bn2d_geometry(P);
    P.part = (double *)workspace;
    P.gamma = gamma; P.beta = beta; P.mean = (float *)save_mean; P.invstd = (float *)save_invstd;
    P.dgamma = dgamma; P.dbeta = dbeta;
    hipLaunchKernelGGL(bn2d_bwd_reduce_kernel, dim3(C * P.S), dim3(256), 0, s, P);
    hipLaunchKernelGGL(bn2d_bwd_apply_kernel, dim3(C * P.S), dim3(256), 0, s, P);
    return sassd_launch_status();
}

extern "C" int sassd_bn2d_stats(const float *x, int B, int C, int HW, const float *gamma, const float *beta,
                                float *running_mean, float *running_var, float momentum, float eps, float *save_mean,
                                float *save_invstd, float *affine, void *workspace, size_t workspace_bytes, void *stream_)
{
    if (!x || !gamma || !beta || !save_mean || !save_invstd || !affine || !workspace || !bn2d_shape_ok(B, C, HW) ||
        (!running_mean) != (!running_var) || ((uintptr_t)x & 15))
        return SASSD_EINVAL;
    if (workspace_bytes < sassd_bn2d_relu_workspace_bytes(C)) return SASSD_ENOSPC;
    hipStream_t s = (hipStream_t)stream_;
    Bn2dArgs P = {};
    P.x = x; P.B = B; P.C = C; P.HW = HW;
    bn2d_geometry(P);
    P.part = (double *)workspace;
    P.gamma = gamma; P.beta = beta; P.mean = save_mean; P.invstd = save_invstd; P.rmean = running_mean; P.rvar = running_var;
    P.momentum = momentum; P.eps = eps;
    hipLaunchKernelGGL(bn2d_stats_kernel, dim3(C * P.S), dim3(256), 0, s, P);
    hipLaunchKernelGGL(bn2d_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, P, affine);
    return sassd_launch_status();
}
