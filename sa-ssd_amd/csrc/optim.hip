// optim.hip -- the 'adam_onecycle' parameter update of the reference's training loop
// (tools/train_utils/__init__.py:57-61: clip_grad_norm_ -> optimizer.step; tools/train_utils/optimization/
// fastai_optim.py:132-148: decoupled weight decay p *= 1 - wd*lr on every group, then torch Adam with
// weight_decay 0, betas (mom, 0.99)) over ONE flat fp32 parameter / gradient buffer.
// HBM-bound streaming kernels: 5.34 M parameters -> 16 B read + 12 B written per element, float4 accesses.
#include "common.h"

#include <algorithm>
#include <cmath>

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) sumsq_kernel(const float *__restrict__ g, long n, float *__restrict__ out)
{
    float acc = 0.f;
    const long n4 = n >> 2;
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = g4[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[(n4 << 2) + threadIdx.x];
        acc += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

struct AdamArgs {
    float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, max_norm, grad_scale;
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const AdamArgs &a, float coef)
{
    g *= coef;
    p *= 1.f - a.wd * a.lr;
    m = a.beta1 * m + (1.f - a.beta1) * g;
    v = a.beta2 * v + (1.f - a.beta2) * g * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p -= (a.lr / a.bc1) * (m / denom);
}

__global__ void __launch_bounds__(256) adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v, long n,
                                                   const float *__restrict__ sumsq, AdamArgs a)
{
    float coef = a.grad_scale;
    if (sumsq && a.max_norm > 0.f) {
        const float norm = sqrtf(*sumsq) * a.grad_scale;
        coef *= fminf(1.f, a.max_norm / (norm + 1e-6f));
    }
    const long n4 = n >> 2;
    f32x4 *p4 = reinterpret_cast<f32x4 *>(p), *m4 = reinterpret_cast<f32x4 *>(m), *v4 = reinterpret_cast<f32x4 *>(v);
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 pp = p4[i], mm = m4[i], vv = v4[i];
        const f32x4 gg = g4[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float ps = pp[c], ms = mm[c], vs = vv[c];
            adam_one(ps, gg[c], ms, vs, a, coef);
            pp[c] = ps; mm[c] = ms; vv[c] = vs;
        }
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        adam_one(p[i], g[i], m[i], v[i], a, coef);
    }
}
}  // namespace

extern "C" int sassd_grad_sumsq(const float *grad, long n, float *out, void *stream_)
{
    if (!grad || !out || n < 0 || ((uintptr_t)grad & 15)) return SASSD_EINVAL;
    hipStream_t s = (hipStream_t)stream_;
    if (hipMemsetAsync(out, 0, sizeof(float), s) != hipSuccess) return sassd_launch_status();
    const int blocks = (int)std::min<long>(1024, std::max<long>(1, (n / 4 + 255) / 256));
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, s, grad, n, out);
    return sassd_launch_status();
}

extern "C" int sassd_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long n,
                               const float *grad_sumsq, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int step, float max_norm, float grad_scale, void *stream_)
{
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return SASSD_EINVAL;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return SASSD_EINVAL;
    AdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.bc1 = (float)(1.0 - std::pow((double)beta1, (double)step));
    a.bc2_sqrt = (float)std::sqrt(1.0 - std::pow((double)beta2, (double)step));
    a.max_norm = max_norm; a.grad_scale = grad_scale;
    const int blocks = (int)std::min<long>(2048, std::max<long>(1, (n / 4 + 255) / 256));
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, param, grad, exp_avg,
                       exp_avg_sq, n, grad_sumsq, a);
    return sassd_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// Re-packing of EVERY kernel-layout weight image after an update, in one launch per destination type: all parameters
// live in one flat buffer and every pack (sparse-conv MFMA fragment order, its transposed twin for the data gradient,
// the direct conv's [chunk][tap][kc][CoutPad], the bf16 [tap][Cin/8][Cout][8] images) is a pure permutation with zero
// padding, so "pack" = gather through a precomputed index map (sassd.train.PackPlan builds the maps once by pushing
// index-valued weights through the individual pack kernels).  map[i] < 0 -> 0.
namespace {
__global__ void gather_pack_f32_kernel(const float *__restrict__ src, const int32_t *__restrict__ map,
                                       float *__restrict__ dst, long n)
{
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int m = map[i];
    dst[i] = m >= 0 ? src[m] : 0.f;
}

__global__ void gather_pack_bf16_kernel(const float *__restrict__ src, const int32_t *__restrict__ map,
                                        unsigned short *__restrict__ dst, long n)
{
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int m = map[i];
    const float v = m >= 0 ? src[m] : 0.f;
    dst[i] = __builtin_bit_cast(unsigned short, (__bf16)v);          // round-to-nearest-even, like the pack kernels
}
}  // namespace

extern "C" int sassd_gather_pack(const float *src, const int32_t *map, void *dst, long n, int bf16, void *stream_)
{
    if (!src || !map || !dst || n < 0) return SASSD_EINVAL;
    if (n == 0) return SASSD_OK;
    hipStream_t s = (hipStream_t)stream_;
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (bf16)
        hipLaunchKernelGGL(gather_pack_bf16_kernel, dim3(grid), dim3(256), 0, s, src, map, (unsigned short *)dst, n);
    else
        hipLaunchKernelGGL(gather_pack_f32_kernel, dim3(grid), dim3(256), 0, s, src, map, (float *)dst, n);
    return sassd_launch_status();
}

