// probe_bf16_mfma.hip -- building block of DESIGN section 8 lead (6): fp32 fragments converted to bf16 in registers
// (v_cvt_pk_bf16_f32, round-to-nearest-even) feeding v_mfma_f32_16x16x32_bf16 -- (1) operand / result lane layout as the
// sparse kernels would use it (lane (m = l % 16, q = l / 16) holds 8 consecutive channels 8 q .. 8 q + 7 of row m; D[row =
// 4 (l / 16) + reg][col = l % 16]), checked against a host product over the same rounded operands; (2) cycles per
// {16 fp32 values -> 2 A operands, 8 MFMAs} step next to the 64 v_mfma_f32_16x16x4_f32 of the fp32 tile it replaces.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/probe_bf16_mfma tools/probe_bf16_mfma.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 to_bf16x8(const float4 lo, const float4 hi)
{
    bf16x8 r;
    r[0] = (__bf16)lo.x; r[1] = (__bf16)lo.y; r[2] = (__bf16)lo.z; r[3] = (__bf16)lo.w;
    r[4] = (__bf16)hi.x; r[5] = (__bf16)hi.y; r[6] = (__bf16)hi.z; r[7] = (__bf16)hi.w;
    return r;
}

// X [16 rows][32 channels] fp32, W [32 channels][16 couts] fp32 -> D [16][16]
__global__ void layout_kernel(const float *x, const float *w, float *d)
{
    const int l = threadIdx.x, m = l & 15, q = l >> 4;
    const float4 *xr = (const float4 *)(x + m * 32 + 8 * q);
    const bf16x8 a = to_bf16x8(xr[0], xr[1]);
    bf16x8 b;
    for (int i = 0; i < 8; ++i) b[i] = (__bf16)w[(8 * q + i) * 16 + m];      // lane (n = m, q): channels 8 q + i of cout n
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[(4 * q + r) * 16 + m] = c[r];
}

// KIND 0: fp32 tile (16 float values per lane -> 64 x 16x16x4 f32); 1: bf16 tile (16 values -> 2 operands -> 8 x 16x16x32)
template <int KIND>
__global__ void rate_kernel(const float *x, float *out, long long *cyc, int iters)
{
    const int l = threadIdx.x & 63;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 v[4];
    for (int i = 0; i < 4; ++i) v[i] = ((const float4 *)x)[l * 4 + i];
    bf16x8 wb[2][4];
    for (int s = 0; s < 2; ++s)
        for (int t = 0; t < 4; ++t)
            for (int i = 0; i < 8; ++i) wb[s][t][i] = (__bf16)(0.001f * (float)(l + s + t + i));
    float wf[4] = {0.5f, 0.25f, 0.125f, 1.0f};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 1) {
            const bf16x8 a0 = to_bf16x8(v[0], v[1]), a1 = to_bf16x8(v[2], v[3]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wb[0][t], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wb[1][t], acc[t], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i].x, wf[t], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i].y, wf[t], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i].z, wf[t], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i].w, wf[t], acc[t], 0, 0, 0);
                }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i].x += acc[i][0] * 1e-30f;      // keeps the conversion inside the loop
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static float bf16_rne(float f)
{
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&f, &u, 4);
    return f;
}

int main()
{
    std::vector<float> x(16 * 32), w(32 * 16), d(256);
    for (int i = 0; i < 16 * 32; ++i) x[i] = sinf(0.37f * i) * (1.f + (i % 7));
    for (int i = 0; i < 32 * 16; ++i) w[i] = cosf(0.11f * i + 0.3f * (i % 5));            // asymmetric
    float *dx, *dw, *dd;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dw, w.size() * 4); hipMalloc(&dd, 256 * 4);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dx, dw, dd);
    hipMemcpy(d.data(), dd, 256 * 4, hipMemcpyDeviceToHost);
    double worst = 0, worst_unrounded = 0;
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) {
            double ref = 0, ref32 = 0;
            for (int k = 0; k < 32; ++k) {
                ref += (double)bf16_rne(x[r * 32 + k]) * (double)bf16_rne(w[k * 16 + c]);
                ref32 += (double)x[r * 32 + k] * (double)w[k * 16 + c];
            }
            worst = fmax(worst, fabs(ref - d[r * 16 + c]));
            worst_unrounded = fmax(worst_unrounded, fabs(ref32 - d[r * 16 + c]));
        }
    printf("layout: max |D - host product over bf16(RNE) operands| = %.3e  (vs unrounded operands %.3e; values ~ %.1f)\n", worst,
           worst_unrounded, 20.0);
    printf("layout %s\n", worst < 1e-4 ? "OK: A lane (m, q) = channels 8q..8q+7 of row m; B lane (n, q) likewise; D[4q + reg][n]" : "MISMATCH");
    const int iters = 2000;
    float *out; long long *cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    float *xin; hipMalloc(&xin, 64 * 16 * 4); hipMemset(xin, 0, 64 * 16 * 4);
    for (int waves = 1; waves <= 2; ++waves)
        for (int kind = 0; kind < 2; ++kind) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int blocks = 1024, threads = 256 * waves;                       // `waves` waves per SIMD on 256 CUs x 4 SIMDs
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (kind) hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks / 4), dim3(threads), 0, 0, xin, out, cyc, iters);
                else hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks / 4), dim3(threads), 0, 0, xin, out, cyc, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
            printf("%s tile step, %d wave(s) per SIMD: %.0f cycles per 16-value step and wave (%.3f ms for %d steps)\n",
                   kind ? "bf16 (16 values -> 2 operands, 8 x 16x16x32_bf16)" : "fp32 (64 x 16x16x4_f32)            ", waves,
                   (double)c0 / iters, ms, iters);
        }
    return 0;
}
