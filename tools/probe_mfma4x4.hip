// probe_mfma4x4.hip -- empirical check of what spconv_gq.h assumes about v_mfma_f32_4x4x1_16b_f32 on gfx950:
// (1) operand / result lane layout and the CBSZ / ABID broadcast of the A operand, (2) issue rate and dependent-chain
// latency next to v_mfma_f32_16x16x4_f32.   hipcc --offload-arch=gfx950 -O3 -o tools/bin/probe_mfma4x4 tools/probe_mfma4x4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID>
__global__ void layout_kernel(const float *a, const float *b, float *d)
{
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, CBSZ, ABID, 0);
    for (int i = 0; i < 4; ++i) d[i * 64 + l] = c[i];
}

template <int CHAINS, int KIND>      // KIND 0: 4x4x1 (cbsz 4), 1: 16x16x4
__global__ void rate_kernel(float *out, long long *cyc, int iters)
{
    const int l = threadIdx.x & 63;
    float a = (float)l * 1e-3f, b = 1.0f + (float)l * 1e-4f;
    f32x4 c[4];
    for (int i = 0; i < 4; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            constexpr int dummy = 0; (void)dummy;
            if (KIND == 0) c[u % CHAINS] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[u % CHAINS], 4, 3, 0);
            else c[u % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[u % CHAINS], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CBSZ, int ABID>
void layout(const char *name)
{
    float *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    std::vector<float> ha(64), hb(64), hd(256);
    for (int l = 0; l < 64; ++l) ha[l] = (float)(l + 1);
    for (int jb : {0, 5, 38}) {
        for (int l = 0; l < 64; ++l) hb[l] = (l == jb) ? 1.f : 0.f;
        hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL((layout_kernel<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, a, b, d);
        hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
        printf("%s B one-hot at lane %d: nonzero D[vgpr][lane] =", name, jb);
        for (int i = 0; i < 4; ++i)
            for (int l = 0; l < 64; ++l)
                if (hd[i * 64 + l] != 0.f) printf(" [%d][%d]=%g", i, l, hd[i * 64 + l]);
        printf("\n");
    }
    hipFree(a); hipFree(b); hipFree(d);
}

template <int CHAINS, int KIND>
void rate(int threads, const char *name)
{
    float *out; long long *cyc;
    hipMalloc(&out, 4 * 1024 * 1024); hipMalloc(&cyc, 8 * 1024);
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate_kernel<CHAINS, KIND>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((rate_kernel<CHAINS, KIND>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * (threads / 64) * iters * 16.0 * (KIND == 0 ? 512.0 : 2048.0);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0;
    for (auto v : h) m += (double)v;
    m /= blocks;
    // __builtin_readcyclecounter = s_memtime: shader-clock ticks (guide); MFMAs per SIMD = waves per SIMD * iters * 16
    const double wps = threads / 256.0 < 1 ? 1 : threads / 256.0;
    printf("%-34s threads %4d: %.1f ticks per MFMA per wave, %.1f per MFMA per SIMD; wall %.3f ms = %.1f TFLOP/s\n", name,
           threads, m / (iters * 16.0), m / (iters * 16.0 * wps), ms, flop / ms * 1e-9);
    hipFree(out); hipFree(cyc);
}

int main()
{
    layout<0, 0>("cbsz0      ");
    layout<4, 0>("cbsz4 abid0");
    layout<4, 9>("cbsz4 abid9");
    layout<3, 2>("cbsz3 abid2");
    for (int threads : {64, 256, 512}) {
        rate<1, 0>(threads, "4x4x1 cbsz4, 1 chain");
        rate<2, 0>(threads, "4x4x1 cbsz4, 2 chains");
        rate<4, 0>(threads, "4x4x1 cbsz4, 4 chains");
        rate<1, 1>(threads, "16x16x4, 1 chain");
        rate<4, 1>(threads, "16x16x4, 4 chains");
    }
    return 0;
}
