// misc.hip -- library bookkeeping + the MFMA lane-map self test used by tests/test_gpu_mfma_probe.py.
#include "common.h"

int g_sassd_last_hip_error = 0;

extern "C" const char *sassd_version(void) { return "sassd-mi355x 0.1 (gfx950)"; }
extern "C" int sassd_last_hip_error(void) { return g_sassd_last_hip_error; }
extern "C" const char *sassd_last_hip_error_string(void)
{
    return hipGetErrorString((hipError_t)g_sassd_last_hip_error);
}

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// a32 [32][2*ks] row-major, b32 [2*ks][32] row-major, d32 [32][32];  a16 [16][4*ks], b16 [4*ks][16], d16 [16][16]
__global__ void __launch_bounds__(64) mfma_probe_kernel(const float *a32, const float *b32, float *d32,
                                                        const float *a16, const float *b16, float *d16, int ks)
{
    const int l = threadIdx.x;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int s = 0; s < ks; ++s) {
        const float a = a32[(l & 31) * (2 * ks) + 2 * s + (l >> 5)];       // A[i = l&31][k = l>>5]
        const float b = b32[(2 * s + (l >> 5)) * 32 + (l & 31)];           // B[k = l>>5][j = l&31]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        d32[row * 32 + col] = acc[r];
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < ks; ++s) {
        const float a = a16[(l & 15) * (4 * ks) + 4 * s + (l >> 4)];       // A[i = l&15][k = l>>4]
        const float b = b16[(4 * s + (l >> 4)) * 16 + (l & 15)];           // B[k = l>>4][j = l&15]
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) d16[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
}  // namespace

extern "C" int sassd_mfma_probe(const float *a32, const float *b32, float *d32, const float *a16, const float *b16,
                                float *d16, int ksteps, void *stream_)
{
    if (!a32 || !b32 || !d32 || !a16 || !b16 || !d16 || ksteps < 1) return SASSD_EINVAL;
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, a32, b32, d32, a16, b16, d16,
                       ksteps);
    return sassd_launch_status();
}


// ------------------------------------------------------------------------------------------------------------------
// hipGraph helpers.  Every entry point of this library is a fixed launch sequence on the caller's stream with all
// data-dependent counts in device memory, so a whole frame (side streams joined through events included) can be
// captured once and replayed: ~80 launches issued through a language binding become one hipGraphLaunch.
// ------------------------------------------------------------------------------------------------------------------
extern "C" int sassd_graph_begin(void *stream)
{
    return sassd_hip(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
}

extern "C" int sassd_graph_end(void *stream, void **graph_exec)
{
    if (!graph_exec) return SASSD_EINVAL;
    hipGraph_t g = nullptr;
    int rc = sassd_hip(hipStreamEndCapture((hipStream_t)stream, &g));
    if (rc) return rc;
    hipGraphExec_t ex = nullptr;
    rc = sassd_hip(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    if (rc) return rc;
    *graph_exec = (void *)ex;
    return SASSD_OK;
}

extern "C" int sassd_graph_launch(void *graph_exec, void *stream)
{
    if (!graph_exec) return SASSD_EINVAL;
    return sassd_hip(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
}

extern "C" int sassd_graph_destroy(void *graph_exec)
{
    if (!graph_exec) return SASSD_OK;
    return sassd_hip(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
}
