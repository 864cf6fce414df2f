// conv2d_wino4.hip -- 3x3 stride-1 pad-1 convolution of the BEV network through Winograd F(4x4, 3x3):
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A   per 4x4 output tile, summed over input channels
// (mmdet/models/necks/cmn.py:240-262: conv0 320->256 and conv1-6 256->256 at 200x176, 93 % of the dense FLOPs).
//
// fp32 MFMA and fp32 VALU share one 157 TF peak on MI355X, so the only lever past a good direct kernel is fewer
// multiplications.  F(2x2,3x3) (conv2d_wino.hip) needs 16 per 4 outputs; F(4x4,3x3) needs 36 per 16 outputs: 4x
// fewer than the direct convolution, 1.78x fewer than F(2x2).  The fused F(2x2) kernel is bound by the VALU work of
// its in-loop input transform (every transformed value feeds only 32 output channels); here the transforms are taken
// OUT of the contraction, so that the hot loop is a plain batched GEMM with no VALU work at all:
//   1. wino4_in_kernel    V[p][ci][t] = (B^T d B)[p]         x [B,Cin,H,W] -> V [36][Cin][Tp]       (HBM stream)
//   2. wino4_gemm_kernel  M[p][co][t] = sum_ci U[p][ci][co] V[p][ci][t]     36 GEMMs 256 x Cin x T   (MFMA)
//   3. wino4_out_kernel   Y = A^T M A, folded BatchNorm / bias / ReLU        M [36][Cout][Tp] -> y    (HBM stream)
// with U = G g G^T packed once per weight update as [36][Cin][Cout].  T = B * (H/4) * (W/4) tiles (2200 per KITTI
// BEV map), padded to a multiple of the GEMM's 64-column block.  Per 256->256 layer: 10.4 GFLOP on the MFMA (41.5
// direct), 81 MB of V and M each; both transforms are elementwise streams.
// Numerics: F(4x4,3x3) amplifies fp32 rounding (cancellation in the transforms); the interpolation points were chosen for
// it (below): ~8x the max-abs error of the direct convolution on a 256-channel layer instead of ~40x with the textbook
// points -- inside the 2e-4 BEV-feature and 1e-4 box parity bars (tests/test_gpu_pipeline.py prints the measured errors).
//
// GEMM kernel, round 4: the fp32 products run on the bf16 MFMA (16 x the fp32 MFMA's rate) over operands split exactly into
// three bf16 pieces in registers -- half the MFMA cycles per fp32 product, the error of the fp32 instruction (2.1e-5 against
// fp64 on a KITTI layer for both), operands still fp32 in HBM and LDS: see "fp32 products on the bf16 MFMA" below.  Workgroup
// = 128 output channels x 128 tiles, four waves of 64 x 64.  The fp32-MFMA form (geometry 1..6 of the cfg word,
// the default of rounds 2-3): workgroup = 128 output channels x 32 WN tiles of one Winograd position, 2 x WN waves, each a 64 x 32
// block = two v_mfma_f32_32x32x2_f32 accumulators sharing one B fragment.  A ([k][128 co]) and B ([k][32 WN t]) chunks
// of 32 input channels are staged by global->LDS DMA in their natural row-major form (both operands are K-major with
// the M / N index contiguous, so fragment reads are 32 consecutive dwords: conflict free), double buffered, one
// barrier per chunk; two workgroups per CU.  WN is picked per launch so that the last round of workgroups is full
// (w4_pick_wn).  Work order: all tile blocks of one (position, channel half) run back to back on ONE XCD
// (blockIdx % 8), so V[p] crosses the fabric twice and U[p] once per layer.  Measured (B=1, 256->256 @200x176):
// MFMA part ~70 % of the fp32 peak, bounded by the workgroup rounds; the two transforms run at HBM speed.
#include "common.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

constexpr int kBM = 128, kKC = 32;           // channel-block / chunk granularity every GEMM geometry divides

// Interpolation points {0, +-5/8, +-3/2, inf} (tools/gen_wino4_matrices.py prints the matrices): in fp32 they give 4.7x
// less max-abs error than the textbook {0, +-1, +-2} (7.4e-6 vs 3.5e-5 on a 256-channel layer with |y| ~ 3; direct 9e-7).
// ---- weights: U[p][ci][co] = (G g G^T)[p],  G = [[1,0,0],[1,5/8,25/64],[1,-5/8,25/64],[1,3/2,9/4],[1,-3/2,9/4],[0,0,1]]
__device__ __forceinline__ void g_row(const float g0, const float g1, const float g2, float (&o)[6])
{
    o[0] = g0;
    const float e = g0 + 0.390625f * g2, f = 0.625f * g1;
    o[1] = e + f;
    o[2] = e - f;
    const float e2 = g0 + 2.25f * g2, f2 = 1.5f * g1;
    o[3] = e2 + f2;
    o[4] = e2 - f2;
    o[5] = g2;
}

__global__ void wino4_pack_kernel(const float *__restrict__ w, int Cout, int Cin, float *__restrict__ U, int ldu)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cout * Cin) return;
    const int co = i % Cout, ci = i / Cout;
    const float *g = w + ((size_t)co * Cin + ci) * 9;
    float t[3][6];                                    // columns of g transformed along the rows: (G g)^T
#pragma unroll
    for (int c = 0; c < 3; ++c) g_row(g[c], g[3 + c], g[6 + c], t[c]);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        float o[6];
        g_row(t[0][r], t[1][r], t[2][r], o);
#pragma unroll
        for (int c = 0; c < 6; ++c) U[((size_t)(r * 6 + c) * Cin + ci) * ldu + co] = o[c];
    }
}

// ---- input transform: B^T = [[1,0,-676/225,0,256/225,0],[0,576/595,4608/2975,-256/595,-2048/2975,0],
//   [0,-576/595,4608/2975,256/595,-2048/2975,0],[0,-25/357,-50/1071,64/357,128/1071,0],
//   [0,25/357,-50/1071,-64/357,128/1071,0],[0,225/256,0,-169/64,0,1]]
__device__ __forceinline__ void bt_vec(const float (&d)[6], float (&o)[6])
{
    o[0] = d[0] - 3.00444444f * d[2] + 1.13777778f * d[4];
    const float s = 1.54890756f * d[2] - 0.688403361f * d[4], t = 0.968067227f * d[1] - 0.430252101f * d[3];
    o[1] = s + t;
    o[2] = s - t;
    const float u = 0.119514472f * d[4] - 0.0466853408f * d[2], v = 0.179271709f * d[3] - 0.0700280112f * d[1];
    o[3] = u + v;
    o[4] = u - v;
    o[5] = 0.87890625f * d[1] - 2.640625f * d[3] + d[5];
}

struct W4Geom { int B, C, H, W, TH, TW, T, Tp; };
constexpr int kTmapHead = 4;       // tile map: [0] active tiles, [1..3] reserved, then tpos[T] (column of tile t or -1), tlist[T]

// one thread = one (channel, tile): 36 loads (neighbouring threads' patches overlap: L1); the 36 x 256 results of a
// workgroup go through LDS so that every plane row leaves as 16-byte stores (dword stores cost ~6x more per byte)
// `tmap` (optional, sassd_wino4_tile_map): only the ACTIVE tiles are transformed, column j of V = tile tlist[j]
__global__ void __launch_bounds__(256) wino4_in_kernel(const float *__restrict__ x, W4Geom G, float *__restrict__ V,
                                                       const int32_t *__restrict__ tmap)
{
    __shared__ __attribute__((aligned(16))) float tr[36 * 256];
    const int t0 = blockIdx.x * 256;
    const int nt = tmap ? tmap[0] : G.T;                  // columns in use
    if (t0 >= nt) return;                                 // (uniform)
    const int j = t0 + threadIdx.x;
    const int c = blockIdx.y;
    if (j < nt) {
        const int t = tmap ? tmap[kTmapHead + G.T + j] : j;
        const int tpi = G.TH * G.TW;
        const int b = t / tpi, r = t - b * tpi;
        const int ty = r / G.TW, tx = r - ty * G.TW;
        const float *src = x + ((size_t)b * G.C + c) * G.H * G.W;
        const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
        float d[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int yy = y0 + i;
            const bool rok = yy >= 0 && yy < G.H;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int xx = x0 + j;
                d[i][j] = (rok && xx >= 0 && xx < G.W) ? src[(size_t)yy * G.W + xx] : 0.f;
            }
        }
        float u[6][6];                                    // u[j] = B^T (column j of d)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
            float o[6];
            bt_vec(col, o);
#pragma unroll
            for (int i = 0; i < 6; ++i) u[i][j] = o[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float o[6];
            bt_vec(u[i], o);                              // (B^T d) B = rows transformed again
#pragma unroll
            for (int j = 0; j < 6; ++j) tr[(i * 6 + j) * 256 + threadIdx.x] = o[j];
        }
    }
    __syncthreads();
    // 36 planes x 64 float4: columns up to the padded width are written (the padding only feeds padding columns of M)
    const size_t plane = (size_t)G.C * G.Tp;
    const int c4 = threadIdx.x & 63;
    if (t0 + c4 * 4 < G.Tp) {
        float *dst = V + (size_t)c * G.Tp + t0 + c4 * 4;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int p = (threadIdx.x >> 6) + 4 * k;
            *reinterpret_cast<float4 *>(dst + (size_t)p * plane) = *reinterpret_cast<const float4 *>(tr + p * 256 + c4 * 4);
        }
    }
}

// ---- output transform: A^T = [[1,1,1,1,1,0],[0,5/8,-5/8,3/2,-3/2,0],[0,25/64,25/64,9/4,9/4,0],
//                                  [0,125/512,-125/512,27/8,-27/8,1]]
__device__ __forceinline__ void at_vec(const float (&m)[6], float (&o)[4])
{
    const float s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4];
    o[0] = m[0] + s1 + s2;
    o[1] = 0.625f * d1 + 1.5f * d2;
    o[2] = 0.390625f * s1 + 2.25f * s2;
    o[3] = 0.244140625f * d1 + 3.375f * d2 + m[5];
}

// (CoutP: channel count of the product planes -- Cout, or Cout padded to the GEMM's channel block for a narrow layer)
__global__ void __launch_bounds__(256) wino4_out_kernel(const float *__restrict__ M, W4Geom G, int Cout, int CoutP,
                                                        const float *__restrict__ scale,
                                                        const float *__restrict__ shift, int relu,
                                                        float *__restrict__ y, const int32_t *__restrict__ tmap)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int co = blockIdx.y;
    if (t >= G.T) return;
    const size_t plane = (size_t)CoutP * G.Tp;
    const int col = tmap ? tmap[kTmapHead + t] : t;       // compacted launch: an inactive tile's products are exactly zero
    const float *src = M + (size_t)co * G.Tp + (col < 0 ? 0 : col);
    float v[4][6];                                    // A^T m : 4 rows x 6 columns
    {
        float m[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) m[i][j] = col < 0 ? 0.f : src[(size_t)(i * 6 + j) * plane];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float col[6] = {m[0][j], m[1][j], m[2][j], m[3][j], m[4][j], m[5][j]};
            float o[4];
            at_vec(col, o);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i][j] = o[i];
        }
    }
    const int tpi = G.TH * G.TW;
    const int b = t / tpi, r = t - b * tpi;
    const int ty = r / G.TW, tx = r - ty * G.TW;
    const float sc = scale ? scale[co] : 1.f, sh = shift ? shift[co] : 0.f;
    float *dst = y + (((size_t)b * Cout + co) * G.H + 4 * ty) * G.W + 4 * tx;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float o[4];
        at_vec(v[i], o);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = o[j] * sc + sh;
            if (relu) o[j] = fmaxf(o[j], 0.f);
        }
        *reinterpret_cast<float4 *>(dst + (size_t)i * G.W) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ---- fused output -> input transform between two chained 3x3 layers ----------------------------------------------------
// Layer l's output transform (A^T M A, folded BatchNorm / ReLU) and layer l+1's input transform (B^T d B) in ONE kernel:
// the activation map of the layer in between is never written to HBM (36 MB written + 40 MB read back per 256-channel
// KITTI map, plus a launch).  One workgroup owns one (channel, image) plane: phase 1 rebuilds the plane's H x W
// activations in LDS (zero border = the convolution's padding) from the 36 product planes, phase 2 reads every 6x6 patch
// back and writes the 36 transformed planes -- four consecutive tiles per thread, so that every plane row leaves as
// 16-byte stores.  LDS: (H + 2) x (W + 2, rounded up to 4) floats = 142 KB for the 200 x 176 KITTI map (one workgroup per
// CU; 256 channels x batch workgroups).
template <int NT>
__global__ void __launch_bounds__(NT) wino4_outin_kernel(const float *__restrict__ M, W4Geom G,
                                                          const float *__restrict__ scale,
                                                          const float *__restrict__ shift, int relu, int pitch,
                                                          float *__restrict__ V, const int32_t *__restrict__ tmap,
                                                          float *__restrict__ y_prev)
{
    extern __shared__ __attribute__((aligned(16))) float w4_plane[];
    const int c = blockIdx.x, b = blockIdx.y;
    const int tpi = G.TH * G.TW;
    const int nflt = (G.H + 2) * pitch;
    for (int i = threadIdx.x; i < nflt / 4; i += NT) ((float4 *)w4_plane)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const size_t plane = (size_t)G.C * G.Tp;
    const float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
    // ---- phase 1: A^T M A + epilogue -> LDS plane (pixel (y, x) at [(y + 1) * pitch + x + 1]) ------------------------
    for (int t = threadIdx.x; t < tpi; t += NT) {
        // products of an inactive tile of the previous layer's (compacted) launch are exactly zero: nothing to load
        const int col = tmap ? tmap[kTmapHead + b * tpi + t] : b * tpi + t;
        const float *src = M + (size_t)c * G.Tp + (col < 0 ? 0 : col);
        float v[4][6];
        {
            float m[6][6];
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) m[i][j] = col < 0 ? 0.f : src[(size_t)(i * 6 + j) * plane];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float col[6] = {m[0][j], m[1][j], m[2][j], m[3][j], m[4][j], m[5][j]};
                float o[4];
                at_vec(col, o);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i][j] = o[i];
            }
        }
        const int ty = t / G.TW, tx = t - ty * G.TW;
        float *dst = w4_plane + (4 * ty + 1) * pitch + 4 * tx + 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float o[4];
            at_vec(v[i], o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = o[j] * sc + sh;
                if (relu) o[j] = fmaxf(o[j], 0.f);
                dst[i * pitch + j] = o[j];
            }
        }
    }
    __syncthreads();
    // (round 6) the previous layer's activation map for its OTHER consumers (conv6 feeds conv7 and the part-sensitive head):
    // the plane is in LDS anyway -- one 16-byte store per pixel quad instead of a separate output-transform launch
    if (y_prev) {
        float *yp = y_prev + ((size_t)b * G.C + c) * G.H * G.W;
        const int w4 = G.W >> 2;
        for (int i = threadIdx.x; i < G.H * w4; i += NT) {
            const int yy = i / w4, x4 = (i - yy * w4) * 4;
            const float *src = w4_plane + (yy + 1) * pitch + x4 + 1;
            *reinterpret_cast<float4 *>(yp + (size_t)yy * G.W + x4) = make_float4(src[0], src[1], src[2], src[3]);
        }
    }
    // ---- phase 2: B^T d B of every tile of this image, quads of consecutive GLOBAL tile indices -----------------------
    const int g0 = b * tpi, g1 = g0 + tpi;
    const int q0 = g0 >> 2, q1 = (g1 + 3) >> 2;
    float *Vc = V + (size_t)c * G.Tp;
    for (int q = q0 + threadIdx.x; q < q1; q += NT) {
        float out[4][36];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int tg = 4 * q + k;
            ok[k] = tg >= g0 && tg < g1;
            const int tl = ok[k] ? tg - g0 : 0;
            const int ty = tl / G.TW, tx = tl - ty * G.TW;
            const float *pp = w4_plane + 4 * ty * pitch + 4 * tx;        // patch row i, column j: pixel (4ty-1+i, 4tx-1+j)
            float d[6][6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float4 a = *(const float4 *)(pp + i * pitch);
                const float2 e = *(const float2 *)(pp + i * pitch + 4);
                d[i][0] = a.x; d[i][1] = a.y; d[i][2] = a.z; d[i][3] = a.w; d[i][4] = e.x; d[i][5] = e.y;
            }
            float u[6][6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
                float o[6];
                bt_vec(col, o);
#pragma unroll
                for (int i = 0; i < 6; ++i) u[i][j] = o[i];
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                float o[6];
                bt_vec(u[i], o);
#pragma unroll
                for (int j = 0; j < 6; ++j) out[k][i * 6 + j] = o[j];
            }
        }
        float *dst = Vc + 4 * (size_t)q;
        if (ok[0] && ok[1] && ok[2] && ok[3]) {
#pragma unroll
            for (int p = 0; p < 36; ++p)
                *reinterpret_cast<float4 *>(dst + (size_t)p * plane) = make_float4(out[0][p], out[1][p], out[2][p], out[3][p]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (ok[k]) {
#pragma unroll
                    for (int p = 0; p < 36; ++p) dst[(size_t)p * plane + k] = out[k][p];
                }
        }
    }
}

// 512 threads (8 waves, one workgroup per CU: 217 VGPRs leave room for exactly that) -- the kernel is bound by the
// memory requests one CU keeps in flight, and the 550 tile quads of a KITTI plane are 2 rounds of 512 threads instead of 3
// rounds of 256.  dbg bit 8 (cfg bit 16): the 256-thread form of rounds 3-5, for the A/B.
static int launch_outin(int dbg, int Cin, int batch, size_t lds, hipStream_t stream, const float *M, W4Geom G,
                        const float *scale, const float *shift, int relu, int pitch, float *V, const int32_t *tmap,
                        float *y_prev)
{
    static std::atomic<unsigned long long> done512{0}, done256{0};
    if (dbg & 256) {
        int rc = sassd_dyn_lds((const void *)wino4_outin_kernel<256>, (size_t)160 * 1024, done256);   // once, for any plane
        if (rc) return rc;
        hipLaunchKernelGGL(wino4_outin_kernel<256>, dim3(Cin, batch), dim3(256), lds, stream, M, G, scale, shift, relu, pitch,
                           V, tmap, y_prev);
    } else {
        int rc = sassd_dyn_lds((const void *)wino4_outin_kernel<512>, (size_t)160 * 1024, done512);
        if (rc) return rc;
        hipLaunchKernelGGL(wino4_outin_kernel<512>, dim3(Cin, batch), dim3(512), lds, stream, M, G, scale, shift, relu, pitch,
                           V, tmap, y_prev);
    }
    return SASSD_OK;
}

// ---- the 36 GEMMs ---------------------------------------------------------------------------------------------------
// NP independent GEMMs  M_p [Cout x N] = U_p^T [Cout x Cin] . V_p [Cin x N]  with K-major operands (U_p [Cin][Cout],
// V_p [Cin][ldv], M_p [Cout][ldm]).  Winograd: p = position, 36 of them.  A 1x1 convolution over NCHW is the same
// problem with p = image, U shared (su = 0), V_p = x[p] [Cin][H*W], M_p = y[p] -- plus the per-channel epilogue.
struct W4Gemm {
    const float *U, *V;
    float *M;
    const float *scale, *shift;      // optional per-output-channel epilogue (1x1 convolution use)
    size_t su, sv, sm;               // per-problem strides (floats)
    int np, Cin, Cout, ldv, ldm, ncols, relu;
    int nmb, nnb;            // channel blocks, column blocks
    int nseg, seglen;        // every (problem, channel block) pair is cut into nseg runs of seglen column blocks
    int pairs_per_xcd;       // (pair, run) items per XCD
    int dbg;                 // ablation: bit0 stage only the first chunk, bit1 no MFMA
    const int32_t *ncols_dev;  // optional: columns in use (device count; column blocks past it leave at once)
};

// ---- fp32 products on the bf16 MFMA (SPLIT = 1) ------------------------------------------------------------------------
// The fp32 MFMA runs at 1/16 of the bf16 MFMA's rate.  An fp32 value is EXACTLY the sum of three bf16 values (8 + 8 + 8
// significand bits, split by truncation: a1 = top 16 bits of a, a2 = top 16 bits of a - a1, a3 = a - a1 - a2, every
// subtraction exact), and a product of two bf16 values is exact in fp32.  v_mfma_f32_32x32x16_bf16 gives every lane 8
// K-slots: lane (row, k-half) fills them with ONE channel's pieces -- eight of the nine piece products, every pair except
// a3 w3 (the exact slot order is the one split_a / split_b below implement and document) --
// so that one instruction (32 cycles for two channels; the fp32 form needs 64) accumulates a w - a3 w3, i.e. the fp32
// product to 2^-30, in fp32 -- the operands stay fp32 in HBM and LDS, the fragment reads are those of the fp32 kernel, and
// the split costs 8 VALU instructions per fragment value (two masks, two subtractions, four byte permutes) that run
// beside the other wave's MFMAs.  tools/probe_bf16x3.hip measures the error against fp64 next to the fp32 MFMA's.
__device__ __forceinline__ void split3(const float a, unsigned &u0, unsigned &u1, unsigned &u2)
{
    u0 = __float_as_uint(a);
    const float r1 = a - __uint_as_float(u0 & 0xffff0000u);
    u1 = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
    u2 = __float_as_uint(r2);                                   // (its low 16 bits are zero: 8 significant bits are left)
}
// v_perm_b32: [hi16(lo) | hi16(hi) << 16]
__device__ __forceinline__ unsigned hi_pair(const unsigned lo, const unsigned hi)
{
    return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
}
// K-slot convention (8 slots of one channel, low half of dword 0 first):
//     A side  a1 a2 | a3 a1 | a1 a2 | a2 a3        B side  w1 w1 | w1 w2 | w3 w2 | w3 w2
// i.e. every product of pieces except a3 w3.  (The A dwords are S0 = [a1|a2], S1 = [a3|a1], S0 again and [a2|a3]: weights
// stored pre-split as the 8 bytes (S0, S1) need 4 VALU instructions instead of 8 -- built and measured in round 4: the
// doubled weight stream through the LDS DMA costs what the VALU saves)
__device__ __forceinline__ bf16x8 split_a(const float a)
{
    unsigned u0, u1, u2;
    split3(a, u0, u1, u2);
    u32x4 d;
    d[0] = hi_pair(u0, u1); d[1] = hi_pair(u2, u0); d[2] = d[0]; d[3] = hi_pair(u1, u2);
    return __builtin_bit_cast(bf16x8, d);
}
__device__ __forceinline__ bf16x8 split_b(const float w)
{
    unsigned u0, u1, u2;
    split3(w, u0, u1, u2);
    u32x4 d;
    d[0] = hi_pair(u0, u0); d[1] = hi_pair(u0, u1); d[2] = hi_pair(u2, u1); d[3] = d[2];
    return __builtin_bit_cast(bf16x8, d);
}

// scheduling groups of one pipelined step: MFMA m, then the split arithmetic (8 VALU instructions per value) of its share
// of the next step's NV fragment values
template <int M, int NM, int NV>
__device__ __forceinline__ void split_sched()
{
    if constexpr (M < NM) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        constexpr int n = (M + 1) * NV / NM - M * NV / NM;
        if constexpr (n > 0) __builtin_amdgcn_sched_group_barrier(0x002, 8 * n, 0);
        split_sched<M + 1, NM, NV>();
    }
}

// WM x WN waves; a wave owns a 64 x (32 NB) block = 2 x NB accumulators of v_mfma_f32_32x32x2_f32 (SPLIT = 0) or of
// v_mfma_f32_32x32x16_bf16 over split operands (SPLIT = 1; 2: ablation without the split arithmetic).
template <int WM, int WN, int NB, int KC, int SPLIT>
__global__ void __launch_bounds__(WM * WN * 64) wino4_gemm_kernel(W4Gemm P)
{
    constexpr int MT = 2;
    constexpr int NWV = WM * WN, WROWS = 32 * MT, BM = WROWS * WM, BN = 32 * NB * WN;
    constexpr int STAGE = (BM + BN) * KC;
    constexpr int NLA = KC * BM / 256, NLB = KC * BN / 256;          // 1 KB wave-loads per chunk
    constexpr int LPW = (NLA + NLB + NWV - 1) / NWV;                 // ... per wave
    extern __shared__ __attribute__((aligned(16))) float w4_lds[];   // [2][A KC x BM | B KC x BN]
    float *lds = w4_lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-local order: blockIdx % 8 = XCD; inside an XCD the tile blocks of one (position, channel block) pair are
    // consecutive, so concurrently running workgroups share U[p] (and V[p] with the other channel blocks' XCDs only)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int q = j / P.seglen, within = j - q * P.seglen;
    const int item = xcd * P.pairs_per_xcd + q;
    if (item >= P.np * P.nmb * P.nseg) return;
    const int pair = item / P.nseg, nb = (item - pair * P.nseg) * P.seglen + within;
    if (nb >= P.nnb) return;
    if (P.ncols_dev && nb * BN >= *P.ncols_dev) return;                    // (uniform) compacted launch: unused column block
    const int p = pair / P.nmb, mb = pair - p * P.nmb;
    const float *Ub = P.U + (size_t)p * P.su + mb * BM;                    // + k * Cout
    const float *Vb = P.V + (size_t)p * P.sv + nb * BN;                    // + k * ldv
    // DMA map: wave-load id t = wave + NWV * i; t < NLA: A rows (BM/4 16-byte pieces per row), else B rows.  The LDS
    // destination of a wave-load is lane-linear, i.e. the natural row-major [k][m] / [k][n] image.
    const float *src[LPW];
    size_t kstride[LPW];
    int dst[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int t = wave + NWV * i;
        if (t < NLA) {
            const int e = t * 64 + lane;
            src[i] = Ub + (size_t)(e / (BM / 4)) * P.Cout + (e % (BM / 4)) * 4;
            kstride[i] = (size_t)KC * P.Cout;
            dst[i] = t * 256;
        } else {
            const int e = (t - NLA) * 64 + lane;
            src[i] = Vb + (size_t)(e / (BN / 4)) * P.ldv + (e % (BN / 4)) * 4;
            kstride[i] = (size_t)KC * P.ldv;
            dst[i] = BM * KC + (t - NLA) * 256;
        }
    }
    auto dma = [&](int chunk, float *stage) {
#pragma unroll
        for (int i = 0; i < LPW; ++i)
            if (wave + NWV * i < NLA + NLB)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[i] + chunk * kstride[i]), (lds_ptr_t)(stage + dst[i]), 16,
                                                 0, 0);
    };
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, kh = lane >> 5;
    f32x16 acc[MT][NB];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int nchunk = P.Cin / KC;
    dma(0, lds);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        float *cur = lds + (c & 1) * STAGE;
        if (c + 1 < nchunk && !(P.dbg & 1)) dma(c + 1, lds + ((c + 1) & 1) * STAGE);
        const float *As = cur + kh * BM + wm * WROWS + l31;                   // + 2*kk*BM
        const float *Bs = cur + BM * KC + kh * BN + wn * 32 * NB + l31;       // + 2*kk*BN
        // (no run-time switches inside the split loop: a second path makes the compiler park the accumulators in the other
        // register file across the chunk loop, 128 v_accvgpr moves per chunk)
        if constexpr (SPLIT != 0) {
            if constexpr (SPLIT == 1) {
                // Software pipeline inside the chunk: the fragment values of step kk + 1 are read and split while the MFMAs of
                // step kk issue, one share of the split arithmetic behind every MFMA (the compiler's own order is all MFMAs
                // of a step back to back, then all the VALU work: a wave cannot issue past its own queued MFMA, so nothing of
                // its split overlaps its MFMAs).
                constexpr int NK = KC / 2, NV = MT + NB, NM = MT * NB;
                float raw[NV];
                bf16x8 fr[2][NV];
                auto rd = [&](const int kk) {
#pragma unroll
                    for (int a = 0; a < MT; ++a) raw[a] = As[2 * kk * BM + 32 * a];
#pragma unroll
                    for (int b = 0; b < NB; ++b) raw[MT + b] = Bs[2 * kk * BN + 32 * b];
                };
                auto mk = [&](const int v) -> bf16x8 { return v < MT ? split_a(raw[v]) : split_b(raw[v]); };
                rd(0);
#pragma unroll
                for (int v = 0; v < NV; ++v) fr[0][v] = mk(v);
#pragma unroll
                for (int kk = 0; kk < NK; ++kk) {
                    const int cur = kk & 1, nxt = cur ^ 1;
                    const bool more = kk + 1 < NK;
                    if (more) rd(kk + 1);
#pragma unroll
                    for (int b = 0; b < NB; ++b)
#pragma unroll
                        for (int a = 0; a < MT; ++a) {
                            const int m = b * MT + a;
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[cur][a], fr[cur][MT + b], acc[a][b], 0, 0, 0);
                            if (more) {
#pragma unroll
                                for (int v = m * NV / NM; v < (m + 1) * NV / NM; ++v) fr[nxt][v] = mk(v);
                            }
                        }
                    if (more) split_sched<0, NM, NV>();
                }
            } else {                                    // ablation (SPLIT = 2): the same reads and MFMAs without the split arithmetic
#pragma unroll
                for (int kk = 0; kk < KC / 2; ++kk) {
                    bf16x8 av[MT];
#pragma unroll
                    for (int a = 0; a < MT; ++a) {
                        const unsigned xa = __float_as_uint(As[2 * kk * BM + 32 * a]);
                        const u32x4 d = {xa, xa, xa, xa};
                        av[a] = __builtin_bit_cast(bf16x8, d);
                    }
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const unsigned xb = __float_as_uint(Bs[2 * kk * BN + 32 * b]);
                        const u32x4 db = {xb, xb, xb, xb};
#pragma unroll
                        for (int a = 0; a < MT; ++a)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[a], __builtin_bit_cast(bf16x8, db), acc[a][b],
                                                                                0, 0, 0);
                    }
                }
            }
        } else if (!(P.dbg & 2)) {
#pragma unroll
            for (int kk = 0; kk < KC / 2; ++kk) {
                float av[MT];
#pragma unroll
                for (int a = 0; a < MT; ++a) av[a] = As[2 * kk * BM + 32 * a];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const float bv = Bs[2 * kk * BN + 32 * b];
#pragma unroll
                    for (int a = 0; a < MT; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv, acc[a][b], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                // (the compiler drains vmcnt: chunk c+1 has landed)
    }
    // Epilogue through LDS (the staging buffers are dead): D[row = (r & 3) + 8 * (r >> 2) + 4 * kh][col = l31] goes to the
    // wave's [32 MT][32 NB] image, then 16-byte row stores (a dword store per accumulator register is ~6x slower per byte)
    constexpr int WCOLS = 32 * NB;
    float *img = lds + wave * WROWS * WCOLS;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                img[(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * WCOLS + b * 32 + l31] = acc[a][b][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private image: no barrier needed
    const int co0 = mb * BM + wm * WROWS;
    float *Mb = P.M + (size_t)p * P.sm + (size_t)co0 * P.ldm + nb * BN + wn * WCOLS;
    constexpr int C4 = WCOLS / 4, RPI = 64 / C4;        // float4 per row, rows per wave-instruction
    const bool epi = P.scale || P.shift || P.relu;
#pragma unroll
    for (int i = 0; i < WROWS / RPI; ++i) {
        const int row = i * RPI + lane / C4, c4 = lane % C4;
        float4 v = *reinterpret_cast<const float4 *>(img + row * WCOLS + c4 * 4);
        if (epi) {
            const float sc = P.scale ? P.scale[co0 + row] : 1.f, sh = P.shift ? P.shift[co0 + row] : 0.f;
            v.x = v.x * sc + sh; v.y = v.y * sc + sh; v.z = v.z * sc + sh; v.w = v.w * sc + sh;
            if (P.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        }
        *reinterpret_cast<float4 *>(Mb + (size_t)row * P.ldm + c4 * 4) = v;
    }
}

// `cfg` of the Winograd entry points (0 in production): bits 0-7 = GEMM geometry (w4_tile), bits 8.. = ablation / stage-skip
// flags (listed at w4_dbg below).  Rounds 2-5 kept both in process-wide ints behind sassd_debug_set_wino4; a per-call word
// cannot be baked into a hipGraph by accident.
inline int w4_geo(int cfg) { return cfg & 0xff; }
inline int w4_dbg(int cfg) { return cfg >> 8; }

template <int WM, int WN, int NB, int KC, int SPLIT = 0>
int launch_w4_gemm(W4Gemm P, int dbg, hipStream_t stream)
{
    constexpr int BM = 64 * WM, BN = 32 * NB * WN;
    static_assert(BM == kBM || (WM == 1 && SPLIT != 0), "the channel block is fixed (w4_pick_wn and the supported() checks price "
                                                        "it); the one exception is the 64-channel block of a narrow tail layer");
    constexpr size_t stage_b = (size_t)2 * (BM + BN) * KC * 4, img_b = (size_t)WM * WN * 64 * 32 * NB * 4;
    constexpr size_t lds = stage_b > img_b ? stage_b : img_b;        // the epilogue image reuses the staging buffers
    if constexpr (SPLIT == 1)
        if (dbg & 4) return launch_w4_gemm<WM, WN, NB, KC, 2>(P, dbg, stream);      // ablation instance
    static std::atomic<unsigned long long> attr_done{0};
    const void *fn = (const void *)wino4_gemm_kernel<WM, WN, NB, KC, SPLIT>;
    int rc = sassd_dyn_lds(fn, lds, attr_done);
    if (rc) return rc;
    P.nmb = P.Cout / BM; P.nnb = P.ncols / BN;
    // few pairs (a 1x1 convolution of one image has two): cut each pair's column blocks into runs so that all 8 XCDs work
    const int pairs = P.np * P.nmb;
    P.nseg = pairs >= 32 ? 1 : cdiv(32, pairs);
    if (P.nseg > P.nnb) P.nseg = P.nnb;
    P.seglen = cdiv(P.nnb, P.nseg);
    P.pairs_per_xcd = cdiv(pairs * P.nseg, 8);
    P.dbg = dbg;
    hipLaunchKernelGGL((wino4_gemm_kernel<WM, WN, NB, KC, SPLIT>), dim3(8 * P.pairs_per_xcd * P.seglen), dim3(WM * WN * 64),
                       lds, stream, P);
    return sassd_launch_status();
}

// Tile-block width of the fp32-MFMA GEMM (32 WN columns, WN waves across): all workgroups do equal work, so the launch takes
// ceil(units / resident slots) rounds -- pick the width whose LAST round is fullest.  KITTI B=1 (2200 tiles): 160
// columns -> 1008 units = 1.97 rounds of 512 slots (64 / 96 / 128 / 192 columns all waste 17 %).
inline int w4_pick_wn(int T, int Cout, int np = 36, bool exact = false)
{
    int best = exact ? 0 : 4;
    double best_cost = 1e30;
    for (int wn = 2; wn <= 6; ++wn) {
        const int bn = 32 * wn;
        if (exact && T % bn) continue;                // (1x1 convolution: no padding columns in an NCHW tensor)
        const size_t lds = (size_t)2 * (kBM + bn) * kKC * 4;
        int wgs = (int)((size_t)160 * 1024 / lds);
        if (wgs > 32 / (2 * wn)) wgs = 32 / (2 * wn);
        if (wgs < 1) wgs = 1;
        const long units = (long)np * (Cout / kBM) * cdiv(T, bn);
        const long rounds = (units + 256L * wgs - 1) / (256L * wgs);
        const double cost = (double)rounds * wgs * bn;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = wn; }
    }
    return best;
}

// Geometry of the Winograd GEMM launch.  Default: fp32 products on the bf16 MFMA over split operands, 128 channels x 128
// tiles per workgroup, four waves of 64 x 64 (measured at 2200 tiles: 142 us per 256 -> 256 layer with its two transform
// launches against 157 for the best fp32-MFMA geometry; 64 columns 160, 192 columns 170).  geometry (cfg bits 0-7): 0 = default, 1 = the
// fp32 MFMA at its picked width (the round-3 default), 2..6 = the fp32 MFMA with 32 cfg columns, 11..13 = split with
// 64 (cfg - 10) columns, 14 = split, 128 columns, 16-channel chunks.
struct W4Tile {
    int split, wn, nb, kc;
    int bn() const { return 32 * nb * wn; }
};
inline W4Tile w4_tile(int T, int Cout, int geo)
{
    const int c = geo;
    if (c >= 2 && c <= 6) return W4Tile{0, c, 1, kKC};
    if (c >= 11 && c <= 13) return W4Tile{1, c - 10, 2, kKC};
    if (c == 14) return W4Tile{1, 2, 2, 16};
    if (c == 1) return W4Tile{0, w4_pick_wn(T, Cout), 1, kKC};
    return W4Tile{1, 2, 2, kKC};
}

inline int w4_tiles_padded(int B, int H, int W, int Cout, int geo)
{
    const int T = B * (H / 4) * (W / 4);
    const int bn = w4_tile(T, Cout, geo).bn();
    return cdiv(T, bn) * bn;
}

inline int w4_launch(const W4Tile t, const W4Gemm &P, int dbg, hipStream_t stream)
{
    if (t.split) {
        if (t.kc == 16) return launch_w4_gemm<2, 2, 2, 16, 1>(P, dbg, stream);
        switch (t.wn) {
        case 1: return launch_w4_gemm<2, 1, 2, kKC, 1>(P, dbg, stream);
        case 2: return launch_w4_gemm<2, 2, 2, kKC, 1>(P, dbg, stream);
        default: return launch_w4_gemm<2, 3, 2, kKC, 1>(P, dbg, stream);
        }
    }
    switch (t.wn) {                                                      // 128 channels x 32 WN tiles, 2 x WN waves of 64 x 32
    case 2: return launch_w4_gemm<2, 2, 1, kKC>(P, dbg, stream);
    case 3: return launch_w4_gemm<2, 3, 1, kKC>(P, dbg, stream);
    case 4: return launch_w4_gemm<2, 4, 1, kKC>(P, dbg, stream);
    case 5: return launch_w4_gemm<2, 5, 1, kKC>(P, dbg, stream);
    default: return launch_w4_gemm<2, 6, 1, kKC>(P, dbg, stream);
    }
}

}  // namespace

// cfg bits 8.. (ablation, tools/run_wino4.py; per-kernel timing on live buffers, bench.py): bit0 stage only the first chunk,
// bit1 no MFMA (fp32-MFMA geometries), bit2 no split arithmetic (split geometries), bits 4 / 5 / 6 / 7 skip the input transform /
// the GEMM / the output transform / the fused output -> input transform
extern "C" int sassd_conv2d_wino4_supported(int Cin, int Cout, int H, int W)
{
    return (Cin >= kKC && Cin % kKC == 0 && Cout >= 256 && Cout % 256 == 0 && H >= 4 && H % 4 == 0 && W >= 4 && W % 4 == 0)
               ? 1 : 0;
}

extern "C" size_t sassd_conv2d_wino4_packed_floats(int Cin, int Cout)
{
    if (Cin < 1 || Cout < 1) return 0;
    return (size_t)36 * Cin * Cout;
}

extern "C" int sassd_conv2d_wino4_pack_weight(const float *w, int Cout, int Cin, float *packed, void *stream_)
{
    if (!w || !packed || Cin < 1 || Cout < 1) return SASSD_EINVAL;
    hipLaunchKernelGGL(wino4_pack_kernel, dim3(cdiv(Cout * Cin, 256)), dim3(256), 0, (hipStream_t)stream_, w, Cout, Cin,
                       packed, Cout);
    return sassd_launch_status();
}

extern "C" size_t sassd_conv2d_wino4_workspace_bytes(int batch, int Cin, int Cout, int H, int W)
{
    if (!sassd_conv2d_wino4_supported(Cin, Cout, H, W) || batch < 1) return 0;
    // (sized for the widest tile block, so that a forced geometry never outgrows a caller's buffer)
    const size_t Tp = (size_t)cdiv(batch * (H / 4) * (W / 4), 256) * 256 + 256;
    return align_up(36 * (size_t)Cin * Tp * 4, 256) + align_up(36 * (size_t)Cout * Tp * 4, 256);
}

extern "C" int sassd_conv2d_wino4_fwd(const float *x, const float *w_packed, const float *scale, const float *shift,
                                      int relu, float *y, int batch, int Cin, int Cout, int H, int W, int cfg,
                                      void *workspace, size_t workspace_bytes, void *stream_)
{
    if (!x || !w_packed || !y || !workspace || batch < 1 || !sassd_conv2d_wino4_supported(Cin, Cout, H, W))
        return SASSD_EINVAL;
    const int geo = w4_geo(cfg), dbg = w4_dbg(cfg);
    if (((uintptr_t)y & 15) || ((uintptr_t)w_packed & 15) || ((uintptr_t)workspace & 15)) return SASSD_EINVAL;
    if (workspace_bytes < sassd_conv2d_wino4_workspace_bytes(batch, Cin, Cout, H, W)) return SASSD_ENOSPC;
    hipStream_t stream = (hipStream_t)stream_;
    W4Geom G;
    G.B = batch; G.C = Cin; G.H = H; G.W = W; G.TH = H / 4; G.TW = W / 4; G.T = batch * G.TH * G.TW;
    G.Tp = w4_tiles_padded(batch, H, W, Cout, geo);
    float *V = (float *)workspace;
    float *M = (float *)((char *)workspace + align_up(36 * (size_t)Cin * G.Tp * 4, 256));
    // the padding columns of V feed padding columns of M that the output transform never reads; they only have to be
    // finite-or-not-read: the GEMM's columns are independent, so stale values cannot leak into real tiles
    if (!(dbg & 16))
        hipLaunchKernelGGL(wino4_in_kernel, dim3(cdiv(G.T, 256), Cin), dim3(256), 0, stream, x, G, V, (const int32_t *)nullptr);
    W4Gemm P;
    P.U = w_packed; P.V = V; P.M = M; P.scale = nullptr; P.shift = nullptr; P.relu = 0; P.ncols_dev = nullptr;
    P.np = 36; P.Cin = Cin; P.Cout = Cout; P.ldv = G.Tp; P.ldm = G.Tp; P.ncols = G.Tp;
    P.su = (size_t)Cin * Cout; P.sv = (size_t)Cin * G.Tp; P.sm = (size_t)Cout * G.Tp;
    int rc = SASSD_OK;
    if (!(dbg & 32)) rc = w4_launch(w4_tile(G.T, Cout, geo), P, dbg, stream);
    if (rc) return rc;
    W4Geom Go = G;
    Go.C = Cout;
    if (!(dbg & 64))
        hipLaunchKernelGGL(wino4_out_kernel, dim3(cdiv(G.T, 256), Cout), dim3(256), 0, stream, (const float *)M, Go, Cout, Cout,
                           scale, shift, relu, y, (const int32_t *)nullptr);
    return sassd_launch_status();
}


// ---- chained 3x3 layers: the activation map between two layers stays in the transform domain -----------------------------
namespace {
inline int w4_plane_pitch(int W) { return (W + 2 + 3) / 4 * 4; }
inline size_t w4_plane_bytes(int H, int W) { return (size_t)(H + 2) * w4_plane_pitch(W) * 4; }
}  // namespace

extern "C" int sassd_conv2d_wino4_chain_supported(int Cin, int Cout, int H, int W)
{
    return sassd_conv2d_wino4_supported(Cin, Cout, H, W) && w4_plane_bytes(H, W) <= (size_t)160 * 1024 ? 1 : 0;
}

extern "C" size_t sassd_conv2d_wino4_chain_workspace_bytes(int batch, int cmax, int H, int W)
{
    if (batch < 1 || cmax < 1 || H < 4 || W < 4 || H % 4 || W % 4) return 0;
    const size_t Tp = (size_t)cdiv(batch * (H / 4) * (W / 4), 256) * 256 + 256;          // widest tile block + slack
    return 2 * align_up(36 * (size_t)cmax * Tp * 4, 256);                              // V | M, each for cmax channels
}

extern "C" int sassd_conv2d_wino4_chain(const float *x, int src_products, const float *prev_scale,
                                        const float *prev_shift, int prev_relu, const float *w_packed,
                                        const float *scale, const float *shift, int relu, float *y, int batch, int Cin,
                                        int Cout, int cmax, int H, int W, const int32_t *tile_map,
                                        const int32_t *prev_tile_map, int cfg, void *workspace,
                                        size_t workspace_bytes, void *stream_)
{
    const int geo = w4_geo(cfg), dbg = w4_dbg(cfg);
    // tile_map: this layer runs on the ACTIVE tiles of its input only (needs the NCHW input, src_products == 0);
    // prev_tile_map: the previous chain call did, and left compacted products (needs src_products != 0)
    if ((tile_map && src_products) || (prev_tile_map && !src_products)) return SASSD_EINVAL;
    if (!w_packed || !workspace || batch < 1 || !sassd_conv2d_wino4_supported(Cin, Cout, H, W) || cmax < Cin ||
        cmax < Cout || (!src_products && !x))
        return SASSD_EINVAL;
    if (src_products && !sassd_conv2d_wino4_chain_supported(Cin, Cout, H, W)) return SASSD_EINVAL;
    if ((y && ((uintptr_t)y & 15)) || ((uintptr_t)w_packed & 15) || ((uintptr_t)workspace & 15)) return SASSD_EINVAL;
    const size_t need = sassd_conv2d_wino4_chain_workspace_bytes(batch, cmax, H, W);
    if (need == 0 || workspace_bytes < need) return SASSD_ENOSPC;
    hipStream_t stream = (hipStream_t)stream_;
    W4Geom G;
    G.B = batch; G.C = Cin; G.H = H; G.W = W; G.TH = H / 4; G.TW = W / 4; G.T = batch * G.TH * G.TW;
    G.Tp = w4_tiles_padded(batch, H, W, Cout, geo);
    // src_products: the previous call (its Cout = this Cin) left M [36][Cin][Tp'] with Tp' padded for ITS column-block
    // width.  The fused transform reads M and writes V with one plane stride: both layers must agree on it (ADVICE r03;
    // every chained SA-SSD layer is 256 -> 256, so they do) -- a mismatch is refused instead of read as garbage.
    if (src_products && w4_tiles_padded(batch, H, W, Cin, geo) != G.Tp) return SASSD_EINVAL;
    float *V = (float *)workspace;
    float *M = (float *)((char *)workspace + need / 2);
    if (!src_products) {
        if (!(dbg & 16))
            hipLaunchKernelGGL(wino4_in_kernel, dim3(cdiv(G.T, 256), Cin), dim3(256), 0, stream, x, G, V, tile_map);
    } else if (!(dbg & 128)) {
        // the previous call left M [36][Cin][Tp] (its Cout = this Cin, same tile geometry) in the workspace
        const size_t lds = w4_plane_bytes(H, W);
        int rc = launch_outin(dbg, Cin, batch, lds, stream, (const float *)M, G, prev_scale, prev_shift, prev_relu,
                              w4_plane_pitch(W), V, prev_tile_map, (float *)nullptr);
        if (rc) return rc;
    }
    W4Gemm P;
    P.U = w_packed; P.V = V; P.M = M; P.scale = nullptr; P.shift = nullptr; P.relu = 0; P.ncols_dev = tile_map;
    P.np = 36; P.Cin = Cin; P.Cout = Cout; P.ldv = G.Tp; P.ldm = G.Tp; P.ncols = G.Tp;
    P.su = (size_t)Cin * Cout; P.sv = (size_t)Cin * G.Tp; P.sm = (size_t)Cout * G.Tp;
    int rc = SASSD_OK;
    if (!(dbg & 32)) rc = w4_launch(w4_tile(G.T, Cout, geo), P, dbg, stream);
    if (rc) return rc;
    if (y && !(dbg & 64)) {
        W4Geom Go = G;
        Go.C = Cout;
        hipLaunchKernelGGL(wino4_out_kernel, dim3(cdiv(G.T, 256), Cout), dim3(256), 0, stream, (const float *)M, Go, Cout, Cout,
                           scale, shift, relu, y, tile_map);
    }
    return sassd_launch_status();
}

// ---- a NARROW layer at the end of a chain (round 6): the part-sensitive head's 3x3 conv 256 -> 28 (ssd_rotate_head.py:424-429)
// reads conv6's output.  As a direct convolution on the fp32 MFMA it took 77 us (28 output channels leave 10 MFMAs per wave
// between barriers); here it is one more chained layer -- the fused transform of conv6's products (which also stores conv6's
// activation map for conv7: no separate output-transform launch), the 36 GEMMs on a 64-channel block (weights zero-padded to
// 64), the output transform of 28 channels.  prev_* = folded BatchNorm / ReLU of the previous layer, y_prev its NCHW map.
extern "C" size_t sassd_conv2d_wino4_narrow_packed_floats(int Cin) { return Cin < 1 ? 0 : (size_t)36 * Cin * 64; }

extern "C" int sassd_conv2d_wino4_pack_weight_narrow(const float *w, int Cout, int Cin, float *packed, void *stream_)
{
    if (!w || !packed || Cin < 1 || Cout < 1 || Cout > 64) return SASSD_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    int rc;
    if ((rc = sassd_hip(hipMemsetAsync(packed, 0, sassd_conv2d_wino4_narrow_packed_floats(Cin) * sizeof(float), stream)))) return rc;
    hipLaunchKernelGGL(wino4_pack_kernel, dim3(cdiv(Cout * Cin, 256)), dim3(256), 0, stream, w, Cout, Cin, packed, 64);
    return sassd_launch_status();
}

extern "C" int sassd_conv2d_wino4_chain_tail(const float *prev_scale, const float *prev_shift, int prev_relu, float *y_prev,
                                             const float *w_packed64, const float *scale, const float *shift, int relu,
                                             float *y, int batch, int Cin, int Cout, int cmax, int H, int W,
                                             const int32_t *prev_tile_map, int cfg, void *workspace, size_t workspace_bytes,
                                             void *stream_)
{
    const int geo = w4_geo(cfg), dbg = w4_dbg(cfg);
    if (!w_packed64 || !y || !workspace || batch < 1 || Cout < 1 || Cout > 64 || cmax < Cin || cmax < 64) return SASSD_EINVAL;
    if (!sassd_conv2d_wino4_chain_supported(Cin, 256, H, W) || geo != 0) return SASSD_EINVAL;     // (default geometry only)
    if (((uintptr_t)y & 15) || (y_prev && ((uintptr_t)y_prev & 15)) || ((uintptr_t)w_packed64 & 15) || ((uintptr_t)workspace & 15))
        return SASSD_EINVAL;
    const size_t need = sassd_conv2d_wino4_chain_workspace_bytes(batch, cmax, H, W);
    if (need == 0 || workspace_bytes < need) return SASSD_ENOSPC;
    hipStream_t stream = (hipStream_t)stream_;
    W4Geom G;
    G.B = batch; G.C = Cin; G.H = H; G.W = W; G.TH = H / 4; G.TW = W / 4; G.T = batch * G.TH * G.TW;
    G.Tp = w4_tiles_padded(batch, H, W, 256, geo);            // the previous (256-channel) layer's plane stride: same 128-column block
    float *V = (float *)workspace;
    float *M = (float *)((char *)workspace + need / 2);
    int rc;
    if (!(dbg & 128) && (rc = launch_outin(dbg, Cin, batch, w4_plane_bytes(H, W), stream, (const float *)M, G, prev_scale,
                                           prev_shift, prev_relu, w4_plane_pitch(W), V, prev_tile_map, y_prev)))
        return rc;
    W4Gemm P;
    P.U = w_packed64; P.V = V; P.M = M; P.scale = nullptr; P.shift = nullptr; P.relu = 0; P.ncols_dev = nullptr;
    P.np = 36; P.Cin = Cin; P.Cout = 64; P.ldv = G.Tp; P.ldm = G.Tp; P.ncols = G.Tp;
    P.su = (size_t)Cin * 64; P.sv = (size_t)Cin * G.Tp; P.sm = (size_t)64 * G.Tp;
    if (!(dbg & 32) && (rc = launch_w4_gemm<1, 2, 2, kKC, 1>(P, dbg, stream))) return rc;
    W4Geom Go = G;
    Go.C = Cout;
    if (!(dbg & 64))
        hipLaunchKernelGGL(wino4_out_kernel, dim3(cdiv(G.T, 256), Cout), dim3(256), 0, stream, (const float *)M, Go, Cout, 64,
                           scale, shift, relu, y, (const int32_t *)nullptr);
    return sassd_launch_status();
}

// ---- active-tile map of a SPARSE input map (BEV conv0 reads the densified sparse tensor: 17 % of its pixels / 56 % of its
// 4x4-tile patches are occupied on a KITTI frame) -------------------------------------------------------------------------
// A tile is active when any occupied pixel lies in its 6x6 input patch; inactive tiles have all-zero transformed input,
// all-zero products and the output relu(shift) -- exactly what the dense launch computes for them, so skipping them is
// bit-identical.  One workgroup: flags in LDS (benign same-value byte stores), ordered compaction by a block scan.
namespace {
__global__ void __launch_bounds__(1024) w4_tile_map_kernel(const int32_t *__restrict__ idx, const int32_t *__restrict__ n_ptr,
                                                           int cap, int B, int H, int W, int TH, int TW,
                                                           int32_t *__restrict__ tmap)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char w4_flags[];
    __shared__ int wsum[17];
    const int T = B * TH * TW;
    for (int i = threadIdx.x; i < (T + 3) / 4; i += 1024) ((unsigned *)w4_flags)[i] = 0u;
    __syncthreads();
    const int n = min(*n_ptr, cap);
    for (int r = threadIdx.x; r < n; r += 1024) {
        const int4 c = ((const int4 *)idx)[r];               // (b, z, y, x)
        if (c.x < 0 || c.x >= B || c.z < 0 || c.z >= H || c.w < 0 || c.w >= W) continue;
        // pixel y lies in the patch rows 4 ty - 1 .. 4 ty + 4 of tile ty = y / 4, of ty - 1 when y % 4 == 0, of ty + 1 when y % 4 == 3
        const int ty = c.z >> 2, tx = c.w >> 2;
        const int y0 = ((c.z & 3) == 0 && ty > 0) ? ty - 1 : ty, y1 = ((c.z & 3) == 3 && ty + 1 < TH) ? ty + 1 : ty;
        const int x0 = ((c.w & 3) == 0 && tx > 0) ? tx - 1 : tx, x1 = ((c.w & 3) == 3 && tx + 1 < TW) ? tx + 1 : tx;
        for (int a = y0; a <= y1; ++a)
            for (int e = x0; e <= x1; ++e) w4_flags[(c.x * TH + a) * TW + e] = 1;
    }
    __syncthreads();
    const int per = (T + 1023) / 1024;
    const int t0 = min((int)threadIdx.x * per, T), t1 = min(t0 + per, T);
    int cnt = 0;
    for (int t = t0; t < t1; ++t) cnt += w4_flags[t];
    int total;
    int j = block_exclusive_scan(cnt, wsum, &total);
    for (int t = t0; t < t1; ++t) {
        const bool on = w4_flags[t] != 0;
        tmap[kTmapHead + t] = on ? j : -1;
        if (on) tmap[kTmapHead + T + j++] = t;
    }
    if (threadIdx.x == 0) { tmap[0] = total; tmap[1] = T; tmap[2] = tmap[3] = 0; }
}
}  // namespace

extern "C" size_t sassd_wino4_tile_map_ints(int batch, int H, int W)
{
    if (batch < 1 || H < 4 || W < 4 || H % 4 || W % 4) return 0;
    const size_t T = (size_t)batch * (H / 4) * (W / 4);
    return T <= 65536 ? kTmapHead + 2 * T : 0;               // (the flags of all tiles live in one workgroup's LDS)
}

extern "C" int sassd_wino4_tile_map(const int32_t *indices, const int32_t *n_ptr, int cap, int batch, int H, int W,
                                    int32_t *tile_map, void *stream_)
{
    if (!indices || !n_ptr || !tile_map || cap < 1 || sassd_wino4_tile_map_ints(batch, H, W) == 0) return SASSD_EINVAL;
    const int TH = H / 4, TW = W / 4, T = batch * TH * TW;
    hipLaunchKernelGGL(w4_tile_map_kernel, dim3(1), dim3(1024), (size_t)align_up((size_t)T, 16), (hipStream_t)stream_, indices,
                       n_ptr, cap, batch, H, W, TH, TW, tile_map);
    return sassd_launch_status();
}


// ---- 1x1 convolution (BEV conv7, cmn.py:262) on the same GEMM kernel: y[b] [Cout x HW] = W [Cout x Cin] . x[b] [Cin x HW]
namespace {
__global__ void conv1x1_pack_kernel(const float *__restrict__ w, int Cout, int Cin, float *__restrict__ U)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cout * Cin) return;
    const int co = i % Cout, ci = i / Cout;
    U[(size_t)ci * Cout + co] = w[(size_t)co * Cin + ci];
}
}  // namespace

extern "C" int sassd_conv1x1_gemm_supported(int Cin, int Cout, int H, int W)
{
    if (!(Cin >= kKC && Cin % kKC == 0 && Cout >= kBM && Cout % kBM == 0 && H > 0 && W > 0)) return 0;
    return w4_pick_wn(H * W, Cout, 1, true) ? 1 : 0;
}

extern "C" int sassd_conv1x1_gemm_pack_weight(const float *w, int Cout, int Cin, float *packed, void *stream_)
{
    if (!w || !packed || Cin < 1 || Cout < 1) return SASSD_EINVAL;
    hipLaunchKernelGGL(conv1x1_pack_kernel, dim3(cdiv(Cout * Cin, 256)), dim3(256), 0, (hipStream_t)stream_, w, Cout, Cin,
                       packed);
    return sassd_launch_status();
}

extern "C" int sassd_conv1x1_gemm_fwd(const float *x, const float *w_packed, const float *scale, const float *shift,
                                      int relu, float *y, int batch, int Cin, int Cout, int H, int W, int cfg,
                                      void *stream_)
{
    if (!x || !w_packed || !y || batch < 1 || !sassd_conv1x1_gemm_supported(Cin, Cout, H, W)) return SASSD_EINVAL;
    const int geo = w4_geo(cfg), dbg = w4_dbg(cfg);
    if (((uintptr_t)y & 15) || ((uintptr_t)w_packed & 15) || ((uintptr_t)x & 15)) return SASSD_EINVAL;
    const int hw = H * W;
    W4Gemm P;
    P.U = w_packed; P.V = x; P.M = y; P.scale = scale; P.shift = shift; P.relu = relu; P.ncols_dev = nullptr;
    P.np = batch; P.Cin = Cin; P.Cout = Cout; P.ldv = hw; P.ldm = hw; P.ncols = hw;
    P.su = 0; P.sv = (size_t)Cin * hw; P.sm = (size_t)Cout * hw;
    hipStream_t stream = (hipStream_t)stream_;
    if (geo != 1 && hw % 128 == 0) return launch_w4_gemm<2, 2, 2, kKC, 1>(P, dbg, stream);     // split operands, as above
    switch (w4_pick_wn(hw, Cout, batch, true)) {
    case 2: return launch_w4_gemm<2, 2, 1, kKC>(P, dbg, stream);
    case 3: return launch_w4_gemm<2, 3, 1, kKC>(P, dbg, stream);
    case 4: return launch_w4_gemm<2, 4, 1, kKC>(P, dbg, stream);
    case 5: return launch_w4_gemm<2, 5, 1, kKC>(P, dbg, stream);
    case 6: return launch_w4_gemm<2, 6, 1, kKC>(P, dbg, stream);
    }
    return SASSD_EINVAL;
}
