// conv2d_wino.hip -- 3x3 stride-1 pad-1 convolution of the BEV network through Winograd F(2x2, 3x3) on the fp32 MFMA
// (mmdet/models/necks/cmn.py:240-262: conv0 320->256 and conv1-6 256->256 at 200x176, 93 % of the dense FLOPs).
//
// fp32 MFMA and fp32 VALU have the same peak on MI355X (157 TF), so the only way past the direct kernel's 71 % is to do
// fewer multiplications: F(2x2,3x3) needs 16 instead of 36 per 2x2 output tile and channel pair (2.25x), at the price
// of cheap add-only transforms:   Y = A^T [ (G g G^T) (.) (B^T d B) ] A,   summed over input channels.
//
// One kernel, everything fused (no transformed tensors in HBM), WAVE-SPECIALISED:
//   * work unit (workgroup, 8 waves) = 64 couts x 32 output tiles (2x2 pixels each, linear tile order -> 8800 tiles
//     = 275 groups, no padding waste) x all 16 Winograd positions.
//   * waves 0-3 are MFMA waves: wave xi owns the 4 positions (xi, nu = 0..3) of both 32-cout blocks = 8 accumulator
//     tiles of v_mfma_f32_32x32x2_f32 (128 VGPRs), 64 MFMAs per 16-channel chunk.  Their only other instructions are
//     8 ds_read_b128 (B operands) and 16 coalesced 16-byte weight loads per chunk: the pre-transformed weights
//     G g G^T are packed offline in MFMA A-operand order and have no reuse inside a workgroup, so they go straight
//     to registers, prefetched half a chunk ahead (cout groups are the slowest grid index: the 1 MB slice in use
//     stays L2-resident).
//   * waves 4-7 are LOADER waves: each thread owns one tile and two channels of the chunk; it loads the two 4x4 input
//     patches straight from global memory (per-lane byte offsets computed once, a scalar base advances per chunk),
//     transforms them in registers (B^T d B: 32 adds each) and writes the 16 values into LDS in exactly the layout
//     the B-operand reads want.  One SIMD = one MFMA wave + one loader wave, so the MFMA pipe never waits behind
//     VMEM issue: measured ablations of the unspecialised versions showed MFMA time (0.165 ms) and load/transform
//     issue time (0.17 ms) adding up almost serially (0.27-0.33 ms), whatever the prefetch depth.
//   * one barrier per chunk; the output transform A^T M A is split: each MFMA wave reduces over nu in registers, the
//     four xi are combined through LDS by all 8 waves, then scale/shift/ReLU and float2 stores of the 2x2 pixels.
// Numerics: F(2,3) in fp32 has a relative error ~1e-6 (transform matrices hold only 0, +-1, +-1/2).
#include "common.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kNT = 32;                  // tiles per workgroup
constexpr int kCoW = 64;                 // couts per workgroup
constexpr int kKC = 16;                  // input channels per chunk
constexpr int kVBuf = 16 * 2 * 2 * kNT * 4;        // floats per V buffer: [pos][half][kh][tile][4]  (8192 = 32 KB)

struct WinoParams {
    const float *x, *wp, *scale, *shift;
    float *y;
    int B, Cin, Cout, H, W;
    int TH, TW, tiles;                   // tile rows / cols per image, total tiles (B * TH * TW)
    int ncb64;                           // cout groups of 64
    int relu;
    int dbg;                             // reserved for ablation builds
};

// w [Cout][Cin][3][3] -> U = G g G^T packed as [cb32][half chunk of 8 ci][pos 16][lane 64][4]:
//   value(lane, e) = U[pos][co = cb32*32 + (lane & 31)][ci = hc*8 + 2*e + (lane >> 5)]
__global__ void wino_pack_kernel(const float *__restrict__ w, int Cout, int Cin, int ncb32, float *__restrict__ wp)
{
    const size_t total = (size_t)ncb32 * (Cin / 8) * 16 * 256;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int e = i & 3, lane = (i >> 2) & 63, pos = (i >> 8) & 15;
    const size_t r = i >> 12;
    const int hc = r % (Cin / 8), cb = r / (Cin / 8);
    const int co = cb * 32 + (lane & 31), ci = hc * 8 + 2 * e + (lane >> 5);
    float u = 0.f;
    if (co < Cout) {
        const float *g = w + ((size_t)co * Cin + ci) * 9;
        const int xi = pos >> 2, nu = pos & 3;
        // row xi of G applied to the 3 filter rows, then row nu of G to the 3 columns
        float t[3];
        for (int c = 0; c < 3; ++c) {
            const float g0 = g[c], g1 = g[3 + c], g2 = g[6 + c];
            t[c] = xi == 0 ? g0 : (xi == 1 ? 0.5f * (g0 + g1 + g2) : (xi == 2 ? 0.5f * (g0 - g1 + g2) : g2));
        }
        u = nu == 0 ? t[0] : (nu == 1 ? 0.5f * (t[0] + t[1] + t[2]) : (nu == 2 ? 0.5f * (t[0] - t[1] + t[2]) : t[2]));
    }
    wp[i] = u;
}

__global__ void __launch_bounds__(512) conv2d_wino_kernel(WinoParams P)
{
    extern __shared__ float smem[];                  // 2 x V buffer (64 KB); reused by the output reduction
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // cout group slowest: all workgroups in flight share ONE 64-cout slice of the transformed weights
    const int ngrp = gridDim.x / P.ncb64;
    const int cg = blockIdx.x / ngrp, grp = blockIdx.x - cg * ngrp;
    const int HW = P.H * P.W;
    const int nchunk = P.Cin / kKC;
    f32x16 acc[4][2];                                // MFMA waves only: [nu][cout block]

    if (wave >= 4) {
        // ================================ loader waves ================================================================
        const int lt = tid - 256;
        const int ltile = lt & 31, cil = lt >> 5;    // tile, channels cil and cil + 8 of the chunk
        const int gt = grp * kNT + ltile;            // global tile index (may run past the end in the last group)
        int tb = 0, ty = 0, tx = 0;
        const bool tile_ok = gt < P.tiles;
        if (tile_ok) {
            tb = gt / (P.TH * P.TW);
            const int r = gt - tb * P.TH * P.TW;
            ty = r / P.TW;
            tx = r - ty * P.TW;
        }
        // 4 patch rows (2ty-1 .. 2ty+2) x 4 columns (2tx-1 .. 2tx+2): per-lane BYTE offsets (batch, channel-in-chunk,
        // row, column; clamped to a valid element where the patch leaves the image) are computed ONCE -- per chunk only
        // a wave-uniform base pointer advances (saddr + 32-bit voffset addressing, no VALU address arithmetic)
        unsigned off[16];
        unsigned vmask = 0;                          // bit (r*4 + c): inside the image
        const int lane_base = (tb * P.Cin + cil) * HW;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int yy = 2 * ty - 1 + r;
            const bool rok = tile_ok && yy >= 0 && yy < P.H;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int xx = 2 * tx - 1 + c;
                const bool ok = rok && xx >= 0 && xx < P.W;
                if (ok) vmask |= 1u << (r * 4 + c);
                off[r * 4 + c] = 4u * (unsigned)(lane_base + (ok ? yy * P.W + xx : 0));
            }
        }
        const bool wave_interior = __ballot(vmask != 0xFFFFu) == 0ull;  // wave-uniform: no zero padding needed
        float rawA[16], rawB[16];
        auto fetch_patches = [&](int chunk) {
            const char *sa = reinterpret_cast<const char *>(P.x + (size_t)chunk * kKC * HW);    // wave-uniform
            const char *sb = sa + (size_t)8 * HW * sizeof(float);
#pragma unroll
            for (int e = 0; e < 16; ++e) rawA[e] = *reinterpret_cast<const float *>(sa + off[e]);
#pragma unroll
            for (int e = 0; e < 16; ++e) rawB[e] = *reinterpret_cast<const float *>(sb + off[e]);
        };
        // B^T d B in registers, then scatter to V[pos][half][kh][tile][s]
        auto transform_store = [&](float *vbuf, float *raw, int h) {
            float t[16];
            if (!wave_interior) {
#pragma unroll
                for (int e = 0; e < 16; ++e) raw[e] = ((vmask >> e) & 1u) ? raw[e] : 0.f;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                t[0 * 4 + c] = raw[0 * 4 + c] - raw[2 * 4 + c];
                t[1 * 4 + c] = raw[1 * 4 + c] + raw[2 * 4 + c];
                t[2 * 4 + c] = raw[2 * 4 + c] - raw[1 * 4 + c];
                t[3 * 4 + c] = raw[1 * 4 + c] - raw[3 * 4 + c];
            }
            const int kh = cil & 1, s = cil >> 1;
            float *dst = vbuf + ((h * 2 + kh) * kNT + ltile) * 4 + s;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dst[(r * 4 + 0) * (2 * 2 * kNT * 4)] = t[r * 4 + 0] - t[r * 4 + 2];
                dst[(r * 4 + 1) * (2 * 2 * kNT * 4)] = t[r * 4 + 1] + t[r * 4 + 2];
                dst[(r * 4 + 2) * (2 * 2 * kNT * 4)] = t[r * 4 + 2] - t[r * 4 + 1];
                dst[(r * 4 + 3) * (2 * 2 * kNT * 4)] = t[r * 4 + 1] - t[r * 4 + 3];
            }
        };
        fetch_patches(0);
        transform_store(smem, rawA, 0);
        transform_store(smem, rawB, 1);
        fetch_patches(min(1, nchunk - 1));           // chunk 1 in flight across the first barrier
        __syncthreads();
        for (int c = 0; c < nchunk; ++c) {
            float *vnxt = smem + ((c + 1) & 1) * kVBuf;          // last read by the MFMA waves during chunk c-1
            if (c + 1 < nchunk) {
                transform_store(vnxt, rawA, 0);
                transform_store(vnxt, rawB, 1);
                if (c + 2 < nchunk) fetch_patches(c + 2);        // in flight during the whole next chunk
            }
            __syncthreads();
        }
    } else {
        // ================================ MFMA waves ==================================================================
        const int xi = wave;
        const int nhc = P.Cin / 8;
        // weights of (cout block cb, position (xi, nu), half chunk hc): float4 at wsrc[((cb*nhc + hc)*16 + nu) * 64]
        const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(P.wp) + ((size_t)(cg * 2) * nhc * 16 + xi * 4) * 64 + lane;
        f32x4 wq[2][2][4];                           // [buffer][cb][nu]
        auto fetch_w = [&](int hc, f32x4 (*wdst)[4]) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) wdst[cb][nu] = wsrc[(((size_t)cb * nhc + hc) * 16 + nu) * 64];
        };
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nu][cb][r] = 0.f;
        fetch_w(0, wq[0]);
        __syncthreads();
        for (int c = 0; c < nchunk; ++c) {
            const float *vcur = smem + (c & 1) * kVBuf;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                fetch_w(min(2 * c + h + 1, nhc - 1), wq[(h + 1) & 1]);       // next half chunk's weights
                const f32x4 *vb = reinterpret_cast<const f32x4 *>(vcur) + ((h * 2 + (lane >> 5)) * kNT + (lane & 31));
                f32x4 bq[4];
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) bq[nu] = vb[(xi * 4 + nu) * (2 * 2 * kNT)];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
                            acc[nu][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[h][cb][nu][s], bq[nu][s], acc[nu][cb],
                                                                               0, 0, 0);
            }
            __syncthreads();
        }
    }

    // ---- output transform: over nu in registers (MFMA waves), over xi through LDS (all waves) --------------------------
    // P_j[xi] = sum_nu M[xi][nu] A[nu][j],  A^T = [[1,1,1,0],[0,1,-1,-1]]
    float *red = smem;                               // [cb 2][xi 4][j 2][reg 16][lane 64]  = 64 KB
    if (wave < 4) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p0 = acc[0][cb][r] + acc[1][cb][r] + acc[2][cb][r];
                const float p1 = acc[1][cb][r] - acc[2][cb][r] - acc[3][cb][r];
                red[(((cb * 4 + wave) * 2 + 0) * 16 + r) * 64 + lane] = p0;
                red[(((cb * 4 + wave) * 2 + 1) * 16 + r) * 64 + lane] = p1;
            }
    }
    __syncthreads();
    // wave w finishes registers 4*(w & 3) .. +3 of cout block (w >> 2): Y[i][j] = sum_xi A^T[i][xi] P_j[xi]
    const int otile = grp * kNT + (lane & 31);
    if (otile < P.tiles) {
        const int ob = otile / (P.TH * P.TW);
        const int orr = otile - ob * P.TH * P.TW;
        const int oty = orr / P.TW, otx = orr - oty * P.TW;
        const int cb = wave >> 2;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = (wave & 3) * 4 + rr;
            const int co = (cg * 2 + cb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (co >= P.Cout) continue;
            float pj[4][2];
#pragma unroll
            for (int x4 = 0; x4 < 4; ++x4)
#pragma unroll
                for (int j = 0; j < 2; ++j) pj[x4][j] = red[(((cb * 4 + x4) * 2 + j) * 16 + r) * 64 + lane];
            const float sc = P.scale ? P.scale[co] : 1.f, sh = P.shift ? P.shift[co] : 0.f;
            float *dst = P.y + ((size_t)ob * P.Cout + co) * HW + (size_t)(2 * oty) * P.W + 2 * otx;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float2 o;
                float y0 = i == 0 ? pj[0][0] + pj[1][0] + pj[2][0] : pj[1][0] - pj[2][0] - pj[3][0];
                float y1 = i == 0 ? pj[0][1] + pj[1][1] + pj[2][1] : pj[1][1] - pj[2][1] - pj[3][1];
                y0 = y0 * sc + sh;
                y1 = y1 * sc + sh;
                if (P.relu) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
                o.x = y0; o.y = y1;
                *reinterpret_cast<float2 *>(dst + (size_t)i * P.W) = o;
            }
        }
    }
}
int g_wino_dbg = 0;
}  // namespace

extern "C" void sassd_debug_set_wino(int flags) { g_wino_dbg = flags; }

extern "C" int sassd_conv2d_wino_supported(int Cin, int Cout, int H, int W)
{
    return (Cin >= 16 && Cin % 16 == 0 && Cout >= 32 && Cout % 32 == 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0)
               ? 1 : 0;
}

extern "C" size_t sassd_conv2d_wino_packed_floats(int Cin, int Cout)
{
    if (Cin < 16 || Cin % 16 || Cout < 1) return 0;
    const int ncb32 = cdiv(cdiv(Cout, 64) * 64, 32);
    return (size_t)ncb32 * (Cin / 8) * 16 * 256;
}

extern "C" int sassd_conv2d_wino_pack_weight(const float *w, int Cout, int Cin, float *packed, void *stream_)
{
    if (!w || !packed || Cin < 16 || Cin % 16 || Cout < 1) return SASSD_EINVAL;
    const int ncb32 = cdiv(cdiv(Cout, 64) * 64, 32);
    const size_t total = (size_t)ncb32 * (Cin / 8) * 16 * 256;
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, w,
                       Cout, Cin, ncb32, packed);
    return sassd_launch_status();
}

extern "C" int sassd_conv2d_wino_fwd(const float *x, const float *w_packed, const float *scale, const float *shift,
                                     int relu, float *y, int batch, int Cin, int Cout, int H, int W, void *stream_)
{
    if (!x || !w_packed || !y || batch < 1 || !sassd_conv2d_wino_supported(Cin, Cout, H, W)) return SASSD_EINVAL;
    if (((uintptr_t)y & 7) || ((uintptr_t)w_packed & 15)) return SASSD_EINVAL;
    if ((size_t)batch * Cin * H * W >= (1u << 29)) return SASSD_EINVAL;         // 32-bit per-lane element offsets
    WinoParams P;
    P.x = x; P.wp = w_packed; P.scale = scale; P.shift = shift; P.y = y;
    P.B = batch; P.Cin = Cin; P.Cout = Cout; P.H = H; P.W = W;
    P.TH = H / 2; P.TW = W / 2; P.tiles = batch * P.TH * P.TW;
    P.ncb64 = cdiv(Cout, kCoW);
    P.relu = relu;
    P.dbg = g_wino_dbg;
    const size_t lds = (size_t)2 * kVBuf * sizeof(float);           // 65 536 B
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)conv2d_wino_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return sassd_launch_status();
        attr_done = true;
    }
    const int grid = cdiv(P.tiles, kNT) * P.ncb64;
    hipLaunchKernelGGL(conv2d_wino_kernel, dim3(grid), dim3(512), lds, (hipStream_t)stream_, P);
    return sassd_launch_status();
}
