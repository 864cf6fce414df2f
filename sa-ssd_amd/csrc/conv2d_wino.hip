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
//   * waves 4-7 are LOADER waves.  The input rows a 32-tile group needs (<= 2 runs of tiles inside one tile row each:
//     4 image rows x <= 72 columns per run and channel) are staged in LDS by 16-byte global->LDS DMA (zero page for
//     everything outside the image), 36 wave-instructions per 16-channel chunk instead of the 128 scalar-dword
//     gathers of a per-thread 4x4 patch fetch; each thread then reads its two 4x4 patches (its tile, channels c and
//     c + 8) from LDS, transforms them in registers (B^T d B: 32 adds each) and writes the 16 values into LDS in
//     exactly the layout the B-operand reads want.  VMEM instruction COUNT is what matters: measured ablations
//     showed MFMA time (0.165 ms) and memory-instruction time (16 cycles per wave-instruction in the texture
//     addresser) adding up almost serially on a CU, whatever the prefetch depth or wave arrangement.
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
constexpr int kGb = 8;                   // tile groups per XCD-local reuse block
constexpr int kVBuf = 16 * 2 * 2 * kNT * 4;        // floats per V buffer: [pos][half][kh][tile][4]  (8192 = 32 KB)
constexpr int kRawW = 72;                // staged columns per run (66 needed + alignment slack), 18 float4
constexpr int kRawBuf = kKC * 2 * 4 * kRawW;       // floats per raw-row buffer: [ci][run][row][col]  (9216 = 36 KB)

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

__device__ __attribute__((aligned(16))) float g_wino_zero[4] = {0.f, 0.f, 0.f, 0.f};
__device__ long long g_wino_prof[16];      // -DSASSD_WINO_PROF builds: per-phase cycle counts of workgroup 0 (P.dbg & 1)

struct WinoParams {
    const float *x, *wp, *scale, *shift;
    float *y;
    int B, Cin, Cout, H, W;
    int TH, TW, tiles;                   // tile rows / cols per image, total tiles (B * TH * TW)
    int ncb64;                           // cout groups of 64
    int ngrp;                            // tile groups of kNT
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
    extern __shared__ float smem[];                  // 2 x V buffer (64 KB; reused by the output reduction) + 2 x raw rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware work order.  Workgroup ids are dealt round-robin to the 8 XCDs (id % 8), each with its own 4 MB L2.
    // Inside one XCD's sequence j = id / 8 the work runs in blocks of kGb tile groups: for each block, the ncb64 cout
    // slices one after the other, each over the block's groups.  So the input rows of a block (kGb x ~0.3 MB) are
    // fetched over the fabric once and re-read from L2 by the other cout slices, and only one 1 MB weight slice is
    // live at a time (the whole 4.2 MB transformed-weight set does not fit the L2 next to the inputs).
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int blk = j / (kGb * P.ncb64), rem = j - blk * (kGb * P.ncb64);
    const int cg = rem / kGb, gi = rem - cg * kGb;
    const int grp = (blk * kGb + gi) * 8 + xcd;
    if (grp >= P.ngrp) return;                       // whole workgroup: padding of the last block
    const int HW = P.H * P.W;
    const int nchunk = P.Cin / kKC;
    f32x16 acc[4][2];                                // MFMA waves only: [nu][cout block]

    float *rawbuf = smem + 2 * kVBuf;                // 2 x raw-row buffer
    if (wave >= 4) {
        // ================================ loader waves ================================================================
        const int lt = tid - 256;
        // the group's 32 linear tiles form at most two runs, each inside one tile row (TW >= 32 is required)
        const int g0 = grp * kNT;
        const int tpi = P.TH * P.TW;
        int rb[2], rty[2], rtx[2], rn[2];            // image, tile row, first tile column, length of each run
        {
            const int b0 = g0 / tpi, r0 = g0 - b0 * tpi;
            rb[0] = b0; rty[0] = r0 / P.TW; rtx[0] = r0 - rty[0] * P.TW;
            rn[0] = min(min(kNT, P.TW - rtx[0]), P.tiles - g0);
            const int g1 = g0 + rn[0];
            rn[1] = max(min(kNT - rn[0], P.tiles - g1), 0);
            const int b1 = g1 / tpi, r1 = g1 - b1 * tpi;
            rb[1] = b1; rty[1] = r1 / P.TW; rtx[1] = 0;
        }
        int cola[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) cola[q] = (2 * rtx[q] - 1) & ~3;        // 16-byte aligned first staged column
        // ---- DMA role: float4 i = lt + 256*k of the raw buffer, i = ((ci*2 + run)*4 + row)*18 + f4 --------------------
        const float *dsrc[9];
        unsigned dvalid = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int i = lt + 256 * k;
            const int f4 = i % 18, row = (i / 18) & 3, run = (i / 72) & 1, ci = i / 144;
            const int yy = 2 * rty[run] - 1 + row, xx = cola[run] + 4 * f4;
            const bool ok = rn[run] > 0 && yy >= 0 && yy < P.H && xx >= 0 && xx < P.W;    // W % 4 == 0: all or nothing
            dsrc[k] = ok ? P.x + ((size_t)(rb[run] * P.Cin + ci) * P.H + yy) * P.W + xx : g_wino_zero;
            if (ok) dvalid |= 1u << k;
        }
        const size_t cstride = (size_t)kKC * HW;
        const int wl = wave - 4;
        auto dma_rows = [&](float *dst) {            // issues the next chunk's rows and advances the pointers
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                __builtin_amdgcn_global_load_lds((glb_ptr_t)dsrc[k], (lds_ptr_t)(dst + (wl * 64 + 256 * k) * 4), 16, 0, 0);
                dsrc[k] += ((dvalid >> k) & 1u) ? cstride : 0;
            }
        };
        // ---- transform role: thread -> (tile = lt & 31, channels cil and cil + 8 of the chunk) ------------------------
        const int ltile = lt & 31, cil = lt >> 5;
        const int trun = ltile < rn[0] ? 0 : 1;
        const int ttx = trun == 0 ? rtx[0] + ltile : ltile - rn[0];
        const int pbase = (trun * 4) * kRawW + (2 * ttx - 1 - cola[trun]);    // + (ci*2*4 + r) * kRawW + c
        // both patches of the thread (channels cil and cil + 8) are read first, then transformed, then written: the
        // LDS read latency is paid once per chunk instead of once per patch (the loader is the critical path:
        // measured 4.9k cycles per chunk for two back-to-back read->transform->write passes vs 4.6k of MFMA issue)
        auto transform_store2 = [&](const float *rawrows, float *vbuf) {
            float raw[2][16], t[16];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float *src = rawrows + ((cil + 8 * h) * 8) * kRawW + pbase;
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) raw[h][r * 4 + c] = src[r * kRawW + c];
            }
            const int kh = cil & 1, s = cil >> 1;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    t[0 * 4 + c] = raw[h][0 * 4 + c] - raw[h][2 * 4 + c];
                    t[1 * 4 + c] = raw[h][1 * 4 + c] + raw[h][2 * 4 + c];
                    t[2 * 4 + c] = raw[h][2 * 4 + c] - raw[h][1 * 4 + c];
                    t[3 * 4 + c] = raw[h][1 * 4 + c] - raw[h][3 * 4 + c];
                }
                float *dst = vbuf + ((h * 2 + kh) * kNT + ltile) * 4 + s;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dst[(r * 4 + 0) * (2 * 2 * kNT * 4)] = t[r * 4 + 0] - t[r * 4 + 2];
                    dst[(r * 4 + 1) * (2 * 2 * kNT * 4)] = t[r * 4 + 1] + t[r * 4 + 2];
                    dst[(r * 4 + 2) * (2 * 2 * kNT * 4)] = t[r * 4 + 2] - t[r * 4 + 1];
                    dst[(r * 4 + 3) * (2 * 2 * kNT * 4)] = t[r * 4 + 1] - t[r * 4 + 3];
                }
            }
        };
        dma_rows(rawbuf);                            // chunk 0 -> raw[0]
        __syncthreads();                             // (the compiler drains vmcnt before every barrier)
        transform_store2(rawbuf, smem);
        if (nchunk > 1) dma_rows(rawbuf + kRawBuf);  // chunk 1 -> raw[1]
        __syncthreads();
#ifdef SASSD_WINO_PROF
        long long lt_tr = 0, lt_dma = 0, lt_bar = 0;
#endif
        for (int c = 0; c < nchunk; ++c) {
            // raw[(c+1)&1] holds chunk c+1 (landed before the last barrier); V[(c+1)&1] was last read during chunk c-1
#ifdef SASSD_WINO_PROF
            const long long t0 = clock64();
#endif
            if (c + 1 < nchunk) {
                const float *rr = rawbuf + ((c + 1) & 1) * kRawBuf;
                float *vnxt = smem + ((c + 1) & 1) * kVBuf;
                transform_store2(rr, vnxt);
            }
#ifdef SASSD_WINO_PROF
            const long long t1 = clock64();
#endif
            if (c + 2 < nchunk) dma_rows(rawbuf + (c & 1) * kRawBuf);        // raw[c&1] was last read in iteration c-1
#ifdef SASSD_WINO_PROF
            const long long t2 = clock64();
#endif
            __syncthreads();
#ifdef SASSD_WINO_PROF
            const long long t3 = clock64();
            lt_tr += t1 - t0; lt_dma += t2 - t1; lt_bar += t3 - t2;
#endif
        }
#ifdef SASSD_WINO_PROF
        if ((P.dbg & 1) && blockIdx.x == 0 && tid == 256) {
            g_wino_prof[4] = lt_tr; g_wino_prof[5] = lt_dma; g_wino_prof[6] = lt_bar;
        }
#endif
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
        __syncthreads();                             // loader prologue: rows of chunk 0 landed
        __syncthreads();                             // V[0] written
#ifdef SASSD_WINO_PROF
        long long mt_mma = 0, mt_bar = 0;
        const long long mt_begin = clock64();
#endif
        for (int c = 0; c < nchunk; ++c) {
            const float *vcur = smem + (c & 1) * kVBuf;
#ifdef SASSD_WINO_PROF
            const long long t0 = clock64();
#endif
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
#ifdef SASSD_WINO_PROF
            const long long t1 = clock64();
#endif
            __syncthreads();
#ifdef SASSD_WINO_PROF
            const long long t2 = clock64();
            mt_mma += t1 - t0; mt_bar += t2 - t1;
#endif
        }
#ifdef SASSD_WINO_PROF
        if ((P.dbg & 1) && blockIdx.x == 0 && tid == 0) {
            g_wino_prof[0] = mt_mma; g_wino_prof[1] = mt_bar; g_wino_prof[2] = clock64() - mt_begin;
        }
#endif
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
extern "C" int sassd_debug_get_wino_prof(long long *out16)
{
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_wino_prof), 16 * sizeof(long long)) == hipSuccess ? 0 : -3;
}

extern "C" int sassd_conv2d_wino_supported(int Cin, int Cout, int H, int W)
{
    // W % 4: zero padding is decided per 16-byte DMA; W >= 64: a 32-tile group spans at most two tile rows
    return (Cin >= 16 && Cin % 16 == 0 && Cout >= 32 && Cout % 32 == 0 && H >= 2 && H % 2 == 0 && W >= 64 && W % 4 == 0)
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
    if (((uintptr_t)y & 7) || ((uintptr_t)w_packed & 15) || ((uintptr_t)x & 15)) return SASSD_EINVAL;
    if ((size_t)batch * Cin * H * W >= (1u << 29)) return SASSD_EINVAL;         // 32-bit per-lane element offsets
    WinoParams P;
    P.x = x; P.wp = w_packed; P.scale = scale; P.shift = shift; P.y = y;
    P.B = batch; P.Cin = Cin; P.Cout = Cout; P.H = H; P.W = W;
    P.TH = H / 2; P.TW = W / 2; P.tiles = batch * P.TH * P.TW;
    P.ncb64 = cdiv(Cout, kCoW);
    P.relu = relu;
    P.dbg = g_wino_dbg;
    const size_t lds = (size_t)(2 * kVBuf + 2 * kRawBuf) * sizeof(float);           // 139 264 B
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)conv2d_wino_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return sassd_launch_status();
        attr_done = true;
    }
    P.ngrp = cdiv(P.tiles, kNT);
    const int per_xcd = cdiv(cdiv(P.ngrp, 8), kGb) * kGb;       // groups per XCD, padded to whole blocks
    const int grid = per_xcd * P.ncb64 * 8;
    hipLaunchKernelGGL(conv2d_wino_kernel, dim3(grid), dim3(512), lds, (hipStream_t)stream_, P);
    return sassd_launch_status();
}
