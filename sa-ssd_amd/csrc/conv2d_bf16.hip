// conv2d_bf16.hip -- 3x3 pad-1 convolution of the BEV trunk on the bf16 MFMA pipe (BASELINE configs[2]: "training,
// batch=2/GPU ... bf16").  Forward of mmdet/models/necks/cmn.py:240-262 (the seven 3x3 convs of BEVNet) and, with the
// weights transposed and the taps mirrored, their data gradient (cuDNN under autocast in the reference).
//   y[b][co][y][x] = shift[co] + sum_{ci,ky,kx} bf16(w[co][ci][ky][kx]) * bf16(x[b][ci][y+ky-1][x+kx-1])
// NCHW fp32 activations in and out (BatchNorm / ReLU between the layers stay fp32), operands rounded to bf16
// (round-to-nearest-even) on their way into LDS / at pack time, fp32 accumulation on v_mfma_f32_32x32x16_bf16.
//
// MFMA-bound (83 GFLOP per 256->256 layer at B = 2 against 2.5 PFLOP/s), tiled as an implicit GEMM
// D[co][pixel] = sum_k W[co][k] X[k][pixel], k = (tap, ci):
//   * workgroup = 4 waves = 8 rows x 16 columns of one image (128 pixels) x 256 couts (COW = 4: a wave owns 64 couts
//     x all 128 pixels = 8 accumulator tiles) or x 128 couts (COW = 2: 64 couts x 64 pixels per wave).  The MFMA's 32
//     pixel columns are 2 image rows x 16 columns: the lane -> pixel map is free because every lane computes its own
//     LDS address, and 176 = 11 x 16 tiles the KITTI map with no padding.
//   * the input tile (10 rows x 20 columns x 32 channels) is staged through LDS as [row][column][channel] bf16 --
//     channel-minor, so a lane's B operand (8 consecutive channels of its pixel) is ONE ds_read_b128 and the nine taps
//     are plain address offsets.  The transposition NCHW -> channel-minor happens in registers: a thread loads 8
//     channel planes x one pixel pair (coalesced along the row) and writes two 16-byte channel vectors.  Pixel pitch
//     80 B and row pitch 1792 B make the 16 lanes of every ds_read_b128 phase hit 16 different bank quads.  Double
//     buffered: the next 32-channel chunk is in flight in registers during the 144 MFMAs of the current one.
//   * weights are packed once per update as bf16 [tap][ci/8][cout][8]: a lane's A operand is one 16-byte global load,
//     a wave reads 512 contiguous bytes, straight from L2 (1.2 MB per layer, shared by every workgroup) into a
//     three-deep register ring, two (tap, k-step) groups ahead of the MFMAs that consume them.
#include "common.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kTR = 8, kTC = 16;               // output tile: rows x columns
constexpr int kKC = 32;                        // input channels per LDS chunk (2 MFMA k-steps per tap)
constexpr int kLR = kTR + 2, kLC = kTC + 4;    // LDS tile: rows r0-1 .. r0+8, columns c0-2 .. c0+17 (even start)
constexpr int kPixB = 80;                      // bytes per pixel: 32 bf16 + 16 B pad
constexpr int kRowB = 1792;                    // bytes per tile row: 20 x 80 = 1600 -> multiple of 256
constexpr int kBufB = kLR * kRowB;             // 17 920 B
constexpr int kItems = kLR * (kLC / 2) * (kKC / 8);   // (row, pixel pair, 8-channel group) = 400

struct BfParams {
    const float *x;
    const unsigned short *wp;
    const float *shift;
    float *y;
    int B, Cin, CinP, Cout, H, W;                  // CinP: Cin rounded up to a whole 32-channel chunk
    int tiles_x, tiles_y;
};

__device__ __forceinline__ unsigned pack2(float lo, float hi)
{
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// w fp32 [Cout][Cin][3][3] -> bf16 [tap][CinP/8][Cout][8], CinP = Cin rounded up to 32 (zero weights for the padding)
__global__ void bf16_pack_kernel(const float *__restrict__ w, int Cout, int Cin, int CinP,
                                 unsigned short *__restrict__ out)
{
    const size_t n = (size_t)9 * CinP * Cout;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = i & 7;
    size_t r = i >> 3;
    const int co = r % Cout; r /= Cout;
    const int c8 = r % (CinP / 8);
    const int tap = r / (CinP / 8);
    const int ci = c8 * 8 + j;
    const float v = ci < Cin ? w[((size_t)co * Cin + ci) * 9 + tap] : 0.f;
    out[i] = __builtin_bit_cast(unsigned short, (__bf16)v);
}

// COW waves along cout (MB blocks of 32 each) x PXW waves along the 4 pixel blocks of the tile.  <8, 1, 1> (256 couts):
// every wave owns 32 couts x all 128 pixels, so no two waves fetch the same weights and one 16-byte weight load feeds
// four MFMAs -- with two waves per weight slice the vector L1 (64 B/clk) was the bottleneck, not the MFMA pipe.
template <int COW, int PXW, int MB>
__global__ void __launch_bounds__(64 * COW * PXW, 2) conv2d_bf16_kernel(BfParams p)
{
    constexpr int NT = 64 * COW * PXW;
    constexpr int NPB = 4 / PXW;                   // 32-pixel blocks (2 rows x 16 columns) per wave
    constexpr int kPer = (kItems + NT - 1) / NT;   // staging items per thread
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kBufB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = wave % COW, wpx = wave / COW;
    const int li = lane & 31, lh = lane >> 5;

    int wg = blockIdx.x;
    const int tx = wg % p.tiles_x; wg /= p.tiles_x;
    const int ty = wg % p.tiles_y; wg /= p.tiles_y;
    const int b = wg % p.B;
    const int cot = wg / p.B;
    const int r0 = ty * kTR, c0 = tx * kTC;
    const int co_w = (cot * COW + wco) * 32 * MB;   // this wave's first cout
    const size_t hw = (size_t)p.H * p.W;
    const float *xb = p.x + (size_t)b * p.Cin * hw;

    // ---- staging: item -> (8-channel group, tile row, pixel pair); loads are unconditional (clamped address), the
    // zero padding of the image border is a select when the values are rounded and stored
    f32x2 st[kPer][8];
    int s_lds[kPer];
    bool s_ok[kPer];
    const float *s_ptr[kPer];
    int s_g[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const int e = tid + NT * i;
        const int pr = e % (kLC / 2), row = (e / (kLC / 2)) % kLR, g = e / (kLR * (kLC / 2));
        const int yy = r0 - 1 + row, xx = c0 - 2 + 2 * pr;                    // xx even, W even: xx+1 valid with xx
        s_ok[i] = e < kItems && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
        s_lds[i] = e < kItems ? row * kRowB + 2 * pr * kPixB + g * 16 : -1;
        s_ptr[i] = xb + (s_ok[i] ? (size_t)yy * p.W + xx : 0);
        s_g[i] = g * 8;
    }
    // channels past Cin (padding of the last chunk) read the last real plane: their weights are zero
    auto fetch = [&](int ci0) {
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ch = s_ok[i] ? min(ci0 + s_g[i] + j, p.Cin - 1) : 0;
                st[i][j] = *(const f32x2 *)(s_ptr[i] + (size_t)ch * hw);
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            if (s_lds[i] < 0) continue;
            u32x4 lo, hi;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lo[j] = s_ok[i] ? pack2(st[i][2 * j][0], st[i][2 * j + 1][0]) : 0u;      // pixel 2*pr
                hi[j] = s_ok[i] ? pack2(st[i][2 * j][1], st[i][2 * j + 1][1]) : 0u;      // pixel 2*pr + 1
            }
            *(u32x4 *)(lds + buf * kBufB + s_lds[i]) = lo;
            *(u32x4 *)(lds + buf * kBufB + s_lds[i] + kPixB) = hi;
        }
    };

    // ---- operands
    // A: packed weights, element ((tap * Cin/8 + c8) * Cout + co) * 8; this lane: co = co_w + m*32 + li, c8 += lh
    const unsigned short *wl = p.wp + ((size_t)lh * p.Cout + co_w + li) * 8;
    const size_t w_c8 = (size_t)p.Cout * 8;                 // elements per 8-channel group
    const size_t w_tap = (size_t)(p.CinP / 8) * w_c8;       // elements per tap
    // B: LDS byte offset of this lane's pixel for block n: rows 2*(wpx*NPB+n) + (li>>4), column (li&15) + 1
    const int b_off = ((wpx * NPB * 2 + (li >> 4)) * kRowB) + ((li & 15) + 1) * kPixB + lh * 16;

    f32x16 acc[MB][NPB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NPB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int nchunk = p.CinP / kKC;
    u32x4 aring[3][MB];
    auto load_a = [&](int slot, int chunk, int s) {          // s = 2*tap + kstep, chunk clamped by the caller
        const int tap = s >> 1, ks = s & 1;
        const unsigned short *q = wl + tap * w_tap + (size_t)(chunk * (kKC / 8) + ks * 2) * w_c8;
#pragma unroll
        for (int m = 0; m < MB; ++m) aring[slot][m] = *(const u32x4 *)(q + m * 32 * 8);
    };

    // Cout not a multiple of the workgroup's cout tile: the waves past Cout only help staging the input tile
    const bool active = co_w < p.Cout;
    fetch(0);
    stash(0);
    if (active) {
        load_a(0, 0, 0);
        load_a(1, 0, 1);
    }
    __syncthreads();

    for (int c = 0; c < nchunk; ++c) {
        const int cn = min(c + 1, nchunk - 1);
        if (c + 1 < nchunk) fetch((c + 1) * kKC);
        const unsigned char *xt = lds + (c & 1) * kBufB + b_off;
        if (active) {
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            // weights two steps ahead (the last two steps of a chunk fetch the next chunk's first two)
            if (s + 2 < 18) load_a((s + 2) % 3, c, s + 2);
            else load_a((s + 2) % 3, cn, s + 2 - 18);
            const int tap = s >> 1, ks = s & 1;
            const int ky = tap / 3, kx = tap % 3;
            u32x4 bf[NPB];
#pragma unroll
            for (int n = 0; n < NPB; ++n)
                bf[n] = *(const u32x4 *)(xt + (2 * n + ky) * kRowB + kx * kPixB + ks * 32);
#pragma unroll
            for (int n = 0; n < NPB; ++n)
#pragma unroll
                for (int m = 0; m < MB; ++m)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aring[s % 3][m]),
                                                                       __builtin_bit_cast(bf16x8, bf[n]), acc[m][n],
                                                                       0, 0, 0);
        }
        }
        if (c + 1 < nchunk) stash((c + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: D row = cout = (r & 3) + 8 * (r >> 2) + 4 * lh, column = pixel li -> (row li >> 4, column li & 15)
    float *yb = p.y + (size_t)b * p.Cout * hw;
    if (!active) return;
#pragma unroll
    for (int n = 0; n < NPB; ++n) {
        const int yy = r0 + 2 * (wpx * NPB + n) + (li >> 4), xx = c0 + (li & 15);
        if (yy >= p.H) continue;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_w + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float sh = p.shift ? p.shift[co] : 0.f;
                yb[(size_t)co * hw + (size_t)yy * p.W + xx] = acc[m][n][r] + sh;
            }
    }
}
}  // namespace

extern "C" int sassd_conv2d_bf16_supported(int Cin, int Cout, int H, int W)
{
    return Cin >= 1 && Cout >= 32 && Cout % 32 == 0 && H >= 1 && W >= 16 && W % 16 == 0;
}

extern "C" size_t sassd_conv2d_bf16_packed_elems(int Cin, int Cout) { return (size_t)9 * align_up(Cin, 32) * Cout; }

extern "C" int sassd_conv2d_bf16_pack_weight(const float *w, int Cout, int Cin, void *packed, void *stream_)
{
    if (!w || !packed || Cin < 1 || Cout < 1) return SASSD_EINVAL;
    const size_t n = sassd_conv2d_bf16_packed_elems(Cin, Cout);
    hipLaunchKernelGGL(bf16_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, w, Cout,
                       Cin, (int)align_up(Cin, 32), (unsigned short *)packed);
    return sassd_launch_status();
}

extern "C" int sassd_conv2d_bf16_fwd(const float *x, const void *w_packed, const float *shift, float *y, int batch,
                                     int Cin, int Cout, int H, int W, void *stream_)
{
    if (!x || !w_packed || !y || batch < 1) return SASSD_EINVAL;
    if (!sassd_conv2d_bf16_supported(Cin, Cout, H, W)) return SASSD_EINVAL;
    BfParams p;
    p.x = x; p.wp = (const unsigned short *)w_packed; p.shift = shift; p.y = y;
    p.B = batch; p.Cin = Cin; p.CinP = (int)align_up(Cin, 32); p.Cout = Cout; p.H = H; p.W = W;
    p.tiles_x = W / kTC; p.tiles_y = cdiv(H, kTR);
    hipStream_t s = (hipStream_t)stream_;
    const long tiles = (long)p.tiles_x * p.tiles_y * batch;
    if (Cout % 256 == 0)
        hipLaunchKernelGGL((conv2d_bf16_kernel<8, 1, 1>), dim3((unsigned)(tiles * (Cout / 256))), dim3(512), 0, s, p);
    else        // 128-cout tiles; the last one may be partly idle (Cout = 320: 3 tiles, 2.5 used)
        hipLaunchKernelGGL((conv2d_bf16_kernel<4, 1, 1>), dim3((unsigned)(tiles * cdiv(Cout, 128))), dim3(256), 0, s, p);
    return sassd_launch_status();
}
