// conv1x1_bf16.hip -- 1x1 convolution over NCHW fp32 maps on the bf16 MFMA (training, set_bev_precision("bf16")).
//
// BEVNet's conv7 (256 -> 256, mmdet/models/necks/cmn.py:262), its data gradient and the data gradient of the fused SSD
// head's 1x1 convs (ssd_rotate_head.py:120-125, 20 -> 256) ran on the fp32-MFMA direct kernel in rounds 1-6: 150 + 150 +
// 64 us per step at batch 2 for layers that move 144 MB each -- an HBM stream (~30 us), not a matrix problem, once the
// products run at the bf16 rate.  Same arithmetic contract as conv2d_bf16.hip: operands rounded to bf16 (nearest even) --
// the weights at pack time, the activations on their way into the MFMA --, fp32 accumulation, fp32 tensors in and out.
//
//   y[b][co][p] = sum_ci W[co][ci] x[b][ci][p] (+ shift[co])          v_mfma_f32_32x32x16_bf16, A = W, B = x
//
// Register-stationary weights, no staging: a wave owns 64 output channels and keeps their whole [64 x CinP] bf16 weight block
// as MFMA A fragments in registers (2 x KS x 4 VGPRs, KS = CinP / 16 <= 16), loaded once; it then walks 64-pixel tiles.  For a
// tile it streams x straight from global memory in B-fragment order -- lane (n, kh) loads the pixel PAIR (2n, 2n + 1) of
// channels 16 ks + 8 kh + 0..7 as 8-byte loads (a half-wave reads one 256-byte row segment per channel), two k-steps per
// group, the next group (or the next tile's first) in flight while the current one is converted and multiplied.  The even
// pixels of the pairs form one MFMA column tile, the odd pixels the other, so the two accumulators of a pixel pair leave as
// one 8-byte LDS write into a wave-private [64 co][64 px] image and the image leaves as 16-byte row stores.  No barriers: the
// four waves of a workgroup (the four 64-channel groups of a 256-channel layer) read the same x tile at about the same time,
// i.e. from L1 / L2; HBM sees every map once.
#include "common.h"

namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kPix = 64;            // pixels per tile
constexpr int kImgPitch = 68;       // floats per row of the epilogue image (16 x 17 bytes: conflict-free 16-byte reads)

struct C1Params {
    const float *x;
    const u32x4 *wp;        // [ng][KS][2][64 lanes] x 16 bytes
    const float *shift;     // optional [Cout]
    float *y;
    int B, Cin, Cout, HW;
    int ng;                 // 64-channel output groups
    int ntile;              // tiles per image
    int T;                  // B * ntile
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// w fp32 [Cout][Cin] (transposed = 0) or [Cin][Cout] read as its transpose (transposed = 1: the data gradient's weights)
// -> A fragments: element j of lane l of (group g, k-step ks, row tile a) = W[64 g + 32 a + (l & 31)][16 ks + 8 (l >> 5) + j]
__global__ void conv1x1_bf16_pack_kernel(const float *__restrict__ w, int Cout, int Cin, int transposed, int KS, int ng,
                                         unsigned short *__restrict__ out)
{
    const size_t total = (size_t)ng * KS * 2 * 64 * 8;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i & 7), l = (int)((i >> 3) & 63), a = (int)((i >> 9) & 1);
    const size_t r = i >> 10;
    const int ks = (int)(r % KS), g = (int)(r / KS);
    const int co = 64 * g + 32 * a + (l & 31), ci = 16 * ks + 8 * (l >> 5) + j;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = transposed ? w[(size_t)ci * Cout + co] : w[(size_t)co * Cin + ci];
    out[i] = __builtin_bit_cast(unsigned short, (__bf16)v);
}

template <int KS>
__global__ void __launch_bounds__(256) conv1x1_bf16_kernel(C1Params P)
{
    constexpr int GS = KS < 2 ? KS : 2;             // k-steps per load group
    constexpr int NG = KS / GS;                     // groups per tile
    static_assert(KS % GS == 0, "whole groups");
    extern __shared__ __attribute__((aligned(16))) float c1_img[];       // [4 waves][64][kImgPitch]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, kh = lane >> 5;
    const int wid = blockIdx.x * 4 + wave;
    const int g = wid % P.ng, s0 = wid / P.ng, nstream = (int)(gridDim.x * 4) / P.ng;
    if (s0 >= nstream) return;                      // (ng = 3: the waves left over)
    float *img = c1_img + wave * 64 * kImgPitch;

    // Addresses: a load reads  x + [(b Cin + 16 ks + i) HW + 64 pt] (wave-uniform: SGPR base)  +  [(8 kh) HW + 2 n] (one VGPR for
    // the whole kernel) -- the saddr form of global_load; 128 hoisted 64-bit lane addresses would not fit the register file.
    // Channels past Cin exist only in the last k-step of a ragged Cin (their weights are zero): its eight lane offsets clamp the
    // channel.  Pixels past the map (partial last tile) are clamped per tile; their results are not stored.
    const unsigned HW = (unsigned)P.HW;
    constexpr int KSF = KS - 1;                     // k-steps that are whole for every supported Cin (16 (KS - 1) < Cin)
    const unsigned chan_off = 8u * kh * HW;
    unsigned last_off[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int ci = 16 * KSF + 8 * kh + i;
        ci = ci < P.Cin ? ci : P.Cin - 1;
        last_off[i] = (unsigned)(ci - 16 * KSF) * HW;
    }
    struct Tile { const float *base; unsigned pix; };
    auto tile_of = [&](int t) -> Tile {
        const int b = t / P.ntile, pt = t - b * P.ntile;
        const int room = P.HW - 2 - pt * kPix;                          // last pixel pair inside the map, tile-relative
        Tile r;
        r.base = P.x + (size_t)b * P.Cin * HW + (size_t)pt * kPix;
        r.pix = (unsigned)min(2 * n, room);
        return r;
    };
    f32x2 raw[2][GS][8];
    auto issue = [&](const Tile &tl, int grp, f32x2 (&dst)[GS][8]) {
#pragma unroll
        for (int q = 0; q < GS; ++q) {
            const int ks = grp * GS + q;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float *sb = tl.base + (size_t)(16 * ks + (ks < KSF ? i : 0)) * HW;           // wave-uniform
                const unsigned vo = (ks < KSF ? chan_off : last_off[i]) + tl.pix;                   // per lane, floats
                dst[q][i] = *reinterpret_cast<const f32x2 *>(reinterpret_cast<const char *>(sb) + (size_t)vo * 4u);
            }
        }
    };
    int t = s0;
    if (t >= P.T) return;
    Tile xt = tile_of(t);
    issue(xt, 0, raw[0]);                           // the first loads go out before the weights

    bf16x8 wf[KS][2];
    {
        const u32x4 *wp = P.wp + (size_t)g * KS * 2 * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int a = 0; a < 2; ++a) wf[ks][a] = __builtin_bit_cast(bf16x8, wp[(ks * 2 + a) * 64]);
    }
    for (; t < P.T; t += nstream) {
        const int tn = t + nstream;
        const Tile xn = tn < P.T ? tile_of(tn) : xt;
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][e][r] = 0.f;
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            // (NG is even or 1: the ring parity restarts with every tile; NG = 1 keeps one buffer and no cross-tile prefetch)
            if (grp + 1 < NG) issue(xt, grp + 1, raw[(grp + 1) & 1]);
            else if (NG > 1 && tn < P.T) issue(xn, 0, raw[0]);
#pragma unroll
            for (int q = 0; q < GS; ++q) {
                const f32x2(&v)[8] = raw[grp & 1][q];
                u32x4 be, bo;
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    be[h] = pack_bf16(v[2 * h][0], v[2 * h + 1][0]);
                    bo[h] = pack_bf16(v[2 * h][1], v[2 * h + 1][1]);
                }
                const int ks = grp * GS + q;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][a], __builtin_bit_cast(bf16x8, be), acc[a][0], 0, 0, 0);
                    acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][a], __builtin_bit_cast(bf16x8, bo), acc[a][1], 0, 0, 0);
                }
            }
        }
        // epilogue: D[row = 32 a + (r & 3) + 8 (r >> 2) + 4 kh][pixel 2n + e] -> wave-private image -> 16-byte row stores
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const f32x2 v = {acc[a][0][r], acc[a][1][r]};
                *reinterpret_cast<f32x2 *>(img + (32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh) * kImgPitch + 2 * n) = v;
            }
        const int b = t / P.ntile, pt = t - b * P.ntile;
        const int c4 = lane & 15, p = pt * kPix + 4 * c4;
        float *yb = P.y + ((size_t)b * P.Cout + 64 * g) * HW + p;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 4 * i + (lane >> 4);
            f32x4 v = *reinterpret_cast<const f32x4 *>(img + row * kImgPitch + 4 * c4);
            if (64 * g + row < P.Cout && p < P.HW) {                      // (HW % 4 == 0: a quad is inside or outside)
                if (P.shift) {
                    const float sh = P.shift[64 * g + row];
                    v += sh;
                }
                *reinterpret_cast<f32x4 *>(yb + (size_t)row * HW) = v;
            }
        }
        if (NG == 1 && tn < P.T) issue(xn, 0, raw[0]);
        xt = xn;
    }
}

int c1_ksteps(int Cin) { return Cin <= 16 ? 1 : Cin <= 32 ? 2 : Cin <= 64 ? 4 : Cin <= 128 ? 8 : 16; }
}  // namespace

extern "C" int sassd_conv1x1_bf16_supported(int Cin, int Cout, int HW)
{
    // (only the LAST k-step of the kernel's power-of-two K may be ragged: Cin in (16 (KS - 1), 16 KS])
    return Cin >= 1 && Cin <= 256 && Cin > 16 * (c1_ksteps(Cin) - 1) && Cout >= 1 && HW >= 4 && HW % 4 == 0;
}

extern "C" size_t sassd_conv1x1_bf16_packed_elems(int Cin, int Cout)
{
    if (!sassd_conv1x1_bf16_supported(Cin, Cout, 4)) return 0;
    return (size_t)cdiv(Cout, 64) * c1_ksteps(Cin) * 2 * 64 * 8;
}

extern "C" int sassd_conv1x1_bf16_pack_weight(const float *w, int Cout, int Cin, int transposed, void *packed, void *stream_)
{
    if (!w || !packed || !sassd_conv1x1_bf16_supported(Cin, Cout, 4)) return SASSD_EINVAL;
    const size_t total = sassd_conv1x1_bf16_packed_elems(Cin, Cout);
    hipLaunchKernelGGL(conv1x1_bf16_pack_kernel, dim3((unsigned)cdiv(total, (size_t)256)), dim3(256), 0, (hipStream_t)stream_, w,
                       Cout, Cin, transposed, c1_ksteps(Cin), cdiv(Cout, 64), (unsigned short *)packed);
    return sassd_launch_status();
}

extern "C" int sassd_conv1x1_bf16_fwd(const float *x, const void *w_packed, const float *shift, float *y, int batch, int Cin,
                                      int Cout, int HW, void *stream_)
{
    if (!x || !w_packed || !y || batch < 1 || !sassd_conv1x1_bf16_supported(Cin, Cout, HW)) return SASSD_EINVAL;
    if (((uintptr_t)x & 7) || ((uintptr_t)y & 15) || ((uintptr_t)w_packed & 15)) return SASSD_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    C1Params P;
    P.x = x; P.wp = (const u32x4 *)w_packed; P.shift = shift; P.y = y;
    P.B = batch; P.Cin = Cin; P.Cout = Cout; P.HW = HW;
    P.ng = cdiv(Cout, 64);
    P.ntile = cdiv(HW, kPix);
    if ((long)batch * P.ntile > 0x7fffffffL) return SASSD_EINVAL;
    P.T = batch * P.ntile;
    const int KS = c1_ksteps(Cin);
    // Workgroups: one per CU for the register-heavy forms (KS >= 8: 128 weight registers), two otherwise (the 70 KB epilogue
    // image); the tile streams -- 4 / ng per workgroup -- get equal runs of tiles.
    int ncu = 256;
    {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (ncu < 1) ncu = 256;
    }
    const int slots = ncu * (KS >= 8 ? 1 : 2);
    const long streams_needed = P.T;                                   // at most one stream per tile
    long nstream = (long)slots * 4 / P.ng;
    if (nstream < 1) nstream = 1;
    if (nstream > streams_needed) nstream = streams_needed;
    const long per = cdiv((long)P.T, nstream);                         // tiles per stream
    nstream = cdiv((long)P.T, per);
    const int grid = (int)cdiv(nstream * P.ng, 4L);
    const size_t lds = (size_t)4 * 64 * kImgPitch * sizeof(float);
    static std::atomic<unsigned long long> done[5];
    const void *fn = nullptr;
    int slot = 0;
#define SASSD_C1_GO(KSV, SLOT)                                                                                             \
    {                                                                                                                      \
        fn = (const void *)conv1x1_bf16_kernel<KSV>; slot = SLOT;                                                          \
        int rc = sassd_dyn_lds(fn, lds, done[slot]);                                                                       \
        if (rc) return rc;                                                                                                 \
        hipLaunchKernelGGL(conv1x1_bf16_kernel<KSV>, dim3(grid), dim3(256), lds, stream, P);                               \
    }
    switch (KS) {
    case 1: SASSD_C1_GO(1, 0) break;
    case 2: SASSD_C1_GO(2, 1) break;
    case 4: SASSD_C1_GO(4, 2) break;
    case 8: SASSD_C1_GO(8, 3) break;
    default: SASSD_C1_GO(16, 4) break;
    }
#undef SASSD_C1_GO
    (void)fn; (void)slot;
    return sassd_launch_status();
}
