// conv2d.hip -- dense NCHW fp32 2-D convolution (3x3 pad 1 / 1x1) as an implicit GEMM on fp32 MFMA.
//
// Replaces nn.Conv2d + nn.BatchNorm2d(eval) + ReLU of BEVNet (mmdet/models/necks/cmn.py:233-282, 306 GFLOP per
// KITTI frame -- the dominant cost of the whole path), the three 1x1 SSD head convs
// (single_stage_heads/ssd_rotate_head.py:120-125) and PSWarpHead.convs (:424-429).
//
// GEMM view: out^T[cout][pixel] = sum_{tap,cin} W[cout][tap,cin] * in[cin][pixel + shift(tap)],
//   MFMA "A" = weights (M = cout), "B" = pixels (N = 32 consecutive linear pixels -> coalesced 128-B stores),
//   v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), 64 cycles, peak 157.3 TF.
// Tiling (wave = 64 lanes, 4 waves / workgroup, 1 workgroup / CU):
//   workgroup tile = 128 couts x 288 linear pixels  (4 x 9 MFMA tiles; 35200 px * 256 couts -> 246 workgroups,
//   i.e. ONE round on 256 CUs at 96 % tile utilisation);  wave w owns cout tile w and all 9 pixel segments
//   (9 x 16 accumulator VGPRs).  Small-Cout layers (28 / 20 channels) use 32 couts x 256 px, 2 segments / wave.
//   K loop: 8 input channels per step.  The input PATCH (8 ch x <=5 rows x (W+2), zero halo) is staged ONCE in LDS
//   and serves all 9 taps through shifted LDS reads -- 9x less L2->LDS traffic than per-tap im2col staging and no
//   border masks in the inner loop.  LDS images are k-major ([k][cout] / [k][pixel]) so both MFMA operand reads
//   are 32 consecutive dwords per half-wave: conflict free, no swizzle.
// Roofline: MFMA (fp32) bound; FLOPs = 2*Cout*Cin*k*k*H*W.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kKC = 8;          // input channels per K step

template <int CT>
struct ConvCfg {
    static constexpr int NSEG = (CT == 4) ? 9 : 8;    // 32-pixel segments per workgroup
    static constexpr int SPW = (CT == 4) ? 9 : 2;     // segments per wave
    static constexpr int BMC = CT * 32;               // couts per workgroup
    static constexpr int PIX = NSEG * 32;
};

struct ConvParams {
    const float *x, *wp, *scale, *shift;
    float *y;
    int B, Cin, Cout, CoutPad, H, W, HW;
    int ntp;        // pixel tiles per image
    int ncg;        // cout groups
    int NPR;        // patch rows allocated in LDS
    int PW;         // patch width = W + 2
    int relu;
};

// w [Cout][Cin][k][k] -> wp [Cin/8][TAPS][8][CoutPad]
__global__ void conv_pack_kernel(const float *__restrict__ w, int Cout, int Cin, int ks, int CoutPad,
                                 float *__restrict__ wp)
{
    const int taps = ks * ks;
    const int CinPad = (Cin + kKC - 1) / kKC * kKC;
    const size_t total = (size_t)CinPad * taps * CoutPad;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int co = i % CoutPad;
    const int kc = (i / CoutPad) % kKC;
    const int tap = (i / ((size_t)CoutPad * kKC)) % taps;
    const int chunk = i / ((size_t)CoutPad * kKC * taps);
    const int ci = chunk * kKC + kc;
    wp[i] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * taps + tap] : 0.f;
}

template <int CT, int TAPS>
__global__ void __launch_bounds__(256) conv2d_kernel(ConvParams P)
{
    using C = ConvCfg<CT>;
    constexpr int SPW = C::SPW, BMC = C::BMC, PIX = C::PIX;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *W_s = smem;                               // [TAPS*8][BMC]
    float *P_s = smem + TAPS * kKC * BMC;            // [8][NPR][PW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int g = blockIdx.x;
    const int cg = g % P.ncg; g /= P.ncg;
    const int pt = g % P.ntp;
    const int b = g / P.ntp;

    const int p0 = pt * PIX;
    const int plast = min(p0 + PIX, P.HW) - 1;
    const int y_first = p0 / P.W;
    const int nprows = plast / P.W - y_first + 3;     // rows y_first-1 .. y_last+1
    const int PW = P.PW;
    const int chs = P.NPR * PW;                       // channel stride inside P_s

    const int ct = (CT == 4) ? wave : 0;
    const int seg0 = (CT == 4) ? 0 : wave * SPW;

    int laddr[SPW];
#pragma unroll
    for (int j = 0; j < SPW; ++j) {
        int p = p0 + (seg0 + j) * 32 + (lane & 31);
        p = min(p, P.HW - 1);
        const int yy = p / P.W, xx = p - yy * P.W;
        laddr[j] = (yy - y_first + 1) * PW + xx + 1 + (lane >> 5) * chs;   // lanes 32..63 read channel kc+1
    }

    f32x16 acc[SPW];
#pragma unroll
    for (int j = 0; j < SPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const float *xb = P.x + (size_t)b * P.Cin * P.HW;
    const int nchunk = (P.Cin + kKC - 1) / kKC;
    const int aoff = (lane >> 5) * BMC + ct * 32 + (lane & 31);

    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();
        // ---- stage weights: TAPS*8 rows x BMC couts ------------------------------------------------
        {
            const float *src = P.wp + (size_t)ch * TAPS * kKC * P.CoutPad + cg * BMC;
            constexpr int V = BMC / 4;
            for (int i = tid; i < TAPS * kKC * V; i += 256) {
                const int row = i / V, c4 = i - row * V;
                const float4 v = *(const float4 *)(src + (size_t)row * P.CoutPad + c4 * 4);
                *(float4 *)(W_s + row * BMC + c4 * 4) = v;
            }
        }
        // ---- stage the input patch with zero halo: rows handled round-robin by the 4 waves ---------
        {
            const int nrows_tot = kKC * nprows;
            for (int r = wave; r < nrows_tot; r += 4) {
                const int kc = r / nprows, pr = r - kc * nprows;
                const int yy = y_first - 1 + pr;
                const bool rowok = (yy >= 0) && (yy < P.H) && (ch * kKC + kc < P.Cin);
                const float *srow = xb + ((size_t)(rowok ? ch * kKC + kc : 0) * P.H + (rowok ? yy : 0)) * P.W;
                float *drow = P_s + kc * chs + pr * PW;
                for (int px = lane; px < PW; px += 64) {
                    const int xx = px - 1;
                    float v = 0.f;
                    if (rowok && xx >= 0 && xx < P.W) v = srow[xx];
                    drow[px] = v;
                }
            }
        }
        __syncthreads();
        // ---- MFMA: 9 taps x 4 k-steps x SPW segments ----------------------------------------------
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int toff = (TAPS == 9) ? ((tap / 3 - 1) * PW + (tap % 3 - 1)) : 0;
#pragma unroll
            for (int s = 0; s < kKC / 2; ++s) {
                const float a = W_s[(tap * kKC + 2 * s) * BMC + aoff];
                const float *pb = P_s + 2 * s * chs + toff;
#pragma unroll
                for (int j = 0; j < SPW; ++j) {
                    const float bv = pb[laddr[j]];
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[j], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: D[row = cout][col = pixel] ----------------------------------------------------------
    float sc[16], sh[16];
    int co[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        co[r] = cg * BMC + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const bool ok = co[r] < P.Cout;
        sc[r] = (ok && P.scale) ? P.scale[co[r]] : 1.f;
        sh[r] = (ok && P.shift) ? P.shift[co[r]] : 0.f;
    }
    float *yb = P.y + (size_t)b * P.Cout * P.HW;
#pragma unroll
    for (int j = 0; j < SPW; ++j) {
        const int p = p0 + (seg0 + j) * 32 + (lane & 31);
        if (p < P.HW) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (co[r] < P.Cout) {
                    float v = acc[j][r] * sc[r] + sh[r];
                    if (P.relu) v = fmaxf(v, 0.f);
                    yb[(size_t)co[r] * P.HW + p] = v;
                }
            }
        }
    }
}

template <int CT, int TAPS>
int launch_conv(ConvParams P, hipStream_t stream)
{
    using C = ConvCfg<CT>;
    P.ntp = cdiv(P.HW, C::PIX);
    P.ncg = P.CoutPad / C::BMC;
    P.PW = P.W + 2;
    P.NPR = (C::PIX - 1 + P.W - 1) / P.W + 1 + 2;
    const size_t lds = ((size_t)TAPS * kKC * C::BMC + (size_t)kKC * P.NPR * P.PW) * sizeof(float);
    if (lds > 160 * 1024) return SASSD_EINVAL;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)conv2d_kernel<CT, TAPS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
        attr_set = true;
    }
    const int grid = P.B * P.ntp * P.ncg;
    hipLaunchKernelGGL((conv2d_kernel<CT, TAPS>), dim3(grid), dim3(256), lds, stream, P);
    return sassd_launch_status();
}

inline int cout_pad(int Cout) { return (Cout > 64) ? cdiv(Cout, 128) * 128 : cdiv(Cout, 32) * 32; }

}  // namespace

extern "C" size_t sassd_conv2d_packed_floats(int Cin, int Cout, int ksize)
{
    return (size_t)cdiv(Cin, kKC) * kKC * ksize * ksize * cout_pad(Cout);
}

extern "C" int sassd_conv2d_pack_weight(const float *w, int Cout, int Cin, int ksize, float *packed, void *stream_)
{
    if (!w || !packed || (ksize != 1 && ksize != 3) || Cin < 1) return SASSD_EINVAL;
    const int cp = cout_pad(Cout);
    const size_t total = (size_t)cdiv(Cin, kKC) * kKC * ksize * ksize * cp;
    hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, w,
                       Cout, Cin, ksize, cp, packed);
    return sassd_launch_status();
}

extern "C" int sassd_conv2d_fwd(const float *x, const float *w_packed, const float *scale, const float *shift,
                                int relu, float *y, int batch, int Cin, int Cout, int H, int W, int ksize,
                                void *stream_)
{
    if (!x || !w_packed || !y || batch < 1 || Cin < 1 || (ksize != 1 && ksize != 3)) return SASSD_EINVAL;
    if (H < 1 || W < 2) return SASSD_EINVAL;
    ConvParams P;
    P.x = x; P.wp = w_packed; P.scale = scale; P.shift = shift; P.y = y;
    P.B = batch; P.Cin = Cin; P.Cout = Cout; P.CoutPad = cout_pad(Cout); P.H = H; P.W = W; P.HW = H * W;
    P.relu = relu;
    hipStream_t stream = (hipStream_t)stream_;
    if (P.CoutPad % 128 == 0) return ksize == 3 ? launch_conv<4, 9>(P, stream) : launch_conv<4, 1>(P, stream);
    return ksize == 3 ? launch_conv<1, 9>(P, stream) : launch_conv<1, 1>(P, stream);
}
