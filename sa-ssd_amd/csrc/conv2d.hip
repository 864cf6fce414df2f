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
//   wide layers (Cout multiple of 128): workgroup tile = 128 couts x 288 linear pixels (4 x 9 MFMA tiles;
//     35200 px * 256 couts -> 246 workgroups = ONE round on 256 CUs at 96 % tile utilisation); wave w owns cout
//     tile w and all 9 pixel segments (9 x 16 accumulator registers).
//   narrow layers (Cout <= 64, e.g. 28 / 20): workgroup tile = 32 couts x 160 px; the 4 waves split the K
//     (input channel) dimension and reduce through LDS at the end.
//   K loop: KC input channels per step (8 for 3x3, 16 for 1x1).  The input PATCH (KC ch x <=5 rows x (W+8), zero
//   halo) is staged ONCE in LDS and serves all 9 taps through shifted LDS reads -- 9x less L2->LDS traffic than
//   per-tap im2col staging, no border masks in the inner loop.  LDS images are k-major ([k][cout] / [k][pixel]) so
//   both MFMA operand reads are 32 consecutive dwords per half-wave: conflict free without a swizzle.
//   Pipelining: LDS is double buffered and filled by direct global->LDS DMA (global_load_lds_dwordx4, 1 KiB per wave
//   instruction, no staging VGPRs, no ds_write): the DMA of chunk i+1 is issued before the 324 MFMAs of chunk i and
//   drained (vmcnt) at the single barrier per chunk.  The LDS image is lane-linear, so the zero halo / out-of-image
//   rows are produced by pointing those lanes' SOURCE address at a 16-byte zero page -- no masks, no memset.
//   (Rows not 16-B aligned, W % 4 != 0, fall back to register staging.)
// Roofline: MFMA (fp32) bound; FLOPs = 2*Cout*Cin*k*k*H*W.
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

__device__ __attribute__((aligned(16))) float g_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

template <int CT, int TAPS, int VEC>
struct ConvCfg {
    static constexpr bool KSPLIT = (CT == 1);
    static constexpr int NSEG = (CT == 4) ? 9 : 5;     // 32-pixel segments per workgroup (all owned by every wave)
    static constexpr int KC = (TAPS == 9) ? 8 : 16;    // input channels per K step
    static constexpr int HALO = (TAPS == 9) ? 1 : 0;
    static constexpr int BMC = CT * 32;                // couts per workgroup
    static constexpr int PIX = NSEG * 32;
    static constexpr int WV4 = TAPS * KC * BMC / 4;    // weight float4 per chunk
    static constexpr int NLW = (WV4 + 255) / 256;      // weight float4 per thread
    static constexpr int MAXLP = (VEC == 4) ? 1 : 40;  // register-staged patch loads per thread (VEC == 1 path)
    static constexpr int WS = TAPS * KC * BMC;         // floats (multiple of 256)
    static constexpr int NWJ = (WS / 256 + 3) / 4;     // weight DMA wave-loads per wave
    static constexpr int MAXPJ = 10;                   // patch DMA wave-loads per wave (upper bound)
};

#ifndef SASSD_DMA_TAPS
#define SASSD_DMA_TAPS 3
#endif
constexpr int kDmaTaps = SASSD_DMA_TAPS;   // taps over which the next chunk's DMA issue is spread

struct ConvParams {
    const float *x, *wp, *scale, *shift;
    float *y;
    int B, Cin, Cout, CoutPad, H, W, HW;
    int ntp;        // pixel tiles per image
    int ncg;        // cout groups
    int NPR;        // patch rows allocated in LDS
    int PW;         // patch row stride (floats): 4 left pad (col 3 = halo) + W + right halo/pad, multiple of 4
    int relu;
    int dbg;        // ablation switches (tools/run_conv.py): 1 = no DMA after the first chunk, 2 = no barrier
};

// w [Cout][Cin][k][k] -> wp [Cin padded to KC][TAPS][CoutPad] grouped as [chunk of KC][tap][kc][CoutPad]
__global__ void conv_pack_kernel(const float *__restrict__ w, int Cout, int Cin, int ks, int CoutPad, int KC,
                                 float *__restrict__ wp)
{
    const int taps = ks * ks;
    const int CinPad = (Cin + KC - 1) / KC * KC;
    const size_t total = (size_t)CinPad * taps * CoutPad;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int co = i % CoutPad;
    const int kc = (i / CoutPad) % KC;
    const int tap = (i / ((size_t)CoutPad * KC)) % taps;
    const int chunk = i / ((size_t)CoutPad * KC * taps);
    const int ci = chunk * KC + kc;
    wp[i] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * taps + tap] : 0.f;
}

template <int CT, int TAPS, int VEC>
__global__ void __launch_bounds__(256) conv2d_kernel(ConvParams P)
{
    using C = ConvCfg<CT, TAPS, VEC>;
    constexpr int NSEG = C::NSEG, BMC = C::BMC, PIX = C::PIX, KC = C::KC, HALO = C::HALO, NLW = C::NLW,
                  MAXLP = C::MAXLP, WS = C::WS;
    constexpr bool KSPLIT = C::KSPLIT;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int g = blockIdx.x;
    const int cg = g % P.ncg; g /= P.ncg;
    const int pt = g % P.ntp;
    const int b = g / P.ntp;

    const int p0 = pt * PIX;
    const int plast = min(p0 + PIX, P.HW) - 1;
    const int y_first = p0 / P.W;
    const int nprows = plast / P.W - y_first + 1 + 2 * HALO;
    const int PW = P.PW;
    const int chs = P.NPR * PW;                       // channel stride inside the patch image
    const int PS = KC * chs;
    const int PSpad = (PS + 255) / 256 * 256;
    const int bufs = WS + PSpad;                      // floats per LDS buffer (1 KiB aligned)
    constexpr bool DMA = (VEC == 4);

    const float *xb = P.x + (size_t)b * P.Cin * P.HW;
    int laddr[NSEG];
#pragma unroll
    for (int j = 0; j < NSEG; ++j) {
        int p = p0 + j * 32 + (lane & 31);
        p = min(p, P.HW - 1);
        const int yy = p / P.W, xx = p - yy * P.W;
        laddr[j] = (yy - y_first + HALO) * PW + 4 + xx + (lane >> 5) * chs;   // lanes 32..63 read channel kc+1
    }

    f32x16 acc[NSEG];
#pragma unroll
    for (int j = 0; j < NSEG; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int nchunk = (P.Cin + KC - 1) / KC;
    const int ct = KSPLIT ? 0 : wave;
    const int aoff = (lane >> 5) * BMC + ct * 32 + (lane & 31);
    const bool ragged = (P.Cin % KC) != 0;

    // ================= staging descriptors ==========================================================
    // DMA path: wave-load j covers LDS float4 slots [64j, 64j+64); lane l owns slot 64j + l.
    // Branch-free issue: every lane of every wave-load is always active.  Each lane keeps a 64-bit source pointer per
    // wave-load that is advanced by a per-lane increment each chunk (0 for lanes fed from the zero page); wave-loads
    // that do not exist for this wave (tail) read the zero page into a 1 KiB dummy LDS slot.
    const float *wptr[C::NWJ];
    const float *pptr[C::MAXPJ];
    unsigned pdata = 0;          // bit i: this lane's patch piece i is real data (pointer advances per chunk)
    int wdst[C::NWJ], pdst[C::MAXPJ];      // wave-uniform LDS float offsets inside a buffer (or the dummy slot)
    int winc[C::NWJ];
    // register path (VEC == 1)
    int goff[MAXLP], loff[MAXLP];
    float preg[MAXLP];
    float4 wreg[NLW];
    int woff_g[NLW], woff_l[NLW];
    const int dummy = 2 * bufs;                     // float offset of the dummy LDS slot
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    if constexpr (DMA) {
#pragma unroll
        for (int i = 0; i < C::NWJ; ++i) {
            const int t = wave_s + 4 * i;
            const int q = 64 * t + lane;
            constexpr int V = BMC / 4;
            const int row = q / V, c4 = q - row * V;
            const bool ok = t * 256 < WS;
            wptr[i] = ok ? P.wp + (size_t)row * P.CoutPad + cg * BMC + c4 * 4 : g_zero_page;
            wdst[i] = ok ? t * 256 : -1;
            winc[i] = ok ? TAPS * KC * P.CoutPad : 0;
        }
#pragma unroll
        for (int i = 0; i < C::MAXPJ; ++i) {
            const int t = wave_s + 4 * i;
            const int e = (64 * t + lane) * 4;
            pptr[i] = g_zero_page;
            pdst[i] = (t * 256 < PSpad) ? WS + t * 256 : -1;
            if (e < PS) {
                const int kc = e / chs, rem = e - kc * chs;
                const int pr = rem / PW, col = rem - pr * PW;
                const int yy = y_first - HALO + pr;
                if (pr < nprows && yy >= 0 && yy < P.H && col >= 4 && col < 4 + P.W) {
                    pptr[i] = xb + (size_t)(kc * P.H + yy) * P.W + col - 4;
                    pdata |= 1u << i;
                }
            }
        }
    } else {
        for (int i = tid; i < 2 * bufs; i += 256) smem[i] = 0.f;      // halos / invalid rows stay zero
        const int WV = P.W;
#pragma unroll
        for (int i = 0; i < MAXLP; ++i) {
            const int f = tid + 256 * i;
            goff[i] = -1;
            loff[i] = 0;
            if (f < KC * nprows * WV) {
                const int r = f / WV, cv = f - r * WV;
                const int kc = r / nprows, pr = r - kc * nprows;
                const int yy = y_first - HALO + pr;
                if (yy >= 0 && yy < P.H) {
                    goff[i] = (kc * P.H + yy) * P.W + cv;
                    loff[i] = kc * chs + pr * PW + 4 + cv;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NLW; ++i) {
            const int f = tid + 256 * i;
            constexpr int V = BMC / 4;
            const int row = f / V, c4 = f - row * V;
            woff_g[i] = (f < C::WV4) ? row * P.CoutPad + cg * BMC + c4 * 4 : -1;
            woff_l[i] = row * BMC + c4 * 4;
        }
        __syncthreads();
    }

    // channel clamp for a ragged last chunk: channels >= Cin read channel Cin-1 (their packed weights are zero)
    auto clamp_off = [&](int ch, int off) -> size_t {
        size_t o = (size_t)ch * KC * P.HW + off;
        if (ragged && ch + 1 == nchunk) {
            const int kc = off / P.HW;
            const int over = max(ch * KC + kc - (P.Cin - 1), 0);
            o -= (size_t)over * P.HW;
        }
        return o;
    };
    // one weight wave-load + one patch wave-load; part in [0, NPARTS).  Chunks are issued strictly in order
    // (0, 1, 2, ...), each part exactly once per chunk, so "advance then load" keeps the pointers in step.
    constexpr int NPARTS = (C::NWJ > C::MAXPJ) ? C::NWJ : C::MAXPJ;
    const int pinc = KC * P.HW;
    auto dma_part = [&](int ch, float *buf, int part) {
        if (part < C::NWJ) {
            __builtin_amdgcn_global_load_lds((glb_ptr_t)wptr[part],
                                             (lds_ptr_t)(wdst[part] >= 0 ? buf + wdst[part] : smem + dummy), 16, 0, 0);
            wptr[part] += winc[part];
        }
        if (part < C::MAXPJ) {
            const float *src = pptr[part];
            if (ragged && ch + 1 == nchunk) {          // uniform, last partial chunk only: clamp channels >= Cin
                if ((pdata >> part) & 1u) {
                    const int e = (64 * (wave_s + 4 * part) + lane) * 4;
                    const int over = max(ch * KC + e / chs - (P.Cin - 1), 0);
                    src -= (size_t)over * P.HW;
                }
            }
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(pdst[part] >= 0 ? buf + pdst[part] : smem + dummy),
                                             16, 0, 0);
            pptr[part] += ((pdata >> part) & 1u) ? pinc : 0;
        }
    };
    auto dma = [&](int ch, float *buf) {
#pragma unroll
        for (int part = 0; part < NPARTS; ++part) dma_part(ch, buf, part);
    };
    auto issue = [&](int ch) {
        const float *wsrc_base = P.wp + (size_t)ch * TAPS * KC * P.CoutPad;
#pragma unroll
        for (int i = 0; i < NLW; ++i)
            if (woff_g[i] >= 0) wreg[i] = *(const float4 *)(wsrc_base + woff_g[i]);
#pragma unroll
        for (int i = 0; i < MAXLP; ++i)
            if (goff[i] >= 0) preg[i] = xb[clamp_off(ch, goff[i])];
    };
    auto commit = [&](float *buf) {
#pragma unroll
        for (int i = 0; i < NLW; ++i)
            if (woff_g[i] >= 0) *(float4 *)(buf + woff_l[i]) = wreg[i];
        float *pbuf = buf + WS;
#pragma unroll
        for (int i = 0; i < MAXLP; ++i)
            if (goff[i] >= 0) pbuf[loff[i]] = preg[i];
    };

    if constexpr (DMA) {
        dma(0, smem);
    } else {
        issue(0);
        commit(smem);
    }
    __syncthreads();

    for (int ch = 0; ch < nchunk; ++ch) {
        const float *W_s = smem + (ch & 1) * bufs;
        const float *P_s = W_s + WS;
        const bool more = ch + 1 < nchunk;
        float *nbuf = smem + ((ch + 1) & 1) * bufs;
        if (more) {
            if constexpr (!DMA) issue(ch + 1);
            else if constexpr (TAPS == 1) dma(ch + 1, nbuf);
        }
        // ---- MFMA: TAPS x (KC/2) k-steps x NSEG segments -------------------------------------------
        // Operand registers are double buffered at tap granularity: the LDS reads of tap t+1 (NSTEP weight +
        // NSTEP*NSEG pixel operands) are issued as one batch in front of tap t's MFMAs and have the whole MFMA
        // block (NSTEP*NSEG x 64 cycles) to land; sched_barrier keeps hipcc from sinking them into the block,
        // where it would emit an s_waitcnt lgkmcnt(0) -- a full LDS round trip -- in front of every MFMA.
        constexpr int NSTEP = KSPLIT ? KC / 8 : KC / 2;
        float a_cur[NSTEP], a_nxt[NSTEP], b_cur[NSTEP][NSEG], b_nxt[NSTEP][NSEG];
        auto read_ops = [&](int tap, float (&av)[NSTEP], float (&bv)[NSTEP][NSEG]) {
            const int toff = (TAPS == 9) ? ((tap / 3 - 1) * PW + (tap % 3 - 1)) : 0;
#pragma unroll
            for (int t = 0; t < NSTEP; ++t) {
                const int s = KSPLIT ? wave + 4 * t : t;
                av[t] = W_s[(tap * KC + 2 * s) * BMC + aoff];
                const float *pb = P_s + 2 * s * chs + toff;
#pragma unroll
                for (int j = 0; j < NSEG; ++j) bv[t][j] = pb[laddr[j]];
            }
        };
        read_ops(0, a_cur, b_cur);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            if constexpr (DMA && TAPS == 9) {
                // spread the next chunk's DMA issue over the first taps so its address math hides under the MFMAs
                if (more && tap < kDmaTaps && !(P.dbg & 1)) {
#pragma unroll
                    for (int part = tap; part < NPARTS; part += kDmaTaps) dma_part(ch + 1, nbuf, part);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < TAPS) read_ops(tap + 1, a_nxt, b_nxt);
#pragma unroll
            for (int t = 0; t < NSTEP; ++t)
#pragma unroll
                for (int j = 0; j < NSEG; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t], b_cur[t][j], acc[j], 0, 0, 0);
            // interleave: one MFMA, then (at most) two LDS reads of the next tap, so the matrix pipe never drains
            // while the operand batch is being issued
#pragma unroll
            for (int i = 0; i < NSTEP * NSEG; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // <= 2 DS reads
            }
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < TAPS) {
#pragma unroll
                for (int t = 0; t < NSTEP; ++t) {
                    a_cur[t] = a_nxt[t];
#pragma unroll
                    for (int j = 0; j < NSEG; ++j) b_cur[t][j] = b_nxt[t][j];
                }
            }
        }
        if constexpr (!DMA) {
            if (more) commit(smem + ((ch + 1) & 1) * bufs);
        }
        if (!(P.dbg & 2)) __syncthreads();   // drains the in-flight LDS-DMA (vmcnt) before the buffers swap
    }

    // ---- epilogue: D[row = cout][col = pixel] ----------------------------------------------------------
    float *yb = P.y + (size_t)b * P.Cout * P.HW;
    if constexpr (!KSPLIT) {
        float sc[16], sh[16];
        int co[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            co[r] = cg * BMC + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const bool ok = co[r] < P.Cout;
            sc[r] = (ok && P.scale) ? P.scale[co[r]] : 1.f;
            sh[r] = (ok && P.shift) ? P.shift[co[r]] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NSEG; ++j) {
            const int p = p0 + j * 32 + (lane & 31);
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
    } else {
        // cross-wave K reduction: red[wave][seg][reg][lane] (the loop's final barrier already passed)
        float *red = smem;
#pragma unroll
        for (int j = 0; j < NSEG; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave * NSEG + j) * 16 + r) * 64 + lane] = acc[j][r];
        __syncthreads();
        for (int j = wave; j < NSEG; j += 4) {
            const int p = p0 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cg * BMC + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = red[((0 * NSEG + j) * 16 + r) * 64 + lane] + red[((1 * NSEG + j) * 16 + r) * 64 + lane] +
                          red[((2 * NSEG + j) * 16 + r) * 64 + lane] + red[((3 * NSEG + j) * 16 + r) * 64 + lane];
                if (p < P.HW && co < P.Cout) {
                    v = v * (P.scale ? P.scale[co] : 1.f) + (P.shift ? P.shift[co] : 0.f);
                    if (P.relu) v = fmaxf(v, 0.f);
                    yb[(size_t)co * P.HW + p] = v;
                }
            }
        }
    }
}

template <int CT, int TAPS, int VEC>
int launch_conv(ConvParams P, hipStream_t stream)
{
    using C = ConvCfg<CT, TAPS, VEC>;
    P.ntp = cdiv(P.HW, C::PIX);
    P.ncg = P.CoutPad / C::BMC;
    P.PW = (P.W + 8 + 3) / 4 * 4;
    P.NPR = (C::PIX - 1 + P.W - 1) / P.W + 1 + 2 * C::HALO;
    const size_t ps = ((size_t)C::KC * P.NPR * P.PW + 255) / 256 * 256;
    const size_t stage = 2 * ((size_t)C::WS + ps) * sizeof(float);
    const size_t red = C::KSPLIT ? (size_t)4 * C::NSEG * 16 * 64 * sizeof(float) : 0;
    const size_t lds = (stage > red ? stage : red) + 1024;      // + 1 KiB dummy DMA slot
    if (lds > 160 * 1024) return SASSD_EINVAL;
    if (VEC == 4 ? (ps > (size_t)C::MAXPJ * 4 * 256) : ((size_t)C::KC * P.NPR * P.W > (size_t)256 * C::MAXLP))
        return SASSD_EINVAL;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)conv2d_kernel<CT, TAPS, VEC>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int grid = P.B * P.ntp * P.ncg;
    hipLaunchKernelGGL((conv2d_kernel<CT, TAPS, VEC>), dim3(grid), dim3(256), lds, stream, P);
    return sassd_launch_status();
}

inline int cout_pad(int Cout) { return (Cout > 64) ? cdiv(Cout, 128) * 128 : cdiv(Cout, 32) * 32; }
inline int kc_of(int ksize) { return ksize == 3 ? 8 : 16; }

}  // namespace

extern "C" size_t sassd_conv2d_packed_floats(int Cin, int Cout, int ksize)
{
    const int KC = kc_of(ksize);
    return (size_t)cdiv(Cin, KC) * KC * ksize * ksize * cout_pad(Cout);
}

extern "C" int sassd_conv2d_pack_weight(const float *w, int Cout, int Cin, int ksize, float *packed, void *stream_)
{
    if (!w || !packed || (ksize != 1 && ksize != 3) || Cin < 1) return SASSD_EINVAL;
    const int cp = cout_pad(Cout);
    const size_t total = sassd_conv2d_packed_floats(Cin, Cout, ksize);
    hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, w,
                       Cout, Cin, ksize, cp, kc_of(ksize), packed);
    return sassd_launch_status();
}

// `cfg` (0 in production; per call since round 6, like the sparse-conv and Winograd entry points): ablation switches of
// tools/run_conv.py -- 1 = no staging DMA after the first chunk, 2 = no chunk barrier
extern "C" int sassd_conv2d_fwd_cfg(const float *x, const float *w_packed, const float *scale, const float *shift,
                                    int relu, float *y, int batch, int Cin, int Cout, int H, int W, int ksize, int cfg,
                                    void *stream_)
{
    if (!x || !w_packed || !y || batch < 1 || Cin < 1 || (ksize != 1 && ksize != 3)) return SASSD_EINVAL;
    if (H < 1 || W < 2) return SASSD_EINVAL;
    ConvParams P;
    P.x = x; P.wp = w_packed; P.scale = scale; P.shift = shift; P.y = y;
    P.B = batch; P.Cin = Cin; P.Cout = Cout; P.CoutPad = cout_pad(Cout); P.H = H; P.W = W; P.HW = H * W;
    P.relu = relu;
    P.dbg = cfg;
    hipStream_t stream = (hipStream_t)stream_;
    const bool wide = P.CoutPad % 128 == 0;
    // 16-byte row loads need W % 4 == 0 and a 16-byte aligned base
    const bool vec = (W % 4 == 0) && (((uintptr_t)x & 15) == 0);
#define SASSD_CONV_GO(CT, TAPS) \
    return vec ? launch_conv<CT, TAPS, 4>(P, stream) : launch_conv<CT, TAPS, 1>(P, stream)
    if (wide) {
        if (ksize == 3) { SASSD_CONV_GO(4, 9); } else { SASSD_CONV_GO(4, 1); }
    } else {
        if (ksize == 3) { SASSD_CONV_GO(1, 9); } else { SASSD_CONV_GO(1, 1); }
    }
#undef SASSD_CONV_GO
}

extern "C" int sassd_conv2d_fwd(const float *x, const float *w_packed, const float *scale, const float *shift,
                                int relu, float *y, int batch, int Cin, int Cout, int H, int W, int ksize,
                                void *stream_)
{
    return sassd_conv2d_fwd_cfg(x, w_packed, scale, shift, relu, y, batch, Cin, Cout, H, W, ksize, 0, stream_);
}

// ------------------------------------------------------------------------------------------------------------------
// 1x1 convolution with FEW output channels (<= 32): the fused SSD head (ssd_rotate_head.py:120-125: 14 + 2 + 4 = 20 maps
// from 256 channels) and the second conv of the part-sensitive head (:424-429, 28 -> 28).  y [Cout x HW] = W [Cout x Cin]
// x [Cin x HW] is 0.36 GFLOP over a 36 MB input: an HBM stream (7 us), not a matrix-core problem -- on the generic MFMA
// kernel (32-cout tile, four waves splitting K, 10 MFMAs between barriers) the two launches took 19-24 us each.  Here a
// workgroup owns 64 consecutive pixels; wave w streams its quarter of the input channels (one coalesced 256-byte row
// segment per channel, 8 in flight) and multiplies each value by the channel's Cout weights, which sit in SGPRs (the weight
// row index is wave-uniform: scalar loads, v_fmac with a scalar operand -- no LDS, no operand shuffles); the four partial
// sums meet in LDS in wave order (a fixed summation order).  wT is the weight TRANSPOSED and zero-padded to the kernel's
// channel count, [Cin][CO] fp32 with CO = sassd_conv1x1_narrow_pad(Cout) (whole 16-byte scalar loads per weight row).
// ------------------------------------------------------------------------------------------------------------------
namespace {
template <int CO>
__global__ void __launch_bounds__(256) conv1x1_narrow_kernel(const float *__restrict__ x, const float *__restrict__ wT,
                                                             const float *__restrict__ scale, const float *__restrict__ shift,
                                                             int relu, float *__restrict__ y, int Cin, int Cout, int HW)
{
    __shared__ float part[4][CO][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int p = blockIdx.x * 64 + lane;
    const bool ok = p < HW;
    const int pc = ok ? p : HW - 1;
    const int cq = (Cin + 3) / 4;
    const int c0 = wave * cq, c1 = min(c0 + cq, Cin);
    const float *xp = x + ((size_t)b * Cin + c0) * HW + pc;
    float acc[CO];
#pragma unroll
    for (int j = 0; j < CO; ++j) acc[j] = 0.f;
    int c = c0;
    for (; c + 8 <= c1; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = xp[(size_t)u * HW];
        xp += (size_t)8 * HW;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float *wr = wT + (size_t)(c + u) * CO;            // wave-uniform: scalar loads
#pragma unroll
            for (int j = 0; j < CO; ++j) acc[j] = fmaf(v[u], wr[j], acc[j]);
        }
    }
    for (; c < c1; ++c) {
        const float v = *xp;
        xp += HW;
        const float *wr = wT + (size_t)c * CO;
#pragma unroll
        for (int j = 0; j < CO; ++j) acc[j] = fmaf(v, wr[j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < CO; ++j) part[wave][j][lane] = acc[j];
    __syncthreads();
    for (int j = wave; j < Cout; j += 4) {
        float s = (part[0][j][lane] + part[1][j][lane]) + (part[2][j][lane] + part[3][j][lane]);
        s = s * (scale ? scale[j] : 1.f) + (shift ? shift[j] : 0.f);
        if (relu) s = fmaxf(s, 0.f);
        if (ok) y[((size_t)b * Cout + j) * HW + p] = s;
    }
}
}  // namespace

extern "C" int sassd_conv1x1_narrow_supported(int Cin, int Cout) { return Cin >= 1 && Cout >= 1 && Cout <= 32 ? 1 : 0; }

// padded channel count of the weight image [Cin][CO] (the kernel instantiations: 8, 16, 20, 24, 28, 32)
extern "C" int sassd_conv1x1_narrow_pad(int Cout)
{
    return Cout <= 8 ? 8 : Cout <= 16 ? 16 : Cout <= 20 ? 20 : Cout <= 24 ? 24 : Cout <= 28 ? 28 : 32;
}

extern "C" int sassd_conv1x1_narrow_fwd(const float *x, const float *wT, const float *scale, const float *shift, int relu,
                                        float *y, int batch, int Cin, int Cout, int H, int W, void *stream_)
{
    if (!x || !wT || !y || batch < 1 || H < 1 || W < 1 || !sassd_conv1x1_narrow_supported(Cin, Cout)) return SASSD_EINVAL;
    const int HW = H * W;
    const dim3 grid((unsigned)cdiv(HW, 64), (unsigned)batch);
    hipStream_t s = (hipStream_t)stream_;
#define SASSD_NARROW(CO) hipLaunchKernelGGL((conv1x1_narrow_kernel<CO>), grid, dim3(256), 0, s, x, wT, scale, shift, relu, y, Cin, Cout, HW)
    if (((uintptr_t)wT) & 15) return SASSD_EINVAL;
    switch (sassd_conv1x1_narrow_pad(Cout)) {
    case 8: SASSD_NARROW(8); break;
    case 16: SASSD_NARROW(16); break;
    case 20: SASSD_NARROW(20); break;
    case 24: SASSD_NARROW(24); break;
    case 28: SASSD_NARROW(28); break;
    default: SASSD_NARROW(32); break;
    }
#undef SASSD_NARROW
    return sassd_launch_status();
}
