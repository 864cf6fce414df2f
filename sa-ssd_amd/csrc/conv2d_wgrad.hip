// conv2d_wgrad.hip -- weight gradient of the dense BEV / head convolutions (training, SURVEY 8 a9/a10/a12 backward):
//   dW[co][ci][ky][kx] = sum_{b,y,x} dY[b][co][y][x] * X[b][ci][y+ky-P][x+kx-P]        (NCHW fp32, stride 1, P = k/2)
// i.e. the autograd of torch.nn.Conv2d at mmdet/models/necks/cmn.py:240-262 and
// mmdet/models/single_stage_heads/ssd_rotate_head.py:120-125,424-429, which the reference gets from cuDNN.
//
// A GEMM whose contraction runs over PIXELS: M = Cout, N = Cin*taps, K = B*H*W (70 400 at B = 2).  MFMA-bound
// (83 GFLOP per 256->256 3x3 layer), so it is tiled for v_mfma_f32_32x32x2_f32:
//   * WG = 4 waves = 64 couts x 64 cins x all taps; a wave owns 32 couts x 32 cins x 9 taps = 9 accumulator tiles
//     (144 AGPRs; 18 tiles would overflow the 256 AGPRs and the compiler then shuffles accumulators through VGPRs
//     on every MFMA) -- every dY value read from LDS feeds 9 MFMAs.  <= 256 VGPRs and 71 KB of LDS per WG keep two
//     WGs resident per CU, so one WG's barriers / LDS refills hide behind the other's MFMAs.
//   * split-K over pixels: a WG walks a strip of 44 columns x R rows of one image; X rows live in a 4-slot LDS ring
//     (each row is loaded once and used by three dY rows), dY rows are double-buffered; the next row is prefetched
//     into registers while the current one is multiplied.
//   * LDS rows are [channel][pixel] with an odd pitch, so the per-lane "channel = lane & 31" operand reads are
//     bank-conflict free without transposing the NCHW data.
//   * partial sums go to a [split][tap][Cout][Cin] workspace (coalesced 128-B rows) and a second kernel reduces the
//     splits in a fixed order -> deterministic, no atomics.
#include "common.h"

#include <algorithm>

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kSeg = 44;                 // strip width in pixels (176 = 4 * 44)
constexpr int kPitchY = kSeg + 1;        // 45 (odd)
constexpr int kPitchX = kSeg + 3;        // 47 (odd): columns x0-1 .. x0+44
constexpr int kCoT = 64, kCiT = 64;
constexpr int kYRowFloats = kCoT * kPitchY;   // 2880
constexpr int kXRowFloats = kCiT * kPitchX;   // 3008
constexpr int kYPer = (kCoT * kSeg + 255) / 256;        // 11 dY values per thread per row
constexpr int kXPer = (kCiT * (kSeg + 2) + 255) / 256;  // 12 X values per thread per row

struct WgradParams {
    const float *x, *dy;
    float *part;
    int B, Cin, Cout, H, W;
    int nseg, nrange, rows_per;          // strips per image row, row ranges per image, rows per range
    int n_ci_t, n_co_t;
    const float *x_aff;                  // bf16 kernel, BN = 1: [3][Cin] mean | invstd * gamma | beta -- X is relu(batchnorm(x))
};

template <int TAPS>
__global__ void __launch_bounds__(256, 2) conv2d_wgrad_kernel(WgradParams p)
{
    extern __shared__ float smem[];
    float *ylds = smem;                              // [2][64][45]
    float *xlds = smem + 2 * kYRowFloats;            // [4][64][47]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;

    int wg = blockIdx.x;
    const int ci_t = wg % p.n_ci_t; wg /= p.n_ci_t;
    const int co_t = wg % p.n_co_t; wg /= p.n_co_t;
    const int split = wg;                            // (b, seg, range)
    const int range = wg % p.nrange; wg /= p.nrange;
    const int seg = wg % p.nseg;
    const int b = wg / p.nseg;
    const int x0 = seg * kSeg;
    const int r0 = range * p.rows_per, r1 = min(r0 + p.rows_per, p.H);
    const int co0 = co_t * kCoT, ci0 = ci_t * kCiT;
    const size_t hw = (size_t)p.H * p.W;
    const float *xb = p.x + (size_t)b * p.Cin * hw;
    const float *yb = p.dy + (size_t)b * p.Cout * hw;

    float yreg[kYPer], xreg[kXPer];

    auto fetch_y = [&](int row) {
#pragma unroll
        for (int i = 0; i < kYPer; ++i) {
            const int e = tid + 256 * i;
            const int c = e / kSeg, px = e - c * kSeg;
            const int co = co0 + c, x = x0 + px;
            const bool ok = e < kCoT * kSeg && co < p.Cout && x < p.W && row < p.H;
            const float v = yb[ok ? (size_t)co * hw + (size_t)row * p.W + x : 0];     // unconditional load + select
            yreg[i] = ok ? v : 0.f;
        }
    };
    auto store_y = [&](int buf) {
#pragma unroll
        for (int i = 0; i < kYPer; ++i) {
            const int e = tid + 256 * i;
            const int c = e / kSeg, px = e - c * kSeg;
            if (e < kCoT * kSeg) ylds[buf * kYRowFloats + c * kPitchY + px] = yreg[i];
        }
    };
    auto fetch_x = [&](int row) {
#pragma unroll
        for (int i = 0; i < kXPer; ++i) {
            const int e = tid + 256 * i;
            const int c = e / (kSeg + 2), px = e - c * (kSeg + 2);
            const int ci = ci0 + c, x = x0 - 1 + px;
            const bool ok = e < kCiT * (kSeg + 2) && ci < p.Cin && row >= 0 && row < p.H && x >= 0 && x < p.W;
            const float v = xb[ok ? (size_t)ci * hw + (size_t)row * p.W + x : 0];
            xreg[i] = ok ? v : 0.f;
        }
    };
    auto store_x = [&](int row) {
        const int slot = row & 3;
#pragma unroll
        for (int i = 0; i < kXPer; ++i) {
            const int e = tid + 256 * i;
            const int c = e / (kSeg + 2), px = e - c * (kSeg + 2);
            if (e < kCiT * (kSeg + 2)) xlds[slot * kXRowFloats + c * kPitchX + px] = xreg[i];
        }
    };

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (r0 < r1) {
        // prologue: X rows r0-1, r0 (and r0+1 is fetched as the first "next" row), dY row r0
        fetch_x(r0 - 1); store_x(r0 - 1 + 4);        // (+4 keeps the slot index non-negative: (r0-1+4) & 3 == (r0-1) & 3)
        fetch_x(r0);     store_x(r0);
        fetch_x(r0 + 1); store_x(r0 + 1);
        fetch_y(r0);     store_y(0);
        __syncthreads();
        for (int y = r0; y < r1; ++y) {
            const int cur = (y - r0) & 1;
            const bool more = (y + 1 < r1);
            if (more) { fetch_y(y + 1); fetch_x(y + 2); }            // global loads in flight during the MFMAs
            const float *yl = ylds + cur * kYRowFloats + (wm * 32 + (lane & 31)) * kPitchY + (lane >> 5);
            const float *xl = xlds + (wn * 32 + (lane & 31)) * kPitchX + (lane >> 5);
            const int s0 = ((y - 1 + 4) & 3) * kXRowFloats, s1 = (y & 3) * kXRowFloats,
                      s2 = ((y + 1) & 3) * kXRowFloats;
            // operands of pixel pair pp+1 are read from LDS while the 9 MFMAs of pair pp issue (double-buffered
            // registers, ds_reads interleaved 1 : 1 with the MFMAs by sched_group_barrier)
            float av[2], bv[2][TAPS];
            auto lds_ops = [&](int pp, float &a, float *bq) {
                a = yl[2 * pp];
                if (TAPS == 9) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int so = t < 3 ? s0 : (t < 6 ? s1 : s2);
                        bq[t % TAPS] = xl[so + 2 * pp + (t % 3)];
                    }
                } else {
                    bq[0] = xl[s1 + 2 * pp + 1];
                }
            };
            lds_ops(0, av[0], bv[0]);
#pragma unroll
            for (int pp = 0; pp < kSeg / 2; ++pp) {
                const int c = pp & 1;
                if (pp + 1 < kSeg / 2) lds_ops(pp + 1, av[c ^ 1], bv[c ^ 1]);
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bv[c][t], acc[t], 0, 0, 0);
                if (pp + 1 < kSeg / 2) {
                    if (TAPS == 9) {                         // 10 ds_reads spread over the 9 MFMAs
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                }
            }
            __syncthreads();                         // everyone is done with row y's dY buffer and X slot (y-1)
            if (more) { store_y(cur ^ 1); store_x(y + 2); }
            __syncthreads();
        }
    }

    // epilogue: part[split][tap][co][ci]; D element (row = co, col = ci): col = lane & 31,
    // row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float *pt = p.part + (size_t)split * TAPS * p.Cout * p.Cin;
    const int ci = ci0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (co < p.Cout && ci < p.Cin) pt[((size_t)t * p.Cout + co) * p.Cin + ci] = acc[t][r];
        }
}

// dw[co][ci][tap] = sum_s part[s][tap][co][ci]  (+ dw when accumulate)
__global__ void conv2d_wgrad_reduce_kernel(const float *__restrict__ part, int nsplit, int taps, int Cout, int Cin,
                                           float *__restrict__ dw, int accumulate)
{
    const size_t n = (size_t)taps * Cout * Cin;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;            // four loads in flight, fixed summation order
    int k = 0;
    for (; k + 4 <= nsplit; k += 4) {
        s0 += part[(size_t)k * n + i];
        s1 += part[(size_t)(k + 1) * n + i];
        s2 += part[(size_t)(k + 2) * n + i];
        s3 += part[(size_t)(k + 3) * n + i];
    }
    for (; k < nsplit; ++k) s0 += part[(size_t)k * n + i];
    const float s = (s0 + s1) + (s2 + s3);
    const int ci = i % Cin;
    const int co = (i / Cin) % Cout;
    const int t = i / ((size_t)Cin * Cout);
    float *o = dw + ((size_t)co * Cin + ci) * taps + t;
    *o = accumulate ? *o + s : s;
}


// ---------------------------------------------------------------------------------------------------------------------
// bf16 variant (BASELINE configs[2]: training in bf16).  Same GEMM, same strip walk and split-K workspace, but the
// operands are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) on their way into LDS and multiplied by
// v_mfma_f32_32x32x16_bf16 (fp32 accumulation): 16 pixels per MFMA instead of 2, on the 2.5 PFLOP/s pipe.
//   * LDS rows are [channel][pixel] bf16 with a 56-element (112 B = 7 x 16 B) pitch: a lane's operand is ONE aligned
//     ds_read_b128 of 8 consecutive pixels of its channel, and the 16 lanes of every ds_read_b128 phase land on 16
//     different 16-byte bank groups.
//   * X column c of a row holds x = x0 - 2 + c, so that the (even) global float2 loads stay 8-byte aligned; the three
//     horizontal taps start 1, 2, 3 elements into a lane's 12-element window (b128 + b64 read): the middle tap is a
//     register rename, the outer two are four v_alignbit_b32 each.
//   * the strip is 44 pixels wide = 3 MFMA k-steps of 16 with 4 zero columns (8 % padding); pad columns are zeroed
//     once (garbage x 0 could be NaN) and never written again.
//   * XCD-aware grid: the 16 (cout tile, cin tile) workgroups that read the same pixel strip run on ONE XCD, so the
//     strip reaches that XCD's L2 once and is re-read from there.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kPitchH = 56;                                   // bf16 elements per LDS row
constexpr int kRowH = kCoT * kPitchH;                          // 3584 elements = 7168 B per (row, 64 channels)
constexpr int kYPairs = kSeg / 2;                              // 22 float2 per dY channel row
constexpr int kXPairs = (kSeg + 4) / 2;                        // 24 float2 per X channel row: x0-2 .. x0+45
constexpr int kYPerH = (kYPairs + 3) / 4;                      // 6 pairs per thread (4 threads per channel row)
constexpr int kXPerH = kXPairs / 4;                            // 6 exactly
static_assert(kCoT * 4 == 256 && kCiT * 4 == 256 && kXPairs % 4 == 0, "4 threads per channel row");

__device__ __forceinline__ unsigned pack_bf16(f32x2 v)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// BN = 1 (round 6): x is the RAW output of the previous convolution; the X operand is relu(batchnorm(x)) with the batch
// statistics of sassd_bn2d_stats, applied when a row is rounded and stored -- the expression of bn2d_apply_kernel, bit for bit
// (the forward convolution's loader waves do the same: the normalised map does not exist in HBM).
template <int TAPS, int BN = 0>
__global__ void __launch_bounds__(256, 2) conv2d_wgrad_bf16_kernel(WgradParams p)
{
    extern __shared__ float smem[];
    unsigned short *ylds = (unsigned short *)smem;             // [2][64][56]
    unsigned short *xlds = ylds + 2 * kRowH;                   // [4][64][56]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;

    // blockIdx -> (strip, channel-tile pair): XCD = blockIdx % 8 keeps all pairs of a strip on one XCD
    const int npair = p.n_ci_t * p.n_co_t;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int split = (local / npair) * 8 + xcd;
    const int pair = local % npair;
    const int nsplit = p.B * p.nseg * p.nrange;
    if (split >= nsplit) return;
    const int ci_t = pair % p.n_ci_t, co_t = pair / p.n_ci_t;
    int wg = split;
    const int range = wg % p.nrange; wg /= p.nrange;
    const int seg = wg % p.nseg;
    const int b = wg / p.nseg;
    const int x0 = seg * kSeg;
    const int r0 = range * p.rows_per, r1 = min(r0 + p.rows_per, p.H);
    const int co0 = co_t * kCoT, ci0 = ci_t * kCiT;
    const size_t hw = (size_t)p.H * p.W;
    const float *xb = p.x + (size_t)b * p.Cin * hw;
    const float *yb = p.dy + (size_t)b * p.Cout * hw;

    for (int i = tid; i < 6 * kRowH / 2; i += 256) ((unsigned *)ylds)[i] = 0u;     // pad columns stay zero

    // (Round 6 measured TWO register sets -- the rows a step stores requested two steps earlier, every fetch unconditional, exact
    // vmcnt(12) waits in the ISA, 244 VGPRs: 0.1262 against 0.1267-0.1284 ms for the 3x3 layer and 0.089 against 0.072 ms for
    // the 1x1 one.  The 2.3 us a row step takes are NOT load latency: 143 VALU / LDS instructions per 27 MFMAs and two
    // barriers per row issue from the same four waves.  Not adopted; profiles/r06_wgrad_prefetch.txt.)
    // the raw fp32 pairs stay in registers across the MFMAs (loads in flight) and are rounded when they are stored;
    // loads are unconditional (clamped address) with a select on the raw value -- a select on the CONVERTED value
    // makes the compiler branch around load + wait + convert, which serialises the twelve loads
    // thread -> (channel c = tid / 4, pair lane q = tid % 4); its i-th pair of a row is pixel pair q + 4 i, so both the
    // global address and the LDS address advance by a constant per i (no per-element index arithmetic in the loop)
    const int fc = tid >> 2, fq = tid & 3;
    const bool y_cok = co0 + fc < p.Cout, x_cok = ci0 + fc < p.Cin;
    const float *ych = yb + (size_t)min(co0 + fc, p.Cout - 1) * hw;
    const float *xch = xb + (size_t)min(ci0 + fc, p.Cin - 1) * hw;
    unsigned short *yst = ylds + fc * kPitchH + 2 * fq;
    unsigned short *xst = xlds + fc * kPitchH + 2 * fq;
    float bn_m = 0.f, bn_s = 1.f, bn_b = 0.f;
    if constexpr (BN) {
        const int c = min(ci0 + fc, p.Cin - 1);
        bn_m = p.x_aff[c]; bn_s = p.x_aff[p.Cin + c]; bn_b = p.x_aff[2 * p.Cin + c];
    }
    f32x2 yreg[kYPerH], xreg[kXPerH];
    auto y_ok = [&](int i, int row) {
        const int px = 2 * (fq + 4 * i);
        return y_cok && px < kSeg && x0 + px < p.W && row < p.H;                        // W is even: x+1 < W too
    };
    auto x_ok = [&](int i, int row) {
        const int x = x0 - 2 + 2 * (fq + 4 * i);                                        // even: x, x+1 valid together
        return x_cok && row >= 0 && row < p.H && x >= 0 && x < p.W;
    };
    auto fetch_y = [&](int row) {
        const float *r = ych + (size_t)min(row, p.H - 1) * p.W + x0 + 2 * fq;
#pragma unroll
        for (int i = 0; i < kYPerH; ++i) yreg[i] = *(const f32x2 *)(y_ok(i, row) ? r + 8 * i : ych);
    };
    auto store_y = [&](int buf, int row) {
#pragma unroll
        for (int i = 0; i < kYPerH; ++i) {
            const bool ok = y_ok(i, row);
            const f32x2 v = {ok ? yreg[i][0] : 0.f, ok ? yreg[i][1] : 0.f};
            if (2 * (fq + 4 * i) < kSeg) *(unsigned *)(yst + buf * kRowH + 8 * i) = pack_bf16(v);
        }
    };
    auto fetch_x = [&](int row) {
        const float *r = xch + (size_t)min(max(row, 0), p.H - 1) * p.W + x0 - 2 + 2 * fq;
#pragma unroll
        for (int i = 0; i < kXPerH; ++i) xreg[i] = *(const f32x2 *)(x_ok(i, row) ? r + 8 * i : xch);
    };
    auto store_x = [&](int row) {
        const int slot = (row + 4) & 3;
#pragma unroll
        for (int i = 0; i < kXPerH; ++i) {
            const bool ok = x_ok(i, row);
            float a0 = xreg[i][0], a1 = xreg[i][1];
            if constexpr (BN) {
                const float z0 = fmaf(a0 - bn_m, bn_s, bn_b), z1 = fmaf(a1 - bn_m, bn_s, bn_b);       // == bn2d_apply_kernel
                a0 = z0 > 0.f ? z0 : 0.f; a1 = z1 > 0.f ? z1 : 0.f;
            }
            const f32x2 v = {ok ? a0 : 0.f, ok ? a1 : 0.f};                                           // zero padding AFTER the ReLU
            *(unsigned *)(xst + slot * kRowH + 8 * i) = pack_bf16(v);
        }
    };

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (r0 < r1) {
        __syncthreads();                                     // zero fill done before the first row stores
        fetch_x(r0 - 1); store_x(r0 - 1);
        fetch_x(r0);     store_x(r0);
        fetch_x(r0 + 1); store_x(r0 + 1);
        fetch_y(r0);     store_y(0, r0);
        __syncthreads();
        for (int y = r0; y < r1; ++y) {
            const int cur = (y - r0) & 1;
            const bool more = (y + 1 < r1);
            if (more) { fetch_y(y + 1); fetch_x(y + 2); }
            const unsigned short *yl = ylds + cur * kRowH + (wm * 32 + (lane & 31)) * kPitchH + 8 * (lane >> 5);
            const unsigned short *xl = xlds + (wn * 32 + (lane & 31)) * kPitchH + 8 * (lane >> 5);
            const int so[3] = {((y - 1 + 4) & 3) * kRowH, (y & 3) * kRowH, ((y + 1) & 3) * kRowH};
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, *(const u32x4 *)(yl + 16 * s));
                if (TAPS == 9) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const u32x4 w = *(const u32x4 *)(xl + so[ky] + 16 * s);
                        const uint2 w2 = *(const uint2 *)(xl + so[ky] + 16 * s + 8);
                        const u32x4 t0 = {__builtin_amdgcn_alignbit(w[1], w[0], 16), __builtin_amdgcn_alignbit(w[2], w[1], 16),
                                          __builtin_amdgcn_alignbit(w[3], w[2], 16), __builtin_amdgcn_alignbit(w2.x, w[3], 16)};
                        const u32x4 t1 = {w[1], w[2], w[3], w2.x};
                        const u32x4 t2 = {__builtin_amdgcn_alignbit(w[2], w[1], 16), __builtin_amdgcn_alignbit(w[3], w[2], 16),
                                          __builtin_amdgcn_alignbit(w2.x, w[3], 16), __builtin_amdgcn_alignbit(w2.y, w2.x, 16)};
                        acc[3 * ky + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, t0), acc[3 * ky + 0], 0, 0, 0);
                        acc[(3 * ky + 1) % TAPS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, t1), acc[(3 * ky + 1) % TAPS], 0, 0, 0);
                        acc[(3 * ky + 2) % TAPS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, t2), acc[(3 * ky + 2) % TAPS], 0, 0, 0);
                    }
                } else {
                    const u32x4 w = *(const u32x4 *)(xl + so[1] + 16 * s);
                    const unsigned w4 = *(const unsigned *)(xl + so[1] + 16 * s + 8);
                    const u32x4 t1 = {w[1], w[2], w[3], w4};
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, t1), acc[0], 0, 0, 0);
                }
            }
            __syncthreads();
            if (more) { store_y(cur ^ 1, y + 1); store_x(y + 2); }
            __syncthreads();
        }
    }

    float *pt = p.part + (size_t)split * TAPS * p.Cout * p.Cin;
    const int ci = ci0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (co < p.Cout && ci < p.Cin) pt[((size_t)t * p.Cout + co) * p.Cin + ci] = acc[t][r];
        }
}

struct WgradPlan { int nseg, nrange, rows_per, n_ci_t, n_co_t, nsplit; };
WgradPlan wgrad_plan(int B, int Cin, int Cout, int H, int W)
{
    WgradPlan q;
    q.nseg = cdiv(W, kSeg);
    q.n_ci_t = cdiv(Cin, kCiT);
    q.n_co_t = cdiv(Cout, kCoT);
    const int base = q.n_ci_t * q.n_co_t * B * q.nseg;
    // row ranges per image: minimise (rounds of 512 resident workgroups, 2 per CU) x (rows per workgroup + ~2 rows
    // of prologue / epilogue)
    long best = -1;
    q.rows_per = H; q.nrange = 1;
    for (int nr = 1; nr <= std::min(H, 64); ++nr) {
        const int rp = cdiv(H, nr), nrr = cdiv(H, rp);
        const long cost = (long)cdiv(base * nrr, 512) * (rp + 2);
        if (best < 0 || cost < best) { best = cost; q.rows_per = rp; q.nrange = nrr; }
    }
    q.nsplit = B * q.nseg * q.nrange;
    return q;
}
}  // namespace

extern "C" size_t sassd_conv2d_wgrad_workspace_bytes(int batch, int Cin, int Cout, int H, int W, int ksize)
{
    if (batch < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1 || (ksize != 1 && ksize != 3)) return 0;
    const WgradPlan q = wgrad_plan(batch, Cin, Cout, H, W);
    return (size_t)q.nsplit * ksize * ksize * Cout * Cin * sizeof(float);
}

namespace {
int wgrad_launch(const float *x, const float *dy, float *dw, int batch, int Cin, int Cout, int H, int W, int ksize,
                 int accumulate, void *workspace, size_t workspace_bytes, void *stream_, bool bf16,
                 const float *x_aff = nullptr)
{
    if (!x || !dy || !dw || !workspace || batch < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1 ||
        (ksize != 1 && ksize != 3) || (bf16 && (W & 1)))
        return SASSD_EINVAL;
    if (workspace_bytes < sassd_conv2d_wgrad_workspace_bytes(batch, Cin, Cout, H, W, ksize)) return SASSD_ENOSPC;
    const WgradPlan q = wgrad_plan(batch, Cin, Cout, H, W);
    hipStream_t s = (hipStream_t)stream_;
    WgradParams p;
    p.x = x; p.dy = dy; p.part = (float *)workspace;
    p.B = batch; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W;
    p.nseg = q.nseg; p.nrange = q.nrange; p.rows_per = q.rows_per; p.n_ci_t = q.n_ci_t; p.n_co_t = q.n_co_t;
    p.x_aff = x_aff;
    if (x_aff && !(bf16 && ksize == 3)) return SASSD_EINVAL;
    if (bf16) {
        const size_t lds = (size_t)6 * kRowH * sizeof(unsigned short);                      // 43 008 B
        const int grid = cdiv(q.nsplit, 8) * 8 * q.n_ci_t * q.n_co_t;
        if (ksize == 3 && x_aff)
            hipLaunchKernelGGL((conv2d_wgrad_bf16_kernel<9, 1>), dim3(grid), dim3(256), lds, s, p);
        else if (ksize == 3)
            hipLaunchKernelGGL(conv2d_wgrad_bf16_kernel<9>, dim3(grid), dim3(256), lds, s, p);
        else
            hipLaunchKernelGGL(conv2d_wgrad_bf16_kernel<1>, dim3(grid), dim3(256), lds, s, p);
    } else {
        const size_t lds = (size_t)(2 * kYRowFloats + 4 * kXRowFloats) * sizeof(float);     // 71 168 B
        const int grid = q.nsplit * q.n_ci_t * q.n_co_t;
        static std::atomic<unsigned long long> done9{0}, done1{0};
        int rc;
        if ((rc = sassd_dyn_lds((const void *)conv2d_wgrad_kernel<9>, lds, done9))) return rc;
        if ((rc = sassd_dyn_lds((const void *)conv2d_wgrad_kernel<1>, lds, done1))) return rc;
        if (ksize == 3)
            hipLaunchKernelGGL(conv2d_wgrad_kernel<9>, dim3(grid), dim3(256), lds, s, p);
        else
            hipLaunchKernelGGL(conv2d_wgrad_kernel<1>, dim3(grid), dim3(256), lds, s, p);
    }
    const int taps = ksize * ksize;
    const size_t n = (size_t)taps * Cout * Cin;
    hipLaunchKernelGGL(conv2d_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       (const float *)workspace, q.nsplit, taps, Cout, Cin, dw, accumulate);
    return sassd_launch_status();
}
}  // namespace

extern "C" int sassd_conv2d_bwd_weight(const float *x, const float *dy, float *dw, int batch, int Cin, int Cout, int H,
                                       int W, int ksize, int accumulate, void *workspace, size_t workspace_bytes,
                                       void *stream_)
{
    return wgrad_launch(x, dy, dw, batch, Cin, Cout, H, W, ksize, accumulate, workspace, workspace_bytes, stream_, false);
}

// Same contract with the operands rounded to bf16 (fp32 accumulation, fp32 dw); W must be even.
extern "C" int sassd_conv2d_bwd_weight_bf16(const float *x, const float *dy, float *dw, int batch, int Cin, int Cout,
                                            int H, int W, int ksize, int accumulate, void *workspace,
                                            size_t workspace_bytes, void *stream_)
{
    return wgrad_launch(x, dy, dw, batch, Cin, Cout, H, W, ksize, accumulate, workspace, workspace_bytes, stream_, true);
}

// ... with X = relu(batchnorm(x)) applied on the way into LDS: x_affine = [3][Cin] (mean | invstd * gamma | beta) as
// sassd_bn2d_stats writes it; 3x3 only.  Bit-identical to the weight gradient over the materialised normalised map.
extern "C" int sassd_conv2d_bwd_weight_bf16_bnrelu(const float *x, const float *x_affine, const float *dy, float *dw, int batch,
                                                   int Cin, int Cout, int H, int W, int ksize, int accumulate,
                                                   void *workspace, size_t workspace_bytes, void *stream_)
{
    if (!x_affine) return SASSD_EINVAL;
    return wgrad_launch(x, dy, dw, batch, Cin, Cout, H, W, ksize, accumulate, workspace, workspace_bytes, stream_, true, x_affine);
}
