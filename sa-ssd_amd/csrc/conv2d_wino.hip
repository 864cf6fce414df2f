// conv2d_wino.hip -- 3x3 stride-1 pad-1 convolution of the BEV network through Winograd F(2x2, 3x3) on the fp32 MFMA
// (mmdet/models/necks/cmn.py:240-262: conv0 320->256 and conv1-6 256->256 at 200x176, 93 % of the dense FLOPs).
//
// fp32 MFMA and fp32 VALU have the same peak on MI355X (157 TF), so the only way past the direct kernel's 71 % is to do
// fewer multiplications: F(2x2,3x3) needs 16 instead of 36 per 2x2 output tile and channel pair (2.25x), at the price
// of cheap add-only transforms:   Y = A^T [ (G g G^T) (.) (B^T d B) ] A,   summed over input channels.
//
// One kernel, everything fused (no transformed tensors in HBM or LDS), WAVE-SPECIALISED (12 waves):
//   * work unit (workgroup) = 64 couts x 32 output tiles (2x2 pixels each, linear tile order -> 8800 tiles = 275
//     groups, no padding waste) x all 16 Winograd positions.
//   * waves 8-11 are LOADER waves: the input rows a 32-tile group needs (<= 2 runs of tiles inside one tile row
//     each: 4 image rows x <= 72 columns per run and channel) are staged in LDS by 16-byte global->LDS DMA (zero page
//     for everything outside the image), 36 wave-instructions per 16-channel chunk, double buffered one chunk ahead.
//   * waves 0-7 are MFMA waves, two per SIMD: wave (xi, cb) owns the 4 positions (xi, nu = 0..3) of one 32-cout block
//     = 4 accumulator tiles of v_mfma_f32_32x32x2_f32.  The input transform is fused INTO the B-operand fetch: row xi
//     of B^T d B needs only two of the four patch rows, so per k-step a lane reads 8 raw floats of its tile's patch
//     from LDS, forms t = a +- b (4 FMAs) and the four nu values (4 adds) and feeds 4 MFMAs -- no transformed-input
//     buffer, no scattered LDS writes.  The pre-transformed weights G g G^T are packed offline in MFMA A-operand
//     order and have no reuse inside a workgroup, so they go straight to registers with coalesced 16-byte loads,
//     prefetched half a chunk ahead.
//   * XCD-local blocked work order (inputs cross the fabric once per block, one 1 MB weight slice live per XCD L2);
//     one barrier per chunk; the output transform A^T M A is split: each MFMA wave reduces over nu in registers, the
//     four xi are combined through LDS, then scale/shift/ReLU and float2 stores of the 2x2 pixels.
// Measured history (B=1, 256->256 @200x176; direct kernel 0.39 ms): per-thread patch gathers + transformed-input buffer
// in LDS, unspecialised 8 waves 0.262; two 4-wave workgroups per CU 0.267; loader/MFMA specialisation 0.272; LDS-DMA row
// staging 0.266; XCD blocking 0.260; transform fused into 4 MFMA waves 0.285; into 8 MFMA waves (this file) 0.259; row DMA
// two chunks ahead 0.264; 8 MFMA + 8 loader waves with a transformed-input buffer 0.264; 32-channel chunks (half the
// barriers, this file) 0.259; accumulators pinned to AGPRs 0.251; + k-step software pipeline pinned with sched_barrier
// 0.251; + no SLP / load-store vectorizer (Makefile: 90 fewer VALU instructions per 64 MFMAs -- VALU never co-executes
// with MFMA, SQ_VALU_MFMA_COEXEC_CYCLES = 0) 0.235; a 128-cout workgroup of 8 MFMA+DMA waves (8 MFMAs per transform)
// 0.257 at B=1 (550 workgroups: 3 rounds) and no gain at B=2.  Time is affine in Cin (tools/wino_scaling.py).
// Numerics: F(2,3) in fp32 has a relative error ~1e-6 (transform matrices hold only 0, +-1, +-1/2).
#include "common.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kNT = 32;                  // tiles per workgroup
constexpr int kCoW = 64;                 // couts per workgroup
constexpr int kKC = 32;                  // input channels per chunk (one barrier per chunk: 64 MFMAs per MFMA wave)
constexpr int kNDma = kKC * 2 * 4 * 18 / 256;      // DMA wave-loads per loader thread and chunk (18)
constexpr int kGb = 8;                   // tile groups per XCD-local reuse block
constexpr int kRawW = 72;                // staged columns per run (66 needed + alignment slack), 18 float4
constexpr int kRawBuf = kKC * 2 * 4 * kRawW;       // floats per raw-row buffer: [ci][run][row][col]  (18432 = 72 KB)

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

__device__ __attribute__((aligned(16))) float g_wino_zero[4] = {0.f, 0.f, 0.f, 0.f};

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

__global__ void __launch_bounds__(768) conv2d_wino_kernel(WinoParams P)
{
    extern __shared__ float smem[];                  // 2 x raw-row buffer (144 KB); reused by the output reduction
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
    // Every workgroup of a cout slice streams the SAME 1 MB of weights; each walks the input-channel chunks in a rotated
    // order so that concurrently running workgroups do not ask the same L2 channel for the same lines at the same time
    // (the sum over chunks is order independent up to fp32 rounding and stays deterministic; measured neutral, +-1 %).
    const int rot = (grp * 5 + cg * 3) % nchunk;

    // the group's 32 linear tiles form at most two runs, each inside one tile row (TW >= 32 is required)
    const int g0 = grp * kNT;
    const int tpi = P.TH * P.TW;
    int rb_[2], rty[2], rtx[2], rn[2];               // image, tile row, first tile column, length of each run
    {
        const int b0 = g0 / tpi, r0 = g0 - b0 * tpi;
        rb_[0] = b0; rty[0] = r0 / P.TW; rtx[0] = r0 - rty[0] * P.TW;
        rn[0] = min(min(kNT, P.TW - rtx[0]), P.tiles - g0);
        const int g1 = g0 + rn[0];
        rn[1] = max(min(kNT - rn[0], P.tiles - g1), 0);
        const int b1 = g1 / tpi, r1 = g1 - b1 * tpi;
        rb_[1] = b1; rty[1] = r1 / P.TW; rtx[1] = 0;
    }
    int cola[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) cola[q] = (2 * rtx[q] - 1) & ~3;            // 16-byte aligned first staged column
    f32x16 acc[4];                                   // MFMA waves only: [nu]

    if (wave >= 8) {
        // ================================ loader waves: DMA only ======================================================
        const int lt = tid - 512;
        // float4 i = lt + 256*k of the raw buffer, i = ((ci*2 + run)*4 + row)*18 + f4
        const float *dsrc[kNDma];
        unsigned dvalid = 0;
#pragma unroll
        for (int k = 0; k < kNDma; ++k) {
            const int i = lt + 256 * k;
            const int f4 = i % 18, row = (i / 18) & 3, run = (i / 72) & 1, ci = i / 144;
            const int yy = 2 * rty[run] - 1 + row, xx = cola[run] + 4 * f4;
            const bool ok = rn[run] > 0 && yy >= 0 && yy < P.H && xx >= 0 && xx < P.W;    // W % 4 == 0: all or nothing
            dsrc[k] = ok ? P.x + ((size_t)(rb_[run] * P.Cin + ci) * P.H + yy) * P.W + xx : g_wino_zero;
            if (ok) dvalid |= 1u << k;
        }
        const size_t cstride = (size_t)kKC * HW;
        const int wl = wave - 8;
        auto dma_rows = [&](float *dst, int chunk) { // issues one chunk's rows
#pragma unroll
            for (int k = 0; k < kNDma; ++k) {
                const float *src = dsrc[k] + (((dvalid >> k) & 1u) ? (size_t)chunk * cstride : 0);
                __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(dst + (wl * 64 + 256 * k) * 4), 16, 0, 0);
            }
        };
        // Two row buffers, DMA issued one chunk ahead at the top of the iteration; __syncthreads() (the compiler drains
        // vmcnt before it) makes the chunk visible.  A chunk is 32 channels = 64 MFMAs per MFMA wave (8.2 k cycles per
        // SIMD), longer than the DMA round trip: with 16-channel chunks every variant of this kernel measured the same
        // ~7.5 k cycles per chunk whatever work the chunk contained.
        auto chunk_of = [&](int q) {
            int c = q + rot;
            c -= c >= nchunk ? nchunk : 0;
            return c;
        };
        dma_rows(smem, chunk_of(0));
        __syncthreads();
        for (int q = 0; q < nchunk; ++q) {
            // raw[(q+1)&1] was last read during step q-1 (barrier since then)
            if (q + 1 < nchunk) dma_rows(smem + ((q + 1) & 1) * kRawBuf, chunk_of(q + 1));
            __syncthreads();
        }
    } else {
        // ================================ MFMA waves ==================================================================
        // 8 MFMA waves: wave (xi = w & 3, cb = w >> 2) owns the 4 positions (xi, nu) of ONE 32-cout block.  Two MFMA
        // waves share a SIMD, so while one waits for its LDS reads / forms its B operands the other issues MFMAs.
        const int xi = wave & 3, cbw = wave >> 2;
        const int nhc = P.Cin / 8;
        // weights of (position (xi, nu), half chunk hc): float4 at wsrc[(hc*16 + nu) * 64]
        const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(P.wp) + ((size_t)(cg * 2 + cbw) * nhc * 16 + xi * 4) * 64 + lane;
        f32x4 wq[2][4];                              // [buffer][nu]
        auto fetch_w = [&](int hc, f32x4 *wdst) {
#pragma unroll
#ifdef SASSD_WINO_ABL_W
            hc &= 1;                                 // ablation (wrong results): the weight stream stays L1 resident
#endif
            for (int nu = 0; nu < 4; ++nu) wdst[nu] = wsrc[((size_t)hc * 16 + nu) * 64];
        };
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nu][r] = 0.f;
        // this lane's B-operand source: tile = lane & 31, channel parity kh = lane >> 5; row xi of B^T d B combines the
        // patch rows (ra, rb) as raw[ra] + sg * raw[rb]:  xi 0: r0 - r2, 1: r1 + r2, 2: r2 - r1, 3: r1 - r3
        const int ltile = lane & 31, kh = lane >> 5;
        const int trun = ltile < rn[0] ? 0 : 1;
        const int ttx = trun == 0 ? rtx[0] + ltile : ltile - rn[0];
        const int ra = xi == 0 ? 0 : (xi == 2 ? 2 : 1), rbw = xi == 3 ? 3 : (xi == 2 ? 1 : 2);
        const float sg = xi == 1 ? 1.f : -1.f;
        const int lbase = (kh * 8 + trun * 4) * kRawW + (2 * ttx - 1 - cola[trun]);      // + ci_even*8*kRawW + r*kRawW + c
        const int offa = lbase + ra * kRawW, offb = lbase + rbw * kRawW;
        fetch_w((kKC / 8) * rot, wq[0]);
        __syncthreads();                             // rows of the first chunk landed
        for (int q = 0; q < nchunk; ++q) {
            int c = q + rot;                         // this workgroup's chunk order starts at `rot`
            c -= c >= nchunk ? nchunk : 0;
            int c1 = c + 1;
            c1 -= c1 >= nchunk ? nchunk : 0;
            const float *rw = smem + (q & 1) * kRawBuf;
            float pa[2][4], pb[2][4];                // raw rows (ra, rb) of the patch, double buffered over k-steps
            auto fetch_raw = [&](int step, float *a, float *b) {    // step = h*4 + s -> channels 2*step + kh
                const float *src = rw + step * (2 * 8 * kRawW);
#pragma unroll
                for (int q = 0; q < 4; ++q) { a[q] = src[offa + q]; b[q] = src[offb + q]; }
            };
            // Software pipeline over the k-steps, pinned with sched_barrier (left alone the compiler sinks every LDS read
            // next to its use: 4 MFMAs per exposed LDS round trip, s_nop hazards between the transform and the MFMA):
            //   step k:  issue the LDS reads of step k+2  |  transform step k+1 (reads issued a step ago)  |  MFMAs of k
            float bv[2][4];
            auto transform = [&](const float *a, const float *b, float *o) {
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = fmaf(sg, b[e], a[e]);
                o[0] = t[0] - t[2]; o[1] = t[1] + t[2]; o[2] = t[2] - t[1]; o[3] = t[1] - t[3];
            };
            fetch_raw(0, pa[0], pb[0]);
            fetch_raw(1, pa[1], pb[1]);
            transform(pa[0], pb[0], bv[0]);
#pragma unroll
            for (int step = 0; step < kKC / 2; ++step) {
                const int hh = step >> 2, h = hh & 1, s = step & 3, cur = step & 1;      // hh: 8-channel slice of the chunk
                if (s == 0)                          // next 8-channel slice's weights (first slice of the next chunk at the end)
                    fetch_w(hh + 1 < kKC / 8 ? (kKC / 8) * c + hh + 1 : (kKC / 8) * c1, wq[(h + 1) & 1]);
                if (step + 2 < kKC / 2) fetch_raw(step + 2, pa[cur], pb[cur]);           // raw set `cur` was consumed last step
                __builtin_amdgcn_sched_barrier(0);
                if (step + 1 < kKC / 2) transform(pa[cur ^ 1], pb[cur ^ 1], bv[cur ^ 1]);
#pragma unroll
                for (int nu = 0; nu < 4; ++nu)
                    acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[h][nu][s], bv[cur][nu], acc[nu], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
    }

    // ---- output transform: over nu in registers (MFMA waves), over xi through LDS (all waves) --------------------------
    // P_j[xi] = sum_nu M[xi][nu] A[nu][j],  A^T = [[1,1,1,0],[0,1,-1,-1]]
    float *red = smem;                               // [cb 2][xi 4][j 2][reg 16][lane 64]  = 64 KB (raw rows are dead)
    if (wave < 8) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p0 = acc[0][r] + acc[1][r] + acc[2][r];
            const float p1 = acc[1][r] - acc[2][r] - acc[3][r];
            red[((wave * 2 + 0) * 16 + r) * 64 + lane] = p0;          // wave = cb * 4 + xi
            red[((wave * 2 + 1) * 16 + r) * 64 + lane] = p1;
        }
    }
    __syncthreads();
    // wave w finishes registers 4*(w & 3) .. +3 of cout block (w >> 2): Y[i][j] = sum_xi A^T[i][xi] P_j[xi]
    const int otile = grp * kNT + (lane & 31);
    if (otile < P.tiles && wave < 8) {
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
                for (int jj = 0; jj < 2; ++jj) pj[x4][jj] = red[(((cb * 4 + x4) * 2 + jj) * 16 + r) * 64 + lane];
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
}  // namespace


extern "C" int sassd_conv2d_wino_supported(int Cin, int Cout, int H, int W)
{
    // W % 4: zero padding is decided per 16-byte DMA; W >= 64: a 32-tile group spans at most two tile rows
    return (Cin >= kKC && Cin % kKC == 0 && Cout >= 32 && Cout % 32 == 0 && H >= 2 && H % 2 == 0 && W >= 64 && W % 4 == 0)
               ? 1 : 0;
}

extern "C" size_t sassd_conv2d_wino_packed_floats(int Cin, int Cout)
{
    if (Cin < kKC || Cin % kKC || Cout < 1) return 0;
    const int ncb32 = cdiv(cdiv(Cout, 64) * 64, 32);
    return (size_t)ncb32 * (Cin / 8) * 16 * 256;
}

extern "C" int sassd_conv2d_wino_pack_weight(const float *w, int Cout, int Cin, float *packed, void *stream_)
{
    if (!w || !packed || Cin < kKC || Cin % kKC || Cout < 1) return SASSD_EINVAL;
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
    P.dbg = 0;
    const size_t lds = (size_t)(2 * kRawBuf) * sizeof(float);                       // 147 456 B
    {
        static std::atomic<unsigned long long> attr_done{0};
        const int rc = sassd_dyn_lds((const void *)conv2d_wino_kernel, lds, attr_done);      // per device, thread safe
        if (rc) return rc;
    }
    P.ngrp = cdiv(P.tiles, kNT);
    const int per_xcd = cdiv(cdiv(P.ngrp, 8), kGb) * kGb;       // groups per XCD, padded to whole blocks
    const int grid = per_xcd * P.ncb64 * 8;
    hipLaunchKernelGGL(conv2d_wino_kernel, dim3(grid), dim3(768), lds, (hipStream_t)stream_, P);
    return sassd_launch_status();
}
