// conv2d_bf16.hip -- 3x3 pad-1 convolution of the BEV trunk on the bf16 MFMA pipe (BASELINE configs[2]: "training,
// batch=2/GPU ... bf16").  Forward of mmdet/models/necks/cmn.py:240-262 (the seven 3x3 convs of BEVNet) and, with the
// weights transposed and the taps mirrored, their data gradient (cuDNN under autocast in the reference).
//   y[b][co][y][x] = shift[co] + sum_{ci,ky,kx} bf16(w[co][ci][ky][kx]) * bf16(x[b][ci][y+ky-1][x+kx-1])
// NCHW fp32 activations in and out (BatchNorm / ReLU between the layers stay fp32), operands rounded to bf16
// (round-to-nearest-even) on their way into LDS / at pack time, fp32 accumulation on v_mfma_f32_32x32x16_bf16.
//
// MFMA-bound (83 GFLOP per 256->256 layer at B = 2 against 2.5 PFLOP/s), tiled as an implicit GEMM
// D[co][pixel] = sum_k W[co][k] X[k][pixel], k = (tap, ci):
//   * workgroup = 10 rows x 16 columns of one image (160 pixels) x 256 (or 128) couts: 8 (4) MMA waves of 32 couts x
//     160 pixels (5 accumulator tiles) + 4 loader waves.  The MFMA's 32 pixel columns are 2 image rows x 16 columns:
//     the lane -> pixel map is free because every lane computes its own LDS address, and 176 = 11 x 16 tiles the
//     KITTI map with no padding.
//   * the input tile (12 rows x 24 columns x 32 channels) is staged through LDS as [row][column][channel] bf16 --
//     channel-minor, so a lane's B operand (8 consecutive channels of its pixel) is ONE ds_read_b128 and the nine taps
//     are plain address offsets.  The transposition NCHW -> channel-minor happens in registers: a thread loads 8
//     channel planes x one aligned pixel quad (16-byte loads, coalesced along the row) and writes four 16-byte channel
//     vectors.  Pixel pitch 80 B and row pitch 2048 B make the 16 lanes of every ds_read_b128 phase hit 16 different bank quads.  Double
//     buffered: the loader waves fill the next 32-channel chunk during the 72 MFMAs per wave of the current one.
//   * weights are packed once per update as bf16 [tap][ci/8][cout][8]: a lane's A operand is one 16-byte global load,
//     a wave reads 512 contiguous bytes, straight from L2 (1.2 MB per layer, shared by every workgroup) into a
//     six-deep register ring, five (tap, k-step) groups ahead of the MFMAs that consume them.
#include "common.h"

namespace {
int g_bf16_dbg = 0;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kTR = 10, kTC = 16;              // output tile: rows x columns (200 = 20 x 10: 440 workgroups at B = 2)
constexpr int kKC = 32;                        // input channels per LDS chunk (2 MFMA k-steps per tap)
constexpr int kLR = kTR + 2, kLC = kTC + 8;    // LDS tile: rows r0-1 .. r0+10, columns c0-4 .. c0+19 (16-byte aligned quads)
constexpr int kColOff = 3;                     // LDS column of output column c, tap kx: c + kx + kColOff
constexpr int kPixB = 80;                      // bytes per pixel: 32 bf16 + 16 B pad
constexpr int kRowB = 2048;                    // bytes per tile row: 24 x 80 = 1920 -> multiple of 256
constexpr int kBufB = kLR * kRowB;             // 24 576 B
constexpr int kItems = kLR * (kLC / 4) * (kKC / 8);   // (row, pixel quad, 8-channel group) = 288

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct BfParams {
    const float *x;
    const unsigned short *wp;
    const float *shift;
    float *y;
    int B, Cin, CinP, Cout, H, W;                  // CinP: Cin rounded up to a whole 32-channel chunk
    int tiles_x, tiles_y;
    int nwg;                                       // real workgroups (the grid is padded to a multiple of 8)
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

// Wave specialisation: COW "MMA" waves (wave w owns 32 couts x all 128 pixels of the tile: no two waves fetch the same
// weights, and one 16-byte weight load feeds four MFMAs) + NLW "loader" waves that stage the NEXT 32-channel input chunk
// (global fp32 -> bf16 -> LDS) while the MMA waves multiply the current one.  The split matters because vector-memory
// loads return in order: with one wave doing both, every wait for a weight fragment (L2 latency) also waited for the
// input-tile loads issued before it (HBM / MALL latency), and the MFMA pipe idled for about half of every chunk.
// CPW = 32-cout blocks per MMA wave.  CPW = 2 (round 3: 4 MMA + 4 loader waves, 2 waves per SIMD, 256 VGPRs each): every
// B fragment read from LDS feeds TWO MFMAs and a SIMD hosts one MMA wave, so the wave's own ten MFMAs per step (320 cycles)
// cover its LDS / weight look-ahead -- with CPW = 1 (8 + 4 waves, 170 VGPRs) eight MMA waves issued their five-fragment
// ds_read bursts together and two of them shared every MFMA pipe (profiles/r03_stall_breakdown.json: MFMA busy 33 %).
template <int COW, int NLW, int DBG, int CPW = 1>
__global__ void __launch_bounds__(64 * (COW + NLW), CPW == 1 ? 3 : 2) conv2d_bf16_kernel(BfParams p)
{
    constexpr int NLT = 64 * NLW;                  // loader threads
    constexpr int NPB = kTR / 2;                   // 32-pixel blocks (2 rows x 16 columns) of the tile, all per MMA wave
    constexpr int RING = CPW == 1 ? 6 : 3;         // weight fragment STEPS in flight per MMA wave (18 % RING == 0)
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kBufB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = wave >= COW;               // wave-uniform
    const int li = lane & 31, lh = lane >> 5;

    // XCD-local tile order: blockIdx % 8 selects the XCD (dispatch order), and every XCD walks its own contiguous band of
    // tiles, so the halo rows / columns two neighbouring tiles share (44 % of a tile's input) meet in that XCD's L2 instead
    // of crossing the fabric twice (round-2 PMC: 2.8 x the algorithmic bytes; without the MFMAs the kernel still took 70 us
    // = 416 MB at the HBM copy rate -- it is bound by that traffic).  The grid is padded to a multiple of 8.
    int wg = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (wg >= p.nwg) return;                               // padding workgroup (uniform: before any barrier)
    const int tx = wg % p.tiles_x; wg /= p.tiles_x;
    const int ty = wg % p.tiles_y; wg /= p.tiles_y;
    const int b = wg % p.B;
    const int cot = wg / p.B;
    const int r0 = ty * kTR, c0 = tx * kTC;
    const size_t hw = (size_t)p.H * p.W;
    const int nchunk = p.CinP / kKC;

    if (loader) {
        // ---- staging: item -> (8-channel group, tile row, pixel pair).  Loads are unconditional (clamped address); the
        // zero padding of the image border is a select when the values are rounded and stored.  Channels past Cin (the
        // padding of the last chunk) read the last real plane: their weights are zero.
        const float *xb = p.x + (size_t)b * p.Cin * hw;
        const int lt = tid - 64 * COW;
        // an item = (8-channel group, tile row, pixel quad): 8 plane loads, then four 16-byte channel vectors into LDS.  With
        // 288 items on 256 loader threads 32 threads own two.  The loads of chunk c+2 are ISSUED before the barrier that ends
        // chunk c and converted / stored after it (hand-issued like the weight ring: the compiler's barrier does not drain
        // loads it does not know about), so a chunk's memory latency has a whole chunk of MFMAs to hide behind and the loader
        // waves reach every barrier early -- `profiles/r03_stall_breakdown.json` showed 48 % of the kernel's wave-cycles
        // parked, with load -> convert -> store -> barrier inside ONE chunk time.
        auto item_geom = [&](int e, bool live, bool &ok, int &dst_off, const float *&q, int &g) {
            const int qd = e % (kLC / 4), row = (e / (kLC / 4)) % kLR;
            g = e / (kLR * (kLC / 4));
            const int yy = r0 - 1 + row, xx = c0 - 4 + 4 * qd;            // xx % 4 == 0, W % 4 == 0: whole quad in or out
            ok = live && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            q = xb + (ok ? (size_t)yy * p.W + xx : 0);
            dst_off = row * kRowB + 4 * qd * kPixB + g * 16;
        };
        auto item_issue = [&](int e, bool live, f32x4 (&st)[8], int ci0) {
            bool ok; int d, g; const float *q;
            item_geom(e, live, ok, d, q, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ch = ok ? min(ci0 + g * 8 + j, p.Cin - 1) : 0;
                const float *a = q + (size_t)ch * hw;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(st[j]) : "v"(a));
            }
        };
        auto item_store = [&](int e, bool live, const f32x4 (&st)[8], int buf) {
            bool ok; int d, g; const float *q;
            item_geom(e, live, ok, d, q, g);
            unsigned char *dst = lds + buf * kBufB + d;
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                u32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = ok ? pack2(st[2 * j][px], st[2 * j + 1][px]) : 0u;
                *(u32x4 *)(dst + px * kPixB) = v;
            }
        };
        static_assert(kItems <= 2 * NLT, "a loader thread owns at most two items");
        const bool one = lt < kItems, two = lt + NLT < kItems;
        f32x4 s0[8], s1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s0[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; s1[j] = s0[j]; }
        auto issue = [&](int ci0) {
            if (one) item_issue(lt, true, s0, ci0);
            if (two) item_issue(lt + NLT, true, s1, ci0);
        };
        auto land = [&](int buf) {      // every hand-issued load of this wave has returned; registers -> LDS
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(s0[0]), "+v"(s0[1]), "+v"(s0[2]), "+v"(s0[3]), "+v"(s0[4]), "+v"(s0[5]), "+v"(s0[6]), "+v"(s0[7]),
                           "+v"(s1[0]), "+v"(s1[1]), "+v"(s1[2]), "+v"(s1[3]), "+v"(s1[4]), "+v"(s1[5]), "+v"(s1[6]), "+v"(s1[7]));
            if (one) item_store(lt, true, s0, buf);
            if (two) item_store(lt + NLT, true, s1, buf);
        };
        issue(0);
        land(0);
        if (nchunk > 1 && !(DBG & 1)) issue(kKC);
        __syncthreads();
        for (int c = 0; c < nchunk; ++c) {
            if (c + 1 < nchunk && !(DBG & 1)) {
                land((c + 1) & 1);                      // chunk c+1 (issued one chunk ago) -> the buffer chunk c-1 vacated
                if (c + 2 < nchunk) issue((c + 2) * kKC);
            }
            __syncthreads();
        }
        return;
    }

    // ---- MMA waves
    const int co_w = (cot * COW + wave) * 32 * CPW;     // this wave's first cout
    // Cout not a multiple of the workgroup's cout tile: the waves past Cout only keep the barriers company
    const bool active = co_w < p.Cout;
    // A: packed weights, element ((tap * CinP/8 + c8) * Cout + co) * 8; this lane: co = co_w + 32 h + li, c8 += lh
    const unsigned short *wl = p.wp + ((size_t)lh * p.Cout + (active ? co_w : 0) + li) * 8;
    const size_t w_c8 = (size_t)p.Cout * 8;                 // elements per 8-channel group
    const size_t w_tap = (size_t)(p.CinP / 8) * w_c8;       // elements per tap
    // B: LDS byte offset of this lane's pixel for block n: rows 2*n + (li>>4), column (li&15) + kColOff (+ kx per tap)
    const int b_off = ((li >> 4) * kRowB) + ((li & 15) + kColOff) * kPixB + lh * 16;

    f32x16 acc[CPW][NPB];
#pragma unroll
    for (int h = 0; h < CPW; ++h)
#pragma unroll
        for (int n = 0; n < NPB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[h][n][r] = 0.f;

    // The weight loads are issued by hand (inline asm + explicit s_waitcnt): left to the compiler, the scheduler sinks
    // every load next to its MFMA to save registers, which turns the ring into load -> wait -> use.  These are the only
    // vector-memory loads of an MMA wave inside the loop, so vmcnt counts exactly the ring (CPW loads per step).
    u32x4 aring[RING][CPW];
    auto load_a = [&](int slot, int chunk, int s) {          // s = 2*tap + kstep, chunk clamped by the caller
        const int tap = s >> 1, ks = s & 1;
        const unsigned short *q = wl + tap * w_tap + (size_t)(chunk * (kKC / 8) + ks * 2) * w_c8;
#pragma unroll
        for (int h = 0; h < CPW; ++h) {
            const unsigned short *qh = q + h * 32 * 8;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(aring[slot][h]) : "v"(qh));
        }
    };
    auto wait_a = [&](int slot) {      // RING * CPW loads are outstanding; the oldest CPW are this step's fragments
        if constexpr (CPW == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(aring[slot][0]) : "n"(RING - 1));
        else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(aring[slot][0]), "+v"(aring[slot][CPW - 1]) : "n"((RING - 1) * CPW));
    };
    static_assert(CPW == 1 || CPW == 2, "one or two cout blocks per MMA wave");
    if (active) {
#pragma unroll
        for (int s = 0; s < RING - 1; ++s) load_a(s, 0, s);
    }
    __syncthreads();

    for (int c = 0; c < nchunk; ++c) {
        const int cn = min(c + 1, nchunk - 1);
        const unsigned char *xt = lds + (c & 1) * kBufB + b_off;
        if (active) {
            // B fragments one step ahead of the MFMAs that consume them (double-buffered registers)
            u32x4 bf[2][NPB];
#pragma unroll
            for (int n = 0; n < NPB; ++n) bf[0][n] = *(const u32x4 *)(xt + (2 * n) * kRowB);
#pragma unroll
            for (int s = 0; s < 18; ++s) {
                // weights RING-1 steps ahead (the last steps of a chunk fetch the first ones of the next chunk)
                const int sa = s + RING - 1;
                if (DBG & 2) load_a(sa % RING, 0, 0);      // ablation: one hot fragment
                else if (sa < 18) load_a(sa % RING, c, sa);
                else load_a(sa % RING, cn, sa - 18);
                if (s + 1 < 18) {
                    const int tap = (s + 1) >> 1, ks = (s + 1) & 1;
                    const int ky = tap / 3, kx = tap % 3;
#pragma unroll
                    for (int n = 0; n < NPB; ++n)
                        bf[(s + 1) & 1][n] = *(const u32x4 *)(xt + (2 * n + ky) * kRowB + kx * kPixB + ks * 32);
                }
                wait_a(s % RING);
                if (!(DBG & 4)) {
#pragma unroll
                for (int n = 0; n < NPB; ++n)
#pragma unroll
                    for (int h = 0; h < CPW; ++h)
                        acc[h][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aring[s % RING][h]),
                                                                           __builtin_bit_cast(bf16x8, bf[s & 1][n]),
                                                                           acc[h][n], 0, 0, 0);
                } else {
#pragma unroll
                for (int n = 0; n < NPB; ++n)
#pragma unroll
                    for (int h = 0; h < CPW; ++h)
                        acc[h][n][0] += __builtin_bit_cast(float, aring[s % RING][h][0] ^ bf[s & 1][n][1]);
                }
            }
        }
        __syncthreads();
    }
    if (active) {     // drain the look-ahead loads of the (clamped) "next" chunk before their registers are reused
        if constexpr (CPW == 1) {
            static_assert(CPW != 1 || RING == 6, "the drain below names the six ring registers");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(aring[0][0]), "+v"(aring[1][0]), "+v"(aring[2][0]), "+v"(aring[3][0]),
                         "+v"(aring[4][0]), "+v"(aring[5][0]));
        } else {
            static_assert(CPW != 2 || RING == 3, "the drain below names the 3 x 2 ring registers");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(aring[0][0]), "+v"(aring[0][CPW - 1]), "+v"(aring[1][0]),
                         "+v"(aring[1][CPW - 1]), "+v"(aring[2][0]), "+v"(aring[2][CPW - 1]));
        }
    }

    // ---- epilogue: D row = cout = (r & 3) + 8 * (r >> 2) + 4 * lh, column = pixel li -> (row li >> 4, column li & 15)
    if (!active) return;
    float *yb = p.y + (size_t)b * p.Cout * hw;
    if (DBG & 8) {      // ablation: one store per lane that depends on every accumulator
        float t = 0.f;
#pragma unroll
        for (int h = 0; h < CPW; ++h)
#pragma unroll
            for (int n = 0; n < NPB; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[h][n][r];
        yb[(size_t)(co_w + lh) * hw + (size_t)(r0 + (li >> 4)) * p.W + c0 + (li & 15)] = t;
        return;
    }
    // Through LDS (the input buffers are dead: every MMA wave is past the last barrier, the loader waves have left): a
    // 32 couts x 32 pixels accumulator tile goes to the wave's private 4.5 KB image and comes back as 16-byte pieces
    // along the pixel rows -- 4 float4 stores per lane and tile instead of 16 dword stores.
    if (DBG & 16) {     // A/B: the dword-store epilogue of round 2
#pragma unroll
        for (int h = 0; h < CPW; ++h)
#pragma unroll
            for (int n = 0; n < NPB; ++n) {
                const int yy = r0 + 2 * n + (li >> 4), xx = c0 + (li & 15);
                if (yy >= p.H || xx >= p.W) continue;           // partial last tile row / column
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co_w + 32 * h + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float sh = p.shift ? p.shift[co] : 0.f;
                    yb[(size_t)co * hw + (size_t)yy * p.W + xx] = acc[h][n][r] + sh;
                }
            }
        return;
    }
    constexpr int kEP = 36;                                     // image row pitch in floats (16-byte aligned rows)
    static_assert(COW * 32 * kEP * 4 <= 2 * kBufB, "epilogue images exceed the LDS input buffers");
    float *img = (float *)lds + wave * 32 * kEP;
#pragma unroll
    for (int h = 0; h < CPW; ++h)
#pragma unroll
        for (int n = 0; n < NPB; ++n) {
#pragma unroll
            for (int r = 0; r < 16; ++r) img[((r & 3) + 8 * (r >> 2) + 4 * lh) * kEP + li] = acc[h][n][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private image: written, now read back
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = (j * 64 + lane) * 4;              // element of the [32 co][32 px] tile
                const int col = e >> 5, px = e & 31;            // cout row of the tile, first pixel of the quad
                const f32x4 v = *(const f32x4 *)(img + col * kEP + px);
                const int co = co_w + 32 * h + col;
                const int yy = r0 + 2 * n + (px >> 4), xx = c0 + (px & 15);
                if (yy < p.H && xx < p.W) {                     // W % 4 == 0: a quad is inside or outside as a whole
                    const float sh = p.shift ? p.shift[co] : 0.f;
                    *(f32x4 *)(yb + (size_t)co * hw + (size_t)yy * p.W + xx) =
                        (f32x4){v[0] + sh, v[1] + sh, v[2] + sh, v[3] + sh};
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads retired before the next tile overwrites the image
        }
}
}  // namespace

// ablation (tools/run_bf16_conv.py --ablate): bit0 stage only the first input chunk, bit1 re-load one hot weight
// fragment, bit2 no MFMA, bit3 no output stores, bit4 (value 16) the round-2 dword-store epilogue (all on the 8 + 4 wave
// kernel), value 32 the 8 + 4 wave kernel itself (the default is 4 MMA waves of 64 couts + 4 loader waves)
extern "C" void sassd_debug_set_bf16(int flags) { g_bf16_dbg = flags; }

extern "C" int sassd_conv2d_bf16_supported(int Cin, int Cout, int H, int W)
{
    // W % 4 == 0: the loader stages whole 4-pixel quads (in or out of the image together, 16-byte aligned); a map whose width is
    // not a multiple of the 16-column tile (the 188-wide Waymo-scale BEV map) gets a partial last tile column with masked stores
    return Cin >= 1 && Cout >= 32 && Cout % 32 == 0 && H >= 1 && W >= 16 && W % 4 == 0;
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
    p.tiles_x = cdiv(W, kTC); p.tiles_y = cdiv(H, kTR);
    hipStream_t s = (hipStream_t)stream_;
    const long tiles = (long)p.tiles_x * p.tiles_y * batch;
    if (Cout % 256 == 0) {
        p.nwg = (int)(tiles * (Cout / 256));
        const dim3 grid((unsigned)(cdiv(p.nwg, 8) * 8));
        switch (g_bf16_dbg) {       // compile-time ablation variants (a run-time switch inside the kernel de-tunes it)
#define SASSD_BF16_VARIANT(D) case D: hipLaunchKernelGGL((conv2d_bf16_kernel<8, 4, D>), grid, dim3(768), 0, s, p); break;
            SASSD_BF16_VARIANT(1) SASSD_BF16_VARIANT(2) SASSD_BF16_VARIANT(3) SASSD_BF16_VARIANT(4)
            SASSD_BF16_VARIANT(8) SASSD_BF16_VARIANT(11) SASSD_BF16_VARIANT(15) SASSD_BF16_VARIANT(16)
#undef SASSD_BF16_VARIANT
            case 32: hipLaunchKernelGGL((conv2d_bf16_kernel<8, 4, 0>), grid, dim3(768), 0, s, p); break;   // A/B: 8 + 4 waves
            default: hipLaunchKernelGGL((conv2d_bf16_kernel<4, 4, 0, 2>), grid, dim3(512), 0, s, p);       // 4 x 64 couts + 4
        }
    } else {    // 128-cout tiles; the last one may be partly idle (Cout = 320: 3 tiles, 2.5 used)
        p.nwg = (int)(tiles * cdiv(Cout, 128));
        hipLaunchKernelGGL((conv2d_bf16_kernel<4, 4, 0>), dim3((unsigned)(cdiv(p.nwg, 8) * 8)), dim3(512), 0, s, p);
    }
    return sassd_launch_status();
}
