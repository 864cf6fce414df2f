// conv2d_bf16.hip -- 3x3 pad-1 convolution of the BEV trunk on the bf16 MFMA pipe (BASELINE configs[2]: "training,
// batch=2/GPU ... bf16").  Forward of mmdet/models/necks/cmn.py:240-262 (the seven 3x3 convs of BEVNet) and, with the
// weights transposed and the taps mirrored, their data gradient (cuDNN under autocast in the reference).
//   y[b][co][y][x] = shift[co] + sum_{ci,ky,kx} bf16(w[co][ci][ky][kx]) * bf16(x[b][ci][y+ky-1][x+kx-1])
// NCHW fp32 activations in and out (BatchNorm / ReLU between the layers stay fp32), operands rounded to bf16
// (round-to-nearest-even) on their way into LDS / at pack time, fp32 accumulation on v_mfma_f32_32x32x16_bf16.
//
// 83 GFLOP per 256->256 layer at B = 2 against 2.5 PFLOP/s nominal; measured 0.089 ms = 0.37 of that, with the MFMA stream
// alone at 53 us and the staging side alone at 43 us (DESIGN.md section 8 has the decomposition).  Tiled as an implicit GEMM
// D[co][pixel] = sum_k W[co][k] X[k][pixel], k = (tap, ci):
//   * a TILE = up to 10 rows x 16 columns of one image (<= 160 pixels) x 256 (or 128) couts: 8 (4) MMA waves of 32 couts x
//     the tile's pixels (<= 5 accumulator tiles) + 4 loader waves.  The MFMA's 32 pixel columns are 2 image rows x 16
//     columns: the lane -> pixel map is free because every lane computes its own LDS address, and 176 = 11 x 16 tiles the
//     KITTI map with no padding.  Workgroups are persistent (one per CU) and walk a run of tiles each -- see the kernel.
//   * the input tile (<= 12 rows x 24 columns x 32 channels) is staged through LDS as [row][column][channel] bf16 --
//     channel-minor, so a lane's B operand (8 consecutive channels of its pixel) is ONE ds_read_b128 and the nine taps
//     are plain address offsets.  The transposition NCHW -> channel-minor happens in registers: a thread loads 8
//     channel planes x one aligned pixel quad (16-byte loads, coalesced along the row) and writes four 16-byte channel
//     vectors.  Pixel pitch 80 B and row pitch 2048 B make the 16 lanes of every ds_read_b128 phase hit 16 different bank
//     quads.  Double buffered: the loader waves fill the next 32-channel chunk during the 18 MFMA steps of the current one.
//   * weights are packed once per update as bf16 [tap][ci/8][cout][8]: a lane's A operand is one 16-byte global load,
//     a wave reads 512 contiguous bytes, straight from L2 (1.2 MB per layer, shared by every workgroup) into a
//     six-deep register ring, five (tap, k-step) groups ahead of the MFMAs that consume them.
#include <type_traits>

#include "common.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kTR = 10, kTC = 16;              // largest output tile: rows x columns
constexpr int kNPB = kTR / 2;                  // ... = 5 MFMA pixel blocks of 2 rows x 16 columns
constexpr int kMaxTiles = 64;                  // tiles of one workgroup's run (the launcher keeps runs <= 41 blocks)
constexpr int kEP = 36;                        // epilogue image row pitch in floats (16-byte aligned rows)
constexpr int kImgB = 32 * kEP * 4;            // one MMA wave's [32 couts][32 pixels] image
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
    int tiles_x;                                   // 16-column strips per image row
    int hb;                                        // 2-row blocks per strip: ceil(H / 2)
    int strips;                                    // cout tiles x B x tiles_x, cout tile slowest
    int wr_map;                                    // loader item order: 1 channel group fastest (default), 0 pixel quad fastest
    int mma_prio;                                  // 1: MMA waves at s_setprio 1 (default); 0: none; 2: loader waves at 2
    const float *in_aff;                           // BN = 1: [3][Cin] mean | invstd * gamma | beta of the BatchNorm + ReLU the
                                                   // loader waves apply to the input on its way into LDS (round 6)
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

// Wave specialisation: COW "MMA" waves (wave w owns 32 couts x all pixels of the tile: no two waves fetch the same weights,
// and one 16-byte weight load feeds up to five MFMAs) + NLW "loader" waves that stage the NEXT 32-channel input chunks
// (global fp32 -> bf16 -> LDS) while the MMA waves multiply the current one.  The split matters because vector-memory
// loads return in order: with one wave doing both, every wait for a weight fragment (L2 latency) also waited for the
// input-tile loads issued before it (HBM / MALL latency), and the MFMA pipe idled for about half of every chunk.
//
// Persistent workgroups (round 3): one workgroup fits a CU (12 waves at 170 VGPRs, 86 KB LDS), and 440 one-tile
// workgroups on 256 CUs ran as two rounds, the second 72 % full.  Now the launch has one workgroup per CU and the work --
// the 2-row MFMA blocks of every (cout tile, image, 16-column strip), G = strips x ceil(H / 2) of them -- is cut into
// gridDim.x equal runs (B = 2, 200 x 176: 2200 blocks -> runs of 8 or 9 instead of 2 x 5).  A run becomes a short list of
// tiles of 1..5 blocks (cut at strip ends, sizes balanced); the loader waves run two chunks ahead ACROSS tile boundaries,
// so a tile's first chunk is already in LDS when the MMA waves finish the previous tile's epilogue (which has its own LDS
// image area for that reason), and the weight ring keeps running from one tile into the next.
// A variant with 64 couts per MMA wave (4 MMA + 4 loader waves, every B fragment feeding two MFMAs) was built and measured
// in round 3: 0.104 ms against 0.104 ms -- neither LDS reads nor MFMA-pipe sharing bound the kernel.
struct LoadTile {                                  // what the loader waves need of a tile
    const float *xb;                               // image base
    int r0, c0, rows;                              // first output row / column, live LDS rows (2 nb + 2)
};

// BN = 1 (round 6): the input is the RAW output of the previous convolution and the loader waves normalise it -- training-mode
// BatchNorm2d + ReLU with the batch statistics sassd_bn2d_stats left in `in_aff`, z = fmaf(x - mean, invstd * gamma, beta),
// max(z, 0): the very expression of bn2d_apply_kernel, so the operand that reaches the MFMA is bit-identical to the one the
// stand-alone apply pass would have written -- before they round it to bf16.  The normalised map is never written to HBM
// (72 MB written + read back per 256-channel layer at batch 2 = the 20.7 us bn2d_apply_kernel took).
template <int COW, int NLW, int DBG, int BN = 0>
__global__ void __launch_bounds__(64 * (COW + NLW), 3) conv2d_bf16_kernel(BfParams p)
{
    constexpr int NLT = 64 * NLW;                  // loader threads
    constexpr int RING = 6;                        // weight fragment steps in flight per MMA wave (18 % RING == 0)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];     // 2 input buffers | COW images | tile list
    float *img_base = (float *)(lds + 2 * kBufB);
    int *tl = (int *)(lds + 2 * kBufB + COW * kImgB);   // [0] = tiles, then (strip, first block, blocks) per tile
    float *aff = (float *)(tl + 4 * ((1 + 3 * kMaxTiles + 3) / 4));   // BN: [3][CinP] (16-byte aligned: 8-channel groups are b128 reads)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // in an SGPR: what depends on it alone stays scalar
    const bool loader = wave >= COW;
    const int li = lane & 31, lh = lane >> 5;

    // XCD-local order: blockIdx % 8 selects the XCD (dispatch order), and every XCD owns one contiguous eighth of the runs,
    // so the halo rows / columns neighbouring tiles share (44 % of a tile's input) meet in that XCD's L2 instead of
    // crossing the fabric twice (round-2 PMC: 2.8 x the algorithmic bytes; without its MFMAs the kernel still took 70 us
    // = 416 MB at the HBM copy rate).  gridDim.x is a multiple of 8.
    const int v = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    const long G = (long)p.strips * p.hb;
    long g0 = (long)v * G / gridDim.x;
    const long g1 = (long)(v + 1) * G / gridDim.x;
    if (g0 >= g1) return;                          // more workgroups than blocks (uniform: before any barrier)
    if (tid == 0) {
        int n = 0;
        while (g0 < g1) {
            const int strip = (int)(g0 / p.hb);
            int blk = (int)(g0 % p.hb);
            const int seg = g1 - g0 < (long)(p.hb - blk) ? (int)(g1 - g0) : p.hb - blk;
            const int nt = (seg + kNPB - 1) / kNPB, base = seg / nt, extra = seg % nt;
            for (int i = 0; i < nt; ++i) {
                const int nb = base + (i < extra);
                tl[1 + 3 * n] = strip; tl[2 + 3 * n] = blk; tl[3 + 3 * n] = nb;
                blk += nb; ++n;
            }
            g0 += seg;
        }
        tl[0] = n;                                 // <= kMaxTiles: the launcher bounds the run length
    }
    if constexpr (BN) {
        for (int i = tid; i < 3 * p.CinP; i += 64 * (COW + NLW)) {
            const int k = i / p.CinP, c = i - k * p.CinP;
            aff[i] = p.in_aff[k * p.Cin + min(c, p.Cin - 1)];      // padding channels: any finite value (their weights are zero)
        }
    }
    __syncthreads();
    const int ntile = tl[0];
    const size_t hw = (size_t)p.H * p.W;
    const int nchunk = p.CinP / kKC;
    const int Q = ntile * nchunk;                  // chunk steps (= barriers) of this workgroup

    if (loader) {
        // ---- staging: item -> (8-channel group, tile row, pixel quad): 8 plane loads, then four 16-byte channel vectors
        // into LDS.  Loads are unconditional inside the tile (clamped address); the zero padding of the image border is a
        // select when the values are rounded and stored.  Channels past Cin (the padding of the last chunk) read the last
        // real plane: their weights are zero.  With 288 items on 256 loader threads 32 threads own two.  The loads of chunk
        // q+2 are ISSUED before the barrier that ends chunk q and converted / stored after it (hand-issued like the weight
        // ring: the compiler's barrier does not drain loads it does not know about), so a chunk's memory latency has a
        // whole chunk of MFMAs to hide behind and the loader waves reach every barrier early.
        const int lt = tid - 64 * COW;
        if (p.mma_prio == 2) __builtin_amdgcn_s_setprio(2);
        auto tile_geom = [&](int t) {
            const int strip = tl[1 + 3 * t], blk = tl[2 + 3 * t], nb = tl[3 + 3 * t];
            const int tx = strip % p.tiles_x, b = (strip / p.tiles_x) % p.B;
            LoadTile T;
            T.xb = p.x + (size_t)b * p.Cin * hw;
            T.r0 = 2 * blk; T.c0 = tx * kTC; T.rows = 2 * nb + 2;
            return T;
        };
        auto item_geom = [&](int e, const LoadTile &T, bool &live, bool &ok, int &dst_off, const float *&q, int &g) {
            // channel group fastest: the 8 lanes of one ds_write_b128 phase hold 4 groups x 2 pixel quads = 8 different
            // 16-byte slots of the 128-byte bank window (slot = (4 (qd & 1) + 5 px + g) mod 8 with the 80-byte pixel pitch).
            // With the quad fastest (round 2) they fell on 2 slots: 4-way conflicts, 32 LDS cycles per store instead of 8.
            int qd, row;
            if (p.wr_map) { g = e % (kKC / 8); qd = (e / (kKC / 8)) % (kLC / 4); row = e / ((kKC / 8) * (kLC / 4)); }
            else { qd = e % (kLC / 4); row = (e / (kLC / 4)) % kLR; g = e / (kLR * (kLC / 4)); }
            const int yy = T.r0 - 1 + row, xx = T.c0 - 4 + 4 * qd;        // xx % 4 == 0, W % 4 == 0: whole quad in or out
            live = row < T.rows;                                           // a short tile leaves the last LDS rows alone
            ok = live && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            q = T.xb + (ok ? (size_t)yy * p.W + xx : 0);
            dst_off = row * kRowB + 4 * qd * kPixB + g * 16;
        };
        auto item_issue = [&](int e, const LoadTile &T, f32x4 (&st)[8], int ci0) {
            bool live, ok; int d, g; const float *q;
            item_geom(e, T, live, ok, d, q, g);
            if (!live) return;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ch = ok ? min(ci0 + g * 8 + j, p.Cin - 1) : 0;
                const float *a = q + (size_t)ch * hw;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(st[j]) : "v"(a));
            }
        };
        auto item_store = [&](int e, const LoadTile &T, const f32x4 (&st)[8], int buf, int ci0) {
            bool live, ok; int d, g; const float *q;
            item_geom(e, T, live, ok, d, q, g);
            if (!live) return;
            unsigned char *dst = lds + buf * kBufB + d;
            float am[8], as[8], ab[8];
            if constexpr (BN) {
                const float *a0 = aff + ci0 + g * 8;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 m4 = *(const f32x4 *)(a0 + 4 * h), s4 = *(const f32x4 *)(a0 + p.CinP + 4 * h),
                                b4 = *(const f32x4 *)(a0 + 2 * p.CinP + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { am[4 * h + j] = m4[j]; as[4 * h + j] = s4[j]; ab[4 * h + j] = b4[j]; }
                }
            }
            auto val = [&](int j, int px) {
                float v = st[j][px];
                if constexpr (BN) {
                    const float z = fmaf(v - am[j], as[j], ab[j]);     // == bn2d_apply_kernel
                    v = z > 0.f ? z : 0.f;
                }
                return v;
            };
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                u32x4 v4;
#pragma unroll
                for (int j = 0; j < 4; ++j) v4[j] = ok ? pack2(val(2 * j, px), val(2 * j + 1, px)) : 0u;   // zero padding AFTER the ReLU
                *(u32x4 *)(dst + px * kPixB) = v4;
            }
        };
        static_assert(kItems <= 2 * NLT, "a loader thread owns at most two items");
        const bool one = lt < kItems, two = lt + NLT < kItems;
        f32x4 s0[8], s1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s0[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; s1[j] = s0[j]; }
        auto issue = [&](const LoadTile &T, int ci0) {
            if (one) item_issue(lt, T, s0, ci0);
            if (two) item_issue(lt + NLT, T, s1, ci0);
        };
        auto land = [&](const LoadTile &T, int buf, int ci0) {   // every hand-issued load of this wave has returned; registers -> LDS
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(s0[0]), "+v"(s0[1]), "+v"(s0[2]), "+v"(s0[3]), "+v"(s0[4]), "+v"(s0[5]), "+v"(s0[6]), "+v"(s0[7]),
                           "+v"(s1[0]), "+v"(s1[1]), "+v"(s1[2]), "+v"(s1[3]), "+v"(s1[4]), "+v"(s1[5]), "+v"(s1[6]), "+v"(s1[7]));
            if (one) item_store(lt, T, s0, buf, ci0);
            if (two) item_store(lt + NLT, T, s1, buf, ci0);
        };
        // two cursors over the (tile, chunk) sequence: the one being landed and the one being issued (one step ahead of it)
        int t_land = 0, c_land = 0, t_iss = 0, c_iss = 0;
        LoadTile Tl = tile_geom(0), Ti = Tl;
        auto advance = [&](int &t, int &c, LoadTile &T) {
            if (++c == nchunk) { c = 0; ++t; if (t < ntile) T = tile_geom(t); }
        };
        issue(Ti, 0);
        land(Tl, 0, 0);
        if (Q > 1 && !(DBG & 1)) { advance(t_iss, c_iss, Ti); issue(Ti, (DBG & 16) ? 0 : c_iss * kKC); }
        __syncthreads();
        for (int q = 0; q < Q; ++q) {
            if (q + 1 < Q && !(DBG & 1)) {
                advance(t_land, c_land, Tl);
                land(Tl, (q + 1) & 1, (DBG & 16) ? 0 : c_land * kKC);   // step q+1 (issued one step ago) -> the buffer step q-1 vacated
                if (q + 2 < Q) { advance(t_iss, c_iss, Ti); issue(Ti, (DBG & 16) ? 0 : c_iss * kKC); }
            }
            if (!(DBG & 64)) __syncthreads();
        }
        return;
    }

    // ---- MMA waves
    // A: packed weights, 16-byte entry (tap * CinP/8 + c8) * Cout + co; this lane: co = co_w + li, c8 += lh.  The lane's
    // part of the address is one 32-bit byte offset; tile, chunk, tap and k-step are a scalar base (global_load saddr form:
    // no vector ALU and no 64-bit address registers per load).  A wave whose couts lie past Cout (Cout not a multiple of
    // the workgroup's cout tile) only keeps the barriers company; cout tiles only grow along a run, so such a wave never
    // becomes active again.
    const unsigned a_lane = (unsigned)(lh * p.Cout + li) * 16u;
    const size_t w_c8 = (size_t)p.Cout * 16;                // bytes per 8-channel group
    const size_t w_tap = (size_t)(p.CinP / 8) * w_c8;       // bytes per tap
    const size_t w_chunk = (size_t)(kKC / 8) * w_c8;        // bytes per 32-channel chunk
    auto tile_cot = [&](int t) { return __builtin_amdgcn_readfirstlane(tl[1 + 3 * t]) / (p.tiles_x * p.B); };
    auto weights_of = [&](int cot, bool &act) {             // scalar: this wave's 32 couts, chunk 0, tap 0
        const int co_w = (cot * COW + wave) * 32;
        act = co_w < p.Cout;
        return (const char *)p.wp + (size_t)(act ? co_w : 0) * 16;
    };
    // B: LDS byte offset of this lane's pixel for block n: rows 2*n + (li>>4), column (li&15) + kColOff (+ kx per tap)
    const int b_off = ((li >> 4) * kRowB) + ((li & 15) + kColOff) * kPixB + lh * 16;

    f32x16 acc[kNPB];
    // The weight loads are issued by hand (inline asm + explicit s_waitcnt): left to the compiler, the scheduler sinks
    // every load next to its MFMA to save registers, which turns the ring into load -> wait -> use.  Inside the chunk loop
    // these are the only vector-memory loads of an MMA wave, so vmcnt counts exactly the ring; the stores (and the shift
    // loads) of an epilogue in between only make the wait that follows them more conservative.
    u32x4 aring[RING];
    auto load_a = [&](int slot, const char *base, int s) {      // base = this wave's weights of one chunk (scalar)
        const int tap = s >> 1, ks = s & 1;
        const char *q = base + tap * w_tap + (size_t)(ks * 2) * w_c8;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(aring[slot]) : "v"(a_lane), "s"(q));
    };
    auto wait_a = [&](int slot) {      // RING loads are outstanding; the oldest is this step's fragment
        if (DBG & 128) return;
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(aring[slot]) : "n"(RING - 1));
    };
    // Every ring register has its data.  A register with a load in flight must not be copied or spilled (the compiler does
    // not know about the load), and it may do either where control flow joins -- between tiles (the tile bodies are separate
    // instantiations) -- so the ring is only ever in flight inside one tile's chunk loop, where it was verified in the ISA.
    auto settle_ring = [&]() {
        static_assert(RING == 6, "names the six ring registers");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(aring[0]), "+v"(aring[1]), "+v"(aring[2]), "+v"(aring[3]), "+v"(aring[4]),
                     "+v"(aring[5]));
    };
    // ... and in registers: a reload from a spill slot (tracked by the compiler) is waited for HERE, not by an
    // s_waitcnt vmcnt(0) at the first use inside the chunk loop, which would drain the ring every chunk
    auto touch_ring = [&]() {
        asm volatile("" : "+v"(aring[0]), "+v"(aring[1]), "+v"(aring[2]), "+v"(aring[3]), "+v"(aring[4]), "+v"(aring[5]));
    };
    bool active;
    const char *wl = weights_of(tile_cot(0), active);
    if (active) {
#pragma unroll
        for (int s = 0; s < RING - 1; ++s) load_a(s, wl, s);
        settle_ring();
    }
    __syncthreads();
    if (p.mma_prio == 1) __builtin_amdgcn_s_setprio(1);              // the MFMA stream before the loader waves' conversions

    int q = 0;                                                  // chunk step: the input buffer is q & 1
    for (int t = 0; t < ntile; ++t) {
        const int strip = __builtin_amdgcn_readfirstlane(tl[1 + 3 * t]);
        const int blk = __builtin_amdgcn_readfirstlane(tl[2 + 3 * t]);
        const int nb = __builtin_amdgcn_readfirstlane(tl[3 + 3 * t]);
        const int tx = strip % p.tiles_x, b = (strip / p.tiles_x) % p.B, cot = strip / (p.tiles_x * p.B);
        const int r0 = 2 * blk, c0 = tx * kTC;
        const int co_w = (cot * COW + wave) * 32;
        bool active_next = active;
        const char *wl_next = t + 1 < ntile ? weights_of(tile_cot(t + 1), active_next) : wl;
#pragma unroll
        for (int n = 0; n < kNPB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

        auto run_tile = [&](auto nbc) {
            constexpr int NB = decltype(nbc)::value;            // 2-row blocks of this tile
            if (!active) {                                      // keep the barriers company
                for (int c = 0; c < nchunk; ++c, ++q)
                    if (!(DBG & 64)) __syncthreads();
                return;
            }
            touch_ring();
            int c = 0;
            do {        // nchunk >= 1: no zero-trip path that would have to keep the ring's pre-loop values alive
                const unsigned char *xt = lds + (q & 1) * kBufB + b_off;
                {
                    const char *wa = wl + c * w_chunk;
                    // the last steps of a chunk fetch the first ones of the next chunk -- of the next tile after the last
                    const char *wb = c + 1 < nchunk ? wa + w_chunk : wl_next;
                    // B fragments one step ahead of the MFMAs that consume them (double-buffered registers)
                    u32x4 bf[2][NB];
#pragma unroll
                    for (int n = 0; n < NB; ++n) bf[0][n] = *(const u32x4 *)(xt + (2 * n) * kRowB);
                    if (DBG & 32) {
#pragma unroll
                        for (int n = 0; n < NB; ++n) bf[1][n] = bf[0][n];
                    }
#pragma unroll
                    for (int s = 0; s < 18; ++s) {
                        const int sa = s + RING - 1;            // weights RING-1 steps ahead
                        if (DBG & 128) {}                       // ablation: no weight loads
                        else if (DBG & 2) load_a(sa % RING, (const char *)p.wp, 0);     // ablation: one hot fragment
                        else if (sa < 18) load_a(sa % RING, wa, sa);
                        else load_a(sa % RING, wb, sa - 18);
                        if (s + 1 < 18 && !(DBG & 32)) {
                            const int tap = (s + 1) >> 1, ks = (s + 1) & 1;
                            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
                            for (int n = 0; n < NB; ++n)
                                bf[(s + 1) & 1][n] = *(const u32x4 *)(xt + (2 * n + ky) * kRowB + kx * kPixB + ks * 32);
                        }
                        wait_a(s % RING);
                        if (!(DBG & 4)) {
#pragma unroll
                            for (int n = 0; n < NB; ++n)
                                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aring[s % RING]),
                                                                                __builtin_bit_cast(bf16x8, bf[s & 1][n]),
                                                                                acc[n], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int n = 0; n < NB; ++n)
                                acc[n][0] += __builtin_bit_cast(float, aring[s % RING][0] ^ bf[s & 1][n][1]);
                        }
                    }
                }
                if (!(DBG & 64)) __syncthreads();
                ++q;
            } while (++c < nchunk);
            settle_ring();      // the first fragments of the next tile (loaded during this tile's last steps)

            // ---- epilogue: D row = cout = (r & 3) + 8 * (r >> 2) + 4 * lh, column = pixel li -> (row li >> 4, column li & 15)
            // The lane id goes through an opaque asm so that the epilogue's per-lane addresses are computed here, per tile:
            // hoisted out of the tile loop they stayed live through the chunk loop, and the spills that followed put
            // compiler-tracked scratch reloads -- each with an s_waitcnt vmcnt(0) that drains the weight ring -- into it.
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int li = lane_e & 31, lh = lane_e >> 5, lane = lane_e;
            float *yb = p.y + (size_t)b * p.Cout * hw;
            if (DBG & 8) {      // ablation: one store per lane that depends on every accumulator
                float tsum = 0.f;
#pragma unroll
                for (int n = 0; n < NB; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tsum += acc[n][r];
                if (r0 + (li >> 4) < p.H && c0 + (li & 15) < p.W)
                    yb[(size_t)(co_w + lh) * hw + (size_t)(r0 + (li >> 4)) * p.W + c0 + (li & 15)] = tsum;
                return;
            }
            // Through LDS: a 32 couts x 32 pixels accumulator tile goes to the wave's private 4.5 KB image and comes back
            // as 16-byte pieces along the pixel rows -- 4 float4 stores per lane and tile instead of 16 dword stores.  The
            // image area is separate from the input buffers: the loader waves are already filling those for the next tile.
            float *img = img_base + wave * 32 * kEP;
#pragma unroll
            for (int n = 0; n < NB; ++n) {
#pragma unroll
                for (int r = 0; r < 16; ++r) img[((r & 3) + 8 * (r >> 2) + 4 * lh) * kEP + li] = acc[n][r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private image: written, now read back
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = (j * 64 + lane) * 4;              // element of the [32 co][32 px] tile
                    const int col = e >> 5, px = e & 31;            // cout row of the tile, first pixel of the quad
                    const f32x4 v4 = *(const f32x4 *)(img + col * kEP + px);
                    const int co = co_w + col;
                    const int yy = r0 + 2 * n + (px >> 4), xx = c0 + (px & 15);
                    if (yy < p.H && xx < p.W) {                     // W % 4 == 0: a quad is inside or outside as a whole
                        const float sh = p.shift ? p.shift[co] : 0.f;
                        *(f32x4 *)(yb + (size_t)co * hw + (size_t)yy * p.W + xx) =
                            (f32x4){v4[0] + sh, v4[1] + sh, v4[2] + sh, v4[3] + sh};
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads retired before the next block overwrites the image
            }
        };
        switch (nb) {
            case 1: run_tile(std::integral_constant<int, 1>{}); break;
            case 2: run_tile(std::integral_constant<int, 2>{}); break;
            case 3: run_tile(std::integral_constant<int, 3>{}); break;
            case 4: run_tile(std::integral_constant<int, 4>{}); break;
            default: run_tile(std::integral_constant<int, 5>{}); break;
        }
        wl = wl_next;
        active = active_next;
    }
}
}  // namespace

// ablation (tools/run_bf16_conv.py --ablate), low byte: bit0 stage only the first input chunk, bit1 re-load one hot weight
// fragment, bit2 no MFMA, bit3 no output stores, bit4 (value 16) the loader waves fetch channel chunk 0 every time (input hot in L2).  flags >> 8, if not
// zero, forces the number of workgroups (tests: long runs with many tiles on small maps).  Bits 5..7 (with 11 = no global
// traffic) take the MMA loop apart: 32 no B-fragment reads, 64 no chunk barriers, 128 no weight loads.  0x10000 in the
// upper half: MMA waves at default priority; 0x20000: the loader waves at priority 2 instead; 0x40000: the round-2
// loader item order (4-way conflicted LDS stores).

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

namespace {
template <int COW, int DBG, int BN = 0>
int launch_bf16(const BfParams &p, int nwg, hipStream_t s)
{
    static std::atomic<unsigned long long> attr_done{0};
    const size_t tl_b = (size_t)4 * ((1 + 3 * kMaxTiles + 3) / 4) * sizeof(int);
    const size_t lds_max = align_up((size_t)2 * kBufB + (size_t)COW * kImgB + tl_b + (BN ? (size_t)3 * 1024 * 4 : 0), 16);
    const size_t lds = align_up((size_t)2 * kBufB + (size_t)COW * kImgB + tl_b + (BN ? (size_t)3 * p.CinP * 4 : 0), 16);
    const int rc = sassd_dyn_lds((const void *)conv2d_bf16_kernel<COW, 4, DBG, BN>, lds_max, attr_done);
    if (rc != SASSD_OK) return rc;
    hipLaunchKernelGGL((conv2d_bf16_kernel<COW, 4, DBG, BN>), dim3((unsigned)nwg), dim3(64 * (COW + 4)), lds, s, p);
    return sassd_launch_status();
}
}  // namespace

static int conv2d_bf16_launch(const float *x, const float *in_aff, const void *w_packed, const float *shift, float *y,
                              int batch, int Cin, int Cout, int H, int W, int cfg, void *stream_)
{
    if (!x || !w_packed || !y || batch < 1) return SASSD_EINVAL;
    if (!sassd_conv2d_bf16_supported(Cin, Cout, H, W)) return SASSD_EINVAL;
    if (in_aff && Cin > 1024) return SASSD_EINVAL;
    BfParams p;
    p.x = x; p.wp = (const unsigned short *)w_packed; p.shift = shift; p.y = y; p.in_aff = in_aff;
    p.B = batch; p.Cin = Cin; p.CinP = (int)align_up(Cin, 32); p.Cout = Cout; p.H = H; p.W = W;
    p.tiles_x = cdiv(W, kTC); p.hb = cdiv(H, 2);
    // 256-cout workgroups (8 MMA waves) when they divide Cout; 160-cout ones (5 MMA waves) when THEY do -- Cout = 320, the data
    // gradient of BEVNet's conv0 (256 -> 320): two full tiles instead of three 128-cout tiles of which 2.5 work (round 6: 172 ->
    // 1xx us at batch 2); else 128-cout ones whose last tile may be partly idle
    const bool wide = Cout % 256 == 0;
    const bool five = !wide && Cout % 160 == 0 && !in_aff && !(cfg & 0x80000);      // (0x80000: the 128-cout form, A/B)
    const long strips = (long)(wide ? Cout / 256 : five ? Cout / 160 : cdiv(Cout, 128)) * batch * p.tiles_x;
    const long G = strips * p.hb;
    if (strips > 0x7fffffffL / p.hb) return SASSD_EINVAL;
    p.strips = (int)strips;
    // one workgroup per CU, each with an equal run of the G blocks; more (a multiple of the CU count) when a run would
    // exceed 40 blocks, fewer when there are not 4 blocks per CU to hand out.  Always a multiple of 8 (XCD order).
    int cus = 0;
    const int rc = sassd_num_cus(&cus);
    if (rc != SASSD_OK) return rc;
    cus = (int)align_up((size_t)cus, 8);
    long nwg = (long)cus * ((G + (long)cus * 40 - 1) / ((long)cus * 40));
    if (G < 4 * nwg) nwg = (long)align_up((size_t)((G + 3) / 4), 8);
    p.wr_map = (cfg & 0x40000) ? 0 : 1;
    p.mma_prio = (cfg & 0x10000) ? 0 : (cfg & 0x20000) ? 2 : 1;
    if ((cfg >> 8) & 0xff) nwg = (long)align_up((size_t)std::max((long)((cfg >> 8) & 0xff), (G + 39) / 40), 8);
    hipStream_t s = (hipStream_t)stream_;
    if (in_aff) return wide ? launch_bf16<8, 0, 1>(p, (int)nwg, s) : launch_bf16<4, 0, 1>(p, (int)nwg, s);
    if (five) return launch_bf16<5, 0>(p, (int)nwg, s);
    if (!wide) return launch_bf16<4, 0>(p, (int)nwg, s);
    switch (cfg & 0xff) {        // compile-time ablation variants (a run-time switch inside the kernel de-tunes it)
#define SASSD_BF16_VARIANT(D) case D: return launch_bf16<8, D>(p, (int)nwg, s);
        SASSD_BF16_VARIANT(1) SASSD_BF16_VARIANT(2) SASSD_BF16_VARIANT(3) SASSD_BF16_VARIANT(4)
        SASSD_BF16_VARIANT(8) SASSD_BF16_VARIANT(11) SASSD_BF16_VARIANT(15) SASSD_BF16_VARIANT(16)
        SASSD_BF16_VARIANT(43) SASSD_BF16_VARIANT(75) SASSD_BF16_VARIANT(171) SASSD_BF16_VARIANT(235)
        SASSD_BF16_VARIANT(172) SASSD_BF16_VARIANT(164) SASSD_BF16_VARIANT(168) SASSD_BF16_VARIANT(40)
#undef SASSD_BF16_VARIANT
        default: return launch_bf16<8, 0>(p, (int)nwg, s);
    }
}

extern "C" int sassd_conv2d_bf16_fwd(const float *x, const void *w_packed, const float *shift, float *y, int batch,
                                     int Cin, int Cout, int H, int W, void *stream_)
{
    return conv2d_bf16_launch(x, nullptr, w_packed, shift, y, batch, Cin, Cout, H, W, 0, stream_);
}

// `cfg` (per call since round 6; 0 in production): low byte = compile-time ablation variant (tools/run_bf16_conv.py), bits 8-15 =
// forced workgroup count (tests: long runs of tiles), 0x10000 / 0x20000 / 0x40000 = wave-priority and loader-order A/B switches,
// 0x80000 = Cout = 320 on 128-cout instead of 160-cout workgroups
extern "C" int sassd_conv2d_bf16_fwd_cfg(const float *x, const void *w_packed, const float *shift, float *y, int batch,
                                         int Cin, int Cout, int H, int W, int cfg, void *stream_)
{
    return conv2d_bf16_launch(x, nullptr, w_packed, shift, y, batch, Cin, Cout, H, W, cfg, stream_);
}

// The same convolution over relu(batchnorm(x)) with the normalisation applied by the loader waves: in_affine = [3][Cin]
// (mean | invstd * gamma | beta), as sassd_bn2d_stats writes it.  Bit-identical to sassd_bn2d_relu_fwd + sassd_conv2d_bf16_fwd.
extern "C" int sassd_conv2d_bf16_bnrelu_fwd(const float *x, const float *in_affine, const void *w_packed, const float *shift,
                                            float *y, int batch, int Cin, int Cout, int H, int W, void *stream_)
{
    if (!in_affine) return SASSD_EINVAL;
    return conv2d_bf16_launch(x, in_affine, w_packed, shift, y, batch, Cin, Cout, H, W, 0, stream_);
}
