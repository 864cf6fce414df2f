// spconv.hip -- output-stationary sparse 3-D convolution forward with LDS-staged gather / scatter.
//
// Replaces spconv v1.0 `indice_conv_fp32` (27 x {gather kernel, cuBLAS sgemm, scatter-add kernel} per layer,
// ~80 launches) + nn.BatchNorm1d + nn.ReLU (mmdet/models/necks/cmn.py:138-173,208-212) by ONE launch per layer.
//
// Work decomposition (wave = 64 lanes, CDNA4): see spconv_fwd_kernel below -- a register-stationary design: every
// wave owns a 16-row x 16-channel output tile for ALL kernel offsets, accumulates in registers with
// v_mfma_f32_16x16x4_f32 (exact fp32; the fp32 MFMA rate equals the fp32 VALU rate but needs one VGPR per operand)
// and gathers its A operand straight from HBM/L2 in fragment order.  An earlier LDS-staged gather/scatter version
// (per-offset compaction + LDS accumulators) was 3-4x slower at KITTI scale: at ~14k rows the layer is latency
// bound, and per-offset barriers / LDS round trips cost more than the zero-padded MFMA work they saved.
// Roofline: HBM/L2 bound (<= 16 FLOP/B); algorithmic bytes B_gs = 4P(Cin+Cout) + 8P + 4K*Cin*Cout + 4*Nout*Cout.
#include "common.h"

#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));


constexpr int kK = 27;

template <int CIN, int COUT>
struct SpShape {
    static constexpr int KS = CIN / 4;            // floats per lane per operand = MFMA steps
    static constexpr int NT = COUT / 16;          // waves per workgroup
    static constexpr int LDA = COUT + 4;          // accumulator row stride (floats), keeps float4 alignment
};

template <int N>
__device__ __forceinline__ void load_vec(const float *__restrict__ p, float (&v)[N])
{
    if constexpr (N >= 4) {
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            const float4 t = ((const float4 *)p)[i];
            v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = p[i];
    }
}

// w [K][CIN][COUT] -> packed [K][CIN/4][COUT][4]: element (k, ci, co) at ((k*CIN/4 + ci/4)*COUT + co)*4 + ci%4 -- four
// consecutive input channels of one output channel are one 16-byte piece, pieces of neighbouring output channels are
// contiguous, so every kernel of this file (16x16x4 fragments: lane (q, m) reads W[q*KS + kk][nt*16 + m]; 4x4x1 rows:
// lane l reads W[ci][l]) fetches its operand registers with coalesced 16-byte loads from ONE image.
// transposed = 1: `w` is the FORWARD weight [K][COUT][CIN] of a Cout->Cin layer and the packed image is W[k]^T, i.e.
// the weights of the data-gradient conv (CIN x COUT here are the gradient conv's own in/out widths).
template <int CIN, int COUT>
__global__ void pack_weight_kernel(const float *__restrict__ w, int K, float *__restrict__ packed, int transposed)
{
    const int total = K * CIN * COUT;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int r = i & 3;
    const int co = (i >> 2) % COUT;
    const int c4 = (i / (4 * COUT)) % (CIN / 4);
    const int k = i / (CIN * COUT);
    const int ci = c4 * 4 + r;
    packed[i] = transposed ? w[((size_t)k * COUT + co) * CIN + ci] : w[((size_t)k * CIN + ci) * COUT + co];
}

// (rounds 1-5 kept the ablation / geometry switches in two process-wide ints behind sassd_debug_set_spconv; since round 6 they
// are the `cfg` argument of the entry points: bits 0-15 ablation flags, bits 16.. geometry -- per call, nothing a hipGraph
// capture could bake in by accident)

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

// Register-stationary sparse conv.  One wave = 16 output rows x ALL Cout channels x every S-th kernel offset;
// the S waves of a row group split the 27 offsets (k = j*S + s) and add their partial sums through LDS at the end.
//   accumulators  NT x f32x4 (x2 chains) per lane in registers: no LDS accumulation, no atomics, deterministic
//                 summation order (within a wave k ascending, then wave 0 + wave 1 + wave 2)
//   rulebook      the lane's row record entries nbr[row][j*S + s] are loaded once into registers
//   A operand     rows gathered straight from HBM/L2 in MFMA fragment order (lane (m,q) reads Cin/4 contiguous
//                 floats of row nbr[m][k]; missing neighbours contribute zeros), each row read ONCE per offset,
//                 prefetched one iteration ahead under the previous iteration's MFMAs
//   B operand     W[k] (all Cout) of the S offsets of an iteration is shared by the workgroup's 4 row groups:
//                 direct global->LDS DMA of the pre-packed fragment image (lane-linear, conflict-free 16-B reads),
//                 double buffered, one barrier per iteration.
//   why S = 3     at ~14 k rows a layer is bound by the serial chain "gather latency x 27 offsets" of a wave, not by
//                 the MFMA or memory pipes; splitting the offsets over 3 waves cuts the chain to 9 iterations and
//                 puts 2.4 waves on every SIMD.
//   offsets with no neighbour in the wave's 16 rows skip their MFMAs (wave-uniform ballot)
constexpr int kSplit = 3;                     // waves per row group
constexpr int kIter = (kK + kSplit - 1) / kSplit;

template <int CIN, int COUT>
__global__ void __launch_bounds__(4 * kSplit * 64)
spconv_fwd_kernel(const float *__restrict__ x, const int32_t *__restrict__ nbr, const int32_t *__restrict__ n_ptr,
                  int cap, const float *__restrict__ wp, int K, const float *__restrict__ scale,
                  const float *__restrict__ shift, int relu, float *__restrict__ y, int dbg)
{
    constexpr int S = kSplit, NW = 4 * S;
    constexpr int KS = CIN / 4, NT = COUT / 16;
    constexpr bool BLDS = (KS >= 4);                // B through LDS-DMA (16-B pieces); KS == 1: plain loads
    constexpr int J = BLDS ? KS / 4 : 1;            // float4 pieces per lane per channel tile
    constexpr int NWL = NT * J;                     // B wave-loads per offset
    constexpr int BOFF = NWL * 256;                 // floats per offset image
    constexpr int BBUF = S * BOFF;                  // floats per iteration buffer
    constexpr int RED = 4 * (S - 1) * 64 * NT * 4;  // floats for the final partial-sum exchange
    constexpr int LDSF = (2 * BBUF > RED) ? 2 * BBUF : RED;
    __shared__ __attribute__((aligned(16))) float bs[LDSF];

    const int n = min(*n_ptr, cap);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((int)blockIdx.x * 64 >= n) return;          // workgroup-uniform
    const int rg = wave / S, sp = wave - rg * S;
    const int rb = (blockIdx.x * 4 + rg) * 16;
    const int q = lane >> 4, m16 = lane & 15;
    const int row = rb + m16;
    const bool rok = row < n;
    const int KK = nbr ? kK : 1;                    // identity rulebook: a single offset

    // ---- this wave's rulebook entries -> registers -------------------------------------------------
    int nb[kIter];
    unsigned act = 0;                               // bit j: some row of this wave has a neighbour at offset j*S+sp
#pragma unroll
    for (int j = 0; j < kIter; ++j) {
        const int k = j * S + sp;
        nb[j] = -1;
        if (k < KK && rok) nb[j] = nbr ? nbr[(size_t)row * kK + k] : row;
        act |= (__ballot(nb[j] >= 0) != 0ull) ? (1u << j) : 0u;
    }

    float a0[KS], a1[KS];
    f32x4 d0[NT], d1[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { d0[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; d1[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    auto fetch_a = [&](int j, float (&af)[KS]) {
        if (nb[j] >= 0) load_vec<KS>(x + (size_t)nb[j] * CIN + q * KS, af);
        else {
#pragma unroll
            for (int i = 0; i < KS; ++i) af[i] = 0.f;
        }
    };
    auto dma_b = [&](int j, float *buf) {           // the S offset images of iteration j
        if constexpr (BLDS) {
#pragma unroll
            for (int i = 0; i < (S * NWL + NW - 1) / NW; ++i) {
                const int t = wave + NW * i;                             // wave-load id: (offset slot, nt, piece)
                if (t < S * NWL) {
                    const int so = t / NWL, r = t - so * NWL;
                    const int tnt = r / J, jj = r - tnt * J;
                    const int k = j * S + so;
                    if (k < KK) {
                        const float *src = wp + (((size_t)k * (CIN / 4) + (lane >> 4) * J + jj) * COUT + tnt * 16 + (lane & 15)) * 4;
                        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(buf + t * 256), 16, 0, 0);
                    }
                }
            }
        }
    };
    auto mma = [&](int j, const float (&af)[KS], const float *buf) {
        const int k = j * S + sp;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float bf[KS];
            if constexpr (BLDS) {
#pragma unroll
                for (int jj = 0; jj < J; ++jj) {
                    const float4 v = *(const float4 *)(buf + sp * BOFF + ((t * J + jj) * 64 + lane) * 4);
                    bf[4 * jj] = v.x; bf[4 * jj + 1] = v.y; bf[4 * jj + 2] = v.z; bf[4 * jj + 3] = v.w;
                }
            } else {
                static_assert(BLDS || KS == 1, "plain weight loads serve the 4-channel input layer only");
                bf[0] = wp[((size_t)k * COUT + t * 16 + m16) * 4 + q];
            }
#pragma unroll
            for (int kk = 0; kk < KS; kk += 2) {
                d0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk], bf[kk], d0[t], 0, 0, 0);
                if (kk + 1 < KS) d1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk + 1], bf[kk + 1], d1[t], 0, 0, 0);
            }
        }
    };

    const int niter = (KK + S - 1) / S;
    dma_b(0, bs);
    if (act & 1u) fetch_a(0, a0);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kIter; j += 2) {
        if (j < niter) {
            // even iteration j: operands in (a0, bs[0]); odd iteration j+1: (a1, bs[BBUF])
            if (j + 1 < niter) { dma_b(j + 1, bs + BBUF); if ((act >> (j + 1)) & 1u) fetch_a(j + 1, a1); }
            if ((act >> j) & 1u) mma(j, a0, bs);
            __syncthreads();
        }
        if (j + 1 < kIter && j + 1 < niter) {
            if (j + 2 < niter) { dma_b(j + 2, bs); if ((act >> (j + 2)) & 1u) fetch_a(j + 2, a0); }
            if ((act >> (j + 1)) & 1u) mma(j + 1, a1, bs + BBUF);
            __syncthreads();
        }
    }

    // ---- combine the S partial sums of a row group (fixed order), then the epilogue ------------------
    if (sp > 0) {
        float *dst = bs + (((rg * (S - 1) + (sp - 1)) * 64 + lane) * NT) * 4;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 v = d0[t] + d1[t];
            *(float4 *)(dst + t * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    __syncthreads();
    if (sp == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 v = d0[t] + d1[t];
#pragma unroll
            for (int o = 0; o < S - 1; ++o) {
                const float4 p = *(const float4 *)(bs + (((rg * (S - 1) + o) * 64 + lane) * NT + t) * 4);
                v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
            }
            const int co = t * 16 + m16;
            const float sc = scale ? scale[co] : 1.f, sh = shift ? shift[co] : 0.f;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = rb + q * 4 + reg;         // D[row = q*4 + reg][col = m16]
                if (r < n) {
                    float o2 = v[reg] * sc + sh;
                    if (relu) o2 = fmaxf(o2, 0.f);
                    y[(size_t)r * COUT + co] = o2;
                }
            }
        }
    }
    (void)dbg; (void)K;
}

template <int CIN, int COUT>
int launch_fwd(const float *x, const int32_t *nbr, const int32_t *n_ptr, int cap, const float *wp, int K,
               const float *scale, const float *shift, int relu, float *y, int dbg, hipStream_t stream)
{
    using S = SpShape<CIN, COUT>;
    hipLaunchKernelGGL((spconv_fwd_kernel<CIN, COUT>), dim3(cdiv(cap, 64)), dim3(4 * kSplit * 64), 0, stream, x, nbr,
                       n_ptr, cap, wp, K, scale, shift, relu, y, dbg);
    return sassd_launch_status();
}

template <int CIN, int COUT>
int launch_pack(const float *w, int K, float *packed, int transposed, hipStream_t stream)
{
    const int total = K * CIN * COUT;
    hipLaunchKernelGGL((pack_weight_kernel<CIN, COUT>), dim3(cdiv(total, 256)), dim3(256), 0, stream, w, K, packed,
                       transposed);
    return sassd_launch_status();
}


// ------------------------------------------------------------------------------------------------------------------
// Gather - GEMM - scatter sparse conv (the production forward / data-gradient kernel for Cin >= 16).
//
// The register-stationary kernel above pads every (16-row tile, offset) to 16 rows: at KITTI sparsity only 40-60 %
// of its MFMA rows carry a rulebook pair (21-24 % on the strided layers) and it pays one workgroup barrier per three
// offsets.  Here a workgroup owns RW consecutive output rows and its waves pull whole kernel OFFSETS from an LDS
// ticket counter (heaviest offsets first: centre, faces, edges, corners):
//   compaction   the wave scans the offset's column of the workgroup's rulebook slice (staged once in LDS), ballots
//                the rows that have a pair and writes the (input row, local output row) list -- the M dimension of
//                the MFMA is filled with PAIRS: ceil(n_k/16) tiles instead of RW/16 (75-85 % full at RW = 64)
//   W operand    W[k] (Cin x Cout, pre-packed fragment order) lives in REGISTERS for the whole offset, loaded once
//                per (workgroup, offset) straight from L2 and double buffered: the next non-empty offset's weights are
//                in flight while the current offset's tiles run
//   X operand    pair rows gathered from HBM/L2 in fragment order, double buffered across tiles
//   accumulate   every wave owns a PRIVATE fp32 slab [RW][Cout] in LDS.  The MFMA is issued transposed
//                (D^T = W^T X^T: operands swapped), so a lane holds 4 consecutive output channels of ONE pair:
//                the tile's C input is one ds_read_b128 per 16 channels from the pair's output row, the result one
//                ds_write_b128 back -- no atomics (LDS float atomics measured ~2.6 cycles per LANE: a first version
//                with ds_add_f32 into a shared slab spent 70 % of its time there), no barrier, and a row occurs at
//                most once per offset, so lanes of one instruction never collide
//   epilogue     ONE barrier, then the NW slabs are summed in wave order (deterministic), scale / shift / ReLU and
//                coalesced 16-B row stores.
// Workgroup -> rows: runs of 8 consecutive row slices go to the same XCD (blockIdx % 8), so the gathers of
// neighbouring voxels hit that XCD's L2.
// ------------------------------------------------------------------------------------------------------------------
__constant__ int c_offset_order[kK] = {13, 4, 10, 12, 14, 16, 22, 1, 3, 5, 7, 9, 11, 15, 17, 19, 21, 23, 25,
                                       0, 2, 6, 8, 18, 20, 24, 26};

// Static offset -> wave assignment (slots of c_offset_order; longest-processing-time packing with weights centre 4, face
// 3, edge 2, corner 1.3 MFMA tiles per 64 rows): every output row's <= 27 contributions are then summed in a FIXED order
// (inside a wave: its slot list; across waves: slab 0, 1, ...), i.e. results are bit-reproducible run to run.  The
// ticket counter (dynamic assignment) balances better on skewed rulebooks but makes the fp32 summation order depend on
// timing.
__constant__ int c_assign8[8][5] = {{0, 15, 23, -1, -1}, {1, 9, 17, -1, -1}, {2, 10, 18, -1, -1}, {3, 11, 19, 25, -1},
                                    {4, 12, 20, 26, -1}, {5, 13, 21, -1, -1}, {6, 14, 22, -1, -1}, {7, 8, 16, 24, -1}};
__constant__ int c_assign4[4][8] = {{0, 7, 8, 12, 16, 21, 25, -1}, {1, 4, 9, 13, 17, 22, 26, -1},
                                    {2, 5, 10, 14, 18, 23, -1, -1}, {3, 6, 11, 15, 19, 20, 24, -1}};

template <int COUT, int RW, int NW, int CS>
constexpr size_t gs_lds_bytes() { return (size_t)(NW * RW * (COUT / CS) + RW * kK + NW * 2 * RW + 4) * 4; }

// CS = output-channel split: a wave accumulates COUT/CS channels (its slab, weight registers and MFMA count shrink by
// CS; waves of different channel groups take offsets from separate ticket counters and gather the same pair rows).
// WPS = waves per SIMD the register allocation must allow (workgroups per CU x NW / 4).
template <int CIN, int COUT, int RW, int NW, int CS, int WPS>
__global__ void __launch_bounds__(NW * 64, WPS)
spconv_gs_kernel(const float *__restrict__ x, const int32_t *__restrict__ nbr, const int32_t *__restrict__ n_ptr,
                 int cap, const float *__restrict__ wp, const float *__restrict__ scale,
                 const float *__restrict__ shift, int relu, float *__restrict__ y, int dbg)
{
    constexpr int KS = CIN / 4, NT = COUT / 16;
    constexpr int CW = COUT / CS, NTW = CW / 16;             // channels / 16-channel tiles per wave
    constexpr int NCH = CW / 4;                              // 16-byte chunks per slab row, XOR-swizzled by the row
    constexpr int NH = RW / 64;
    static_assert(NW % CS == 0 && NTW >= 1, "bad channel split");
    extern __shared__ __attribute__((aligned(16))) float gs_lds[];
    float *slabs = gs_lds;                                   // [NW][RW][CW]
    int *nbr_s = (int *)(gs_lds + NW * RW * CW);             // [RW][27]
    int *lists = nbr_s + RW * kK;                            // [NW][2][RW]
    int *next_slot = lists + NW * 2 * RW;                    // [CS]

    // Workgroup -> row slice: runs of 8 consecutive slices go to one XCD (blockIdx % 8), round-robin over the XCDs.  The
    // map depends on the capacity only, so the slice's rulebook rows are requested BEFORE the device row count is
    // known (one dependent memory round trip less on the launch's critical path; rows past the count are masked).
    const int g = (int)(blockIdx.x >> 3), xcd = (int)(blockIdx.x & 7);
    const int slice = (((g >> 3) << 3) + xcd) * 8 + (g & 7);
    const int r0 = slice * RW;
    const int rows_cap = min(RW, cap - r0);
    if (rows_cap <= 0) return;                               // workgroup-uniform, host-known
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, m16 = lane & 15;
    const int half = wave % CS;                              // this wave's channel group
    constexpr int NST = (RW * kK + NW * 64 - 1) / (NW * 64);
    int stage[NST];
#pragma unroll
    for (int j = 0; j < NST; ++j) {
        const int i = tid + j * NW * 64;
        stage[j] = (i < rows_cap * kK) ? nbr[(size_t)r0 * kK + i] : -1;
    }
    float b0[NTW][KS], b1[NTW][KS];
    float a0[KS], a1[KS];
    auto load_w = [&](int k, float (&b)[NTW][KS]) {             // fragment order, 16-B pieces, coalesced
#pragma unroll
        for (int u = 0; u < NTW; ++u)
#pragma unroll
            for (int k4 = 0; k4 < KS / 4; ++k4) {
                const float4 v = *(const float4 *)(wp + (((size_t)k * (CIN / 4) + q * (KS / 4) + k4) * COUT +
                                                         (half * NTW + u) * 16 + m16) * 4);
                b[u][4 * k4] = v.x; b[u][4 * k4 + 1] = v.y; b[u][4 * k4 + 2] = v.z; b[u][4 * k4 + 3] = v.w;
            }
    };
    // every wave starts on a fixed offset (the heaviest ones), so its weights are requested at kernel entry as well
    constexpr int WPH = NW / CS;                             // waves per channel group
    constexpr bool HAS_TABLE = (WPH == 8 || WPH == 4);
    const bool fixed = HAS_TABLE && !(dbg & 16);             // static assignment (deterministic) unless switched off
    const int wpos = wave / CS;
    int nxt = 1;                                             // next entry of this wave's static slot list
    int k = c_offset_order[wpos];
    load_w(k, b0);
    const int n = min(*n_ptr, cap);
    if (r0 >= n) return;                                     // workgroup-uniform
    const int rows = min(RW, n - r0);
#pragma unroll
    for (int j = 0; j < NST; ++j) {
        const int i = tid + j * NW * 64;
        if (i < RW * kK) nbr_s[i] = (i < rows * kK) ? stage[j] : -1;
    }
    for (int i = tid; i < NW * RW * CW / 4; i += NW * 64) ((float4 *)slabs)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < CS) next_slot[tid] = NW / CS;
    __syncthreads();

    float *slab = slabs + wave * RW * CW;
    int *lin = lists + wave * 2 * RW, *lout = lin + RW;

    // next kernel offset (heaviest first) that has at least one pair in this workgroup's rows, or -1
    auto grab = [&]() -> int {
        for (;;) {
            int slot = 0;
            if (fixed) {
                slot = (WPH == 8) ? c_assign8[wpos & 7][nxt < 4 ? nxt : 4] : c_assign4[wpos & 3][nxt < 7 ? nxt : 7];
                ++nxt;
                if (slot < 0) return -1;
            } else {
                if (lane == 0) slot = atomicAdd(next_slot + half, 1);
                slot = __builtin_amdgcn_readfirstlane(slot);
            }
            if (slot >= kK) return -1;
            const int k = c_offset_order[slot];
            bool any = false;
#pragma unroll
            for (int h = 0; h < NH; ++h) any = any || (__ballot(nbr_s[(h * 64 + lane) * kK + k] >= 0) != 0ull);
            if (any) return k;
        }
    };
    // pair rows of tile t; the padding of the last tile points at input row 0: an invalid pair only feeds ITS OWN column
    // of D^T, which is never written back, so no zero fill is needed
    auto fetch_a = [&](int t, float (&af)[KS]) {
        const int in = lin[t * 16 + m16];
        load_vec<KS>(x + (size_t)in * CIN + q * KS, af);
    };
    // one tile of 16 pairs: D^T[cout = u*16 + q*4 + r][pair = m16] += W[k]^T X^T, accumulated in the pair's slab row
    auto tile = [&](int t, int nk, const float (&af)[KS], const float (&b)[NTW][KS]) {
        const bool valid = t * 16 + m16 < nk;
        const int orow = lout[t * 16 + m16];
        float *row = slab + orow * CW;
        const int sw = orow & (NCH - 1);
        f32x4 d[NTW];
#pragma unroll
        for (int u = 0; u < NTW; ++u) {                      // padded pairs read slab row 0 and drop the result
            const float4 c = *(const float4 *)(row + (((u * 4 + q) ^ sw) << 2));
            d[u] = (f32x4){c.x, c.y, c.z, c.w};
        }
        if (!(dbg & 4)) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int u = 0; u < NTW; ++u) d[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[u][kk], af[kk], d[u], 0, 0, 0);
        }
        if (valid && !(dbg & 2)) {
#pragma unroll
            for (int u = 0; u < NTW; ++u)
                *(float4 *)(row + (((u * 4 + q) ^ sw) << 2)) = make_float4(d[u][0], d[u][1], d[u][2], d[u][3]);
        }
    };
    auto process = [&](int k, const float (&b)[NTW][KS]) {
        // ---- compaction of offset k over the workgroup's rows ----------------------------------------
        int nk = 0;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int r = h * 64 + lane;
            const int v = nbr_s[r * kK + k];
            const unsigned long long mk = __ballot(v >= 0);
            const int pos = nk + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0));
            if (v >= 0) { lin[pos] = v; lout[pos] = r; }
            nk += __popcll(mk);
        }
        nk = __builtin_amdgcn_readfirstlane(nk);
        const int ntile = (nk + 15) >> 4;
        if (lane < ntile * 16 - nk) { lin[nk + lane] = 0; lout[nk + lane] = 0; }      // pad the last tile
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own list writes precede its list reads
        // (round 4) the look-ahead gathers are unconditional -- past the end the last tile is requested again: under a
        // branch the number of loads in flight is path-dependent and the compiler then drains vmcnt(0), i.e. waits for
        // the gather it has just issued, in front of every tile
        fetch_a(0, a0);
        for (int t = 0; t < ntile; t += 2) {
            fetch_a(min(t + 1, ntile - 1), a1);
            tile(t, nk, a0, b);
            if (t + 1 < ntile) {
                fetch_a(min(t + 2, ntile - 1), a0);
                tile(t + 1, nk, a1, b);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // list / slab accesses retired before the lists are rewritten
    };

    {
        bool any = false;
#pragma unroll
        for (int h = 0; h < NH; ++h) any = any || (__ballot(nbr_s[(h * 64 + lane) * kK + k] >= 0) != 0ull);
        if (!any || wave / CS >= kK) {                       // the pre-assigned offset has no pair here: take a ticket
            k = grab();
            if (k >= 0) load_w(k, b0);
        }
    }
    while (k >= 0) {
        int kn = grab();
        load_w(kn >= 0 ? kn : k, b1);
        process(k, b0);
        k = kn;
        if (k < 0) break;
        kn = grab();
        load_w(kn >= 0 ? kn : k, b0);
        process(k, b1);
        k = kn;
    }
    __syncthreads();

    // ---- epilogue: sum the wave slabs in wave order, folded BatchNorm / bias / ReLU, 16-B row stores ----------
    constexpr int C4 = COUT / 4;
    for (int i = tid; i < rows * C4; i += NW * 64) {
        const int r = i / C4, c4 = i - r * C4;
        const int hh = c4 / NCH, ch = (c4 - hh * NCH) ^ (r & (NCH - 1));
        const float *src = slabs + (hh * RW + r) * CW + ch * 4;          // slab of wave hh, then hh + CS, ...
        float4 v = *(const float4 *)src;
#pragma unroll
        for (int w = 1; w < NW / CS; ++w) {
            const float4 p = *(const float4 *)(src + w * CS * RW * CW);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        const float4 sc = scale ? *(const float4 *)(scale + c4 * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 sh = shift ? *(const float4 *)(shift + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *(float4 *)(y + (size_t)(r0 + r) * COUT + c4 * 4) = v;
    }
}

#include "spconv_gq.h"


template <int CIN, int COUT, int RW, int NW, int CS, int WPS>
int launch_gs_cfg(const float *x, const int32_t *nbr, const int32_t *n_ptr, int cap, const float *wp,
                  const float *scale, const float *shift, int relu, float *y, int dbg, hipStream_t stream)
{
    constexpr size_t lds = gs_lds_bytes<COUT, RW, NW, CS>();
    static_assert(lds <= 160 * 1024, "workgroup slabs exceed the 160 KB LDS");
    static std::atomic<unsigned long long> attr_done{0};
    const void *fn = (const void *)spconv_gs_kernel<CIN, COUT, RW, NW, CS, WPS>;
    int rc = sassd_dyn_lds(fn, lds, attr_done);
    if (rc) return rc;
    const int grid = 64 * cdiv(cdiv(cap, RW), 64);
    hipLaunchKernelGGL((spconv_gs_kernel<CIN, COUT, RW, NW, CS, WPS>), dim3(grid), dim3(NW * 64), lds, stream, x, nbr,
                       n_ptr, cap, wp, scale, shift, relu, y, dbg & 31);
    return sassd_launch_status();
}

template <int CIN, int COUT>
int launch_gs(const float *x, const int32_t *nbr, const int32_t *n_ptr, int cap, const float *wp, int K,
              const float *scale, const float *shift, int relu, float *y, int cfg, hipStream_t stream)
{
    (void)K;
    const int geo = cfg >> 16, dbg = cfg & 0xFFFF;
    // forced geometries (cfg bits 16+): exactly the ones the default dispatch below picks by layer shape
    // and capacity, so that tests / tools can run each of them on any input size.  (Rounds 2-4 carried nine more -- 128-row
    // slices, channel-split wave pairs, 8-wave and consecutive-slice forms of the balanced kernel -- as measured-and-
    // rejected alternatives; removed in round 5, their timings are in profiles/r04_spconv_layers_*.txt.)
    //   1 = spconv_gs_kernel, 4 waves, two workgroups per CU      5 = spconv_gs_kernel, 8 waves (10 = 1 or 5 by capacity)
    //   8 = balanced kernel, 4x4x1 quads, 4 waves                 9 = balanced kernel, 16x16x4 tiles, 4 waves
    constexpr int Q = (COUT >= 32) ? 1 : 0;                  // (16-channel outputs have no quad form: 16x16x4 tile)
    if (geo == 1) return launch_gs_cfg<CIN, COUT, 64, 4, 1, 2>(x, nbr, n_ptr, cap, wp, scale, shift, relu, y, dbg, stream);
    if (geo == 5) return launch_gs_cfg<CIN, COUT, 64, 8, 1, 2>(x, nbr, n_ptr, cap, wp, scale, shift, relu, y, dbg, stream);
    if (geo == 8) return launch_gq_cfg<CIN, COUT, 4, 2, Q, 1>(x, nbr, n_ptr, cap, wp, scale, shift, relu, y, dbg, stream);
    if (geo == 9) return launch_gq_cfg<CIN, COUT, 4, 2, 0, 1>(x, nbr, n_ptr, cap, wp, scale, shift, relu, y, dbg, stream);
    if (geo != 0 && geo != 10) return SASSD_EINVAL;
    // default (round 4, measured per layer shape; profiles/r04_spconv_layers_*.txt).  The balanced kernel pays for its
    // cooperative compaction and pays off where a layer is long enough to be bound by its heaviest workgroup: the
    // 64 -> 64 layers.  KITTI-scale single frames (one round of workgroups; level capacity 40 k rows): 16x16x4 tiles on 4
    // waves, interleaved slices -- 22 / 31 us instead of 33 / 37 us for the 14.6 k / 13.3 k-row submanifold layers.
    // Batches of frames (training batch 2: 37 / 55 us instead of 48 / 65 us; multi_cfg batch 8: strided layers 91 instead
    // of 111 us at 106 k rows): 4x4x1 quads.  Waymo-scale levels (13-17 pairs per row: 16-pair tiles are full) and every
    // narrower layer stay on the round-3 geometry.
    if constexpr (CIN == 64 && COUT == 64) {
        if (geo != 10) {
            if (cap <= 65536) return launch_gq_cfg<CIN, COUT, 4, 2, 0, 1>(x, nbr, n_ptr, cap, wp, scale, shift, relu, y, dbg, stream);
            if (cap <= 400000) return launch_gq_cfg<CIN, COUT, 4, 2, 1, 1>(x, nbr, n_ptr, cap, wp, scale, shift, relu, y, dbg, stream);
        }
    }
    // round-3 geometry (cfg 10 forces it everywhere): KITTI-scale single frames (capacity <= 64 k rows) one 8-wave
    // workgroup per CU, larger batches / frames two 4-wave workgroups per CU
    if (cap > 65536) return launch_gs_cfg<CIN, COUT, 64, 4, 1, 2>(x, nbr, n_ptr, cap, wp, scale, shift, relu, y, dbg, stream);
    return launch_gs_cfg<CIN, COUT, 64, 8, 1, 2>(x, nbr, n_ptr, cap, wp, scale, shift, relu, y, dbg, stream);
}

// forward / data-gradient dispatch: gather-GEMM-scatter for 27-offset layers with Cin >= 16, the streaming kernel for the
// 1x1x1 layer, the row-per-16-threads kernel for the 4-channel input layer, the register-stationary kernel when the legacy
// switch (debug bit 8) is set
template <int CIN, int COUT>
int launch_conv(const float *x, const int32_t *nbr, const int32_t *n_ptr, int cap, const float *wp, int K,
                const float *scale, const float *shift, int relu, float *y, int cfg, hipStream_t stream)
{
    if constexpr (CIN >= 16) {
        if (nbr && !(cfg & 256)) return launch_gs<CIN, COUT>(x, nbr, n_ptr, cap, wp, K, scale, shift, relu, y, cfg, stream);
        if (!nbr && !(cfg & 256)) return launch_pw<CIN, COUT>(x, n_ptr, cap, wp, scale, shift, relu, y, stream);
    }
    if constexpr (CIN == 4) {
        if (nbr && !(cfg & 256)) return launch_c4<COUT>(x, nbr, n_ptr, cap, wp, scale, shift, relu, y, stream);
    }
    return launch_fwd<CIN, COUT>(x, nbr, n_ptr, cap, wp, K, scale, shift, relu, y, cfg & 0xFFFF, stream);
}

#define SP_DISPATCH(FN, ...)                                                     \
    if (Cin == 4 && Cout == 16) return FN<4, 16>(__VA_ARGS__);                   \
    if (Cin == 16 && Cout == 16) return FN<16, 16>(__VA_ARGS__);                 \
    if (Cin == 16 && Cout == 32) return FN<16, 32>(__VA_ARGS__);                 \
    if (Cin == 32 && Cout == 32) return FN<32, 32>(__VA_ARGS__);                 \
    if (Cin == 32 && Cout == 64) return FN<32, 64>(__VA_ARGS__);                 \
    if (Cin == 64 && Cout == 64) return FN<64, 64>(__VA_ARGS__);                 \
    if (Cin == 32 && Cout == 16) return FN<32, 16>(__VA_ARGS__);   /* data-gradient shapes */ \
    if (Cin == 64 && Cout == 32) return FN<64, 32>(__VA_ARGS__);                 \
    return SASSD_EINVAL;

__global__ void densify_kernel(const float *__restrict__ feats, const int32_t *__restrict__ idx,
                               const int32_t *__restrict__ n_ptr, int cap, int C, int D, int H, int W, int order,
                               float *__restrict__ out)
{
    const int n = min(*n_ptr, cap);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * C) return;
    // consecutive threads -> consecutive ROWS for a fixed channel: writes of neighbouring voxels (ascending x)
    // land in the same cache line of one channel plane
    const int c = t / n, row = t - c * n;
    const int4 p = ((const int4 *)idx)[row];
    const int ch = order ? (p.y * C + c) : (c * D + p.y);
    out[(((size_t)p.x * C * D + ch) * H + p.z) * W + p.w] = feats[(size_t)row * C + c];
}


// ------------------------------------------------------------------------------------------------------------------
// Backward (training; replaces spconv `indice_conv_backward_fp32`):
//   dX[i] = sum_{(i,o,k) in R} dY[o] . W[k]^T      -> the FORWARD kernel on the transposed gather table
//                                                     nbrT[i][k] = o  and transposed-packed weights
//   dW[k] = sum_{(i,o) in R_k} X[i]^T (x) dY[o]    -> spconv_wgrad_kernel below
// ------------------------------------------------------------------------------------------------------------------
__global__ void nbr_transpose_kernel(const int32_t *__restrict__ nbr, const int32_t *__restrict__ n_out_ptr, int cap_out,
                                     int cap_in, int32_t *__restrict__ nbrT)
{
    const int n = min(*n_out_ptr, cap_out);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * kK) return;
    const int o = t / kK, k = t - o * kK;
    const int i = nbr[t];
    if (i >= 0 && i < cap_in) nbrT[(size_t)i * kK + k] = o;      // each (i,k) has at most one o
}

constexpr int kWgRows = 128;       // output rows per weight-gradient workgroup (~2 workgroups per CU at 29k rows)
// rows per workgroup of the offset-per-wave kernel: 128 up to 64 k rows, then capacity / 512 (multiple of 64, <= 2048)
static inline int wgrad_rows_per_wg(int cap)
{
    int r = ((cap + 511) / 512 + 63) / 64 * 64;
    return r < kWgRows ? kWgRows : (r > 2048 ? 2048 : r);
}

// One wave = one 16x16 tile of dW[k] (ci tile x co tile) for all 27 offsets; the MFMA K dimension runs over rows:
//   D[ci][co] += sum_rows X[nbr[row][k]][ci] * dY[row][co]      (4 rows per v_mfma_f32_16x16x4_f32)
// Per-workgroup partial sums are written to `part` and reduced in fixed order by wgrad_reduce_kernel (deterministic).
// At most 8 waves (tiles) per workgroup so that every wave keeps its 108 accumulators + the 2 x 27 prefetched
// operands in <= 256 VGPRs; 64x64 layers use two workgroups (blockIdx.y) per row chunk.
template <int CIN, int COUT>
__global__ void __launch_bounds__((((CIN + 15) / 16) * (COUT / 16) < 8 ? ((CIN + 15) / 16) * (COUT / 16) : 8) * 64)
spconv_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy, const int32_t *__restrict__ nbr,
                    const int32_t *__restrict__ n_ptr, int cap, float *__restrict__ part)
{
    constexpr int MT = (CIN + 15) / 16, NTT = COUT / 16;
    constexpr int WPW = MT * NTT < 8 ? MT * NTT : 8;          // waves (16x16 tiles) per workgroup
    __shared__ int nbr_s[kWgRows * kK];
    const int n = min(*n_ptr, cap);
    const int r0 = blockIdx.x * kWgRows;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) + blockIdx.y * WPW;
    const int cit = wave / NTT, cot = wave - cit * NTT;
    const int q = lane >> 4, m16 = lane & 15;
    f32x4 acc[kK];
#pragma unroll
    for (int k = 0; k < kK; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (r0 < n) {                                   // workgroup-uniform
        const int rows = min(kWgRows, n - r0);
        for (int i = tid; i < rows * kK; i += WPW * 64) nbr_s[i] = nbr[(size_t)r0 * kK + i];
        __syncthreads();
        const int ci = cit * 16 + m16;
        // two-stage software pipeline over 4-row steps: the 27 gathered X values (and the dY value) of step s+1 are
        // in flight while the MFMAs of step s issue -- the gather latency, not the arithmetic, bounds this kernel
        float a_buf[2][kK], b_buf[2];
        unsigned long long live[2];
        auto fetch = [&](int s0, float *a, float &b, unsigned long long &lv) {
            const int rl = s0 + q;
            const bool rok = rl < rows;
            b = rok ? dy[(size_t)(r0 + rl) * COUT + cot * 16 + m16] : 0.f;
            lv = 0ull;
#pragma unroll
            for (int k = 0; k < kK; ++k) {
                const int in = rok ? nbr_s[rl * kK + k] : -1;
                const bool ok = in >= 0 && ci < CIN;
                const float v = x[ok ? (size_t)in * CIN + ci : 0];          // unconditional load + select
                a[k] = ok ? v : 0.f;
                if (__ballot(in >= 0) != 0ull) lv |= 1ull << k;             // wave-uniform: offset k has a pair
            }
        };
        fetch(0, a_buf[0], b_buf[0], live[0]);
        for (int s0 = 0; s0 < rows; s0 += 8) {
            if (s0 + 4 < rows) fetch(s0 + 4, a_buf[1], b_buf[1], live[1]);
#pragma unroll
            for (int k = 0; k < kK; ++k)
                if ((live[0] >> k) & 1ull) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_buf[0][k], b_buf[0], acc[k], 0, 0, 0);
            if (s0 + 4 >= rows) break;
            if (s0 + 8 < rows) fetch(s0 + 8, a_buf[0], b_buf[0], live[0]);
#pragma unroll
            for (int k = 0; k < kK; ++k)
                if ((live[1] >> k) & 1ull) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_buf[1][k], b_buf[1], acc[k], 0, 0, 0);
        }
    }
    // D[row = q*4 + reg][col = m16] -> dW[k][ci = cit*16 + q*4 + reg][co = cot*16 + m16]
    float *dst = part + (size_t)blockIdx.x * kK * CIN * COUT;
#pragma unroll
    for (int k = 0; k < kK; ++k)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int ci = cit * 16 + q * 4 + reg;
            if (ci < CIN) dst[((size_t)k * CIN + ci) * COUT + cot * 16 + m16] = acc[k][reg];
        }
}

// Second formulation of the same partial sums (default): ONE wave per kernel offset instead of one wave per 16x16 tile.
// The MFMA work of a weight gradient is tiny (~1 GFLOP per layer); what bounded the tile-per-wave kernel above was that
// every lane issued 27 four-byte gathers per 4-row step and that ~2/3 of its MFMA steps multiplied rows without a pair.
// Here a workgroup is still one 128-row chunk, but its 8 waves split the 27 OFFSETS (wave 0: the centre offset, which
// pairs every row of a submanifold layer; waves 1-7: three or four of the others -- about one "row chunk" of pairs each).
// A wave compacts the pairs of its offset with ballots (in / out row lists in LDS), then runs dense 16x16x4 MFMA steps
// over four pairs at a time for the whole Cin x Cout tile (<= 16 accumulator tiles): 64-byte row segments per load,
// two steps of operands in flight, no empty steps.  Partial sums land in the same [chunk][offset][Cin][Cout] layout.
template <int CIN, int COUT>
__global__ void __launch_bounds__(512) spconv_wgrad_offset_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                                  const int32_t *__restrict__ nbr,
                                                                  const int32_t *__restrict__ n_ptr, int cap,
                                                                  float *__restrict__ part, int wg_rows)
{
    constexpr int MT = (CIN + 15) / 16, NTT = COUT / 16;
    extern __shared__ int wg_lists[];                            // [8 waves][2][wg_rows]
    const int n = min(*n_ptr, cap);
    const int r0 = blockIdx.x * wg_rows;
    if (r0 >= n) return;                                        // workgroup-uniform
    const int rows = min(wg_rows, n - r0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, m16 = lane & 15;
    int *lin = wg_lists + wave * 2 * wg_rows, *lout = lin + wg_rows;
    float *dst = part + (size_t)blockIdx.x * kK * CIN * COUT;
    // offsets of this wave: wave 0 -> 13; wave w >= 1 -> the (w-1)-th, (w+6)-th, ... of the other 26
#pragma unroll 1
    for (int t = 0; t < (wave == 0 ? 1 : 4); ++t) {
        int k;
        if (wave == 0) {
            k = 13;
        } else {
            const int j = (wave - 1) + 7 * t;                   // index among the 26 non-centre offsets
            if (j >= 26) break;
            k = j < 13 ? j : j + 1;
        }
        // ---- compact the (in, out) pairs of offset k over the chunk's rows (ascending row order: deterministic)
        int cnt = 0;
        for (int rb = 0; rb < rows; rb += 64) {
            const int rl = rb + lane;
            const int in = rl < rows ? nbr[(size_t)(r0 + rl) * kK + k] : -1;
            const unsigned long long m = __ballot(in >= 0);
            if (in >= 0) {
                const int pos = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                lin[pos] = in;
                lout[pos] = r0 + rl;
            }
            cnt += __popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
        f32x4 acc[MT][NTT];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NTT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // ---- dense MFMA steps, 4 pairs each; the operands of step s+1 are in flight during the MFMAs of step s
        float av[2][MT], bv[2][NTT];
        auto fetch = [&](int s, float *a, float *b) {
            const int pidx = 4 * s + q;
            const bool ok = pidx < cnt;
            const int in = ok ? lin[pidx] : 0, out = ok ? lout[pidx] : 0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int ci = mt * 16 + m16;
                const bool cok = ok && ci < CIN;
                const float v = x[cok ? (size_t)in * CIN + ci : 0];
                a[mt] = cok ? v : 0.f;
            }
#pragma unroll
            for (int nt = 0; nt < NTT; ++nt) {
                const float v = dy[ok ? (size_t)out * COUT + nt * 16 + m16 : 0];
                b[nt] = ok ? v : 0.f;
            }
        };
        const int nsteps = (cnt + 3) >> 2;
        // (round 4) the look-ahead fetch is UNCONDITIONAL -- past the end the last step is requested again: a load under a
        // branch makes the number of loads in flight path-dependent and the compiler answered with s_waitcnt vmcnt(0)
        // after every group of loads (no step's operands were ever in flight during the previous step's MFMAs)
        if (nsteps > 0) {
            fetch(0, av[0], bv[0]);
#pragma unroll 1
            for (int s = 0; s < nsteps; s += 2) {
                fetch(min(s + 1, nsteps - 1), av[1], bv[1]);
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NTT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][a], bv[0][b], acc[a][b], 0, 0, 0);
                if (s + 1 >= nsteps) break;
                fetch(min(s + 2, nsteps - 1), av[0], bv[0]);
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NTT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][a], bv[1][b], acc[a][b], 0, 0, 0);
            }
        }
        // D[row = q*4 + reg][col = m16] -> dW[k][ci = a*16 + q*4 + reg][co = b*16 + m16]
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NTT; ++b)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int ci = a * 16 + q * 4 + reg;
                    if (ci < CIN) dst[((size_t)k * CIN + ci) * COUT + b * 16 + m16] = acc[a][b][reg];
                }
        __builtin_amdgcn_wave_barrier();                       // the lists are rewritten for the next offset
    }
}

// dW = sum of the per-chunk partials, fixed order.  The sum over the chunks is a chain of dependent memory round trips
// (228 chunks at 29 k rows), the same length for a 16 x 16 and a 64 x 64 layer: 8 threads per element take every 8th chunk
// (two accumulators each), lanes 0-31 / 32-63 of a wave read 32 consecutive elements of ONE chunk (coalesced), and the
// eight sums meet in LDS.
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float *__restrict__ part, const int32_t *__restrict__ n_ptr,
                                                           int cap, int per, float *__restrict__ dw, int accumulate,
                                                           int wg_rows)
{
    __shared__ float red[8][32];
    const int n = min(*n_ptr, cap);
    const int nwg = (n + wg_rows - 1) / wg_rows;
    const int e = threadIdx.x & 31, pt = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + e;
    float s0 = 0.f, s1 = 0.f;
    if (i < per) {
        int g = pt;
        for (; g + 8 < nwg; g += 16) {
            s0 += part[(size_t)g * per + i];
            s1 += part[(size_t)(g + 8) * per + i];
        }
        if (g < nwg) s0 += part[(size_t)g * per + i];
    }
    red[pt][e] = s0 + s1;
    __syncthreads();
    if (pt == 0 && i < per) {
        float s = accumulate ? dw[i] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += red[j][e];
        dw[i] = s;
    }
}

template <int CIN, int COUT>
int launch_wgrad(const float *x, const float *dy, const int32_t *nbr, const int32_t *n_ptr, int cap, float *part,
                 float *dw, int accumulate, int cfg, hipStream_t stream)
{
    constexpr int MT = (CIN + 15) / 16, NTT = COUT / 16;
    constexpr int WPW = MT * NTT < 8 ? MT * NTT : 8;
    int wg_rows = kWgRows;
    if (cfg & 32) {    // cfg bit 5: the tile-per-wave formulation
        hipLaunchKernelGGL((spconv_wgrad_kernel<CIN, COUT>), dim3(cdiv(cap, kWgRows), MT * NTT / WPW), dim3(WPW * 64), 0,
                           stream, x, dy, nbr, n_ptr, cap, part);
    } else {
        // every workgroup leaves one [27][Cin][Cout] partial (442 KB at 64 x 64): beyond ~512 workgroups the partials
        // (1.1 GB per layer at Waymo scale, 317 k rows) cost more than the gradient itself, so the rows per workgroup grow
        // with the capacity (the pair lists live in dynamic LDS: 64 bytes per row)
        wg_rows = wgrad_rows_per_wg(cap);
        size_t lds = (size_t)8 * 2 * wg_rows * sizeof(int);
        static std::atomic<unsigned long long> attr_done{0};
        // the opt-in is made once per device with the largest size any capacity can ask for (128 KB); a device that
        // refuses it still runs every capacity whose lists fit the 64 KB no kernel has to ask for
        // (a refusal leaves HIP's sticky last-error set -- cleared here, or sassd_launch_status() would report
        // the kernel that then launches fine as failed -- and is remembered per device, so it is asked once, not per call)
        static std::atomic<unsigned long long> attr_refused{0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return SASSD_EHIP;
        const unsigned long long dbit = 1ull << (dev & 63);
        int rc = SASSD_EHIP;
        if (!(attr_refused.load(std::memory_order_acquire) & dbit)) {
            rc = sassd_dyn_lds((const void *)spconv_wgrad_offset_kernel<CIN, COUT>, (size_t)8 * 2 * 2048 * sizeof(int), attr_done);
            if (rc) {
                (void)hipGetLastError();
                attr_refused.fetch_or(dbit, std::memory_order_release);
            }
        }
        if (rc && lds > 64 * 1024) return rc;
        hipLaunchKernelGGL((spconv_wgrad_offset_kernel<CIN, COUT>), dim3(cdiv(cap, wg_rows)), dim3(512), lds, stream, x, dy,
                           nbr, n_ptr, cap, part, wg_rows);
    }
    const int per = kK * CIN * COUT;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(per, 32)), dim3(256), 0, stream, (const float *)part, n_ptr, cap,
                       per, dw, accumulate, wg_rows);
    return sassd_launch_status();
}

}  // namespace

extern "C" int sassd_rulebook_transpose(const int32_t *nbr, const int32_t *n_out_ptr, int cap_out, int32_t *nbrT,
                                        int cap_in, void *stream_)
{
    if (!nbr || !n_out_ptr || !nbrT || cap_out <= 0 || cap_in <= 0) return SASSD_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    int rc;
    if ((rc = sassd_hip(hipMemsetAsync(nbrT, 0xFF, (size_t)cap_in * kK * sizeof(int32_t), stream)))) return rc;
    hipLaunchKernelGGL(nbr_transpose_kernel, dim3(cdiv(cap_out * kK, 256)), dim3(256), 0, stream, nbr, n_out_ptr,
                       cap_out, cap_in, nbrT);
    return sassd_launch_status();
}

extern "C" int sassd_spconv_pack_weight_t(const float *w, int K, int Cin, int Cout, float *packed, void *stream_)
{
    // w is the forward weight [K, Cin, Cout]; the packed image drives the Cout -> Cin data-gradient conv
    if (!w || !packed || K < 1 || K > kK) return SASSD_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    { const int t = Cin; Cin = Cout; Cout = t; }
    SP_DISPATCH(launch_pack, w, K, packed, 1, stream)
}

extern "C" int sassd_spconv_bwd_data(const float *dy, const int32_t *nbrT, const int32_t *n_in_ptr, int cap_in,
                                     const float *wT_packed, int K, int Cin, int Cout, float *dx, int cfg, void *stream_)
{
    // dx [cap_in, Cin] = sum_k dy[nbrT[i,k]] @ W[k]^T : the forward kernel with (Cin', Cout') = (Cout, Cin)
    if (!dy || !n_in_ptr || !wT_packed || !dx || cap_in <= 0) return SASSD_EINVAL;
    if (nbrT ? (K != kK) : (K != 1)) return SASSD_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    { const int t = Cin; Cin = Cout; Cout = t; }
    SP_DISPATCH(launch_conv, dy, nbrT, n_in_ptr, cap_in, wT_packed, K, nullptr, nullptr, 0, dx, cfg, stream)
}

extern "C" size_t sassd_spconv_bwd_weight_workspace_bytes(int cap_out, int K, int Cin, int Cout)
{
    return (size_t)cdiv(cap_out > 0 ? cap_out : 1, kWgRows) * K * Cin * Cout * sizeof(float);      // (sized for 128-row groups)
}

extern "C" int sassd_spconv_bwd_weight(const float *x, const float *dy, const int32_t *nbr, const int32_t *n_out_ptr,
                                       int cap_out, int K, int Cin, int Cout, float *dw, int accumulate, int cfg,
                                       void *workspace, size_t workspace_bytes, void *stream_)
{
    if (!x || !dy || !nbr || !n_out_ptr || !dw || !workspace || cap_out <= 0 || K != kK) return SASSD_EINVAL;
    if (workspace_bytes < sassd_spconv_bwd_weight_workspace_bytes(cap_out, K, Cin, Cout)) return SASSD_ENOSPC;
    hipStream_t stream = (hipStream_t)stream_;
    float *part = (float *)workspace;
    SP_DISPATCH(launch_wgrad, x, dy, nbr, n_out_ptr, cap_out, part, dw, accumulate, cfg, stream)
}

namespace {
}  // namespace

// `cfg` of sassd_spconv_fwd / _bwd_data / _bwd_weight (0 in production; tools/ablate_spconv.py, tests): bit2 no MFMA, bit5
// (forward) every gather reads row 0 / (weight gradient) the tile-per-wave formulation, bit6 one weight image for every offset,
// bit8 the legacy register-stationary kernel; bits 16.. force a workgroup geometry (launch_gs).  Per call: no process-wide
// switch selects a kernel (ADVICE r04 / VERDICT r05 item 8).

extern "C" size_t sassd_spconv_packed_floats(int K, int Cin, int Cout) { return (size_t)K * Cin * Cout; }

extern "C" int sassd_spconv_pack_weight(const float *w, int K, int Cin, int Cout, float *packed, void *stream_)
{
    if (!w || !packed || K < 1 || K > kK) return SASSD_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    SP_DISPATCH(launch_pack, w, K, packed, 0, stream)
}

extern "C" int sassd_spconv_fwd(const float *x, const int32_t *nbr, const int32_t *n_out_ptr, int cap_out,
                                const float *w_packed, int K, int Cin, int Cout, const float *scale,
                                const float *shift, int relu, float *y, int cfg, void *stream_)
{
    if (!x || !n_out_ptr || !w_packed || !y || cap_out <= 0) return SASSD_EINVAL;
    if (nbr ? (K != kK) : (K != 1)) return SASSD_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    SP_DISPATCH(launch_conv, x, nbr, n_out_ptr, cap_out, w_packed, K, scale, shift, relu, y, cfg, stream)
}

extern "C" int sassd_densify(const float *feats, const int32_t *indices, const int32_t *n_ptr, int cap, int C,
                             int D, int H, int W, int batch_size, int channel_order, float *out, void *stream_)
{
    if (!feats || !indices || !n_ptr || !out || cap <= 0) return SASSD_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    int rc;
    // (a fill KERNEL where the alignment allows -- every map of the pipeline: hipMemsetAsync becomes a memset node in a captured
    // frame graph, the one node type of the frame that is not a kernel; see profiles/r06_late_experiments.txt, experiment 9)
    const size_t obytes = (size_t)batch_size * C * D * H * W * sizeof(float);
    if ((((uintptr_t)out | obytes) & 15) == 0) {
        if ((rc = sassd_fill2(out, obytes, 0, out, 0, 0, stream))) return rc;
    } else if ((rc = sassd_hip(hipMemsetAsync(out, 0, obytes, stream)))) return rc;
    hipLaunchKernelGGL(densify_kernel, dim3(cdiv(cap * C, 256)), dim3(256), 0, stream, feats, indices, n_ptr, cap, C, D,
                       H, W, channel_order, out);
    return sassd_launch_status();
}
