// spconv_gq.h -- balanced gather-GEMM-scatter sparse conv (round 4).  Included by spconv.hip inside its namespace.
//
// Same contract as spconv_gs_kernel (spconv v1.0 `indice_conv_fp32` + BatchNorm1d + ReLU of one layer,
// mmdet/models/necks/cmn.py:138-173,197-212, in ONE launch; wave-private fp32 LDS slabs, no atomics, results
// bit-reproducible run to run).  What changed, and why (profiles/r03_stall_breakdown.json: MFMA pipe 1/3 busy,
// 48 % of the wave-cycles parked; the critical-path model in tools/spconv_balance_model.py):
//
// the round-2/3 kernel runs ONE workgroup per CU in ONE round at KITTI scale, so a layer takes as long as its
// slowest workgroup -- and a workgroup as long as its slowest SIMD.  On a K21 frame the heaviest 64-row slice of a
// 64-channel submanifold layer carries 1.4-2.0x the mean pair count, only 224-233 of the 256 CUs get a slice, whole
// kernel offsets were dealt to waves by a static table (slowest SIMD 1.2x the mean), and a 16-pair MFMA tile was
// 50-84 % full.  Three changes, each deterministic:
//   (B) rows -> workgroups   the rows are cut into 8*m BLOCKS of <= 2048 consecutive rows (block j -> XCD j % 8, so a
//       block's gathers meet in one L2) and a block into 32 INTERLEAVED slices: slice s owns rows s, s+32, s+64, ...
//       of its block (<= 64).  Every slice of a block sees the block's average density, all 256 CUs get a slice at
//       KITTI scale (14.9 k rows: 58-59 rows each instead of 233 x 64).
//   (A) offsets -> waves     the workgroup's waves first compact ALL 27 offsets cooperatively (packed (input row,
//       local output row) lists + counts in LDS, one barrier); the units of work (UNIT pairs of one offset, heaviest
//       offset first) then form one list which is cut into NW equal contiguous ranges -- a wave's share differs from
//       another's by at most one unit, an offset may be split between two waves (each accumulates into its own
//       slab), and since every list is ready before the first tile, a wave prefetches its gathers ACROSS offsets.
//       The partition depends on the pair counts only: the fp32 summation order is a function of the rulebook.
//   (C) tile granularity     QUAD = 1: `v_mfma_f32_4x4x1_16b_f32` with the A operand (4 pairs x 1 input channel)
//       of ONE block broadcast to all 16 blocks (CBSZ / ABID) and B = W[k][cin][4 output channels per block]: one
//       instruction is 4 pairs x 64 output channels x 1 input channel (COUT = 32: CBSZ = 3, 2 x 4 pairs x 32
//       channels) at the same 64 FLOP/clk/SIMD as the 16x16x4 form, so work is issued per 4 (8) pairs: 89-96 % of the
//       issued rows are pairs.  The gathered tile keeps the (pair, channel quarter) register layout of the 16x16x4
//       path; D comes out as [pair i = VGPR][output channel = lane], i.e. a slab row is one conflict-free
//       ds_read_b32 / ds_write_b32 per pair, no swizzle.  QUAD = 0 keeps the transposed 16x16x4 tile of
//       spconv_gs_kernel (XOR-swizzled 16-byte slab chunks) on the new work distribution.
// Weights: W[k] lives in registers for a whole offset (CIN VGPRs), double buffered against the wave's next offset;
// pack layout [K][CIN/4][COUT][4] (every kernel of this file reads it).
// LDS: NW slabs [64][COUT] f32 + rulebook slice [64][27] + lists [27][64] + counts: 142 KB (COUT 64, 8 waves; one
// workgroup per CU), 78 KB (COUT 32: two per CU).

template <int COUT, int NW>
constexpr size_t gq_lds_bytes() { return (size_t)(NW * 64 * COUT + 64 * kK + kK * 64 + 32) * 4; }

static inline int gq_grid(int cap, int ilv) { return ilv ? 256 * (cap <= 16384 ? 1 : cdiv(cap, 16384)) : 64 * cdiv(cdiv(cap, 64), 64); }

// ILV = 1: interleaved slices of XCD-local blocks (change B above; KITTI-scale single frames, where a layer is ONE round of
// workgroups and lasts as long as its heaviest one).  ILV = 0: 64 consecutive rows per workgroup in the XCD-aware order
// of spconv_gs_kernel -- many rounds of workgroups per CU balance themselves, and consecutive rows share their gathered
// neighbours in the CU's vector cache.
// (Rounds 3-4 carried an opt-in bf16 form of the 16x16 tile -- operands rounded in registers onto v_mfma_f32_16x16x32_bf16,
// +4.6 % training samples/s -- behind a process-wide switch.  Removed in round 5: it doubled the gradient noise of the bench
// workload, failed its own whole-step bars, covered only the <= 64 k-row dispatch of the 64 -> 64 layers, and a process-wide
// arithmetic switch is the wrong shape for an ABI.  DESIGN.md appendix A keeps the measurements.)
template <int CIN, int COUT, int NW, int WPS, int QUAD, int ILV>
__global__ void __launch_bounds__(NW * 64, WPS)
spconv_gq_kernel(const float *__restrict__ x, const int32_t *__restrict__ nbr, const int32_t *__restrict__ n_ptr,
                 int cap, const float *__restrict__ wp, const float *__restrict__ scale,
                 const float *__restrict__ shift, int relu, float *__restrict__ y, int dbg)
{
    static_assert(!QUAD || COUT == 64 || COUT == 32, "the 4x4x1 path covers 32 / 64 output channels");
    constexpr int RW = 64, T = NW * 64;
    constexpr int KSEG = CIN / 4;                            // input channels per lane of a gathered tile
    constexpr int NH2 = QUAD ? 64 / COUT : 1;                // pair quads one 4x4x1 instruction covers (1 or 2)
    constexpr int PPS = 4 * NH2;                             // pairs per instruction set
    constexpr int NS = 16 / PPS;                             // instruction sets per 16-pair tile
    constexpr int BG = 16 / NH2;                             // MFMA blocks that share one broadcast A block
    constexpr int CBSZ = (BG == 16) ? 4 : 3;
    constexpr int UNIT = QUAD ? PPS : 16;                    // pairs per unit of the balanced partition
    constexpr int UPT = 16 / UNIT;                           // units per tile
    constexpr int NTW = COUT / 16, NCH = COUT / 4;           // 16x16x4 path: channel tiles, 16-byte chunks per slab row
    extern __shared__ __attribute__((aligned(16))) float gq_lds[];
    float *slabs = gq_lds;                                   // [NW][RW][COUT]
    int *nbr_s = (int *)(gq_lds + NW * RW * COUT);           // [RW][27]
    int *lists = nbr_s + RW * kK;                            // [27][RW]  (input row << 6 | local output row)
    int *cnt = lists + kK * RW;                              // [27]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- workgroup -> (block, interleaved slice) -------------------------------------------------------------------
    constexpr int RS = ILV ? 32 : 1;                         // row stride of a slice: local row r <-> row0 + RS * r
    const int xcd = (int)(blockIdx.x & 7), t_ = (int)(blockIdx.x >> 3);
    const int n = min(*n_ptr, cap);
    int rows, row0;
    if constexpr (ILV) {
        const int sl = t_ & 31, j8 = t_ >> 5;
        const int nb8 = n <= 16384 ? 1 : (n + 16383) / 16384;
        if (j8 >= nb8) return;                               // workgroup-uniform
        const int bs = (n + 8 * nb8 - 1) / (8 * nb8);        // rows per block, <= 2048
        const int base = (j8 * 8 + xcd) * bs;
        const int brows = min(bs, n - base);
        if (brows <= sl) return;
        rows = (brows - sl + 31) >> 5;                       // <= 64
        row0 = base + sl;
    } else {
        const int slice = (((t_ >> 3) << 3) + xcd) * 8 + (t_ & 7);   // runs of 8 consecutive slices on one XCD
        row0 = slice * RW;
        if (row0 >= n) return;
        rows = min(RW, n - row0);
    }
    // the slice's rulebook rows are requested first; the slabs are cleared while they are on their way
    constexpr int NST = (RW * kK + T - 1) / T;
    int stage[NST];
#pragma unroll
    for (int j = 0; j < NST; ++j) {
        const int i = tid + j * T;
        const int r = i / kK, kk = i - r * kK;
        stage[j] = (r < rows) ? nbr[(size_t)(row0 + RS * r) * kK + kk] : -1;
    }
    for (int i = tid; i < NW * RW * COUT / 4; i += T) ((float4 *)slabs)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NST; ++j) {
        const int i = tid + j * T;
        if (i < RW * kK) nbr_s[i] = stage[j];
    }
    __syncthreads();
    // ---- cooperative compaction of the 27 offsets ---------------------------------------------------------------
    for (int k = wave; k < kK; k += NW) {
        const int v = nbr_s[lane * kK + k];
        const unsigned long long mk = __ballot(v >= 0);
        const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0));
        if (v >= 0) lists[k * RW + pos] = (v << 6) | lane;
        if (lane == 0) cnt[k] = __popcll(mk);
    }
    __syncthreads();

    // ---- balanced partition: units of UNIT pairs, offsets heaviest first, NW equal contiguous ranges -------------
    const int k_l = c_offset_order[lane < kK ? lane : 0];             // lane = slot (heaviest offsets first)
    const int c_l = (lane < kK) ? cnt[k_l] : 0;
    const int u_l = (c_l + UNIT - 1) / UNIT;
    int incl = u_l;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const int total = __builtin_amdgcn_readlane(incl, kK - 1);
    const int lo = (int)(((long long)wave * total) / NW), hi = (int)(((long long)(wave + 1) * total) / NW);

    float *slab = slabs + wave * RW * COUT;
    // lane -> (pair of the tile, input-channel quarter): block b = lane / 4 = h * BG + a, a = G * NS + set
    const int blk = lane >> 2;
    const int a_ = blk % BG, h_ = blk / BG;
    const int G_l = QUAD ? a_ / NS : lane >> 4;
    const int pl = QUAD ? PPS * (a_ % NS) + 4 * h_ + (lane & 3) : (lane & 15);
    const int cl = lane % COUT;                              // QUAD: this lane's output channel
    const int hq = QUAD ? lane / COUT : 0;                   // QUAD: which quad of an instruction set this lane's D rows are

    struct Tile { int slot, k, pb, np; };
    auto locate = [&](int g, Tile &tl) -> int {              // tile starting at unit g (< hi); returns units taken
        const unsigned long long m = __ballot(incl > g);
        const int slot = __builtin_ctzll(m);
        const int cs = __builtin_amdgcn_readlane(c_l, slot), is = __builtin_amdgcn_readlane(incl, slot);
        const int us = __builtin_amdgcn_readlane(u_l, slot);
        const int uo = g - (is - us);
        const int tu = min(UPT, min(us - uo, hi - g));
        tl.slot = slot;
        tl.k = __builtin_amdgcn_readlane(k_l, slot);
        tl.pb = uo * UNIT;
        tl.np = min(tu * UNIT, cs - tl.pb);
        return tu;
    };
    // first unit of the first non-empty slot after `slot` if it still belongs to this wave, else -1
    auto next_slot_k = [&](int slot) -> int {
        const int is = __builtin_amdgcn_readlane(incl, slot);
        if (is >= hi) return -1;
        const unsigned long long m = __ballot(incl > is);
        return __builtin_amdgcn_readlane(k_l, __builtin_ctzll(m));
    };
    auto fetch_a = [&](const Tile &tl, float (&af)[KSEG]) {
        int e = lists[tl.k * RW + tl.pb + (pl < tl.np ? pl : 0)];
        if (dbg & 32) e &= 63;                               // ablation: every gather reads row 0 (same loads, cache hits)
        load_vec<KSEG>(x + (size_t)(e >> 6) * CIN + G_l * KSEG, af);
    };
    constexpr int NB = QUAD ? CIN : NTW * KSEG;              // weight registers of one offset
    auto load_w = [&](int k, float (&b)[NB]) {
        if (dbg & 64) k = 0;                                 // ablation: one weight image for every offset
        const float *wk = wp + (size_t)k * CIN * COUT;
        if constexpr (QUAD) {                                // b[c] = W[k][c][cl]
#pragma unroll
            for (int c4 = 0; c4 < CIN / 4; ++c4) {
                const float4 v = *(const float4 *)(wk + ((size_t)c4 * COUT + cl) * 4);
                b[4 * c4] = v.x; b[4 * c4 + 1] = v.y; b[4 * c4 + 2] = v.z; b[4 * c4 + 3] = v.w;
            }
        } else {                                             // b[u*KSEG + kk] = W[k][q*KSEG + kk][u*16 + m16]
            const int q = lane >> 4, m16 = lane & 15;
#pragma unroll
            for (int u = 0; u < NTW; ++u)
#pragma unroll
                for (int k4 = 0; k4 < KSEG / 4; ++k4) {
                    const float4 v = *(const float4 *)(wk + ((size_t)(q * (KSEG / 4) + k4) * COUT + u * 16 + m16) * 4);
                    b[u * KSEG + 4 * k4] = v.x; b[u * KSEG + 4 * k4 + 1] = v.y;
                    b[u * KSEG + 4 * k4 + 2] = v.z; b[u * KSEG + 4 * k4 + 3] = v.w;
                }
        }
    };
    auto tile = [&](const Tile &tl, const float (&af)[KSEG], const float (&b)[NB]) {
        const int *lst = lists + tl.k * RW + tl.pb;
        if constexpr (QUAD) {
            // one instruction set = PPS pairs x COUT channels x CIN: four accumulator chains (one per input-channel
            // quarter), started from zero and added to the slab rows at the end, so the LDS reads hide behind the MFMAs
            auto qset = [&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s < NS) {
                    if (s * PPS < tl.np) {                              // wave-uniform
                        float *rowp[4];                                 // this lane's D rows: pairs s*PPS + 4*hq + i
                        bool ok[4];
                        float c0[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int p = s * PPS + 4 * hq + i;
                            ok[i] = p < tl.np;
                            const int e = lst[ok[i] ? p : 0];
                            rowp[i] = slab + (e & 63) * COUT + cl;
                            c0[i] = *rowp[i];
                        }
                        f32x4 d0 = (f32x4){0.f, 0.f, 0.f, 0.f}, d1 = d0, d2 = d0, d3 = d0;
                        if (!(dbg & 4)) {                               // ablation: no MFMAs
#pragma unroll
                            for (int kk = 0; kk < KSEG; ++kk) {
                                d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(af[kk], b[0 * KSEG + kk], d0, CBSZ, 0 * NS + s, 0);
                                d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(af[kk], b[1 * KSEG + kk], d1, CBSZ, 1 * NS + s, 0);
                                d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(af[kk], b[2 * KSEG + kk], d2, CBSZ, 2 * NS + s, 0);
                                d3 = __builtin_amdgcn_mfma_f32_4x4x1f32(af[kk], b[3 * KSEG + kk], d3, CBSZ, 3 * NS + s, 0);
                            }
                        }
                        const f32x4 sum = (d0 + d1) + (d2 + d3);
                        {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (ok[i]) *rowp[i] = c0[i] + sum[i];
                        }
                    }
                }
            };
            qset(std::integral_constant<int, 0>{});
            qset(std::integral_constant<int, 1>{});
            qset(std::integral_constant<int, 2>{});
            qset(std::integral_constant<int, 3>{});
        } else {
            // transposed 16x16x4 tile: D^T[cout = u*16 + q*4 + r][pair = m16] += W[k]^T X^T in the pair's slab row
            const int q = lane >> 4, m16 = lane & 15;
            const bool valid = m16 < tl.np;
            const int orow = lst[valid ? m16 : 0] & 63;
            float *row = slab + orow * COUT;
            const int sw = orow & (NCH - 1);
            f32x4 d[NTW];
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                const float4 c = *(const float4 *)(row + (((u * 4 + q) ^ sw) << 2));
                d[u] = (f32x4){c.x, c.y, c.z, c.w};
            }
            if (!(dbg & 4)) {
#pragma unroll
                for (int kk = 0; kk < KSEG; ++kk)
#pragma unroll
                    for (int u = 0; u < NTW; ++u)
                        d[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[u * KSEG + kk], af[kk], d[u], 0, 0, 0);
            }
            if (valid) {
#pragma unroll
                for (int u = 0; u < NTW; ++u)
                    *(float4 *)(row + (((u * 4 + q) ^ sw) << 2)) = make_float4(d[u][0], d[u][1], d[u][2], d[u][3]);
            }
        }
    };

    if (lo < hi) {
        float b0[NB], b1[NB];
        float a0[KSEG], a1[KSEG];
        Tile cur, nxt;
        int g = lo;
        g += locate(g, cur);
        fetch_a(cur, a0);
        load_w(cur.k, b0);
        // one offset (slot) of this wave on the weight registers `b`; on entry its first tile is in a0 (requested),
        // on exit the first tile of the wave's next offset is in a0 (requested) and described by `cur`; the weights of
        // the offset after this one go to `bn` once the first tile's MFMAs are issued.  Returns false at the end.
        // The prefetches are UNCONDITIONAL (past the end the wave re-requests its current tile / offset): a load under
        // a branch makes the number of loads in flight path-dependent, and the compiler then waits for vmcnt(0) -- i.e.
        // for the prefetch it has just issued -- in front of every tile (that is what parked the round-2/3 kernel).
        auto run_slot = [&](const float (&b)[NB], float (&bn)[NB]) -> bool {
            const int slot = cur.slot;
            bool first = true;
            for (;;) {
                bool more = g < hi;
                nxt = cur;
                if (more) g += locate(g, nxt);
                fetch_a(nxt, a1);
                tile(cur, a0, b);
                if (first) {
                    const int kn = next_slot_k(slot);
                    load_w(kn >= 0 ? kn : cur.k, bn);
                    first = false;
                }
                if (!more) return false;
                cur = nxt;
                if (cur.slot != slot) {
#pragma unroll
                    for (int i = 0; i < KSEG; ++i) a0[i] = a1[i];
                    return true;
                }
                more = g < hi;
                nxt = cur;
                if (more) g += locate(g, nxt);
                fetch_a(nxt, a0);
                tile(cur, a1, b);
                if (!more) return false;
                cur = nxt;
                if (cur.slot != slot) return true;
            }
        };
        for (;;) {
            if (!run_slot(b0, b1)) break;
            if (!run_slot(b1, b0)) break;
        }
    }
    __syncthreads();

    // ---- epilogue: sum the wave slabs in wave order, folded BatchNorm / bias / ReLU, 16-byte row stores --------------
    constexpr int C4 = COUT / 4;
    for (int i = tid; i < rows * C4; i += T) {
        const int r = i / C4, c4 = i - r * C4;
        const int ch = QUAD ? c4 : (c4 ^ (r & (NCH - 1)));
        const float *src = slabs + r * COUT + ch * 4;
        float4 v = *(const float4 *)src;
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const float4 p = *(const float4 *)(src + w * RW * COUT);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        const float4 sc = scale ? *(const float4 *)(scale + c4 * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 sh = shift ? *(const float4 *)(shift + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *(float4 *)(y + (size_t)(row0 + RS * r) * COUT + c4 * 4) = v;
    }
}

template <int CIN, int COUT, int NW, int WPS, int QUAD, int ILV>
int launch_gq_cfg(const float *x, const int32_t *nbr, const int32_t *n_ptr, int cap, const float *wp,
                  const float *scale, const float *shift, int relu, float *y, int dbg, hipStream_t stream)
{
    constexpr size_t lds = gq_lds_bytes<COUT, NW>();
    static_assert(lds <= 160 * 1024, "workgroup slabs exceed the 160 KB LDS");
    if (cap >= (1 << 25)) return SASSD_EINVAL;               // packed list entries: input row << 6 | local row
    static std::atomic<unsigned long long> attr_done{0};
    const void *fn = (const void *)spconv_gq_kernel<CIN, COUT, NW, WPS, QUAD, ILV>;
    int rc = sassd_dyn_lds(fn, lds, attr_done);
    if (rc) return rc;
    hipLaunchKernelGGL((spconv_gq_kernel<CIN, COUT, NW, WPS, QUAD, ILV>), dim3(gq_grid(cap, ILV)), dim3(NW * 64), lds, stream, x,
                       nbr, n_ptr, cap, wp, scale, shift, relu, y, dbg & 127);
    return sassd_launch_status();
}


// ------------------------------------------------------------------------------------------------------------------
// 1 x 1 x 1 sparse conv (spconv's `SparseConv3d(64, 64, (1,1,1))` shortcut = a dense [rows, CIN] x [CIN, COUT] product,
// cmn.py:208-212 `extra_conv`; the same shape is the data gradient of that layer).  The register-stationary kernel ran it
// through its 27-offset machinery (41 us at 106 k rows = 1.3 TB/s); this is a streaming kernel: a wave keeps W (pack
// [CIN/4][COUT][4]) in registers and walks 16-row tiles with the transposed 16x16x4 MFMA -- a lane ends up with four
// consecutive output channels of one row, i.e. 16-byte stores -- with the next tile's rows in flight (unconditionally:
// past the end the last tile is requested again).  HBM-bound: 8 * rows * 64 bytes.
// ------------------------------------------------------------------------------------------------------------------
template <int CIN, int COUT>
__global__ void __launch_bounds__(256)
spconv_pw_kernel(const float *__restrict__ x, const int32_t *__restrict__ n_ptr, int cap, const float *__restrict__ wp,
                 const float *__restrict__ scale, const float *__restrict__ shift, int relu, float *__restrict__ y)
{
    constexpr int KSEG = CIN / 4, NTW = COUT / 16;
    const int n = min(*n_ptr, cap);
    const int lane = threadIdx.x & 63, q = lane >> 4, m16 = lane & 15;
    const int wave = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)), nwaves = (int)gridDim.x * 4;
    const int ntile = (n + 15) >> 4;
    if (wave >= ntile) return;
    float b[NTW * KSEG];
#pragma unroll
    for (int u = 0; u < NTW; ++u)
#pragma unroll
        for (int k4 = 0; k4 < KSEG / 4; ++k4) {
            const float4 v = *(const float4 *)(wp + ((size_t)(q * (KSEG / 4) + k4) * COUT + u * 16 + m16) * 4);
            b[u * KSEG + 4 * k4] = v.x; b[u * KSEG + 4 * k4 + 1] = v.y; b[u * KSEG + 4 * k4 + 2] = v.z; b[u * KSEG + 4 * k4 + 3] = v.w;
        }
    float4 sc[NTW], sh[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        sc[u] = scale ? *(const float4 *)(scale + u * 16 + q * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
        sh[u] = shift ? *(const float4 *)(shift + u * 16 + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto fetch = [&](int t, float (&af)[KSEG]) {
        const int r = min(t * 16 + m16, n - 1);
        load_vec<KSEG>(x + (size_t)r * CIN + q * KSEG, af);
    };
    auto tile = [&](int t, const float (&af)[KSEG]) {
        f32x4 d[NTW];
#pragma unroll
        for (int u = 0; u < NTW; ++u) d[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KSEG; ++kk)
#pragma unroll
            for (int u = 0; u < NTW; ++u) d[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[u * KSEG + kk], af[kk], d[u], 0, 0, 0);
        const int r = t * 16 + m16;
        if (r < n) {
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                float4 v = make_float4(d[u][0] * sc[u].x + sh[u].x, d[u][1] * sc[u].y + sh[u].y, d[u][2] * sc[u].z + sh[u].z,
                                       d[u][3] * sc[u].w + sh[u].w);
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *(float4 *)(y + (size_t)r * COUT + u * 16 + q * 4) = v;
            }
        }
    };
    float a0[KSEG], a1[KSEG];
    fetch(wave, a0);
    for (int t = wave; t < ntile; t += 2 * nwaves) {
        fetch(min(t + nwaves, ntile - 1), a1);
        tile(t, a0);
        if (t + nwaves < ntile) {
            fetch(min(t + 2 * nwaves, ntile - 1), a0);
            tile(t + nwaves, a1);
        }
    }
}

template <int CIN, int COUT>
int launch_pw(const float *x, const int32_t *n_ptr, int cap, const float *wp, const float *scale, const float *shift,
              int relu, float *y, hipStream_t stream)
{
    int cus = 0;
    int rc = sassd_num_cus(&cus);
    if (rc) return rc;
    const int grid = min(cdiv(cdiv(cap, 16), 4), cus * 8);
    hipLaunchKernelGGL((spconv_pw_kernel<CIN, COUT>), dim3(grid), dim3(256), 0, stream, x, n_ptr, cap, wp, scale, shift, relu, y);
    return sassd_launch_status();
}


// ------------------------------------------------------------------------------------------------------------------
// The 4-channel input layer (cmn.py:197 conv0[0]: SubMConv3d(4, 16, 3) on the per-voxel means).  108 multiply-adds per
// output value: nothing for a matrix core to do -- the register-stationary MFMA kernel padded 4 input channels into
// 16x16x4 steps and took 31 us for 129 k rows.  Here 16 consecutive threads own one output row (one output channel
// each): the row's rulebook record and its neighbours' 16-byte feature rows are read once per 16 threads (the same
// address in all of them: one request), W[k][0..3][co] is one conflict-free 16-byte LDS read (the pack [K][1][COUT][4]
// IS that layout), k ascending -- a fixed summation order.  Latency is hidden by occupancy (few registers, no LDS
// beyond the 6.9 KB of weights), not by a software pipeline.
// ------------------------------------------------------------------------------------------------------------------
template <int COUT>
__global__ void __launch_bounds__(256)
spconv_c4_kernel(const float *__restrict__ x, const int32_t *__restrict__ nbr, const int32_t *__restrict__ n_ptr, int cap,
                 const float *__restrict__ wp, const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                 float *__restrict__ y)
{
    __shared__ float4 ws[kK * COUT];
    for (int i = threadIdx.x; i < kK * COUT; i += 256) ws[i] = ((const float4 *)wp)[i];
    __syncthreads();
    const int n = min(*n_ptr, cap);
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int row = t / COUT, co = t - row * COUT;
    if (row >= n) return;
    const int32_t *nb = nbr + (size_t)row * kK;
    float acc = 0.f;
#pragma unroll 9
    for (int k = 0; k < kK; ++k) {
        const int in = nb[k];
        if (in >= 0) {
            const float4 xv = *(const float4 *)(x + (size_t)in * 4);
            const float4 wv = ws[k * COUT + co];
            acc = fmaf(xv.w, wv.w, fmaf(xv.z, wv.z, fmaf(xv.y, wv.y, fmaf(xv.x, wv.x, acc))));
        }
    }
    float o = acc * (scale ? scale[co] : 1.f) + (shift ? shift[co] : 0.f);
    if (relu) o = fmaxf(o, 0.f);
    y[(size_t)row * COUT + co] = o;
}

template <int COUT>
int launch_c4(const float *x, const int32_t *nbr, const int32_t *n_ptr, int cap, const float *wp, const float *scale,
              const float *shift, int relu, float *y, hipStream_t stream)
{
    hipLaunchKernelGGL((spconv_c4_kernel<COUT>), dim3(cdiv(cap * COUT, 256)), dim3(256), 0, stream, x, nbr, n_ptr, cap, wp,
                       scale, shift, relu, y);
    return sassd_launch_status();
}
