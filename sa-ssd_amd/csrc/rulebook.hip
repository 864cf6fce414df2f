// rulebook.hip -- hash/bitmap rulebook construction for submanifold and strided sparse 3-D convolution.
//
// Replaces spconv v1.0 get_indice_pairs (prepareSubMGridKernel/getSubMIndicePairsKernel and
// prepareIndicePairsKernel -> thrust sort/unique -> assign*Kernel) as reached from
// mmdet/models/necks/cmn.py:147-173.  spconv v1.0 memsets a dense int32 grid of B*D*H*W cells per rulebook
// (360 MB at level 0); here
//   * coordinate -> row lookup is an open-addressing hash (8 B/row * 2),
//   * the strided conv's sorted-unique output set comes from a 1-bit-per-cell bitmap + popcount prefix sum
//     (ascending linear (b,z,y,x) order falls out of the scan -- no sort),
//   * the rulebook is an output-stationary gather table nbr[row_out][27] (one coalesced 108-B record per row).
// Algorithmic HBM bytes: subm 16*Nin + 8*P ; strided 16*Nin + 16*Nout + 8*P.
#include "common.h"

namespace {


__global__ void hash_build_kernel(const int32_t *__restrict__ idx, const int32_t *__restrict__ n_ptr, int cap, int D,
                                  int H, int W, HashView hv, int32_t *status)
{
    const int n = min(*n_ptr, cap);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = ((const int4 *)idx)[i];
    const unsigned key = (((unsigned)c.x * D + c.y) * H + c.z) * W + c.w;
    const int e = hash2_insert(hv.ent, hv.mask, key);
    if (e < 0) { if (status) atomicOr(status, SASSD_ST_HASH_FULL); return; }
    hv.ent[e].y = (unsigned)i;
}

// one thread per (output row, kernel offset): nbr[row*27+k] = row of voxel at c + (k-1) or -1
__global__ void nbr_subm_kernel(const int32_t *__restrict__ idx, const int32_t *__restrict__ n_ptr, int cap, int D,
                                int H, int W, HashView hv, int32_t *__restrict__ nbr)
{
    const int n = min(*n_ptr, cap);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 27) return;
    const int row = t / 27, k = t - row * 27;
    const int4 c = ((const int4 *)idx)[row];
    const int z = c.y + k / 9 - 1, y = c.z + (k / 3) % 3 - 1, x = c.w + k % 3 - 1;
    int r = -1;
    if (k == 13) r = row;
    else if (z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W) {
        const unsigned key = (((unsigned)c.x * D + z) * H + y) * W + x;
        r = hash2_lookup(hv.ent, hv.mask, key);
    }
    nbr[t] = r;
}

struct DownDims { int D, H, W, OD, OH, OW, B; };

// per axis: input coord i reaches outputs (i+1-kk)/2 for kk in {0,1,2} with i+1-kk even and in range
__device__ __forceinline__ int axis_outs(int i, int od, int o[2])
{
    int n = 0;
    if (i & 1) {
        int a = (i + 1) >> 1, b = (i - 1) >> 1;
        if (a < od) o[n++] = a;
        if (b >= 0 && b < od) o[n++] = b;
    } else {
        int a = i >> 1;
        if (a < od) o[n++] = a;
    }
    return n;
}

__global__ void down_mark_kernel(const int32_t *__restrict__ idx, const int32_t *__restrict__ n_ptr, int cap,
                                 DownDims d, unsigned *__restrict__ bitmap)
{
    const int n = min(*n_ptr, cap);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = ((const int4 *)idx)[i];
    int oz[2], oy[2], ox[2];
    const int nz = axis_outs(c.y, d.OD, oz), ny = axis_outs(c.z, d.OH, oy), nx = axis_outs(c.w, d.OW, ox);
    for (int a = 0; a < nz; ++a)
        for (int b = 0; b < ny; ++b)
            for (int e = 0; e < nx; ++e) {
                const unsigned lin = (((unsigned)c.x * d.OD + oz[a]) * d.OH + oy[b]) * d.OW + ox[e];
                atomicOr(&bitmap[lin >> 5], 1u << (lin & 31));
            }
}

constexpr int kWordsPerBlock = 1024;     // 256 threads x 4 words

__global__ void __launch_bounds__(256) bitmap_count_kernel(const unsigned *__restrict__ bitmap, int nwords,
                                                           int *__restrict__ bsum)
{
    __shared__ int wsum[17];
    const int base = blockIdx.x * kWordsPerBlock + threadIdx.x * 4;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k < nwords) s += __popc(bitmap[base + k]);
    int tot;
    block_exclusive_scan(s, wsum, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(1024) blocksum_scan_kernel(const int *__restrict__ bsum, int *__restrict__ bbase,
                                                             int nblk, int32_t *n_out_ptr, int cap_out,
                                                             int32_t *status)
{
    __shared__ int wsum[17];
    int running = 0;
    for (int b0 = 0; b0 < nblk; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        int v = (b < nblk) ? bsum[b] : 0;
        int tot;
        int ex = block_exclusive_scan(v, wsum, &tot);
        if (b < nblk) bbase[b] = running + ex;
        running += tot;
    }
    if (threadIdx.x == 0) {
        if (running > cap_out) { if (status) atomicOr(status, SASSD_ST_VOXEL_OVERFLOW); running = cap_out; }
        *n_out_ptr = running;
    }
}

// emits out_indices in ascending linear order and records, per bitmap word, the row of its first set bit
__global__ void __launch_bounds__(256) down_emit_kernel(const unsigned *__restrict__ bitmap, int nwords,
                                                        const int *__restrict__ bbase, DownDims d, int cap_out,
                                                        int32_t *__restrict__ out_idx)
{
    __shared__ int wsum[17];
    const int base = blockIdx.x * kWordsPerBlock + threadIdx.x * 4;
    unsigned w[4];
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { w[k] = (base + k < nwords) ? bitmap[base + k] : 0u; s += __popc(w[k]); }
    int tot;
    int row = bbase[blockIdx.x] + block_exclusive_scan(s, wsum, &tot);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned m = w[k];
        while (m) {
            const int bit = __ffs(m) - 1;
            m &= m - 1;
            if (row < cap_out) {
                unsigned lin = (unsigned)(base + k) * 32u + bit;
                const int x = lin % d.OW; lin /= d.OW;
                const int y = lin % d.OH; lin /= d.OH;
                const int z = lin % d.OD; lin /= d.OD;
                ((int4 *)out_idx)[row] = make_int4((int)lin, z, y, x);
            }
            ++row;
        }
    }
}

// one thread per (out row, k): input coord = 2*o - 1 + kk per axis
__global__ void nbr_down_kernel(const int32_t *__restrict__ out_idx, const int32_t *__restrict__ n_out_ptr,
                                int cap_out, DownDims d, HashView hv, int32_t *__restrict__ nbr)
{
    const int n = min(*n_out_ptr, cap_out);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 27) return;
    const int row = t / 27, k = t - row * 27;
    const int4 c = ((const int4 *)out_idx)[row];
    const int z = 2 * c.y - 1 + k / 9, y = 2 * c.z - 1 + (k / 3) % 3, x = 2 * c.w - 1 + k % 3;
    int r = -1;
    if (z >= 0 && z < d.D && y >= 0 && y < d.H && x >= 0 && x < d.W) {
        const unsigned key = (((unsigned)c.x * d.D + z) * d.H + y) * d.W + x;
        r = hash2_lookup(hv.ent, hv.mask, key);
    }
    nbr[t] = r;
}

// spconv-format pairs: one block per kernel offset, ordered compaction over output rows
__global__ void __launch_bounds__(1024) pairs_kernel(const int32_t *__restrict__ nbr, const int32_t *__restrict__ n_ptr,
                                                     int cap, int K, int32_t *__restrict__ pairs,
                                                     int32_t *__restrict__ pair_num)
{
    __shared__ int wsum[17];
    const int n = min(*n_ptr, cap);
    const int k = blockIdx.x;
    int32_t *pin = pairs + (size_t)k * 2 * cap, *pout = pin + cap;
    int running = 0;
    for (int r0 = 0; r0 < n; r0 += 1024) {
        const int r = r0 + threadIdx.x;
        const int v = (r < n) ? nbr[(size_t)r * K + k] : -1;
        int tot;
        const int ex = block_exclusive_scan(v >= 0 ? 1 : 0, wsum, &tot);
        if (v >= 0) { pin[running + ex] = v; pout[running + ex] = r; }
        running += tot;
    }
    if (threadIdx.x == 0) pair_num[k] = running;
}

inline bool lin_fits(int D, int H, int W, int B) { return (double)D * H * W * B < 4294967294.0; }


// ------------------------------------------------------------------------------------------------------------------
// Fused rulebook pyramid: every gather table of a submanifold / strided-conv stack (level l: SubMConv3d rulebook;
// l -> l+1: SparseConv3d(k3,s2,p1) rulebook + the output coordinates) in 2 + 3*(L-1) launches and two memsets
// (11 kernel launches instead of 30 for the four VxNet levels of cmn.py:147-173).
//   coordinate -> row lookup   level 0 (voxelizer rows, first-touch order): the open-addressing hash.  Levels >= 1
//                     are emitted in ascending linear order, so their occupancy BITMAP plus a per-word rank array is a
//                     perfect index: row(c) = rank[c >> 5] + popcount(bitmap[c >> 5] below bit c) -- two independent
//                     loads, no probing, no inserts (hash inserts from the emit kernel were a chain of dependent
//                     atomic round trips per thread: 50-110 us per level).
//   rb_level_kernel   one thread per (row of level l, (kz, ky)): three x-offsets at a time of the subm table of level l
//                     and of the strided table INTO level l (lookup in level l-1; the three cells share a bitmap word 15
//                     times out of 16) and -- threads g < 8 -- the marks of the row's <= 8 strided outputs in the bitmap
//                     of level l+1.  All three only read finished structures.
//   rb_count_kernel   set bits per 256-word bitmap block + per 64-block super-block.
//   rb_emit_kernel    ordered compaction of the bitmap: every workgroup sums the (super-)counters before it (a few
//                     hundred L2-resident ints; no single-workgroup scan pass and no inter-workgroup waiting -- a
//                     chained single-pass scan with tagged words measured 70-150 us here), writes the rank array and
//                     emits its coordinates in ascending linear order: one word per thread, the word's first cell
//                     decomposed once, its <= 32 set bits walked with carries.
// ------------------------------------------------------------------------------------------------------------------
struct Lookup {
    const unsigned *bitmap;    // nullptr -> hash
    const int *rank;
    HashView hv;
    int cap;                   // rows past the level's capacity do not exist
};

__device__ __forceinline__ int lookup_row(const Lookup &L, unsigned key)
{
    if (L.bitmap) {
        const unsigned w = L.bitmap[key >> 5], b = 1u << (key & 31);
        if (!(w & b)) return -1;
        const int r = L.rank[key >> 5] + __popc(w & (b - 1));
        return r < L.cap ? r : -1;
    }
    return hash2_lookup(L.hv.ent, L.hv.mask, key);
}

struct LevelArgs {
    const int32_t *idx; const int32_t *n_ptr; int cap; int D, H, W;
    Lookup cur;                // this level's index (complete)
    int32_t *nbr_subm;         // [cap,27] or nullptr
    Lookup prev;               // previous (finer) level's index
    int pD, pH, pW;            // previous level's dims
    int32_t *nbr_down;         // [cap,27] strided table into this level, or nullptr
    unsigned *bitmap_next;     // marks of level l+1's outputs, or nullptr
    int OD, OH, OW;
};

// three lookups along x (keys base + x0 .. base + x0 + 2, columns outside [0, W) = no voxel) in one go: with the bitmap
// index the three cells share one 32-cell word 15 times out of 16, so a row's 27 neighbours cost 9-10 word + rank loads
// instead of 54; the hash index (level 0) has nothing to share
__device__ __forceinline__ void lookup3(const Lookup &L, unsigned base, int x0, int W, int (&r)[3])
{
    r[0] = r[1] = r[2] = -1;
    if (L.bitmap) {
        const int xa = max(x0, 0), xc = min(x0 + 2, W - 1);
        if (xa > xc) return;
        const unsigned ka = base + (unsigned)xa, kc = base + (unsigned)xc;
        const unsigned wa = L.bitmap[ka >> 5];
        const int ra = L.rank[ka >> 5];
        unsigned wc = wa;
        int rc = ra;
        if ((kc >> 5) != (ka >> 5)) { wc = L.bitmap[kc >> 5]; rc = L.rank[kc >> 5]; }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int x = x0 + d;
            if (x < 0 || x >= W) continue;
            const unsigned key = base + (unsigned)x;
            const bool first = (key >> 5) == (ka >> 5);
            const unsigned w = first ? wa : wc, b = 1u << (key & 31);
            if (w & b) {
                const int row = (first ? ra : rc) + __popc(w & (b - 1));
                r[d] = row < L.cap ? row : -1;
            }
        }
    } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int x = x0 + d;
            if (x >= 0 && x < W) r[d] = hash2_lookup(L.hv.ent, L.hv.mask, base + (unsigned)x);
        }
    }
}

// one thread per (row of level l, (kz, ky)): the three x-offsets of the subm table of level l and of the strided table
// INTO level l, and -- threads g < 8 -- the marks of the row's <= 8 strided outputs in the bitmap of level l+1
__global__ void __launch_bounds__(256) rb_level_kernel(LevelArgs A)
{
    const int n = min(*A.n_ptr, A.cap);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 9) return;
    const int row = t / 9, g = t - row * 9;
    const int4 c = ((const int4 *)A.idx)[row];
    const int kz = g / 3, ky = g - kz * 3;
    if (A.nbr_subm) {
        const int z = c.y + kz - 1, y = c.z + ky - 1;
        int r[3] = {-1, -1, -1};
        if (z >= 0 && z < A.D && y >= 0 && y < A.H)
            lookup3(A.cur, (((unsigned)c.x * A.D + z) * A.H + y) * A.W, c.w - 1, A.W, r);
        if (g == 4) r[1] = row;                              // the centre offset (k = 13) is the row itself
        int32_t *dst = A.nbr_subm + (size_t)row * 27 + g * 3;
        dst[0] = r[0]; dst[1] = r[1]; dst[2] = r[2];
    }
    if (A.nbr_down) {
        const int z = 2 * c.y - 1 + kz, y = 2 * c.z - 1 + ky;
        int r[3] = {-1, -1, -1};
        if (z >= 0 && z < A.pD && y >= 0 && y < A.pH)
            lookup3(A.prev, (((unsigned)c.x * A.pD + z) * A.pH + y) * A.pW, 2 * c.w - 1, A.pW, r);
        int32_t *dst = A.nbr_down + (size_t)row * 27 + g * 3;
        dst[0] = r[0]; dst[1] = r[1]; dst[2] = r[2];
    }
    if (A.bitmap_next && g < 8) {
        int oz[2], oy[2], ox[2];
        const int nz = axis_outs(c.y, A.OD, oz), ny = axis_outs(c.z, A.OH, oy), nx = axis_outs(c.w, A.OW, ox);
        const int a = g & 1, b = (g >> 1) & 1, e = g >> 2;
        if (a < nz && b < ny && e < nx) {
            const unsigned lin = (((unsigned)c.x * A.OD + oz[a]) * A.OH + oy[b]) * A.OW + ox[e];
            atomicOr(&A.bitmap_next[lin >> 5], 1u << (lin & 31));
        }
    }
}

constexpr int kEmitWords = 256;          // bitmap words per workgroup of the pyramid's count / emit kernels (1 / thread)
constexpr int kSupShift = 6;             // super-counter = 64 blocks

__global__ void __launch_bounds__(256) rb_count_kernel(const unsigned *__restrict__ bitmap, int nwords,
                                                       int *__restrict__ bcnt, int *__restrict__ sup)
{
    __shared__ int wsum[17];
    const int i = blockIdx.x * kEmitWords + threadIdx.x;
    const int s = i < nwords ? __popc(bitmap[i]) : 0;
    int tot;
    block_exclusive_scan(s, wsum, &tot);
    if (threadIdx.x == 0) {
        bcnt[blockIdx.x] = tot;
        if (tot) atomicAdd(&sup[blockIdx.x >> kSupShift], tot);
    }
}

__global__ void __launch_bounds__(256) rb_emit_kernel(const unsigned *__restrict__ bitmap, int nwords, int nblk,
                                                      const int *__restrict__ bcnt, const int *__restrict__ sup,
                                                      int *__restrict__ rank, int OD, int OH, int OW, int cap_out,
                                                      int32_t *__restrict__ out_idx, int32_t *n_out_ptr,
                                                      int32_t *status)
{
    __shared__ int wsum[17];
    const int blk = blockIdx.x;
    // rows emitted before this block = whole super-blocks + the blocks of its own super-block before it
    const int nsup = blk >> kSupShift;
    int part = 0;
    for (int j = threadIdx.x; j < nsup; j += 256) part += sup[j];
    if ((int)threadIdx.x < blk - (nsup << kSupShift)) part += bcnt[(nsup << kSupShift) + threadIdx.x];
    int before;
    block_exclusive_scan(part, wsum, &before);
    const int wi = blk * kEmitWords + threadIdx.x;
    unsigned m = wi < nwords ? bitmap[wi] : 0u;
    int tot;
    const int ex = block_exclusive_scan(__popc(m), wsum, &tot);
    int row = before + ex;
    if (wi < nwords) rank[wi] = row;                  // rank of a word = set bits before it
    if (blk == nblk - 1 && threadIdx.x == 0) {
        int total = before + tot;
        if (total > cap_out) { if (status) atomicOr(status, SASSD_ST_VOXEL_OVERFLOW); total = cap_out; }
        *n_out_ptr = total;
    }
    if (!m) return;
    // coordinates of the word's first cell once (three divisions), then walk the <= 32 set bits with carries
    unsigned lin = (unsigned)wi * 32u;
    int x = lin % OW; lin /= OW;
    int y = lin % OH; lin /= OH;
    int z = lin % OD; lin /= OD;
    int b = (int)lin, prev = 0;
    while (m) {
        const int bit = __ffs(m) - 1;
        m &= m - 1;
        x += bit - prev; prev = bit;
        while (x >= OW) { x -= OW; if (++y == OH) { y = 0; if (++z == OD) { z = 0; ++b; } } }
        if (row < cap_out) ((int4 *)out_idx)[row] = make_int4(b, z, y, x);
        ++row;
    }
}

constexpr int kMaxLevels = 8;

struct PyramidLayout {
    size_t keys, vals, bitmap[kMaxLevels], bcnt[kMaxLevels], sup[kMaxLevels], rank[kMaxLevels];
    size_t keys_end, zero_begin, zero_end, total;
    int nwords[kMaxLevels], nblk[kMaxLevels];
    int dims[kMaxLevels][3];
};

bool pyramid_layout(int levels, const int *caps, int D, int H, int W, int B, PyramidLayout &L)
{
    if (levels < 1 || levels > kMaxLevels) return false;
    size_t o = 0;
    L.keys = o; o += align_up((size_t)hash_cap(caps[0]) * 8, 256);          // level-0 hash only: {key, value} entries
    L.keys_end = o;
    L.vals = o;
    L.zero_begin = o;
    L.dims[0][0] = D; L.dims[0][1] = H; L.dims[0][2] = W;
    if (!lin_fits(D, H, W, B)) return false;
    for (int l = 0; l + 1 < levels; ++l) {
        for (int a = 0; a < 3; ++a) L.dims[l + 1][a] = (L.dims[l][a] - 1) / 2 + 1;
        const size_t cells = (size_t)B * L.dims[l + 1][0] * L.dims[l + 1][1] * L.dims[l + 1][2];
        L.nwords[l] = (int)((cells + 31) / 32);
        L.nblk[l] = cdiv(L.nwords[l], kEmitWords);
        L.bitmap[l] = o; o += align_up((size_t)L.nwords[l] * 4, 256);
        L.sup[l] = o;    o += align_up((size_t)((L.nblk[l] >> kSupShift) + 1) * 4, 256);
    }
    L.zero_end = o;
    for (int l = 0; l + 1 < levels; ++l) { L.bcnt[l] = o; o += align_up((size_t)L.nblk[l] * 4, 256); }
    for (int l = 0; l + 1 < levels; ++l) { L.rank[l] = o; o += align_up((size_t)L.nwords[l] * 4, 256); }
    L.total = o;
    return true;
}

}  // namespace

extern "C" size_t sassd_hash_bytes(int cap_rows) { return (size_t)hash_cap(cap_rows) * 8; }

extern "C" int sassd_hash_build(const int32_t *indices, const int32_t *n_ptr, int cap, int D, int H, int W,
                                int batch_size, void *table, size_t table_bytes, int32_t *status, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!indices || !n_ptr || !table || cap <= 0 || !lin_fits(D, H, W, batch_size)) return SASSD_EINVAL;
    if (table_bytes < sassd_hash_bytes(cap)) return SASSD_ENOSPC;
    HashView hv = hash_view(table, cap);
    int rc;
    if ((rc = sassd_hip(hipMemsetAsync(hv.ent, 0xFF, (size_t)(hv.mask + 1) * 8, stream)))) return rc;
    hipLaunchKernelGGL(hash_build_kernel, dim3(cdiv(cap, 256)), dim3(256), 0, stream, indices, n_ptr, cap, D, H, W,
                       hv, status);
    return sassd_launch_status();
}

extern "C" int sassd_rulebook_subm(const int32_t *indices, const int32_t *n_ptr, int cap, int D, int H, int W,
                                   int batch_size, const void *table, size_t table_bytes, int32_t *nbr,
                                   void *stream_)
{
    if (!indices || !n_ptr || !table || !nbr || cap <= 0 || !lin_fits(D, H, W, batch_size)) return SASSD_EINVAL;
    if (table_bytes < sassd_hash_bytes(cap)) return SASSD_ENOSPC;
    HashView hv = hash_view(table, cap);
    hipLaunchKernelGGL(nbr_subm_kernel, dim3(cdiv(cap * 27, 256)), dim3(256), 0, (hipStream_t)stream_, indices, n_ptr,
                       cap, D, H, W, hv, nbr);
    return sassd_launch_status();
}

namespace {
struct DownLayout { size_t bitmap, bsum, bbase, total; int nwords, nblk; };
DownLayout down_layout(int D, int H, int W, int B)
{
    DownLayout L;
    const int OD = (D - 1) / 2 + 1, OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const size_t cells = (size_t)B * OD * OH * OW;
    L.nwords = (int)((cells + 31) / 32);
    L.nblk = cdiv(L.nwords, kWordsPerBlock);
    size_t o = 0;
    L.bitmap = o; o += align_up((size_t)L.nwords * 4, 256);
    L.bsum = o;   o += align_up((size_t)L.nblk * 4, 256);
    L.bbase = o;  o += align_up((size_t)L.nblk * 4, 256);
    L.total = o;
    return L;
}
}  // namespace

extern "C" size_t sassd_rulebook_conv_workspace_bytes(int D, int H, int W, int batch_size)
{
    return down_layout(D, H, W, batch_size).total;
}

extern "C" int sassd_rulebook_conv(const int32_t *in_indices, const int32_t *n_in_ptr, int cap_in, int D, int H,
                                   int W, int batch_size, const void *in_table, size_t in_table_bytes,
                                   int32_t *out_indices, int32_t *n_out_ptr, int cap_out, int32_t *nbr,
                                   int32_t *status, void *workspace, size_t workspace_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!in_indices || !n_in_ptr || !in_table || !out_indices || !n_out_ptr || !nbr || !workspace) return SASSD_EINVAL;
    if (cap_in <= 0 || cap_out <= 0 || !lin_fits(D, H, W, batch_size)) return SASSD_EINVAL;
    if (in_table_bytes < sassd_hash_bytes(cap_in)) return SASSD_ENOSPC;
    const DownLayout L = down_layout(D, H, W, batch_size);
    if (workspace_bytes < L.total) return SASSD_ENOSPC;
    DownDims d;
    d.D = D; d.H = H; d.W = W; d.B = batch_size;
    d.OD = (D - 1) / 2 + 1; d.OH = (H - 1) / 2 + 1; d.OW = (W - 1) / 2 + 1;
    char *w = (char *)workspace;
    unsigned *bitmap = (unsigned *)(w + L.bitmap);
    int *bsum = (int *)(w + L.bsum), *bbase = (int *)(w + L.bbase);
    HashView hv = hash_view(in_table, cap_in);
    int rc;
    if ((rc = sassd_hip(hipMemsetAsync(bitmap, 0, (size_t)L.nwords * 4, stream)))) return rc;
    hipLaunchKernelGGL(down_mark_kernel, dim3(cdiv(cap_in, 256)), dim3(256), 0, stream, in_indices, n_in_ptr, cap_in,
                       d, bitmap);
    hipLaunchKernelGGL(bitmap_count_kernel, dim3(L.nblk), dim3(256), 0, stream, bitmap, L.nwords, bsum);
    hipLaunchKernelGGL(blocksum_scan_kernel, dim3(1), dim3(1024), 0, stream, bsum, bbase, L.nblk, n_out_ptr, cap_out,
                       status);
    hipLaunchKernelGGL(down_emit_kernel, dim3(L.nblk), dim3(256), 0, stream, bitmap, L.nwords, bbase, d, cap_out,
                       out_indices);
    hipLaunchKernelGGL(nbr_down_kernel, dim3(cdiv(cap_out * 27, 256)), dim3(256), 0, stream, out_indices, n_out_ptr,
                       cap_out, d, hv, nbr);
    return sassd_launch_status();
}

extern "C" int sassd_rulebook_pairs(const int32_t *nbr, const int32_t *n_out_ptr, int cap_out, int K,
                                    int32_t *pairs, int32_t *pair_num, void *stream_)
{
    if (!nbr || !n_out_ptr || !pairs || !pair_num || cap_out <= 0 || K <= 0) return SASSD_EINVAL;
    hipLaunchKernelGGL(pairs_kernel, dim3(K), dim3(1024), 0, (hipStream_t)stream_, nbr, n_out_ptr, cap_out, K, pairs,
                       pair_num);
    return sassd_launch_status();
}


extern "C" size_t sassd_rulebook_pyramid_workspace_bytes(int levels, const int *caps, int D, int H, int W,
                                                         int batch_size)
{
    PyramidLayout L;
    if (!caps || !pyramid_layout(levels, caps, D, H, W, batch_size, L)) return 0;
    return L.total;
}

extern "C" int sassd_rulebook_pyramid(int levels, int32_t *const *indices, int32_t *const *n_ptrs, const int *caps,
                                      int D, int H, int W, int batch_size, int32_t *const *nbr_subm,
                                      int32_t *const *nbr_down, int level_begin, int level_end, int32_t *status,
                                      void *workspace, size_t workspace_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!indices || !n_ptrs || !caps || !nbr_subm || !nbr_down || !workspace) return SASSD_EINVAL;
    PyramidLayout L;
    if (!pyramid_layout(levels, caps, D, H, W, batch_size, L)) return SASSD_EINVAL;
    if (workspace_bytes < L.total) return SASSD_ENOSPC;
    if (level_begin < 0 || level_end > levels || level_begin >= level_end) return SASSD_EINVAL;
    for (int l = 0; l < levels; ++l) {
        if (!indices[l] || !n_ptrs[l] || caps[l] <= 0) return SASSD_EINVAL;
        if (l > 0 && !nbr_down[l]) return SASSD_EINVAL;
    }
    char *w = (char *)workspace;
    int rc;
    Lookup look[kMaxLevels];
    for (int l = 0; l < levels; ++l) {
        look[l].cap = caps[l];
        if (l == 0) {
            look[l].bitmap = nullptr; look[l].rank = nullptr;
            look[l].hv.ent = (uint2 *)(w + L.keys);
            look[l].hv.mask = hash_cap(caps[0]) - 1;
        } else {
            look[l].bitmap = (const unsigned *)(w + L.bitmap[l - 1]);
            look[l].rank = (const int *)(w + L.rank[l - 1]);
            look[l].hv = look[0].hv;
        }
    }
    for (int l = level_begin; l < level_end; ++l) {
        if (l == 0) {
            if ((rc = sassd_fill2(w, L.keys_end, 0xFF, w + L.zero_begin,
                                  L.zero_end > L.zero_begin ? L.zero_end - L.zero_begin : 0, 0x00, stream))) return rc;
            hipLaunchKernelGGL(hash_build_kernel, dim3(cdiv(caps[0], 256)), dim3(256), 0, stream, indices[0],
                               n_ptrs[0], caps[0], L.dims[0][0], L.dims[0][1], L.dims[0][2], look[0].hv, status);
        } else {
            const int p = l - 1;
            hipLaunchKernelGGL(rb_emit_kernel, dim3(L.nblk[p]), dim3(256), 0, stream,
                               (const unsigned *)(w + L.bitmap[p]), L.nwords[p], L.nblk[p],
                               (const int *)(w + L.bcnt[p]), (const int *)(w + L.sup[p]), (int *)(w + L.rank[p]),
                               L.dims[l][0], L.dims[l][1], L.dims[l][2], caps[l], indices[l], n_ptrs[l], status);
        }
        LevelArgs A;
        A.idx = indices[l]; A.n_ptr = n_ptrs[l]; A.cap = caps[l];
        A.D = L.dims[l][0]; A.H = L.dims[l][1]; A.W = L.dims[l][2];
        A.cur = look[l];
        A.nbr_subm = nbr_subm[l];
        const int pl = l > 0 ? l - 1 : 0;
        A.prev = look[pl];
        A.pD = L.dims[pl][0]; A.pH = L.dims[pl][1]; A.pW = L.dims[pl][2];
        A.nbr_down = l > 0 ? nbr_down[l] : nullptr;
        const bool last = (l + 1 == levels);
        A.bitmap_next = last ? nullptr : (unsigned *)(w + L.bitmap[l]);
        A.OD = last ? 1 : L.dims[l + 1][0]; A.OH = last ? 1 : L.dims[l + 1][1]; A.OW = last ? 1 : L.dims[l + 1][2];
        if (A.nbr_subm || A.nbr_down || A.bitmap_next)
            hipLaunchKernelGGL(rb_level_kernel, dim3(cdiv(caps[l] * 9, 256)), dim3(256), 0, stream, A);
        if (!last)
            hipLaunchKernelGGL(rb_count_kernel, dim3(L.nblk[l]), dim3(256), 0, stream,
                               (const unsigned *)(w + L.bitmap[l]), L.nwords[l], (int *)(w + L.bcnt[l]),
                               (int *)(w + L.sup[l]));
    }
    return sassd_launch_status();
}
