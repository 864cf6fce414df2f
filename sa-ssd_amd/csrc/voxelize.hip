// voxelize.hip -- hash-based hard voxelisation + fused SimpleVoxel mean for gfx950.
//
// Replaces mmdet/ops/points_op/points_ops.py:5-50,104-164 (serial numba loop over a dense 360 MB
// coor_to_voxelidx grid) and mmdet/models/backbones/vxnet.py:110-116.  Bit-exact with the serial code:
//   * voxel id  = first-touch rank  -> per voxel the MIN point index (slot 0 of an atomicMin cascade),
//                 flag "i is the first point of its voxel", exclusive scan of the flags in point order;
//   * <= T points per voxel in arrival order -> the cascade keeps the T smallest point indices, sorted;
//   * the max_voxels `break` (:41-42)  -> cutoff = index of the first-touch point whose rank == max_voxels;
//                 every point index >= cutoff is ignored;
//   * c = floor((p - lo) / vs) with IEEE f32 subtract and a correctly rounded f32 divide (:31).
// HBM traffic per cloud (algorithmic): 16 N read + 8 N scratch + (16+16+4) M written (mean, coors, num).
#include "common.h"

namespace {

constexpr int kVoxEmpty = 0x7F7F7F7F;      // slot sentinel (hipMemsetAsync byte pattern 0x7F)
constexpr int kPtsPerThread = 4;
constexpr int kScanThreads = 256;
constexpr int kPtsPerBlock = kScanThreads * kPtsPerThread;

struct VoxParams {
    float lo[3], vs[3];
    int grid[3];               // x,y,z cells
    int n, ndim, T, max_voxels, batch_idx, coors_cols, nfeat, cap;
    unsigned hmask;
    const int32_t *n_dev;      // device point count (<= n, the capacity) or nullptr: the count is n
};

__device__ __forceinline__ int vox_n(const VoxParams &P) { return P.n_dev ? min(max(*P.n_dev, 0), P.n) : P.n; }

struct VoxWs {
    unsigned *keys;            // [hcap]
    int *slots;                // [hcap * T]
    int *ent;                  // [n]       hash slot of point i or -1
    int *bsum;                 // [nblk]
    int *bbase;                // [nblk]
    int *scal;                 // [0]=voxel_num  [1]=cutoff
};

__global__ void __launch_bounds__(256) vox_insert_kernel(const float *__restrict__ pts, VoxParams P, VoxWs ws,
                                                         int32_t *status)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool took = false;         // this point became the minimum of its voxel's slot 0
    int lost = -1;             // ... displacing this point
    const bool live = i < vox_n(P);
    const float *p = pts + (size_t)(live ? i : 0) * P.ndim;
    int c[3];
    bool ok = live;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float d = p[j] - P.lo[j];
        float q = __fdiv_rn(d, P.vs[j]);
        float f = floorf(q);
        ok = ok && (f >= 0.0f) && (f < (float)P.grid[j]);
        c[j] = (int)f;
    }
    int e = -1;
    if (ok) {
        unsigned key = ((unsigned)c[2] * (unsigned)P.grid[1] + (unsigned)c[1]) * (unsigned)P.grid[0] + (unsigned)c[0];
        e = hash_insert(ws.keys, P.hmask, key);
        if (e < 0) {
            if (status) atomicOr(status, SASSD_ST_HASH_FULL);
        } else {
            // atomicMin cascade: slot s ends up holding the (s+1)-th smallest point index of this voxel
            int v = i;
            int *s = ws.slots + (size_t)e * P.T;
            for (int t = 0; t < P.T; ++t) {
                int old = atomicMin(&s[t], v);
                if (t == 0) { took = old > i; lost = (old != kVoxEmpty && old > i) ? old : -1; }
                if (old == kVoxEmpty) break;
                v = old > v ? old : v;
            }
        }
    }
    if (live) ws.ent[i] = e;
    // first-touch points per 1024-point block, maintained as the slot-0 minima move (round 6: replaces the count kernel):
    // a point that becomes its voxel's minimum counts for its block, the point it displaces stops counting for its own.
    // bsum starts at -1 (cleared with the key table's 0xFF bytes): the count of block b is bsum[b] + 1.
    const unsigned long long m = __ballot(took);
    if (m && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(&ws.bsum[i / kPtsPerBlock], __popcll(m));
    if (lost >= 0) atomicAdd(&ws.bsum[lost / kPtsPerBlock], -1);
}

__device__ __forceinline__ int is_first(const VoxWs &ws, int T, int n, int i)
{
    if (i >= n) return 0;
    int e = ws.ent[i];
    return (e >= 0 && ws.slots[(size_t)e * T] == i) ? 1 : 0;
}

// What the scan kernel of rounds 1-5 computed, by every emit block for itself (round 6: the voxelizer is fill + insert + emit):
// the first-touch rank at which this block starts = the counts of the blocks before it (a few dozen to a few hundred ints),
// the voxel count M = min(F, max_voxels) -- block 0 writes it out -- and, when the cloud has more voxels than max_voxels, the
// cutoff: the index of the (max_voxels + 1)-th first-touch point (points_ops.py:36-38 `break`: later points are dropped).
struct VoxBase { int base, cutoff; };
__device__ __forceinline__ VoxBase vox_block_base(const VoxParams &P, const VoxWs &ws, int nblk, int *wsum, int32_t *row_offset,
                                                  int32_t *voxel_num)
{
    __shared__ int s_cut;
    const int blk = blockIdx.x;
    int before = 0, all = 0;
    for (int b = threadIdx.x; b < nblk; b += kScanThreads) {
        const int c = ws.bsum[b] + 1;
        all += c;
        if (b < blk) before += c;
    }
    int base, F;
    block_exclusive_scan(before, wsum, &base);
    block_exclusive_scan(all, wsum, &F);
    int cutoff = kVoxEmpty;
    if (F > P.max_voxels) {                                   // uniform; rare (the max_voxels break)
        if (threadIdx.x == 0) {
            int bb = 0, sb = 0;
            for (int b = 0; b < nblk; ++b) {                  // the block that holds rank max_voxels
                const int c = ws.bsum[b] + 1;
                if (bb <= P.max_voxels && P.max_voxels < bb + c) { sb = b; break; }
                bb += c;
            }
            s_cut = sb; wsum[16] = bb;
        }
        __syncthreads();
        const int sb = s_cut, bb = wsum[16];
        __syncthreads();
        const int n = vox_n(P);
        const int i0 = sb * kPtsPerBlock + threadIdx.x * kPtsPerThread;
        int f[kPtsPerThread], cnt = 0;
#pragma unroll
        for (int k = 0; k < kPtsPerThread; ++k) { f[k] = is_first(ws, P.T, n, i0 + k); cnt += f[k]; }
        int tot;
        int r = bb + block_exclusive_scan(cnt, wsum, &tot);
        if (threadIdx.x == 0) s_cut = kVoxEmpty;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPtsPerThread; ++k) {
            if (f[k] && r == P.max_voxels) s_cut = i0 + k;
            r += f[k];
        }
        __syncthreads();
        cutoff = s_cut;
        __syncthreads();
    }
    if (blk == 0 && threadIdx.x == 0) {
        const int M = F < P.max_voxels ? F : P.max_voxels;
        if (voxel_num) *voxel_num = M;
        if (row_offset) row_offset[1] = row_offset[0] + M;
    }
    VoxBase o;
    o.base = base; o.cutoff = cutoff;
    return o;
}

__global__ void __launch_bounds__(kScanThreads) vox_emit_kernel(const float *__restrict__ pts, VoxParams P, VoxWs ws,
                                                                int32_t *row_offset, int32_t *voxel_num, float *voxels,
                                                                int32_t *coors, int32_t *num_points, float *mean,
                                                                int32_t *status)
{
    __shared__ int wsum[17];
    const VoxBase vb = vox_block_base(P, ws, (int)gridDim.x, wsum, row_offset, voxel_num);
    const int base = blockIdx.x * kPtsPerBlock + threadIdx.x * kPtsPerThread;
    int f[kPtsPerThread];
    int s = 0;
    const int n = vox_n(P);
#pragma unroll
    for (int k = 0; k < kPtsPerThread; ++k) { f[k] = is_first(ws, P.T, n, base + k); s += f[k]; }
    int tot;
    int ex = block_exclusive_scan(s, wsum, &tot);
    int r = vb.base + ex;
    const int cutoff = vb.cutoff;
    const int row0 = row_offset ? row_offset[0] : 0;
#pragma unroll
    for (int k = 0; k < kPtsPerThread; ++k) {
        if (!f[k]) continue;
        const int rank = r++;
        if (rank >= P.max_voxels) continue;
        const int row = row0 + rank;
        if (row >= P.cap) { if (status) atomicOr(status, SASSD_ST_VOXEL_OVERFLOW); continue; }
        const int e = ws.ent[base + k];
        const unsigned key = ws.keys[e];
        const int x = key % (unsigned)P.grid[0];
        const int y = (key / (unsigned)P.grid[0]) % (unsigned)P.grid[1];
        const int z = key / ((unsigned)P.grid[0] * (unsigned)P.grid[1]);
        int32_t *cr = coors + (size_t)row * P.coors_cols;
        if (P.coors_cols == 4) { cr[0] = P.batch_idx; cr[1] = z; cr[2] = y; cr[3] = x; }
        else { cr[0] = z; cr[1] = y; cr[2] = x; }
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
        int cnt = 0;
        const int *sl = ws.slots + (size_t)e * P.T;
        for (int t = 0; t < P.T; ++t) {
            const int j = sl[t];
            if (j >= cutoff) break;                 // kVoxEmpty >= cutoff too (slots are ascending)
            const float *pj = pts + (size_t)j * P.ndim;
            if (voxels) {
                float *dst = voxels + ((size_t)row * P.T + cnt) * P.ndim;
                for (int q = 0; q < P.ndim; ++q) dst[q] = pj[q];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) if (q < P.nfeat) acc[q] += pj[q];
            ++cnt;
        }
        if (voxels)
            for (int t = cnt; t < P.T; ++t) {
                float *dst = voxels + ((size_t)row * P.T + t) * P.ndim;
                for (int q = 0; q < P.ndim; ++q) dst[q] = 0.f;
            }
        if (num_points) num_points[row] = cnt;
        if (mean) {
            const float inv = (float)cnt;
#pragma unroll
            for (int q = 0; q < 8; ++q) if (q < P.nfeat) mean[(size_t)row * P.nfeat + q] = __fdiv_rn(acc[q], inv);
        }
    }
}

// The same outputs for ndim == 4 and T <= 8 (KITTI: 5), with the dependent loads of a voxel issued LEVEL BY LEVEL instead of
// point by point (round 5).  vox_emit_kernel walks, per first-touch point, ent -> key -> slot t -> point t
// one after the other: <= 4 voxels x 5 slots x 2 dependent round trips per thread on 21 workgroups -- 39.5 us for 16 k voxels
// that move 1.1 MB.  Here a thread fetches the hash slots of its 4 points together, then their keys and all 4 x T slot
// entries together, then (per voxel) all its points as 16-byte loads; sums and stores keep the order of vox_emit_kernel, so
// voxels, counts, coordinates and means are bit-identical.
__global__ void __launch_bounds__(kScanThreads) vox_emit8_kernel(const float *__restrict__ pts, VoxParams P, VoxWs ws,
                                                                 int32_t *row_offset, int32_t *voxel_num, float *voxels,
                                                                 int32_t *coors, int32_t *num_points, float *mean,
                                                                 int32_t *status)
{
    constexpr int TM = 8;
    __shared__ int wsum[17];
    const VoxBase vb = vox_block_base(P, ws, (int)gridDim.x, wsum, row_offset, voxel_num);
    const int base = blockIdx.x * kPtsPerBlock + threadIdx.x * kPtsPerThread;
    int f[kPtsPerThread], e[kPtsPerThread];
    int s = 0;
    const int n = vox_n(P);
    // level 1: hash slot of every point of this thread (is_first needs it anyway)
#pragma unroll
    for (int k = 0; k < kPtsPerThread; ++k) e[k] = (base + k < n) ? ws.ent[base + k] : -1;
    // level 2: the voxel's key and its whole slot list (slot 0 decides first touch)
    unsigned key[kPtsPerThread];
    int sl[kPtsPerThread][TM];
#pragma unroll
    for (int k = 0; k < kPtsPerThread; ++k) {
        const int ee = e[k] >= 0 ? e[k] : 0;
        key[k] = ws.keys[ee];
#pragma unroll
        for (int t = 0; t < TM; ++t) sl[k][t] = (t < P.T) ? ws.slots[(size_t)ee * P.T + t] : kVoxEmpty;
    }
#pragma unroll
    for (int k = 0; k < kPtsPerThread; ++k) { f[k] = (e[k] >= 0 && sl[k][0] == base + k) ? 1 : 0; s += f[k]; }
    int tot;
    int ex = block_exclusive_scan(s, wsum, &tot);
    int r = vb.base + ex;
    const int cutoff = vb.cutoff;
    const int row0 = row_offset ? row_offset[0] : 0;
    const float4 *p4 = (const float4 *)pts;
#pragma unroll
    for (int k = 0; k < kPtsPerThread; ++k) {
        if (!f[k]) continue;
        const int rank = r++;
        if (rank >= P.max_voxels) continue;
        const int row = row0 + rank;
        if (row >= P.cap) { if (status) atomicOr(status, SASSD_ST_VOXEL_OVERFLOW); continue; }
        const int x = key[k] % (unsigned)P.grid[0];
        const int y = (key[k] / (unsigned)P.grid[0]) % (unsigned)P.grid[1];
        const int z = key[k] / ((unsigned)P.grid[0] * (unsigned)P.grid[1]);
        int32_t *cr = coors + (size_t)row * P.coors_cols;
        if (P.coors_cols == 4) { cr[0] = P.batch_idx; cr[1] = z; cr[2] = y; cr[3] = x; }
        else { cr[0] = z; cr[1] = y; cr[2] = x; }
        int cnt = 0;                                    // slots are ascending and kVoxEmpty >= cutoff: a prefix passes
#pragma unroll
        for (int t = 0; t < TM; ++t) cnt += (sl[k][t] < cutoff) ? 1 : 0;
        // level 3: the voxel's points, all requested before the first is used
        float4 pv[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int j = (t < cnt) ? sl[k][t] : base + k;      // (past the count: a harmless re-read of the thread's own point)
            pv[t] = p4[j];
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            if (t < P.T) {
                const bool in = t < cnt;
                if (voxels)
                    *(float4 *)(voxels + ((size_t)row * P.T + t) * 4) = in ? pv[t] : make_float4(0.f, 0.f, 0.f, 0.f);
                if (in) {
                    if (0 < P.nfeat) acc[0] += pv[t].x;
                    if (1 < P.nfeat) acc[1] += pv[t].y;
                    if (2 < P.nfeat) acc[2] += pv[t].z;
                    if (3 < P.nfeat) acc[3] += pv[t].w;
                }
            }
        }
        if (num_points) num_points[row] = cnt;
        if (mean) {
            const float inv = (float)cnt;
#pragma unroll
            for (int q = 0; q < 4; ++q) if (q < P.nfeat) mean[(size_t)row * P.nfeat + q] = __fdiv_rn(acc[q], inv);
        }
    }
}

__global__ void voxel_mean_kernel(const float *__restrict__ voxels, const int32_t *__restrict__ num, int m, int T,
                                  int ndim, int nfeat, float *__restrict__ mean)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * nfeat) return;
    const int v = i / nfeat, f = i % nfeat;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += voxels[((size_t)v * T + t) * ndim + f];
    mean[i] = __fdiv_rn(s, (float)num[v]);
}

size_t vox_layout(int n, int T, unsigned *hcap_out, size_t off[6])
{
    unsigned hcap = next_pow2((unsigned)(n > 0 ? n : 1) * 2u);
    if (hcap < 1024) hcap = 1024;
    const int nblk = cdiv(n > 0 ? n : 1, kPtsPerBlock);
    size_t o = 0;
    off[0] = o; o += align_up((size_t)hcap * 4, 256);
    off[3] = o; o += align_up((size_t)nblk * 4, 256);          // first-touch counts: right behind the keys, cleared with them (0xFF = -1)
    off[1] = o; o += align_up((size_t)hcap * T * 4, 256);
    off[2] = o; o += align_up((size_t)(n > 0 ? n : 1) * 4, 256);
    off[4] = o; o += 256;
    off[5] = o; o += 256;
    if (hcap_out) *hcap_out = hcap;
    return o;
}

}  // namespace

extern "C" size_t sassd_voxelize_workspace_bytes(int n_points, int max_points)
{
    size_t off[6];
    return vox_layout(n_points, max_points, nullptr, off);
}

static int voxelize_impl(const float *points, int n_points, const int32_t *n_points_dev, int ndim,
                         const float *voxel_size, const float *coors_range, int max_points, int max_voxels,
                         int batch_idx, float *voxels, int32_t *coors, int coors_cols, int32_t *num_points,
                         float *mean, int nfeat, int32_t *row_offset, int32_t *voxel_num, int cap, int32_t *status,
                         void *workspace, size_t workspace_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (n_points < 0 || ndim < 3 || !voxel_size || !coors_range || !coors || !workspace) return SASSD_EINVAL;
    if (max_points < 1 || max_points > SASSD_MAX_POINTS_PER_VOXEL || max_voxels < 1) return SASSD_EINVAL;
    if (coors_cols != 3 && coors_cols != 4) return SASSD_EINVAL;
    if (nfeat < 0 || nfeat > 8 || nfeat > ndim) return SASSD_EINVAL;
    size_t off[6];
    unsigned hcap;
    const size_t need = vox_layout(n_points, max_points, &hcap, off);
    if (workspace_bytes < need) return SASSD_ENOSPC;

    VoxParams P;
    double vol = 1.0;
    for (int j = 0; j < 3; ++j) {
        P.lo[j] = coors_range[j];
        P.vs[j] = voxel_size[j];
        // grid_size = round((hi - lo) / vs) in f32, half-to-even (points_ops.py:24)
        volatile float g = (coors_range[3 + j] - coors_range[j]) / voxel_size[j];
        P.grid[j] = (int)__builtin_rintf(g);
        if (P.grid[j] <= 0) return SASSD_EINVAL;
        vol *= P.grid[j];
    }
    if (vol >= 4294967294.0) return SASSD_EINVAL;
    P.n = n_points; P.ndim = ndim; P.T = max_points; P.max_voxels = max_voxels; P.batch_idx = batch_idx;
    P.coors_cols = coors_cols; P.nfeat = mean ? nfeat : 0; P.cap = cap; P.hmask = hcap - 1;
    P.n_dev = n_points_dev;

    char *w = (char *)workspace;
    VoxWs ws;
    ws.keys = (unsigned *)(w + off[0]);
    ws.slots = (int *)(w + off[1]);
    ws.ent = (int *)(w + off[2]);
    ws.bsum = (int *)(w + off[3]);
    ws.bbase = (int *)(w + off[4]);
    ws.scal = (int *)(w + off[5]);

    int rc;
    // three launches (round 6; rounds 2-5: fill, insert, count, scan, emit): ONE fill clears the key table AND the per-block
    // first-touch counters behind it (0xFF bytes: keys empty, counters -1) and the slot table (0x7F); the insert kernel keeps
    // the counters; every emit block sums the counters before it itself
    const int nblk = cdiv(n_points > 0 ? n_points : 1, kPtsPerBlock);
    const size_t ff_bytes = (size_t)((char *)ws.bsum - (char *)ws.keys) + align_up((size_t)nblk * 4, 256);
    if ((rc = sassd_fill2(ws.keys, ff_bytes, 0xFF, ws.slots, (size_t)hcap * max_points * 4, 0x7F, stream))) return rc;
    if (n_points > 0)
        hipLaunchKernelGGL(vox_insert_kernel, dim3(cdiv(n_points, 256)), dim3(256), 0, stream, points, P, ws, status);
    if (ndim == 4 && max_points <= 8 && !((uintptr_t)points & 15) && !((uintptr_t)voxels & 15))
        hipLaunchKernelGGL(vox_emit8_kernel, dim3(nblk), dim3(kScanThreads), 0, stream, points, P, ws, row_offset, voxel_num,
                           voxels, coors, num_points, mean, status);
    else
        hipLaunchKernelGGL(vox_emit_kernel, dim3(nblk), dim3(kScanThreads), 0, stream, points, P, ws, row_offset, voxel_num,
                           voxels, coors, num_points, mean, status);
    return sassd_launch_status();
}

extern "C" int sassd_voxelize(const float *points, int n_points, int ndim, const float *voxel_size,
                              const float *coors_range, int max_points, int max_voxels, int batch_idx,
                              float *voxels, int32_t *coors, int coors_cols, int32_t *num_points, float *mean,
                              int nfeat, int32_t *row_offset, int32_t *voxel_num, int cap, int32_t *status,
                              void *workspace, size_t workspace_bytes, void *stream)
{
    return voxelize_impl(points, n_points, nullptr, ndim, voxel_size, coors_range, max_points, max_voxels, batch_idx,
                         voxels, coors, coors_cols, num_points, mean, nfeat, row_offset, voxel_num, cap, status,
                         workspace, workspace_bytes, stream);
}

extern "C" int sassd_voxelize_dev(const float *points, int points_cap, const int32_t *n_points_dev, int ndim,
                                  const float *voxel_size, const float *coors_range, int max_points, int max_voxels,
                                  int batch_idx, float *voxels, int32_t *coors, int coors_cols, int32_t *num_points,
                                  float *mean, int nfeat, int32_t *row_offset, int32_t *voxel_num, int cap,
                                  int32_t *status, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!n_points_dev || points_cap <= 0) return SASSD_EINVAL;
    return voxelize_impl(points, points_cap, n_points_dev, ndim, voxel_size, coors_range, max_points, max_voxels,
                         batch_idx, voxels, coors, coors_cols, num_points, mean, nfeat, row_offset, voxel_num, cap,
                         status, workspace, workspace_bytes, stream);
}

extern "C" int sassd_voxel_mean(const float *voxels, const int32_t *num_points, int m, int max_points, int ndim,
                                int nfeat, float *mean, void *stream_)
{
    if (m < 0 || !voxels || !num_points || !mean || nfeat > ndim) return SASSD_EINVAL;
    if (m == 0) return SASSD_OK;
    hipLaunchKernelGGL(voxel_mean_kernel, dim3(cdiv(m * nfeat, 256)), dim3(256), 0, (hipStream_t)stream_, voxels,
                       num_points, m, max_points, ndim, nfeat, mean);
    return sassd_launch_status();
}
