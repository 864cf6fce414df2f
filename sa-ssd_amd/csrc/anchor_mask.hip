// anchor_mask.hip -- device-side anchors_mask (the step immediately before the path).
//
// Replaces, per frame, mmdet/datasets/kitti.py:333-343: sparse_sum_for_anchors_mask (geometry.py:676-682),
// cumsum(0).cumsum(1) (a 9 MB float integral image built on the host) and fused_get_anchors_area (:685-710).
// Counts are exact integers (the reference's float32 cumsum is exact below 2^24), the anchor->cell arithmetic is
// the same fp32 sequence (subtract, IEEE divide, floor, clamp); the missing "-1" on the lower corner of the
// integral-image lookup (geometry.py:693-709) is reproduced literally.
#include "common.h"

namespace {

// blockIdx.z = sample in every kernel of this file (per-sample grid / segment / mask slices of the workspace)
__global__ void am_scatter_kernel(const int32_t *__restrict__ coors, const int32_t *__restrict__ rb,
                                  const int32_t *__restrict__ re, int W0, unsigned *__restrict__ grid, size_t gstride)
{
    const int b = blockIdx.z;
    const int begin = rb ? rb[b] : 0, end = re[b];
    const int i = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= end) return;
    const int4 c = ((const int4 *)coors)[i];
    atomicAdd(&grid[b * gstride + (size_t)c.z * W0 + c.w], 1u);
}

// inclusive row scan, one block per row
__global__ void __launch_bounds__(256) am_rowscan_kernel(unsigned *__restrict__ grid, int W0, size_t gstride)
{
    __shared__ int wsum[17];
    unsigned *row = grid + blockIdx.z * gstride + (size_t)blockIdx.x * W0;
    if ((W0 & 3) == 0 && W0 <= 2048 && !((uintptr_t)row & 15)) {
        // two 16-byte loads per thread, kept in registers across the block scan (round 6: one read of the row instead of two,
        // 16-byte accesses instead of 4-byte ones; integer sums, same result)
        const int x0 = threadIdx.x * 8;
        uint4 a = make_uint4(0u, 0u, 0u, 0u), b = a;
        if (x0 < W0) a = *(const uint4 *)(row + x0);
        if (x0 + 4 < W0) b = *(const uint4 *)(row + x0 + 4);
        const int s = (int)(a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w);
        int tot;
        unsigned run = (unsigned)block_exclusive_scan(s, wsum, &tot);
        a.x += run; a.y += a.x; a.z += a.y; a.w += a.z;
        b.x += a.w; b.y += b.x; b.z += b.y; b.w += b.z;
        if (x0 < W0) *(uint4 *)(row + x0) = a;
        if (x0 + 4 < W0) *(uint4 *)(row + x0 + 4) = b;
        return;
    }
    const int per = (W0 + 255) / 256;
    const int x0 = threadIdx.x * per;
    int s = 0;
    for (int k = 0; k < per; ++k) if (x0 + k < W0) s += (int)row[x0 + k];
    int tot;
    int run = block_exclusive_scan(s, wsum, &tot);
    for (int k = 0; k < per; ++k)
        if (x0 + k < W0) { run += (int)row[x0 + k]; row[x0 + k] = (unsigned)run; }
}

constexpr int kRB = 32;   // rows per column segment

// per (row segment, column): in-place inclusive column scan inside the segment, segment total -> seg[rs][x]
__global__ void __launch_bounds__(256) am_colseg_kernel(unsigned *__restrict__ grid, int H0, int W0,
                                                        unsigned *__restrict__ seg, size_t gstride, size_t sstride)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W0) return;
    grid += blockIdx.z * gstride;
    seg += blockIdx.z * sstride;
    const int y0 = blockIdx.y * kRB;
    // all 32 rows requested before the first is used (round 6: the load -> add -> store chain per row was 32 dependent
    // round trips on 300 workgroups: 18 us for a 9 MB pass; integer sums, same result)
    unsigned v[kRB];
#pragma unroll
    for (int k = 0; k < kRB; ++k) v[k] = (y0 + k < H0) ? grid[(size_t)(y0 + k) * W0 + x] : 0u;
    unsigned run = 0;
#pragma unroll
    for (int k = 0; k < kRB; ++k) {
        run += v[k];
        if (y0 + k < H0) grid[(size_t)(y0 + k) * W0 + x] = run;
    }
    seg[(size_t)blockIdx.y * W0 + x] = run;
}

__global__ void __launch_bounds__(256) am_segscan_kernel(unsigned *__restrict__ seg, int nseg, int W0, size_t sstride)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W0) return;
    seg += blockIdx.z * sstride;
    unsigned run = 0;
    for (int s0 = 0; s0 < nseg; s0 += 64) {      // 64 segment totals in flight at a time (KITTI: 50, one round trip instead of 50)
        unsigned v[64];
#pragma unroll
        for (int k = 0; k < 64; ++k) v[k] = (s0 + k < nseg) ? seg[(size_t)(s0 + k) * W0 + x] : 0u;
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            if (s0 + k < nseg) seg[(size_t)(s0 + k) * W0 + x] = run;           // exclusive
            run += v[k];
        }
    }
}

struct AmParams { float vs0, vs1, off0, off1, thr; int H0, W0, n; };

__device__ __forceinline__ unsigned am_I(const unsigned *grid, const unsigned *seg, int W0, int y, int x)
{
    return grid[(size_t)y * W0 + x] + seg[(size_t)(y / kRB) * W0 + x];
}

__global__ void __launch_bounds__(256) am_anchor_kernel(const float *__restrict__ bv, AmParams P,
                                                        const unsigned *__restrict__ grid,
                                                        const unsigned *__restrict__ seg, uint8_t *__restrict__ mask,
                                                        size_t gstride, size_t sstride)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n) return;
    grid += blockIdx.z * gstride;
    seg += blockIdx.z * sstride;
    mask += (size_t)blockIdx.z * P.n;
    const float4 a = ((const float4 *)bv)[i];
    int c0 = (int)floorf(__fdiv_rn(a.x - P.off0, P.vs0));
    int c1 = (int)floorf(__fdiv_rn(a.y - P.off1, P.vs1));
    int c2 = (int)floorf(__fdiv_rn(a.z - P.off0, P.vs0));
    int c3 = (int)floorf(__fdiv_rn(a.w - P.off1, P.vs1));
    c0 = max(c0, 0); c1 = max(c1, 0);
    c2 = min(c2, P.W0 - 1); c3 = min(c3, P.H0 - 1);
    // like numba's unchecked indexing the reference would wrap negatives; clamp defensively instead
    c2 = max(c2, 0); c3 = max(c3, 0); c0 = min(c0, P.W0 - 1); c1 = min(c1, P.H0 - 1);
    const int ID = (int)am_I(grid, seg, P.W0, c3, c2);
    const int IA = (int)am_I(grid, seg, P.W0, c1, c0);
    const int IB = (int)am_I(grid, seg, P.W0, c3, c0);
    const int IC = (int)am_I(grid, seg, P.W0, c1, c2);
    mask[i] = ((float)(ID - IB - IC + IA) > P.thr) ? 1 : 0;
}

}  // namespace

extern "C" size_t sassd_anchor_mask_workspace_bytes(int H0, int W0)
{
    const int nseg = cdiv(H0, kRB);
    return align_up((size_t)H0 * W0 * 4, 256) + align_up((size_t)nseg * W0 * 4, 256);
}

namespace {
// One launch sequence for `batch` samples: sample b owns the coordinate rows [rb[b], re[b]) of `coors` (rb == NULL: 0),
// mask[b*n_anchors ..] and one sassd_anchor_mask_workspace_bytes slice of the workspace.
int anchor_mask_run(const int32_t *coors, const int32_t *rb, const int32_t *re, int batch, int H0, int W0,
                    const float *anchors_bv, int n_anchors, const float *voxel_size, const float *coors_range,
                    float area_threshold, uint8_t *mask, void *workspace, size_t workspace_bytes, hipStream_t stream)
{
    const size_t per = sassd_anchor_mask_workspace_bytes(H0, W0);
    if (workspace_bytes < per * batch) return SASSD_ENOSPC;
    const size_t gbytes = align_up((size_t)H0 * W0 * 4, 256);
    unsigned *grid = (unsigned *)workspace;                                   // [batch] grids, then [batch] segment sums
    unsigned *seg = (unsigned *)((char *)workspace + gbytes * batch);
    const size_t gstride = gbytes / 4, sstride = (per - gbytes) / 4;
    const int nseg = cdiv(H0, kRB);
    int rc;
    if ((((uintptr_t)grid | (gbytes * batch)) & 15) == 0) {                   // a fill kernel, not a memset node (see sassd_densify)
        if ((rc = sassd_fill2(grid, gbytes * batch, 0, grid, 0, 0, stream))) return rc;
    } else if ((rc = sassd_hip(hipMemsetAsync(grid, 0, gbytes * batch, stream)))) return rc;
    // the voxel count of one cloud never exceeds H0*W0*D; launch for a generous fixed bound and exit early
    const int max_rows = 1 << 18;
    hipLaunchKernelGGL(am_scatter_kernel, dim3(cdiv(max_rows, 256), 1, batch), dim3(256), 0, stream, coors, rb, re, W0,
                       grid, gstride);
    hipLaunchKernelGGL(am_rowscan_kernel, dim3(H0, 1, batch), dim3(256), 0, stream, grid, W0, gstride);
    hipLaunchKernelGGL(am_colseg_kernel, dim3(cdiv(W0, 256), nseg, batch), dim3(256), 0, stream, grid, H0, W0, seg, gstride,
                       sstride);
    hipLaunchKernelGGL(am_segscan_kernel, dim3(cdiv(W0, 256), 1, batch), dim3(256), 0, stream, seg, nseg, W0, sstride);
    AmParams P;
    P.vs0 = voxel_size[0]; P.vs1 = voxel_size[1]; P.off0 = coors_range[0]; P.off1 = coors_range[1];
    P.thr = area_threshold; P.H0 = H0; P.W0 = W0; P.n = n_anchors;
    hipLaunchKernelGGL(am_anchor_kernel, dim3(cdiv(n_anchors, 256), 1, batch), dim3(256), 0, stream, anchors_bv, P,
                       (const unsigned *)grid, (const unsigned *)seg, mask, gstride, sstride);
    return sassd_launch_status();
}
}  // namespace

// `batch` samples in ONE launch sequence: sample b owns coordinate rows [row_offsets[b], row_offsets[b+1]) of `coors`
extern "C" int sassd_anchor_mask_batch(const int32_t *coors, const int32_t *row_offsets, int batch, int H0, int W0,
                                       const float *anchors_bv, int n_anchors, const float *voxel_size,
                                       const float *coors_range, float area_threshold, uint8_t *mask, void *workspace,
                                       size_t workspace_bytes, void *stream_)
{
    if (!coors || !row_offsets || !anchors_bv || !voxel_size || !coors_range || !mask || !workspace) return SASSD_EINVAL;
    if (H0 < 1 || W0 < 1 || n_anchors < 1 || batch < 1 || batch > 65535) return SASSD_EINVAL;
    return anchor_mask_run(coors, row_offsets, row_offsets + 1, batch, H0, W0, anchors_bv, n_anchors, voxel_size,
                           coors_range, area_threshold, mask, workspace, workspace_bytes, (hipStream_t)stream_);
}

extern "C" int sassd_anchor_mask(const int32_t *coors, const int32_t *row_begin_ptr, const int32_t *row_end_ptr,
                                 int H0, int W0, const float *anchors_bv, int n_anchors, const float *voxel_size,
                                 const float *coors_range, float area_threshold, uint8_t *mask, void *workspace,
                                 size_t workspace_bytes, void *stream_)
{
    if (!coors || !row_end_ptr || !anchors_bv || !voxel_size || !coors_range || !mask || !workspace) return SASSD_EINVAL;
    if (H0 < 1 || W0 < 1 || n_anchors < 1) return SASSD_EINVAL;
    return anchor_mask_run(coors, row_begin_ptr, row_end_ptr, 1, H0, W0, anchors_bv, n_anchors, voxel_size, coors_range,
                           area_threshold, mask, workspace, workspace_bytes, (hipStream_t)stream_);
}
